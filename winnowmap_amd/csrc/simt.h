// simt.h — thin wave64 vocabulary used by the gfx950 kernels in this directory.
//
// On the device every "per-lane value" V<T> is just T and the helpers lower to single CDNA instructions
// (DPP shifts, v_readlane, ballots). The kernels are written against this vocabulary so that the test
// suite can also compile them against tests/simt_emu/simt.h, a host-side lock-step emulator of one
// wavefront (test infrastructure only; never part of the product build, which always includes THIS file).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WM_DEV __device__ __forceinline__
// keeps a uniform branch a branch: without it the compiler turns `if (uniform) x = f(x)` into selects that every iteration pays
#define WM_KEEP_BRANCH() asm volatile("")
#define WM_IF(c) if (c) {
#define WM_ELSE } else {
#define WM_END }
#define WM_EMU_ASSERT(x) ((void)0)

typedef unsigned long long wm_mbox_t;      // a mailbox word of the chained-workgroup kernels: {value, stamp}, one 64-bit atomic (below)

namespace simt {

template <class T> using V = T;
using vbool = bool;

WM_DEV int lane() { return (int)(threadIdx.x & 63u); }
WM_DEV int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
WM_DEV void block_sync() { __syncthreads(); }
// LDS-only ordering: DS operations of a wave complete in order, so cross-lane LDS traffic only needs the counter wait
// (and a compiler barrier); outstanding GLOBAL stores are deliberately not waited for.
WM_DEV void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
WM_DEV void block_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class T> WM_DEV T sel(bool c, T a, T b) { return c ? a : b; }
template <class T, class U> WM_DEV T cast(U a) { return (T)a; }
WM_DEV int vmax(int a, int b) { return a > b ? a : b; }
WM_DEV int vmin(int a, int b) { return a < b ? a : b; }
WM_DEV int vmax3(int a, int b, int c) { return vmax(vmax(a, b), c); }          // -> v_max3_i32
WM_DEV int add3(int a, int b, int c) { return (int)((unsigned)a + (unsigned)b + (unsigned)c); } // -> v_add3_u32 (wrapping)
WM_DEV int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }     // wrapping 32-bit add
WM_DEV int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }     // wrapping 32-bit sub

// ---- packed 2 x 16-bit lanes inside one 32-bit register (VOP3P: one instruction works on both halves) ----------------
typedef short wm_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short wm_u16x2 __attribute__((ext_vector_type(2)));
WM_DEV wm_s16x2 as_s16x2(int a) { return __builtin_bit_cast(wm_s16x2, a); }
WM_DEV wm_u16x2 as_u16x2(int a) { return __builtin_bit_cast(wm_u16x2, a); }
WM_DEV int pk_add(int a, int b) { return __builtin_bit_cast(int, as_u16x2(a) + as_u16x2(b)); }                          // v_pk_add_u16 (wrapping)
WM_DEV int pk_sub(int a, int b) { return __builtin_bit_cast(int, as_u16x2(a) - as_u16x2(b)); }                          // v_pk_sub_u16 (wrapping)
WM_DEV int pk_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(as_s16x2(a), as_s16x2(b))); } // v_pk_max_i16
WM_DEV int pk_min(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_min(as_s16x2(a), as_s16x2(b))); } // v_pk_min_i16
WM_DEV int pk_minu(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_min(as_u16x2(a), as_u16x2(b))); } // v_pk_min_u16
WM_DEV int pk_subsat(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_sub_sat(as_s16x2(a), as_s16x2(b))); } // v_pk_sub_i16 clamp
WM_DEV int pk_lshr(int a, int k) { return __builtin_bit_cast(int, as_u16x2(a) >> (unsigned short)k); }                  // v_pk_lshrrev_b16
WM_DEV int pk_mad(int a, int b, int c) { return __builtin_bit_cast(int, as_u16x2(a) * as_u16x2(b) + as_u16x2(c)); }     // v_pk_mad_u16 (wrapping)
// byte permute: result byte k = byte sel[k] of the 8 bytes {hi word a : lo word b} (selector 0..3 -> b, 4..7 -> a, 0x0c -> 0x00)
WM_DEV int perm(int a, int b, int sel) { return (int)__builtin_amdgcn_perm((unsigned)a, (unsigned)b, (unsigned)sel); }
WM_DEV int bfi(int mask, int a, int b) { return b ^ ((a ^ b) & mask); }                                               // (a & mask) | (b & ~mask): v_bfi_b32 (this form has no shared ~mask for the selector to split off)
WM_DEV int alignbit(int hi, int lo, int sh) { return (int)__builtin_amdgcn_alignbit((unsigned)hi, (unsigned)lo, (unsigned)sh); }   // ({hi,lo} >> sh)[31:0]
WM_DEV int lshr(int a, int k) { return (int)((unsigned)a >> k); }

// DPP cross-lane moves (gfx9 encodings): one VALU operation each, no LDS crossbar round trip.
// Lanes without a source lane (or in a row that row_mask disables) receive `old`.
template <int CTRL, int ROW_MASK> WM_DEV int dpp_mov(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, 0xf, false); }
// value of lane-1 (lane 0 receives `fill`): wave_shr:1
WM_DEV int shr1(int x, int fill) { return dpp_mov<0x138, 0xf>(fill, x); }
// value of lane-1, lane 0 receives lane 63: wave_ror:1
WM_DEV int ror1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x13c, 0xf, 0xf, true); }
// inclusive wave scans: row_shr:1,2,4,8 inside the 16-lane rows, then row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3)
#define WM_SCAN(NAME, OP, ID) \
WM_DEV int NAME(int x) \
{ \
	int t; \
	t = dpp_mov<0x111, 0xf>(ID, x); x = OP(x, t); \
	t = dpp_mov<0x112, 0xf>(ID, x); x = OP(x, t); \
	t = dpp_mov<0x114, 0xf>(ID, x); x = OP(x, t); \
	t = dpp_mov<0x118, 0xf>(ID, x); x = OP(x, t); \
	t = dpp_mov<0x142, 0xa>(ID, x); x = OP(x, t); \
	t = dpp_mov<0x143, 0xc>(ID, x); x = OP(x, t); \
	return x; \
}
WM_DEV int scan_op_max(int a, int b) { return a > b ? a : b; }
WM_DEV int scan_op_min(int a, int b) { return a < b ? a : b; }
WM_DEV int scan_op_add(int a, int b) { return a + b; }
WM_SCAN(wave_scan_max, scan_op_max, (-0x7fffffff - 1))
WM_SCAN(wave_scan_min, scan_op_min, 0x7fffffff)
WM_SCAN(wave_scan_add, scan_op_add, 0)
#undef WM_SCAN
// lane j receives lane j+k (uniform k); lanes >= 64-k receive `fill`
WM_DEV int shift_down(int x, int k, int fill)
{
	int y = __shfl_down(x, (unsigned)k, 64);
	return lane() + k >= 64 ? fill : y;
}
// lane j receives lane (j+k) & 63 (uniform k): a rotation through the LDS crossbar (ds_bpermute, no memory)
WM_DEV int rot_down(int x, int k) { return __shfl(x, (int)((threadIdx.x + (unsigned)k) & 63u), 64); }
// wave-wide maximum (all lanes receive it)
WM_DEV int wave_max_i32(int x) { return __builtin_amdgcn_readlane(wave_scan_max(x), 63); }
// value of lane-o for a uniform o (lanes < o receive their own value; callers mask them)
WM_DEV int shr_n(int x, int o) { return __shfl_up(x, (unsigned)o, 64); }
WM_DEV int readlane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
WM_DEV double readlane(double x, int l)
{
	const long long b = __builtin_bit_cast(long long, x);
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
	return __builtin_bit_cast(double, (long long)((unsigned long long)hi << 32 | lo));
}
WM_DEV int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
WM_DEV long long uniform(long long x)
{
	unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32));
	return (long long)((unsigned long long)hi << 32 | lo);
}
WM_DEV unsigned long long ballot(bool c) { return __ballot(c); }
WM_DEV bool any(bool c) { return __any(c); }

// wave-wide maximum of a 64-bit key (all lanes receive the result)
WM_DEV long long wave_max_i64(long long k)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		long long t = __shfl_xor(k, o, 64);
		k = t > k ? t : k;
	}
	return k;
}
WM_DEV int wave_sum_i32(int k)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) k += __shfl_xor(k, o, 64);
	return k;
}

// memory (global or LDS pointers alike)
template <class T> WM_DEV T gld(const T *p, long long i) { return p[i]; }
template <class T> WM_DEV void gst(T *p, long long i, T v) { p[i] = v; }

// "coherent" scratch accessors: data written by one lane and read by another lane of the SAME wave through global
// memory must not be served from a stale L1 line, so these go to L2 (relaxed agent-scope atomics = sc1 accesses).
// loads_land(): s_waitcnt vmcnt(0) — every outstanding vector load of this wave has returned. Placed INSIDE the uniform branch that issues a
// rare load: the compiler otherwise waits at the join of the branch, i.e. on every trip of the surrounding loop, and on gfx9 that wait also
// drains the loop's outstanding STORES (one counter for both) — one store round trip per DP row (ksw_packed_kernel.h).
#ifndef WM_LOADS_LAND
#define WM_LOADS_LAND 1       // 0: the round-2 code (wait at the join) for A/B runs
#endif
WM_DEV void loads_land() { if (WM_LOADS_LAND) __builtin_amdgcn_s_waitcnt(0x0f70); }       // (simm16: vmcnt = 0, expcnt = 7, lgkmcnt = 15 = "do not wait")
WM_DEV void mem_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// agent-scope release + acquire: one lane's plain global stores become visible to the plain loads of every other lane of the wave (no stale L1 line)
WM_DEV void mem_sync_agent() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
WM_DEV int cld8(signed char *p, long long i) { return (int)__hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WM_DEV void cst8(signed char *p, long long i, int v) { __hip_atomic_store(p + i, (signed char)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WM_DEV int cld(int *p, long long i) { return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WM_DEV void cst(int *p, long long i, int v) { __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- additions for the fused window kernels (window_kernel.h) ----------------------------------------------------------------
// a section executed by lane 0 only, written as plain scalar C++ on raw pointers (no V<> values inside): the sequential parts of the
// reference's algorithms (the cycle-leader permutation of radix_sort_128x, chain backtracking) run literally, one lane per wavefront
#define WM_LANE0_BEGIN if ((threadIdx.x & 63u) == 0u) {
#define WM_LANE0_END }
WM_DEV uint64_t readlane(uint64_t x, int l)
{
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(x >> 32), l);
	return (uint64_t)hi << 32 | lo;
}
WM_DEV uint64_t wave_or_u64(uint64_t k)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) k |= (uint64_t)__shfl_xor((long long)k, o, 64);
	return k;
}
WM_DEV uint64_t wave_and_u64(uint64_t k)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) k &= (uint64_t)__shfl_xor((long long)k, o, 64);
	return k;
}
WM_DEV int popc64(uint64_t m) { return __builtin_popcountll(m); }
// per-lane atomic add on an LDS (or global) counter; the old value is not needed
WM_DEV void atomic_inc(int *p, int idx) { atomicAdd(p + idx, 1); }
// wave-uniform bump allocation: lane 0 adds `n` to the global counter, every lane receives the old value
WM_DEV uint64_t wave_alloc(uint64_t *counter, uint64_t n)
{
	uint64_t old = 0;
	if ((threadIdx.x & 63u) == 0u) old = (uint64_t)atomicAdd((unsigned long long*)counter, (unsigned long long)n);
	const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)old), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(old >> 32));
	return (uint64_t)hi << 32 | lo;
}
// wave-uniform append to a device list: returns the slot index
WM_DEV int wave_append(int *counter)
{
	int old = 0;
	if ((threadIdx.x & 63u) == 0u) old = atomicAdd(counter, 1);
	return __builtin_amdgcn_readfirstlane(old);
}

// number of set bits of `mask` below this lane (v_mbcnt)
WM_DEV int mbcnt(uint64_t mask) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u)); }

WM_DEV int vclz(int v) { return __builtin_clz((unsigned)v); }                                   // per lane, v != 0
WM_DEV int vpopc64(uint64_t m) { return __builtin_popcountll(m); }                     // per lane
WM_DEV uint64_t lanemask_lt() { return ((uint64_t)1 << (threadIdx.x & 63u)) - 1; }      // bits of the lanes below this one

// ---- cross-wavefront hand-over through LDS without a barrier (ksw_stripe_kernel.h) --------------------------------------------
// DS operations of one wavefront are issued and executed in order and LDS is one memory per CU, so "payload stores, then the stamp store"
// by the writer and "stamp load, then payload loads" by the reader need no counter wait between them — only the compiler must keep the
// order (volatile accesses + barriers). lds_st_rel: lane 0 stores a uniform value after everything before it; lds_ld_acq: a uniform load
// that nothing after it may overtake. spin_pause: yield the issue slot while polling.
// Every access goes through an address-space-3 pointer: a pointer whose address space the compiler cannot prove becomes a FLAT access, which (a) waits
// for vmcnt(0) — the row's outstanding traceback stores — and (b) travels through the vector-memory path and is NOT ordered with the DS
// instructions around it: a payload written FLAT and a stamp written with ds_write can become visible in the wrong order (measured in round 4:
// 41 flat_ instructions in the first build of ksw_stripe_kernel, a hang under load).
typedef __attribute__((address_space(3))) int wm_lds_int;
WM_DEV int lds_ld(const int *p, long long i) { return ((const wm_lds_int*)p)[i]; }                 // plain load (uniform or per-lane index)
WM_DEV void lds_st(int *p, long long i, int v) { ((wm_lds_int*)p)[i] = v; }                         // plain per-lane store under the current exec mask
WM_DEV void lds_st_rel(int *p, long long i, int v)
{
	asm volatile("" ::: "memory");
	if ((threadIdx.x & 63u) == 0u) ((volatile wm_lds_int*)p)[i] = v;
	asm volatile("" ::: "memory");
}
WM_DEV int lds_ld_acq(const int *p, long long i)
{
	asm volatile("" ::: "memory");
	const int v = __builtin_amdgcn_readfirstlane(((const volatile wm_lds_int*)p)[i]);
	asm volatile("" ::: "memory");
	return v;
}
// a message slot (16-byte aligned): the stamp p[0] first, then the 8 payload ints p[4..11] as two 128-bit loads — three DS instructions issued back
// to back, executed in that order, ONE wait. If the stamp read is the expected one, the payload — written before the stamp — is too.
WM_DEV int lds_ld_msg(const int *p, int (&o)[8])
{
	typedef int wm_i4 __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(3))) wm_i4 wm_lds_i4;
	asm volatile("" ::: "memory");
	const int s = ((const volatile wm_lds_int*)p)[0];
	const wm_i4 a = *(const volatile wm_lds_i4*)((const wm_lds_int*)p + 4), b = *(const volatile wm_lds_i4*)((const wm_lds_int*)p + 8);
	asm volatile("" ::: "memory");
	o[0] = __builtin_amdgcn_readfirstlane(a.x); o[1] = __builtin_amdgcn_readfirstlane(a.y); o[2] = __builtin_amdgcn_readfirstlane(a.z); o[3] = __builtin_amdgcn_readfirstlane(a.w);
	o[4] = __builtin_amdgcn_readfirstlane(b.x); o[5] = __builtin_amdgcn_readfirstlane(b.y); o[6] = __builtin_amdgcn_readfirstlane(b.z); o[7] = __builtin_amdgcn_readfirstlane(b.w);
	return __builtin_amdgcn_readfirstlane(s);
}
// Split-phase forms: the DS instructions are issued where lds_*_issue stands, the wait (s_waitcnt) lands where the value is first USED — lds_uniform /
// lds_msg_take — so an LDS round trip hides behind whatever is computed in between. The value is the one LDS held when the load was issued.
WM_DEV int lds_ld_issue(const int *p, long long i)
{
	asm volatile("" ::: "memory");
	const int v = ((const volatile wm_lds_int*)p)[i];
	asm volatile("" ::: "memory");
	return v;                                   // (per-lane copy of a uniform word: lds_uniform makes it scalar)
}
WM_DEV int lds_uniform(int raw) { return __builtin_amdgcn_readfirstlane(raw); }
struct lds_msg_raw { int s; int a __attribute__((ext_vector_type(4))); int b __attribute__((ext_vector_type(4))); };
WM_DEV void lds_ld_msg_issue(const int *p, lds_msg_raw &m)
{
	typedef int wm_i4 __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(3))) wm_i4 wm_lds_i4;
	asm volatile("" ::: "memory");
	m.s = ((const volatile wm_lds_int*)p)[0];
	m.a = *(const volatile wm_lds_i4*)((const wm_lds_int*)p + 4); m.b = *(const volatile wm_lds_i4*)((const wm_lds_int*)p + 8);
	asm volatile("" ::: "memory");
}
WM_DEV int lds_msg_take(const lds_msg_raw &m, int (&o)[8])
{
	o[0] = __builtin_amdgcn_readfirstlane(m.a.x); o[1] = __builtin_amdgcn_readfirstlane(m.a.y); o[2] = __builtin_amdgcn_readfirstlane(m.a.z); o[3] = __builtin_amdgcn_readfirstlane(m.a.w);
	o[4] = __builtin_amdgcn_readfirstlane(m.b.x); o[5] = __builtin_amdgcn_readfirstlane(m.b.y); o[6] = __builtin_amdgcn_readfirstlane(m.b.z); o[7] = __builtin_amdgcn_readfirstlane(m.b.w);
	return __builtin_amdgcn_readfirstlane(m.s);
}
WM_DEV void spin_pause() { __builtin_amdgcn_s_sleep(1); }
WM_DEV void long_pause() { __builtin_amdgcn_s_sleep(127); }      // ~8 000 cycles
// the pause between two polls of a word in memory: short at first, then ~0.5 us, then ~1.7 us (n = polls so far)
WM_DEV void poll_pause(int n) { if (n < 4) __builtin_amdgcn_s_sleep(2); else if (n < 32) __builtin_amdgcn_s_sleep(20); else __builtin_amdgcn_s_sleep(64); }

// ---- cross-WORKGROUP hand-over through global memory (ksw_chain_kernel.h) --------------------------------------------------------
// The wavefronts of one alignment run as workgroups of their own, possibly on different XCDs (each XCD has its own L2). Every mailbox word is 64 bits
// = {value, stamp} and travels as ONE relaxed agent-scope atomic: such an access is single-copy atomic and coherent across the XCDs (sc1 accesses go
// past the non-coherent caches), so a reader that finds the stamp it expects has the value that was written with it — no fence on either side, nothing
// to write back or invalidate (an agent-scope release / acquire pair costs an L2 write-back + invalidate: DESIGN "agent-scope fences").
// four consecutive LDS ints (16-byte aligned index) in one ds_read_b128
WM_DEV void lds_ld4(const int *p, int i, int (&o)[4])
{
	typedef int wm_i4 __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(3))) wm_i4 wm_lds_i4;
	const wm_i4 a = *(const wm_lds_i4*)((const wm_lds_int*)p + i);
	o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
}
WM_DEV long long mbox_pack(int val, int stamp) { return (long long)(((unsigned long long)(unsigned)stamp << 32) | (unsigned)val); }
WM_DEV int mbox_val(long long w) { return (int)(unsigned)(unsigned long long)w; }
WM_DEV int mbox_stamp(long long w) { return (int)(unsigned)((unsigned long long)w >> 32); }
WM_DEV long long mbox_ld(const wm_mbox_t *p, int i) { return (long long)__hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // per lane
WM_DEV void mbox_st(wm_mbox_t *p, int i, long long v) { __hip_atomic_store(p + i, (wm_mbox_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }       // per lane, under the exec mask
// A load that stays in flight across loop iterations must not live in a variable the compiler sees: its register allocator copies such a value at the joins of
// the loop's branches, and every copy of a pending load is a wait (measured: one memory round trip per row, profiles/r06_chain_timing_v2.txt). The prefetch
// therefore lands in two ACCUMULATION registers the compiler never touches (this library uses no MFMA and spills nothing there), issued and collected by hand:
// mbox_prefetch starts the per-lane 64-bit load, mbox_prefetched waits for every outstanding vector-memory operation of the wave and returns the word.
WM_DEV void mbox_prefetch(const wm_mbox_t *p, int i)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const wm_mbox_t *q = p + i;
	asm volatile("global_load_dwordx2 a[0:1], %0, off sc1" : : "v"(q) : "a0", "a1", "memory");
#endif
}
WM_DEV long long mbox_prefetched()
{
	unsigned lo = 0, hi = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1" : "=v"(lo), "=v"(hi) : : "memory");
#endif
	return (long long)(((unsigned long long)hi << 32) | lo);
}
// uniform 32-bit control words (progress, stop): every lane reads the same address / lane 0 writes
WM_DEV int mbox_ld_word(const int *p, int i) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
WM_DEV void mbox_st_word(int *p, int i, int v) { if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// acc with lane L (a constant) replaced by the uniform value v: v_writelane_b32
// (this compiler has no __builtin_amdgcn_writelane: the instruction itself; both sources are scalar registers / inline constants)
template <int L> WM_DEV int wrlane(int acc, int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	// (one scalar register at most: the lane is an inline constant. readfirstlane: a value the compiler keeps in a vector register — it does so for some
	// uniform values — must not reach the "s" operand as it is; on a value that already is scalar the builtin folds away)
	asm("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(__builtin_amdgcn_readfirstlane(v)), "n"(L));
#endif
	return acc;
}

} // namespace simt
