// tests/simt_emu/simt.h — TEST INFRASTRUCTURE ONLY.
//
// Host-side lock-step emulator of ONE gfx950 wavefront (64 lanes + an EXEC mask) exposing the same
// vocabulary as winnowmap_amd/csrc/simt.h. The test build puts this directory first on the include path
// so the very same kernel headers (ksw_kernel.h, chain_kernel.h, sketch_kernel.h) compile for the CPU and
// can be checked bit-for-bit against the oracle WITHOUT a GPU (tests/test_kernels_emu.py). It exists to
// debug kernel logic cheaply; it is never compiled into, linked with or reachable from the product library.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>

#define WM_SIMT_EMU 1
#define WM_DEV inline
#define WM_KEEP_BRANCH() ((void)0)
#define WM_EMU_ASSERT(x) do { if (!(x)) { fprintf(stderr, "EMU ASSERT %s:%d: %s\n", __FILE__, __LINE__, #x); abort(); } } while (0)

typedef unsigned long long wm_mbox_t;      // a mailbox word of the chained-workgroup kernels: {value, stamp}, one 64-bit atomic (below)

namespace simt {
constexpr int WAVE = 64;
inline uint64_t &exec_mask() { static thread_local uint64_t m = ~0ull; return m; }
inline bool on(int i) { return exec_mask() >> i & 1; }

template <class T> struct V {
	T v[WAVE];
	V() { memset(v, 0xCD, sizeof(v)); }
	V(T s) { for (int i = 0; i < WAVE; ++i) v[i] = s; }
	V(const V &o) = default;
	template <class U> explicit V(const V<U> &o) { for (int i = 0; i < WAVE; ++i) v[i] = (T)o.v[i]; }
	V &operator=(const V &o) { for (int i = 0; i < WAVE; ++i) if (on(i)) v[i] = o.v[i]; return *this; }
	V &operator=(T s) { for (int i = 0; i < WAVE; ++i) if (on(i)) v[i] = s; return *this; }
#define WM_CASSIGN(op) \
	V &operator op(const V &o) { for (int i = 0; i < WAVE; ++i) if (on(i)) v[i] op o.v[i]; return *this; } \
	V &operator op(T s) { for (int i = 0; i < WAVE; ++i) if (on(i)) v[i] op s; return *this; }
	WM_CASSIGN(+=) WM_CASSIGN(-=) WM_CASSIGN(*=) WM_CASSIGN(&=) WM_CASSIGN(|=) WM_CASSIGN(^=) WM_CASSIGN(<<=) WM_CASSIGN(>>=)
#undef WM_CASSIGN
};
using vbool = V<bool>;

#define WM_BINOP(op, R) \
	template <class T> V<R> operator op(const V<T> &a, const V<T> &b) { V<R> r; for (int i = 0; i < WAVE; ++i) r.v[i] = a.v[i] op b.v[i]; return r; } \
	template <class T> V<R> operator op(const V<T> &a, T b) { V<R> r; for (int i = 0; i < WAVE; ++i) r.v[i] = a.v[i] op b; return r; } \
	template <class T> V<R> operator op(T a, const V<T> &b) { V<R> r; for (int i = 0; i < WAVE; ++i) r.v[i] = a op b.v[i]; return r; }
WM_BINOP(+, T) WM_BINOP(-, T) WM_BINOP(*, T) WM_BINOP(/, T) WM_BINOP(%, T) WM_BINOP(&, T) WM_BINOP(|, T) WM_BINOP(^, T)
WM_BINOP(==, bool) WM_BINOP(!=, bool) WM_BINOP(<, bool) WM_BINOP(>, bool) WM_BINOP(<=, bool) WM_BINOP(>=, bool)
#undef WM_BINOP
// shifts: the count may be a lane vector or a plain int of any integral type
template <class T> V<T> operator<<(const V<T> &a, int b) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = a.v[i] << b; return r; }
template <class T> V<T> operator>>(const V<T> &a, int b) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = a.v[i] >> b; return r; }
template <class T, class U> V<T> operator<<(const V<T> &a, const V<U> &b) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = a.v[i] << b.v[i]; return r; }
template <class T, class U> V<T> operator>>(const V<T> &a, const V<U> &b) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = a.v[i] >> b.v[i]; return r; }
template <class T> V<T> operator-(const V<T> &a) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = -a.v[i]; return r; }
template <class T> V<T> operator~(const V<T> &a) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = ~a.v[i]; return r; }
inline vbool operator!(const vbool &a) { vbool r; for (int i = 0; i < WAVE; ++i) r.v[i] = !a.v[i]; return r; }
inline vbool operator&&(const vbool &a, const vbool &b) { vbool r; for (int i = 0; i < WAVE; ++i) r.v[i] = a.v[i] && b.v[i]; return r; }
inline vbool operator||(const vbool &a, const vbool &b) { vbool r; for (int i = 0; i < WAVE; ++i) r.v[i] = a.v[i] || b.v[i]; return r; }
inline vbool operator&&(const vbool &a, bool b) { return a && vbool(b); }
inline vbool operator&&(bool a, const vbool &b) { return vbool(a) && b; }
inline vbool operator||(const vbool &a, bool b) { return a || vbool(b); }
inline vbool operator||(bool a, const vbool &b) { return vbool(a) || b; }

template <class T, class U> V<T> cast(const V<U> &a) { return V<T>(a); }
template <class T, class U> T cast(U a) { return (T)a; }

// ---- divergent control flow -------------------------------------------------------------------------
struct MaskScope {
	uint64_t saved, cond;
	explicit MaskScope(const vbool &c) : saved(exec_mask()), cond(0) { for (int i = 0; i < WAVE; ++i) if (c.v[i]) cond |= 1ull << i; exec_mask() = saved & cond; }
	explicit MaskScope(bool c) : saved(exec_mask()), cond(c ? ~0ull : 0ull) { exec_mask() = saved & cond; }
	void flip() { exec_mask() = saved & ~cond; }
	bool some() const { return exec_mask() != 0; }
	~MaskScope() { exec_mask() = saved; }
};
#define WM_IF(c) { simt::MaskScope _wm_ms(c); if (_wm_ms.some()) {
#define WM_ELSE } _wm_ms.flip(); if (_wm_ms.some()) {
#define WM_END } }

// multi-wave workgroups: the driver runs one host thread per wave and installs a shared barrier
inline int &wave_slot() { static thread_local int w = 0; return w; }
inline pthread_barrier_t *&block_barrier() { static pthread_barrier_t *b = 0; return b; }
inline int wave_in_block() { return wave_slot(); }
inline void block_sync() { if (block_barrier()) pthread_barrier_wait(block_barrier()); }
inline void lds_sync() {}
inline void block_sync_lds() { block_sync(); }
inline V<int> lane() { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = i; return r; }

template <class T> V<T> sel(const vbool &c, const V<T> &a, const V<T> &b) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
template <class T> V<T> sel(const vbool &c, const V<T> &a, T b) { return sel(c, a, V<T>(b)); }
template <class T> V<T> sel(const vbool &c, T a, const V<T> &b) { return sel(c, V<T>(a), b); }
template <class T> V<T> sel(const vbool &c, T a, T b) { return sel(c, V<T>(a), V<T>(b)); }
template <class T> V<T> sel(bool c, const V<T> &a, const V<T> &b) { return c ? a : b; }
template <class T> V<T> sel(bool c, const V<T> &a, T b) { return c ? a : V<T>(b); }
template <class T> V<T> sel(bool c, T a, const V<T> &b) { return c ? V<T>(a) : b; }
inline int sel(bool c, int a, int b) { return c ? a : b; }

inline V<int> vmax(const V<int> &a, const V<int> &b) { return sel(a > b, a, b); }
inline V<int> vmax(const V<int> &a, int b) { return vmax(a, V<int>(b)); }
inline V<int> vmin(const V<int> &a, const V<int> &b) { return sel(a < b, a, b); }
inline V<int> vmin(const V<int> &a, int b) { return vmin(a, V<int>(b)); }
inline int vmax(int a, int b) { return a > b ? a : b; }
inline int vmin(int a, int b) { return a < b ? a : b; }
inline V<int> vmax3(const V<int> &a, const V<int> &b, const V<int> &c) { return vmax(vmax(a, b), c); }
inline V<int> wadd(const V<int> &a, const V<int> &b) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = (int)((unsigned)a.v[i] + (unsigned)b.v[i]); return r; }
inline V<int> wadd(const V<int> &a, int b) { return wadd(a, V<int>(b)); }
inline V<int> wsub(const V<int> &a, const V<int> &b) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = (int)((unsigned)a.v[i] - (unsigned)b.v[i]); return r; }
inline V<int> wsub(const V<int> &a, int b) { return wsub(a, V<int>(b)); }
inline V<int> wsub(int a, const V<int> &b) { return wsub(V<int>(a), b); }
inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
inline int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
inline V<int> add3(const V<int> &a, const V<int> &b, int c) { return wadd(wadd(a, b), c); }
inline V<int> add3(const V<int> &a, const V<int> &b, const V<int> &c) { return wadd(wadd(a, b), c); }

// ---- packed 2 x 16-bit (the device versions are single VOP3P instructions) ---------------------------------------------
inline int pk2(int lo, int hi) { return (int)(((unsigned)hi & 0xffffu) << 16 | ((unsigned)lo & 0xffffu)); }
inline int s_lo(int a) { return (int)(short)(a & 0xffff); }
inline int s_hi(int a) { return (int)(short)((unsigned)a >> 16); }
inline int u_lo(int a) { return a & 0xffff; }
inline int u_hi(int a) { return (int)((unsigned)a >> 16); }
inline int sat16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }
inline int pk_add(int a, int b) { return pk2(u_lo(a) + u_lo(b), u_hi(a) + u_hi(b)); }
inline int pk_sub(int a, int b) { return pk2(u_lo(a) - u_lo(b), u_hi(a) - u_hi(b)); }
inline int pk_max(int a, int b) { return pk2(s_lo(a) > s_lo(b) ? s_lo(a) : s_lo(b), s_hi(a) > s_hi(b) ? s_hi(a) : s_hi(b)); }
inline int pk_min(int a, int b) { return pk2(s_lo(a) < s_lo(b) ? s_lo(a) : s_lo(b), s_hi(a) < s_hi(b) ? s_hi(a) : s_hi(b)); }
inline int pk_minu(int a, int b) { return pk2(u_lo(a) < u_lo(b) ? u_lo(a) : u_lo(b), u_hi(a) < u_hi(b) ? u_hi(a) : u_hi(b)); }
inline int pk_subsat(int a, int b) { return pk2(sat16(s_lo(a) - s_lo(b)), sat16(s_hi(a) - s_hi(b))); }
inline int pk_lshr(int a, int k) { return pk2(u_lo(a) >> k, u_hi(a) >> k); }
inline int pk_mad(int a, int b, int c) { return pk2(u_lo(a) * u_lo(b) + u_lo(c), u_hi(a) * u_hi(b) + u_hi(c)); }
inline int perm(int a, int b, int sel)
{
	const uint64_t src = (uint64_t)(unsigned)a << 32 | (unsigned)b;
	unsigned r = 0;
	for (int k = 0; k < 4; ++k) {
		const int s = sel >> (8 * k) & 0xff;
		WM_EMU_ASSERT(s <= 7 || s == 0x0c);
		const unsigned byte = s == 0x0c ? 0u : (unsigned)(src >> (8 * s) & 0xff);
		r |= byte << (8 * k);
	}
	return (int)r;
}
inline int bfi(int mask, int a, int b) { return (a & mask) | (b & ~mask); }
inline int alignbit(int hi, int lo, int sh) { return (int)(((uint64_t)(unsigned)hi << 32 | (unsigned)lo) >> (sh & 31)); }
inline int lshr(int a, int k) { return (int)((unsigned)a >> k); }
#define WM_LIFT2(NAME) \
	inline V<int> NAME(const V<int> &a, const V<int> &b) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = NAME(a.v[i], b.v[i]); return r; } \
	inline V<int> NAME(const V<int> &a, int b) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = NAME(a.v[i], b); return r; } \
	inline V<int> NAME(int a, const V<int> &b) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = NAME(a, b.v[i]); return r; }
WM_LIFT2(pk_add) WM_LIFT2(pk_sub) WM_LIFT2(pk_max) WM_LIFT2(pk_min) WM_LIFT2(pk_minu) WM_LIFT2(pk_subsat)
#undef WM_LIFT2
inline V<int> pk_lshr(const V<int> &a, int k) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = pk_lshr(a.v[i], k); return r; }
inline V<int> lshr(const V<int> &a, int k) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = lshr(a.v[i], k); return r; }
inline V<int> pk_mad(const V<int> &a, int b, const V<int> &c) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = pk_mad(a.v[i], b, c.v[i]); return r; }
inline V<int> pk_mad(const V<int> &a, int b, int c) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = pk_mad(a.v[i], b, c); return r; }
inline V<int> perm(const V<int> &a, const V<int> &b, const V<int> &sel) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = perm(a.v[i], b.v[i], sel.v[i]); return r; }
inline V<int> perm(const V<int> &a, const V<int> &b, int sel) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = perm(a.v[i], b.v[i], sel); return r; }
inline V<int> bfi(const V<int> &m, const V<int> &a, const V<int> &b) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = bfi(m.v[i], a.v[i], b.v[i]); return r; }
inline V<int> bfi(int m, const V<int> &a, const V<int> &b) { return bfi(V<int>(m), a, b); }
inline V<int> bfi(const V<int> &m, int a, const V<int> &b) { return bfi(m, V<int>(a), b); }
inline V<int> bfi(int m, int a, const V<int> &b) { return bfi(V<int>(m), V<int>(a), b); }
inline V<int> alignbit(const V<int> &hi, const V<int> &lo, int sh) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = alignbit(hi.v[i], lo.v[i], sh); return r; }

// ---- cross-lane -------------------------------------------------------------------------------------
// NB (hardware semantics kept): cross-lane reads see the registers of ALL lanes, active or not.
template <class T> V<T> shr1(const V<T> &x, T fill) { V<T> r; r.v[0] = fill; for (int i = 1; i < WAVE; ++i) r.v[i] = x.v[i - 1]; return r; }
template <class T> V<T> shr1(const V<T> &x, const V<T> &fill) { V<T> r; r.v[0] = fill.v[0]; for (int i = 1; i < WAVE; ++i) r.v[i] = x.v[i - 1]; return r; }
template <class T> V<T> shift_down(const V<T> &x, int k, T fill) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = i + k < WAVE ? x.v[i + k] : fill; return r; }
template <class T> V<T> shift_down(const V<T> &x, int k, const V<T> &fill) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = i + k < WAVE ? x.v[i + k] : fill.v[i]; return r; }
template <class T> V<T> ror1(const V<T> &x) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = x.v[(i + WAVE - 1) & (WAVE - 1)]; return r; }
template <class T> V<T> rot_down(const V<T> &x, int k) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = x.v[(i + k) & (WAVE - 1)]; return r; }
template <class T> V<T> shr_n(const V<T> &x, int o) { V<T> r; for (int i = 0; i < WAVE; ++i) r.v[i] = i >= o ? x.v[i - o] : x.v[i]; return r; }
template <class T> T readlane(const V<T> &x, int l) { WM_EMU_ASSERT(l >= 0 && l < WAVE); return x.v[l]; }
inline int readlane(int x, int) { return x; }
template <class T> T uniform(const V<T> &x) { for (int i = 0; i < WAVE; ++i) if (on(i)) return x.v[i]; return x.v[0]; }
inline int uniform(int x) { return x; }
inline uint64_t ballot(const vbool &c) { uint64_t m = 0; for (int i = 0; i < WAVE; ++i) if (on(i) && c.v[i]) m |= 1ull << i; return m; }
inline uint64_t ballot(bool c) { return c ? exec_mask() : 0; }
inline bool any(const vbool &c) { return ballot(c) != 0; }
inline bool any(bool c) { return c && exec_mask(); }
inline V<long long> wave_max_i64(const V<long long> &k) { WM_EMU_ASSERT(exec_mask() == ~0ull); long long m = k.v[0]; for (int i = 1; i < WAVE; ++i) if (k.v[i] > m) m = k.v[i]; return V<long long>(m); }
inline V<int> wave_scan_max(const V<int> &k) { WM_EMU_ASSERT(exec_mask() == ~0ull); V<int> r = k; for (int i = 1; i < WAVE; ++i) r.v[i] = r.v[i - 1] > k.v[i] ? r.v[i - 1] : k.v[i]; return r; }
inline int wave_max_i32(const V<int> &k) { WM_EMU_ASSERT(exec_mask() == ~0ull); int m = k.v[0]; for (int i = 1; i < WAVE; ++i) if (k.v[i] > m) m = k.v[i]; return m; }
inline V<int> wave_scan_min(const V<int> &k) { WM_EMU_ASSERT(exec_mask() == ~0ull); V<int> r = k; for (int i = 1; i < WAVE; ++i) r.v[i] = r.v[i - 1] < k.v[i] ? r.v[i - 1] : k.v[i]; return r; }
inline V<int> wave_scan_add(const V<int> &k) { WM_EMU_ASSERT(exec_mask() == ~0ull); V<int> r = k; for (int i = 1; i < WAVE; ++i) r.v[i] = r.v[i - 1] + k.v[i]; return r; }
inline V<int> wave_sum_i32(const V<int> &k) { WM_EMU_ASSERT(exec_mask() == ~0ull); int s = 0; for (int i = 0; i < WAVE; ++i) s += k.v[i]; return V<int>(s); }

// ---- memory -----------------------------------------------------------------------------------------
template <class T, class I> V<T> gld(const T *p, const V<I> &idx) { V<T> r(T(0)); for (int i = 0; i < WAVE; ++i) if (on(i)) r.v[i] = p[idx.v[i]]; return r; }
template <class T> T gld(const T *p, long long idx) { return p[idx]; }
template <class T, class I> void gst(T *p, const V<I> &idx, const V<T> &v) { for (int i = 0; i < WAVE; ++i) if (on(i)) p[idx.v[i]] = v.v[i]; }
template <class T, class I> void gst(T *p, const V<I> &idx, T v) { for (int i = 0; i < WAVE; ++i) if (on(i)) p[idx.v[i]] = v; }
template <class T> void gst(T *p, long long idx, T v) { if (exec_mask()) p[idx] = v; }

inline void loads_land() {}
inline void mem_sync() {}
inline void mem_sync_agent() {}
template <class I> V<int> cld8(signed char *p, const V<I> &idx) { V<int> r(0); for (int i = 0; i < WAVE; ++i) if (on(i)) r.v[i] = p[idx.v[i]]; return r; }
inline int cld8(signed char *p, long long idx) { return p[idx]; }
template <class I> void cst8(signed char *p, const V<I> &idx, const V<int> &v) { for (int i = 0; i < WAVE; ++i) if (on(i)) p[idx.v[i]] = (signed char)v.v[i]; }
template <class I> V<int> cld(int *p, const V<I> &idx) { return gld((const int*)p, idx); }
inline int cld(int *p, long long idx) { return p[idx]; }
template <class I> void cst(int *p, const V<I> &idx, const V<int> &v) { gst(p, idx, v); }
inline void cst(int *p, long long idx, int v) { if (exec_mask()) p[idx] = v; }

// ---- additions for the fused window kernels ---------------------------------------------------------------------------------
#define WM_LANE0_BEGIN {
#define WM_LANE0_END }
inline uint64_t wave_or_u64(const V<uint64_t> &k) { WM_EMU_ASSERT(exec_mask() == ~0ull); uint64_t m = 0; for (int i = 0; i < WAVE; ++i) m |= k.v[i]; return m; }
inline uint64_t wave_and_u64(const V<uint64_t> &k) { WM_EMU_ASSERT(exec_mask() == ~0ull); uint64_t m = ~0ull; for (int i = 0; i < WAVE; ++i) m &= k.v[i]; return m; }
inline int popc64(uint64_t m) { return __builtin_popcountll(m); }
template <class I> void atomic_inc(int *p, const V<I> &idx) { for (int i = 0; i < WAVE; ++i) if (on(i)) ++p[idx.v[i]]; }
inline uint64_t wave_alloc(uint64_t *counter, uint64_t n) { return __atomic_fetch_add(counter, n, __ATOMIC_RELAXED); }
inline int wave_append(int *counter) { return __atomic_fetch_add(counter, 1, __ATOMIC_RELAXED); }

inline V<int> mbcnt(uint64_t mask) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = __builtin_popcountll(mask & ((i ? ((uint64_t)1 << i) : (uint64_t)1) - 1)); return r; }

inline V<int> vclz(const V<int> &v) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = v.v[i] ? __builtin_clz((unsigned)v.v[i]) : 32; return r; }
inline V<int> vpopc64(const V<uint64_t> &m) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = __builtin_popcountll(m.v[i]); return r; }
inline V<uint64_t> lanemask_lt() { V<uint64_t> r; for (int i = 0; i < WAVE; ++i) r.v[i] = ((uint64_t)1 << i) - 1; return r; }

// cross-wavefront hand-over through LDS (one host thread per emulated wavefront): release / acquire atomics
inline void lds_st_rel(int *p, long long i, int v) { if (exec_mask()) __atomic_store_n(p + i, v, __ATOMIC_SEQ_CST); }
inline int lds_ld_acq(const int *p, long long i) { return __atomic_load_n(p + i, __ATOMIC_SEQ_CST); }
inline int lds_ld_msg(const int *p, int (&o)[8])
{
	const int s = __atomic_load_n(p, __ATOMIC_SEQ_CST);
	for (int i = 0; i < 8; ++i) o[i] = __atomic_load_n(p + 4 + i, __ATOMIC_SEQ_CST);
	return s;
}
// split-phase forms (csrc/simt.h): the emulator takes the snapshot where the load is issued
inline int lds_ld_issue(const int *p, long long i) { return __atomic_load_n(p + i, __ATOMIC_SEQ_CST); }
inline int lds_uniform(int raw) { return raw; }
struct lds_msg_raw { int s; int o[8]; };
inline void lds_ld_msg_issue(const int *p, lds_msg_raw &m) { m.s = lds_ld_msg(p, m.o); }
inline int lds_msg_take(const lds_msg_raw &m, int (&o)[8]) { for (int i = 0; i < 8; ++i) o[i] = m.o[i]; return m.s; }
inline void spin_pause() { sched_yield(); }
inline void long_pause() { sched_yield(); }
inline void poll_pause(int) { sched_yield(); }
// cross-workgroup mailbox words (csrc/simt.h): 64-bit {value, stamp}, one atomic each
inline void lds_ld4(const int *p, const V<int> &idx, V<int> (&o)[4]) { for (int k = 0; k < 4; ++k) for (int i = 0; i < WAVE; ++i) if (on(i)) { WM_EMU_ASSERT((idx.v[i] & 3) == 0); o[k].v[i] = p[idx.v[i] + k]; } }
inline V<long long> mbox_pack(const V<int> &val, const V<int> &stamp) { V<long long> r; for (int i = 0; i < WAVE; ++i) r.v[i] = (long long)(((unsigned long long)(unsigned)stamp.v[i] << 32) | (unsigned)val.v[i]); return r; }
inline V<long long> mbox_pack(const V<int> &val, int stamp) { V<long long> r; for (int i = 0; i < WAVE; ++i) r.v[i] = (long long)(((unsigned long long)(unsigned)stamp << 32) | (unsigned)val.v[i]); return r; }
inline V<int> mbox_val(const V<long long> &w) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = (int)(unsigned)(unsigned long long)w.v[i]; return r; }
inline V<int> mbox_stamp(const V<long long> &w) { V<int> r; for (int i = 0; i < WAVE; ++i) r.v[i] = (int)(unsigned)((unsigned long long)w.v[i] >> 32); return r; }
inline V<long long> mbox_ld(const wm_mbox_t *p, const V<int> &idx) { V<long long> r(0); for (int i = 0; i < WAVE; ++i) if (on(i)) r.v[i] = (long long)__atomic_load_n(p + idx.v[i], __ATOMIC_SEQ_CST); return r; }
inline void mbox_st(wm_mbox_t *p, const V<int> &idx, const V<long long> &v) { for (int i = 0; i < WAVE; ++i) if (on(i)) __atomic_store_n(p + idx.v[i], (wm_mbox_t)v.v[i], __ATOMIC_SEQ_CST); }
inline V<long long> &mbox_prefetch_reg() { static thread_local V<long long> r(0); return r; }
inline void mbox_prefetch(const wm_mbox_t *p, const V<int> &idx) { WM_EMU_ASSERT(exec_mask() == ~0ull); mbox_prefetch_reg() = mbox_ld(p, idx); }      // (the emulator takes the snapshot where the load is issued)
inline V<long long> mbox_prefetched() { return mbox_prefetch_reg(); }
inline int mbox_ld_word(const int *p, int i) { return __atomic_load_n(p + i, __ATOMIC_SEQ_CST); }
inline void mbox_st_word(int *p, int i, int v) { if (exec_mask()) __atomic_store_n(p + i, v, __ATOMIC_SEQ_CST); }
template <int L> V<int> wrlane(const V<int> &acc, int v) { static_assert(L >= 0 && L < WAVE, "lane"); V<int> r = acc; r.v[L] = v; return r; }
inline int lds_ld(const int *p, long long i) { return p[i]; }
template <class I> void lds_st(int *p, const V<I> &idx, const V<int> &v) { gst(p, idx, v); }
template <class I> void lds_st(int *p, const V<I> &idx, int v) { gst(p, idx, v); }

} // namespace simt
