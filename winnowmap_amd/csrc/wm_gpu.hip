// wm_gpu.hip — libwmgpu.so: C-ABI shim (include/wm_gpu.h) + gfx950 kernel entry points.
// Host side: plans a batch (kernel class, traceback pitch, arena offsets), uploads, launches on the
// context's stream, measures kernel time with HIP events on that stream, and gathers results.
// There is no CPU compute path in this file: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>          // device radix sort + run-length encode for the k-mer counter (wm_write_repetitive_kmers_gpu)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <new>
#include <algorithm>
#include <numeric>
#include <thread>
#include <chrono>
#include <atomic>
#include <mutex>
#include <unistd.h>
#include "host/wm_core.h"
// large staging buffers: no value-initialisation (a std::vector would memset hundreds of MB per batch). When a context is
// given, the buffer comes from that context's PINNED host slab (LIFO bump allocation): device copies to/from pinned memory run
// at full PCIe rate and truly asynchronously, pageable memory is bounced through the runtime's staging buffers.
struct wm_ctx_s;
static void *pin_take(wm_ctx_s *c, size_t bytes, size_t *mark);
static void pin_release(wm_ctx_s *c, size_t mark);
template <class T> struct UBuf {
	T *p; size_t n; wm_ctx_s *c; size_t mark; bool pinned;
	explicit UBuf(size_t n_, wm_ctx_s *c_ = 0) : p(0), n(n_), c(c_), mark(0), pinned(false)
	{
		const size_t bytes = (n_ ? n_ : 1) * sizeof(T);
		if (c) { p = (T*)pin_take(c, bytes, &mark); pinned = p != 0; }
		if (!p) p = (T*)malloc(bytes);
		if (!p) throw std::bad_alloc();         // (caught where the batched calls are issued: reported as an error, never abort())
	}
	~UBuf() { if (pinned) pin_release(c, mark); else free(p); }
	UBuf(const UBuf&) = delete; UBuf &operator=(const UBuf&) = delete;
	T *data() { return p; } const T *data() const { return p; }
	size_t size() const { return n; }
	T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
	T *begin() { return p; } T *end() { return p + n; }
};
static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#include "simt.h"
#include "ksw_kernel.h"
#include "ksw_packed_kernel.h"
#include "ksw_packed_multi_kernel.h"
#include "ksw_stripe_kernel.h"
#include "ksw_chain_kernel.h"
#include "ksw_exts2_kernel.h"
#include "ksw_plan.h"
#include "cigar_walk.h"
#include "reads2bit.h"
#include "sketch_kernel.h"
#include "seedchain_kernel.h"
#include "window_kernel.h"

// ======================================================================================================
// kernels
// ======================================================================================================
// Wave priority (s_setprio: the SIMD's arbiter issues the ready wave with the highest priority first). The path mixes two kinds of kernels on
// one chip: the bulk DP classes — tens of thousands of independent waves, throughput work — and LATENCY-bound serial chains that a whole
// batched call waits for (one alignment over thousands of rows with a barrier per row, the window kernels' lane-0 sections, the traceback walk).
// Sharing a SIMD with seven bulk waves slows a serial chain several times while it costs the bulk nothing to yield: the chains run at
// raised priority, the bulk at the default 0 (profiles/r03c_window_profile.txt: a kernel's duration under load vs alone). WM_PRIO=0 (build define) turns it off for A/B.
#ifndef WM_PRIO
#define WM_PRIO 1
#endif
#ifndef WM_STRIPE_PRIO
#define WM_STRIPE_PRIO 2      // the stripe-pipelined classes: a few hundred wavefronts per launch, each a chain of dependent rows (3: above the exact / clipped register classes, for A/B)
#endif
#if WM_PRIO
#define WM_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define WM_SETPRIO(n) ((void)0)
#endif
// WM_OCC_HINT (build define, A/B): ask the register allocator for one more wavefront per SIMD where a kernel sits just above a step of the register file
// (512 / 4 = 128 registers: ksw_dpp_kernel<8, true, *, true> 130-132, ksw_chain_kernel<2, *, *, true> 122-138; tools/kernel_regs.py) at the price of 3-6 spilled values
#ifndef WM_OCC_HINT
#define WM_OCC_HINT 0
#endif
// register classes: one wave per alignment, two DP cells per lane (ksw_dp_packed, ksw_packed_kernel.h)
template <int BP, bool CLIP, bool HASN, bool EXACT>
__global__ __launch_bounds__(64, (WM_OCC_HINT && BP == 8 && CLIP && EXACT) ? 4 : 1) void ksw_dpp_kernel(wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs,
                                                      const int *__restrict__ order, const uint8_t *__restrict__ seqs,
                                                      uint8_t *__restrict__ tb, wm_ksw_dres_t *__restrict__ res)
{
	if constexpr (EXACT || CLIP || BP >= 16) WM_SETPRIO(2);
	const int j = order[blockIdx.x];
	wmk::ksw_dp_packed<BP, CLIP, HASN, EXACT>(sc, jobs[j], seqs, tb, res + j);
}

// generic class: per-lane state in a global scratch slab (7*T int8 + T int32 per job), one wave per alignment
__global__ __launch_bounds__(64) void ksw_generic_kernel(wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs, const int *__restrict__ order,
                                                          const uint8_t *__restrict__ seqs, uint8_t *__restrict__ tb, uint8_t *scratch,
                                                          const uint64_t *__restrict__ scratch_off, wm_ksw_dres_t *__restrict__ res)
{
	WM_SETPRIO(3);
	const int j = order[blockIdx.x];
	const wm_ksw_djob_t jb = jobs[j];
	const int T = (jb.tlen + 15) / 16 * 16;
	signed char *mem = (signed char*)(scratch + scratch_off[blockIdx.x]);
	int *Hm = (int*)(mem + (size_t)8 * T);
	wmk::ksw_dp_generic<true>(sc, jb, seqs, tb, mem, Hm, res + j);
}

// block classes: NWV waves per alignment (ksw_dp_block). The per-lane state window is in LDS (BLOCK, BLOCK2) or in a global
// scratch slab of 3*WN ints per job (BLOCK3: any hull up to 16 sweeps per row); dynamic LDS = [state window,] publish
// area, then the staged sequences (if they fit in seq_cap bytes)
template <int K, int WN>
__global__ __launch_bounds__(64 * WM_KSW_BLK_NWV) void ksw_block_kernel(wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs, const int *__restrict__ order,
                                                                         const uint8_t *__restrict__ seqs, uint8_t *__restrict__ tb, wm_ksw_dres_t *__restrict__ res, int seq_cap,
                                                                         int *gstate, const uint64_t *__restrict__ gstate_off)
{
	WM_SETPRIO(3);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	constexpr bool GLOBAL = WN == 0;
	const int j = order[blockIdx.x];
	const wm_ksw_djob_t jb = jobs[j];
	const int wn = GLOBAL ? (int)wm_ksw_blk3_wn(jb.tlen) : WN;
	int *W0 = GLOBAL ? gstate + gstate_off[blockIdx.x] : (int*)smem, *W1 = W0 + wn, *Hm = W1 + wn;
	int *pub = GLOBAL ? (int*)smem : Hm + wn;
	uint8_t *sq = (uint8_t*)(pub + WM_KSW_BLK_PUB);
	const int qpad = (jb.qlen + 15) & ~15;
	if (qpad + jb.tlen <= seq_cap) {
		uint8_t *st = sq + qpad;
		for (int i = threadIdx.x; i < jb.qlen; i += blockDim.x) sq[i] = seqs[jb.q_off + i];
		for (int i = threadIdx.x; i < jb.tlen; i += blockDim.x) st[i] = seqs[jb.t_off + i];
		__syncthreads();
		wmk::ksw_dp_block<WM_KSW_BLK_NWV, K, GLOBAL>(sc, jb, sq, st, tb, W0, W1, Hm, wn, pub, res + j);
	} else
		wmk::ksw_dp_block<WM_KSW_BLK_NWV, K, GLOBAL>(sc, jb, seqs + jb.q_off, seqs + jb.t_off, tb, W0, W1, Hm, wn, pub, res + j);
}

// BLOCK / BLOCK2 classes (and, with WM_KSW_PMULTI >= 2, the 16-pair register classes on <4,4>): the packed two-cells-per-lane machine over
// 8 wavefronts (ksw_dp_pmulti<4,8>: 4096 lanes, <8,8>: 8192 lanes). Dynamic LDS: exchange areas, then the staged sequences (if they fit).
// It replaced the unpacked ksw_dp_multi<8, 8|16> in round 3 (24 / 3 GCUPS, the <16> form spilled 377 VGPRs; profiles/r03a_first_run.txt)
// CLIP / HASN: the 16-pair register classes know both per class (ksw_plan.h) and get the lean machine when the band never clips or no operand
// holds an N (most alignments of 1009..2032 lanes: stage-2 fills and extensions inside the 3001-wide band); the BLOCK classes run <true, true>.
template <int BP, int NWV, bool CLIP = true, bool HASN = true>
__global__ __launch_bounds__(64 * NWV) void ksw_pmulti_kernel(wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs, const int *__restrict__ order,
                                                                         const uint8_t *__restrict__ seqs, uint8_t *__restrict__ tb, wm_ksw_dres_t *__restrict__ res, int seq_cap)
{
	WM_SETPRIO(3);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	typedef wmk::ksw_pmulti_lds<BP, NWV> L;
	int *lds = (int*)smem;
	uint8_t *sq = (uint8_t*)(lds + L::INTS);
	const int j = order[blockIdx.x];
	const wm_ksw_djob_t jb = jobs[j];
	const int qpad = (jb.qlen + 15) & ~15;
	// Two copies of the machine, one per address space of the sequences: a pointer that is LDS or global at run time would make every query
	// fetch of every row a FLAT load, whose s_waitcnt vmcnt(0) also drains the row's traceback stores — a memory round trip per row
	// (6 us per row measured, profiles/r03c_window_profile.txt: 22 GCUPS). With the LDS copy the row loop never waits for its stores.
	if (qpad + jb.tlen <= seq_cap) {
		uint8_t *st = sq + qpad;
		for (int i = threadIdx.x; i < jb.qlen; i += blockDim.x) sq[i] = seqs[jb.q_off + i];
		for (int i = threadIdx.x; i < jb.tlen; i += blockDim.x) st[i] = seqs[jb.t_off + i];
		__syncthreads();
		if (jb.flag & KSW_F_APPROX_MAX) wmk::ksw_dp_pmulti<BP, NWV, CLIP, HASN, false>(sc, jb, sq, st, tb, lds, res + j);
		else wmk::ksw_dp_pmulti<BP, NWV, CLIP, HASN, true>(sc, jb, sq, st, tb, lds, res + j);
	} else {
		const uint8_t *qp = seqs + jb.q_off, *tp = seqs + jb.t_off;
		if (jb.flag & KSW_F_APPROX_MAX) wmk::ksw_dp_pmulti<BP, NWV, CLIP, HASN, false>(sc, jb, qp, tp, tb, lds, res + j);
		else wmk::ksw_dp_pmulti<BP, NWV, CLIP, HASN, true>(sc, jb, qp, tp, tb, lds, res + j);
	}
}

// stripe classes (ksw_plan.h: WM_KSW_STRIPE..): NWV wavefronts per alignment, every wavefront a fixed stripe of 128 * BP target lanes in registers,
// row-stamped messages through LDS instead of a barrier per row (ksw_stripe_kernel.h). Static LDS only (the rings: < 6 KB).
template <int BP, int NWV, bool CLIP, bool HASN>
__global__ __launch_bounds__(64 * NWV) void ksw_stripe_kernel(wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs, const int *__restrict__ order,
                                                               const uint8_t *__restrict__ seqs, uint8_t *__restrict__ tb, wm_ksw_dres_t *__restrict__ res)
{
	WM_SETPRIO(WM_STRIPE_PRIO);
	__shared__ int lds[wmk::ksw_stripe_lds<BP, NWV>::INTS];
	const int j = order[blockIdx.x];
	const wm_ksw_djob_t jb = jobs[j];
	if (jb.flag & KSW_F_APPROX_MAX) wmk::ksw_dp_stripe<BP, NWV, CLIP, HASN, false>(sc, jb, seqs, tb, lds, res + j);
	else wmk::ksw_dp_stripe<BP, NWV, CLIP, HASN, true>(sc, jb, seqs, tb, lds, res + j);
}

// chained-workgroup classes (ksw_plan.h: WM_KSW_CHAIN..; ksw_chain_kernel.h): one 64-thread workgroup per WAVEFRONT of an alignment. A workgroup takes a
// ticket when it starts and the ticket names (job, wavefront) — `cmap[ticket]` = {index into `order`, wavefront, mailbox offset in 128-byte units, wavefronts
// of the job}, jobs largest first, the wavefronts of a job in consecutive tickets: every lower ticket is running or done, whatever the dispatcher's order.
template <int BP, bool CLIP, bool HASN, bool EXACT>
__global__ __launch_bounds__(64, (WM_OCC_HINT && BP == 2 && EXACT) ? 4 : 1) void ksw_chain_kernel(wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs, const int *__restrict__ order,
                                                        const uint8_t *__restrict__ seqs, uint8_t *__restrict__ tb, wm_ksw_dres_t *__restrict__ res,
                                                        const uint4 *__restrict__ cmap, int *ticket, wm_mbox_t *mail)
{
	WM_SETPRIO(WM_STRIPE_PRIO);
	__shared__ int tbs[wm_chain_box::GROUP * 32 * BP];            // traceback rows of the group being staged
	int tk = 0;
	if (threadIdx.x == 0) tk = atomicAdd(ticket, 1);
	tk = __builtin_amdgcn_readfirstlane(tk);
	const uint4 m = cmap[tk];
	const int j = order[m.x];
	const wm_ksw_djob_t jb = jobs[j];
	wm_mbox_t *mb = mail + (size_t)m.z * 16;
	wmk::ksw_dp_chain<BP, CLIP, HASN, EXACT>(sc, jb, seqs, tb, mb, (int)m.w, (int)m.y, tbs, res + j);      // (the launcher's ticket table holds jobs of one kind: 82..94 registers without the exact maximum, 122..138 with it)
}

// jobs the reference returns from before it does anything (an empty operand, a mismatch that can never be seen: src/ksw2_extd2_sse.c:68,92) get the
// result of ksw_reset_extz (src/ksw2.h:153-158): one launch over the job table instead of one copy per such job
__global__ __launch_bounds__(256) void ksw_reset_kernel(int n, const wm_ksw_djob_t *__restrict__ jobs, wm_ksw_dres_t *__restrict__ res)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n || jobs[j].klass >= 0) return;
	wm_ksw_dres_t z;
	z.max = 0; z.zdropped = 0; z.max_q = z.max_t = z.mqe_t = z.mte_q = -1; z.score = z.mqe = z.mte = KSW_NEG_INF; z.reach_end = 0; z.n_cigar = 0; z.bt_i = z.bt_j = -1;
	res[j] = z;
}

// operands of position jobs (wm_ksw_batch_pos): expand query and target of job blockIdx.x into the batch's sequence slab. Query = two-strand
// space of a (sub)read of the resident read codes (src/align.c:871-877), target = 4-bit packed reference (mm_idx_getseq, src/index.c:161-171).
__global__ __launch_bounds__(64) void ksw_expand_kernel(const wm_ksw_djob_t *__restrict__ jobs, const wm_ksw_dsrc_t *__restrict__ src,
                                                         const uint64_t *__restrict__ reads_pk, const uint64_t *__restrict__ reads_nm, const uint32_t *__restrict__ S,
                                                         uint8_t *__restrict__ seqs)
{
	WM_SETPRIO(1);
	const int j = blockIdx.x;
	const wm_ksw_djob_t jb = jobs[j];
	if (jb.klass < 0) return;                   // degenerate job (an empty operand, src/ksw2_extd2_sse.c:68): it has no slot in the slab
	const wm_ksw_dsrc_t sr = src[j];
	const int64_t L = sr.qwin_len;
	uint8_t *q = seqs + jb.q_off, *t = seqs + jb.t_off;
	for (int i = threadIdx.x; i < jb.qlen; i += 64) {
		const int64_t p = (int64_t)sr.q_pos + (int64_t)i * sr.step;
		uint8_t c = 4;
		if (p >= 0 && p < L) c = (uint8_t)wmk::rd2_code(reads_pk, reads_nm, (long long)(sr.qwin_off + p));
		else if (p >= L && p < 2 * L) { c = (uint8_t)wmk::rd2_code(reads_pk, reads_nm, (long long)(sr.qwin_off + (2 * L - 1 - p))); c = c < 4 ? 3 - c : 4; }
		q[i] = c;
	}
	for (int i = threadIdx.x; i < jb.tlen; i += 64) {
		const int64_t p = sr.t_base + (int64_t)i * sr.step;
		t[i] = (uint8_t)(S[p >> 3] >> ((p & 7) << 2) & 0xf);
	}
}

// one thread per alignment: walk the traceback, write run-length ops (backtrack order) into the job's slot
__global__ __launch_bounds__(64) void ksw_backtrack_kernel(int n, const wm_ksw_djob_t *__restrict__ jobs, const uint8_t *__restrict__ tb,
                                                            wm_ksw_dres_t *__restrict__ res, uint32_t *__restrict__ cig_scratch, int *__restrict__ err)
{
	WM_SETPRIO(3);
	const int j = blockIdx.x * 64 + threadIdx.x;
	if (j >= n) return;
	wm_ksw_dres_t r = res[j];
	int nc = 0;
	if (r.bt_i == KSW_BT_WATCHDOG) atomicMax(err, 2);
	if (r.bt_i >= 0) {
		nc = wmk::ksw_backtrack_thread(jobs[j], tb, r.bt_i, r.bt_j, cig_scratch + jobs[j].cig_off, jobs[j].cig_cap);
		if (nc < 0) { atomicMax(err, 1); nc = 0; }
	}
	res[j].n_cigar = nc;
}

// the same walk by one wavefront per alignment, traceback tiles through LDS (ksw_backtrack_wave); opt-in: WM_KSW_COOP_BT=1
__global__ __launch_bounds__(64) void ksw_backtrack_coop_kernel(int n, const wm_ksw_djob_t *__restrict__ jobs, const uint8_t *__restrict__ tb,
                                                                 wm_ksw_dres_t *__restrict__ res, uint32_t *__restrict__ cig_scratch, int *__restrict__ err)
{
	WM_SETPRIO(3);
	__shared__ uint8_t tile[KSW_BT_ROWS * 64];
	const int j = blockIdx.x;
	const int bt_i = res[j].bt_i, bt_j = res[j].bt_j;
	int nc = 0;
	if (bt_i == KSW_BT_WATCHDOG && threadIdx.x == 0) atomicMax(err, 2);
	if (bt_i >= 0) {
		nc = wmk::ksw_backtrack_wave(jobs[j], tb, bt_i, bt_j, cig_scratch + jobs[j].cig_off, jobs[j].cig_cap, tile);
		if (nc < 0) { if (threadIdx.x == 0) atomicMax(err, 1); nc = 0; }
	}
	if (threadIdx.x == 0) res[j].n_cigar = nc;
}

// exclusive prefix sum of n_cigar (single block; n is at most a few hundred thousand)
__global__ __launch_bounds__(1024) void ksw_scan_kernel(int n, const wm_ksw_dres_t *__restrict__ res, uint32_t *__restrict__ off, uint32_t *__restrict__ total)
{
	WM_SETPRIO(3);
	__shared__ uint32_t part[1024];
	const int tid = threadIdx.x, per = (n + 1023) / 1024, b = tid * per, e = b + per < n ? b + per : n;
	uint32_t s = 0;
	for (int i = b; i < e; ++i) s += (uint32_t)res[i].n_cigar;
	part[tid] = s;
	__syncthreads();
	if (tid == 0) {
		uint32_t acc = 0;
		for (int i = 0; i < 1024; ++i) { uint32_t t = part[i]; part[i] = acc; acc += t; }
		*total = acc;
	}
	__syncthreads();
	s = part[tid];
	for (int i = b; i < e; ++i) { off[i] = s; s += (uint32_t)res[i].n_cigar; }
}

// compact (and un-reverse) the per-job op lists into one dense pool
__global__ __launch_bounds__(64) void ksw_gather_kernel(const wm_ksw_djob_t *__restrict__ jobs, const wm_ksw_dres_t *__restrict__ res,
                                                         const uint32_t *__restrict__ off, const uint32_t *__restrict__ cig_scratch,
                                                         uint32_t *__restrict__ pool, uint32_t pool_cap)
{
	WM_SETPRIO(3);
	const int j = blockIdx.x, n = res[j].n_cigar;
	const uint32_t *src = cig_scratch + jobs[j].cig_off;
	const bool rev = (jobs[j].flag & KSW_F_REV_CIGAR) != 0;
	for (int i = threadIdx.x; i < n; i += 64) {
		const uint32_t dst = off[j] + (uint32_t)i;
		if (dst < pool_cap) pool[dst] = src[rev ? i : n - 1 - i];
	}
}

// mm_test_zdrop's scan (src/align.c:32-66) over the finished alignments of the jobs that ask for it (WM_KSW_F_ZDWALK: the gap fills, whose z-drop the
// mapper judges after every first pass, src/align.c:736): one thread per job walks the job's ops in the dense pool and its operands in the batch's
// slab — the same function body the host compiles (cigar_walk.h). sc: the scores as the CALLER gave them (q, e not swapped).
__global__ __launch_bounds__(64) void ksw_zdwalk_kernel(int n, wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs, const wm_ksw_dres_t *__restrict__ res,
                                                         const uint32_t *__restrict__ off, const uint32_t *__restrict__ pool, const uint8_t *__restrict__ seqs,
                                                         wm_zd_t *__restrict__ zd)
{
	WM_SETPRIO(3);
	const int j = blockIdx.x * 64 + threadIdx.x;
	if (j >= n) return;
	wm_zd_t z = { 0, -1, -1, -1, -1 };
	if ((jobs[j].flag & WM_KSW_F_ZDWALK) && !(jobs[j].flag & KSW_F_REV_CIGAR))
		wm_zdrop_walk(seqs + jobs[j].q_off, seqs + jobs[j].t_off, pool + off[j], res[j].n_cigar, sc.match, sc.mismatch, sc.sc_ambi, sc.q, sc.e, &z);
	zd[j] = z;
}

// ======================================================================================================
// host
// ======================================================================================================
static thread_local char g_err[512] = "";
static int set_err(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err(WM_ENODEV, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

struct wm_ctx_s {
	int device;
	hipStream_t stream;
	hipStream_t kstream[4];                     // own side streams (created on first use): kernel classes of one batch run concurrently
	// the mapper's contexts draw their side streams from ONE pool instead, sized so that all streams of the mapper fit the hardware queues
	// (GPU_MAX_HW_QUEUES = 20 since round 4 — 6 main streams + 14 side streams; 16 / 24 / 32 measured slower, profiles/r04g_sched_sweep.txt;
	// more streams than queues share queues and serialise, profiles/r02f_stream_conc.txt)
	hipStream_t *side_pool; int n_side_pool; std::atomic<unsigned> *side_next;
	std::vector<hipStream_t> owned_pool; std::atomic<unsigned> owned_next[3];   // (the pool lives in the mapper's first context; the others point at it; one cursor per weight class)
	hipEvent_t kev[5];
	hipEvent_t cev[WM_KSW_NCLASS][2];           // per-class start/stop (on the stream the class was launched on)
	double k_ms[WM_KSW_NCLASS]; uint64_t k_cells[WM_KSW_NCLASS], k_launches[WM_KSW_NCLASS];   // accumulated per kernel class
	// every launch of a class as an interval on the device's clock (ms since the process-wide base event): launches of one class overlap on
	// different streams, so their SUMMED durations are residency, not time — the union of the intervals is (wm_mapper_kernel_union)
	std::vector<std::pair<float, float>> k_iv[WM_KSW_NCLASS];
	std::mutex iv_mu;                           // k_iv: appended by the batched call that holds the context, read by wm_mapper_kernel_union from any thread
	uint8_t *arena;
	size_t arena_bytes, arena_used;
	hipEvent_t ev[4];
	hipEvent_t sync_ev;                         // blocking-sync event: waiting threads sleep instead of spinning (the host cores are the scarce resource)
	float last_ms, aux_ms;
	uint64_t acc_cells; double t_prep, t_run, t_fetch;
	// flat index in HBM (wm_index_upload)
	uint64_t *d_hkey, *d_hval, *d_P;
	uint8_t *d_bloom;
	uint32_t *d_S;                              // packed reference (4 bits per base), for position jobs
	std::vector<uint64_t> seq_off; std::vector<uint32_t> seq_len;       // contig table of the uploaded index (bounds of position jobs)
	// the read codes of the current mini-batch(es), resident: 2 bits per base in d_reads, the ambiguity bitmap in d_reads_nm (reads2bit.h; one allocation);
	// reads_bytes = bases a job may address, reads_cap = bases the allocation holds (wm_reads_upload / GpuOps::load_reads)
	uint64_t *d_reads, *d_reads_nm; size_t reads_bytes, reads_cap; bool owns_reads;
	int hbits;
	wm_sketch_params_t skp;
	bool have_index, owns_index;
	bool owns_filter;                           // d_bloom came from wm_sketch_set_filter (no index on this context)
	int host_threads;                           // threads the batched entry points may use for their host-side packing / sorting
	uint8_t *pin; size_t pin_bytes, pin_used;   // pinned host slab for staging (allocated on first use)
	int *pin_small;                             // a few pinned words for scalar read-backs (an async copy into pageable memory makes the caller spin until the stream gets there)
};

static void *pin_take(wm_ctx_s *c, size_t bytes, size_t *mark)
{
	if (!c->pin) {
		if (c->pin_bytes == (size_t)-1) return 0;                        // allocation failed before: stay pageable
		const size_t want = (size_t)(getenv("WM_PINNED_MB") ? atoll(getenv("WM_PINNED_MB")) : 3072) << 20;
		if (want == 0 || hipHostMalloc((void**)&c->pin, want, hipHostMallocDefault) != hipSuccess) { c->pin = 0; c->pin_bytes = (size_t)-1; (void)hipGetLastError(); return 0; }
		c->pin_bytes = want; c->pin_used = 0;
	}
	const size_t off = (c->pin_used + 255) & ~(size_t)255;
	if (off + bytes > c->pin_bytes) return 0;
	*mark = c->pin_used;
	c->pin_used = off + bytes;
	return c->pin + off;
}
static void pin_release(wm_ctx_s *c, size_t mark) { c->pin_used = mark; }

struct wm_ksw_dev_batch_s {
	int n_jobs;
	wm_ksw_score_t sc, sc_in;                   // sc: the cheaper gap piece first (src/ksw2_extd2_sse.c:70); sc_in: as the caller gave them
	wm_zd_t *d_zd = 0;                          // z-drop scans of the jobs flagged WM_KSW_F_ZDWALK (0: no job asked)
	std::vector<wm_ksw_djob_t> jobs;            // host copy
	std::vector<int> order[WM_KSW_NCLASS];      // job indices per class, largest first
	std::vector<int> ord;                       // the classes' orders back to back (what the device sees)
	std::vector<int> degenerate;                // jobs the reference returns from early (src/ksw2_extd2_sse.c:68,92)
	std::vector<std::string> dumped;            // WM_KSW_DUMP: files written for this batch's stripe launches
	// device pointers (inside the arena)
	uint8_t *d_gscratch; uint64_t *d_goff; std::vector<uint64_t> goff;
	uint8_t *d_b3state; uint64_t *d_b3off; std::vector<uint64_t> b3off;
	// chained-workgroup classes: ticket -> (job, wavefront, mailbox) per class, the mailboxes (filled with 0xff before the launches), one ticket counter per class
	// ([kind]: 0 = approximate maximum, 1 = exact maximum + z-drop: two kernels per class)
	std::vector<uint4> cmap[WM_KSW_NCLASS - WM_KSW_CHAIN][2]; uint4 *d_cmap[WM_KSW_NCLASS - WM_KSW_CHAIN][2]; wm_mbox_t *d_mail = 0; size_t mail_words = 0; int *d_tickets = 0;
	wm_ksw_djob_t *d_jobs; int *d_order; uint8_t *d_seqs, *d_tb; wm_ksw_dres_t *d_res; uint32_t *d_cig, *d_off, *d_total, *d_pool; int *d_err;
	size_t pool_cap, arena_mark, slab_bytes;
	uint64_t cells, tb_bytes;
	uint64_t class_cells[WM_KSW_NCLASS];
	float dp_ms, bt_ms;
	uint32_t total_ops;
	int h_err;
};

extern "C" const char *wm_last_error(void) { return g_err; }
#ifndef WM_BUILD_DEFINES
#define WM_BUILD_DEFINES ""
#endif
extern "C" const char *wm_build_defines(void) { return WM_BUILD_DEFINES; }      // kernel-variant defines this library was compiled with (winnowmap_amd/build.py)

// wait for everything queued on the context's stream WITHOUT burning a host core: hipStreamSynchronize — and, on the GPU boxes, also
// hipEventSynchronize on a blocking-sync event (thread CPU time == wall time inside the batched calls, profiles/r02c_bench_hub.json) —
// spin for as long as the kernels run, and the container's CPU quota is the scarce resource. So: record an event, poll it, sleep in
// between (50 us doubling to 1 ms; the batches take tens of milliseconds). WM_SPIN_SYNC=1 restores hipStreamSynchronize (A/B).
static hipError_t ctx_sync(wm_ctx_s *c)
{
	static const bool spin = getenv("WM_SPIN_SYNC") != 0;
	if (spin) return hipStreamSynchronize(c->stream);
	hipError_t e = hipEventRecord(c->sync_ev, c->stream);
	if (e != hipSuccess) return e;
	int us = 50;
	for (;;) {
		e = hipEventQuery(c->sync_ev);
		if (e != hipErrorNotReady) return e;
		std::this_thread::sleep_for(std::chrono::microseconds(us));
		if (us < 1000) us *= 2;
	}
}

// ROCm maps HIP streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and reads it when the runtime initialises: a library
// constructor sets the default the mapper is tuned for (6 contexts + 14 side streams) before any HIP call of this process can have happened
// through this library; a value given by the user wins. (Python callers get the same default from winnowmap_amd/__init__.py.)
// (this is the one setting that has to happen at load time — the HIP runtime reads the variable when it initialises. WM_NO_PROCESS_DEFAULTS=1
// leaves the process alone; include/wm_gpu.h documents both process-wide settings.)
// Round 6: no library constructor any more — loading the library changes nothing in the process. The default is set by the first
// wm_ctx_create / wm_device_count of the process, immediately before this library's first HIP call; if the host program has already initialised the HIP
// runtime by then (it read the variable at that moment), the setting is simply too late and the program's own environment rules.
static void wm_default_hw_queues()
{
	static std::once_flag once;
	std::call_once(once, [] { if (!getenv("WM_NO_PROCESS_DEFAULTS")) setenv("GPU_MAX_HW_QUEUES", "20", 0); });
}

// The mapping calls allocate and free their per-call tables (tens of MB per batched call, from 16+ worker threads) at a rate at which glibc's defaults
// turn into system calls: a worker's malloc arena grows in 128-KB steps (one mprotect each), gives the memory back as soon as it is free, deletes and
// re-creates its 64-MB heaps, and serves anything above the mmap threshold by mmap / munmap. The sampling profile of a bench run had 58 % of the
// host's CPU samples inside mprotect (profiles/r04l_host_sampling_profile.txt) — with the address-space lock held, i.e. with every other thread's page
// faults waiting. Keep the memory instead: grow in 64-MB steps, never trim, allocate up to 32 MB from the arenas: -15 % host CPU, +9 % throughput in one
// GPU call (profiles/r04m_malloc_tuning.txt). Process-wide, like GPU_MAX_HW_QUEUES; WM_MALLOPT=0 or any MALLOC_* tunable of the caller's own wins.
// Applied when the first mapper of the process is created (not at load time: a program that only links the library for its batched operations keeps
// glibc's defaults — ADVICE r4); WM_MALLOPT=0 / WM_NO_PROCESS_DEFAULTS=1 switch it off.
static void wm_default_malloc()
{
	static std::once_flag once;
	std::call_once(once, [] {
		const char *off = getenv("WM_MALLOPT");
		if ((off && atoi(off) == 0) || getenv("WM_NO_PROCESS_DEFAULTS")) return;
		if (!getenv("MALLOC_TOP_PAD_")) mallopt(M_TOP_PAD, 64 << 20);
		if (!getenv("MALLOC_TRIM_THRESHOLD_")) mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
		if (!getenv("MALLOC_MMAP_THRESHOLD_")) mallopt(M_MMAP_THRESHOLD, 32 << 20);
	});
}

// one recorded event per device: the zero of the interval clock above (hipEventElapsedTime works between events of different streams)
static hipEvent_t device_base_event(int device)
{
	static std::mutex mu;
	static std::vector<hipEvent_t> ev;
	std::lock_guard<std::mutex> lk(mu);
	if ((int)ev.size() <= device) ev.resize(device + 1, (hipEvent_t)0);
	if (!ev[device]) {
		hipEvent_t e;
		if (hipSetDevice(device) != hipSuccess || hipEventCreate(&e) != hipSuccess) return 0;
		if (hipEventRecord(e, 0) != hipSuccess || hipEventSynchronize(e) != hipSuccess) { hipEventDestroy(e); return 0; }
		ev[device] = e;
	}
	return ev[device];
}

// how the mapper's side-stream pool of P streams is divided among light | heavy | huge ksw calls (WM_SIDE_SPLIT=light,heavy; the rest = huge)
static void side_split(int P, int *light, int *heavy)
{
	int a = (P * 3 + 3) / 7, h = (P * 2 + 3) / 7;           // 14 streams: 6 | 4 | 4 (profiles/r04g_sched_sweep.txt); 10: 4 | 3 | 3
	if (const char *e = getenv("WM_SIDE_SPLIT")) sscanf(e, "%d,%d", &a, &h);
	*light = std::max(0, a); *heavy = std::max(0, h);
}

extern "C" int wm_device_count(void)
{
	wm_default_hw_queues();
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" int wm_ctx_create(int device, size_t arena_bytes, wm_ctx_t **out)
{
	int n = 0;
	*out = 0;
	wm_default_hw_queues();
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_err(WM_ENODEV, "no HIP device visible (libwmgpu has no CPU fallback)");
	if (device < 0 || device >= n) return set_err(WM_EINVAL, "device %d out of range (%d visible)", device, n);
	HIPCHK(hipSetDevice(device));
	wm_ctx_t *c = new wm_ctx_t();
	c->device = device;
	if (arena_bytes == 0) {
		size_t fr = 0, tot = 0;
		HIPCHK(hipMemGetInfo(&fr, &tot));
		arena_bytes = fr / 4 < ((size_t)24 << 30) ? fr / 4 : ((size_t)24 << 30);
	}
	c->arena_bytes = arena_bytes;
	HIPCHK(hipMalloc((void**)&c->arena, arena_bytes));
	HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	for (int i = 0; i < 4; ++i) c->kstream[i] = 0;
	c->side_pool = 0; c->n_side_pool = 0; c->side_next = 0;
	for (int i = 0; i < 3; ++i) c->owned_next[i] = 0;
	for (int i = 0; i < 5; ++i) HIPCHK(hipEventCreateWithFlags(&c->kev[i], hipEventDisableTiming));
	for (int k = 0; k < WM_KSW_NCLASS; ++k) { HIPCHK(hipEventCreate(&c->cev[k][0])); HIPCHK(hipEventCreate(&c->cev[k][1])); c->k_ms[k] = 0; c->k_cells[k] = c->k_launches[k] = 0; }
	for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&c->ev[i]));
	HIPCHK(hipEventCreateWithFlags(&c->sync_ev, hipEventBlockingSync | hipEventDisableTiming));
	c->arena_used = 0; c->last_ms = 0; c->aux_ms = 0; c->host_threads = 1; c->pin = 0; c->pin_bytes = c->pin_used = 0; c->have_index = false; c->owns_index = false; c->d_hkey = c->d_hval = c->d_P = 0; c->d_bloom = 0;
	c->owns_filter = false;
	c->pin_small = 0;
	if (hipHostMalloc((void**)&c->pin_small, 256, hipHostMallocDefault) != hipSuccess) { c->pin_small = 0; (void)hipGetLastError(); }
	c->d_S = 0; c->d_reads = 0; c->d_reads_nm = 0; c->reads_bytes = c->reads_cap = 0; c->owns_reads = false;
	*out = c;
	return WM_OK;
}

extern "C" void wm_ctx_destroy(wm_ctx_t *c)
{
	if (!c) return;
	hipSetDevice(c->device);
	hipStreamSynchronize(c->stream);
	for (int i = 0; i < 4; ++i) hipEventDestroy(c->ev[i]);
	hipEventDestroy(c->sync_ev);
	hipStreamDestroy(c->stream);
	if (c->pin) hipHostFree(c->pin);
	if (c->pin_small) hipHostFree(c->pin_small);
	for (int i = 0; i < 4; ++i) if (c->kstream[i]) hipStreamDestroy(c->kstream[i]);
	for (hipStream_t st : c->owned_pool) hipStreamDestroy(st);
	for (int i = 0; i < 5; ++i) hipEventDestroy(c->kev[i]);
	for (int k = 0; k < WM_KSW_NCLASS; ++k) { hipEventDestroy(c->cev[k][0]); hipEventDestroy(c->cev[k][1]); }
	hipFree(c->arena);
	if (c->have_index && c->owns_index) { hipFree(c->d_hkey); hipFree(c->d_hval); hipFree(c->d_P); hipFree(c->d_bloom); hipFree(c->d_S); }
	if (c->d_reads && c->owns_reads) hipFree(c->d_reads);
	if (!c->have_index && c->owns_filter && c->d_bloom) hipFree(c->d_bloom);
	delete c;
}

extern "C" float wm_last_kernel_ms(const wm_ctx_t *c) { return c ? c->last_ms : 0.f; }
extern "C" int wm_ctx_device(const wm_ctx_t *c) { return c ? c->device : -1; }

static inline double now_ms();
static void *arena_take(wm_ctx_t *c, size_t bytes)
{
	size_t a = (c->arena_used + 255) & ~(size_t)255;
	if (a + bytes > c->arena_bytes) return 0;
	c->arena_used = a + bytes;
	return c->arena + a;
}

// releases what a batched call took from the arena when the call returns
struct ArenaMark { wm_ctx_t *c; size_t m; ArenaMark(wm_ctx_t *c_) : c(c_), m(c_->arena_used) {} ~ArenaMark() { c->arena_used = m; } };

// variant = EXACT*4 + CLIP*2 + HASN; jobs with an N run on the CLIP instantiation (a superset: exact emulation of stale lanes)
template <int BP> static void launch_dpp(int variant, int n, hipStream_t s, const wm_ksw_score_t &sc, const wm_ksw_djob_t *jobs, const int *order,
                                         const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res)
{
	dim3 g(n), b(64);
	const bool exact = variant & 4, clip = (variant & 2) || (variant & 1), hasn = variant & 1;
	if (exact) {
		if (hasn) hipLaunchKernelGGL((ksw_dpp_kernel<BP, true, true, true>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
		else if (clip) hipLaunchKernelGGL((ksw_dpp_kernel<BP, true, false, true>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
		else hipLaunchKernelGGL((ksw_dpp_kernel<BP, false, false, true>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
	} else {
		if (hasn) hipLaunchKernelGGL((ksw_dpp_kernel<BP, true, true, false>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
		else if (clip) hipLaunchKernelGGL((ksw_dpp_kernel<BP, true, false, false>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
		else hipLaunchKernelGGL((ksw_dpp_kernel<BP, false, false, false>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
	}
}

// WM_KSW_STRIPE_ROWS4 / WM_KSW_STRIPE_ROWS8: alignments of the 4- / 8-pair register classes with at least this many DP rows run on four wavefronts
// (ksw_plan.h: wm_ksw_route); 0 = never. WM_KSW_STRIPE=0: no stripe classes at all (the round-3 kernels; A/B). wm_ksw_set_routing overrides.
static std::atomic<int> g_stripe_on(-1), g_stripe_rows4(-1), g_stripe_rows8(-1), g_stripe_wide16(0);
static int stripe_min_rows(int bp)
{
	if (g_stripe_on.load(std::memory_order_relaxed) < 0) {
		g_stripe_rows4 = getenv("WM_KSW_STRIPE_ROWS4") ? std::max(0, atoi(getenv("WM_KSW_STRIPE_ROWS4"))) : 0;
		g_stripe_rows8 = getenv("WM_KSW_STRIPE_ROWS8") ? std::max(0, atoi(getenv("WM_KSW_STRIPE_ROWS8"))) : 4096;      // (3 000-row extensions are faster on one wavefront, 10 000-row ones on four: profiles/r04c_probe.txt)
		g_stripe_wide16 = getenv("WM_KSW_STRIPE16") ? atoi(getenv("WM_KSW_STRIPE16")) & 3 : 1;      // bits: 1 = <2,16> instead of <4,8> (default since round 5: profiles/r05_sched.txt), 2 = <1,16> for long narrow jobs (measured worse; ksw_plan.h)
		g_stripe_on = !(getenv("WM_KSW_STRIPE") && atoi(getenv("WM_KSW_STRIPE")) == 0);
	}
	return !g_stripe_on.load(std::memory_order_relaxed) ? 0 : bp == 4 ? g_stripe_rows4.load(std::memory_order_relaxed) : bp == 8 ? g_stripe_rows8.load(std::memory_order_relaxed) : 1;   // (bp == 0: are the stripe classes on at all)
}
extern "C" void wm_ksw_set_routing(int on, int rows4, int rows8)
{
	stripe_min_rows(0);
	if (on >= 0) { g_stripe_on = on != 0; g_stripe_wide16 = on == 2 ? 3 : on == 3 ? 0 : 1; }      // (1: the default routing, <2,16> for the 1793..3840-lane hulls; 2: both sixteen-wavefront geometries; 3: none of them)
	if (rows4 >= 0) g_stripe_rows4 = rows4;
	if (rows8 >= 0) g_stripe_rows8 = rows8;
}

// variant = CLIP * 2 + HASN (0, 2, 3)
template <int BP, int NWV> static void launch_stripe(int variant, int n, hipStream_t s, const wm_ksw_score_t &sc, const wm_ksw_djob_t *jobs, const int *order,
                                                     const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res)
{
	dim3 g(n), b(64 * NWV);
	if (variant & 1) hipLaunchKernelGGL((ksw_stripe_kernel<BP, NWV, true, true>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
	else if (variant & 2) hipLaunchKernelGGL((ksw_stripe_kernel<BP, NWV, true, false>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
	else hipLaunchKernelGGL((ksw_stripe_kernel<BP, NWV, false, false>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
}

// variant = CLIP * 2 + HASN (0, 2, 3); n = wavefronts (= workgroups) of the class
template <int BP, bool EXACT> static void launch_chain(int variant, int n, hipStream_t s, const wm_ksw_score_t &sc, const wm_ksw_djob_t *jobs, const int *order,
                                                       const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res, const uint4 *cmap, int *ticket, wm_mbox_t *mail)
{
	dim3 g(n), b(64);
	if (variant & 1) hipLaunchKernelGGL((ksw_chain_kernel<BP, true, true, EXACT>), g, b, 0, s, sc, jobs, order, seqs, tb, res, cmap, ticket, mail);
	else if (variant & 2) hipLaunchKernelGGL((ksw_chain_kernel<BP, true, false, EXACT>), g, b, 0, s, sc, jobs, order, seqs, tb, res, cmap, ticket, mail);
	else hipLaunchKernelGGL((ksw_chain_kernel<BP, false, false, EXACT>), g, b, 0, s, sc, jobs, order, seqs, tb, res, cmap, ticket, mail);
}
// WM_KSW_CHAIN (wm_ksw_route_chain's mode): bit 0 = the stripe classes and the old wide-hull kernels' jobs run on the chained-workgroup kernels, bit 1 = long exact
// extensions of the 8-pair register classes too (from WM_KSW_CHAIN_ROWS rows on); 0 = none (the round-5 routing, A/B). WM_KSW_CHAIN_BP=4: 512-lane stripes.
static std::atomic<int> g_chain_mode(-1), g_chain_rows(2048), g_chain_geom(0);
static int chain_mode()
{
	if (g_chain_mode.load(std::memory_order_relaxed) < 0) {
		g_chain_rows = getenv("WM_KSW_CHAIN_ROWS") ? std::max(1, atoi(getenv("WM_KSW_CHAIN_ROWS"))) : 2048;
		g_chain_geom = getenv("WM_KSW_CHAIN_BP") && atoi(getenv("WM_KSW_CHAIN_BP")) == 4 ? 1 : 0;
		g_chain_mode = getenv("WM_KSW_CHAIN") ? atoi(getenv("WM_KSW_CHAIN")) & 7 : 1;
	}
	return g_chain_mode.load(std::memory_order_relaxed);
}
extern "C" void wm_ksw_set_chain_routing(int mode, int min_rows_exact, int bp)
{
	chain_mode();
	if (mode >= 0) g_chain_mode = mode & 7;
	if (min_rows_exact > 0) g_chain_rows = min_rows_exact;
	if (bp == 2 || bp == 4) g_chain_geom = bp == 4 ? 1 : 0;
}

// Plans one batch: kernel class, traceback pitch and arena offsets per job; uploads the job table and the operands. Operands are either
// bytes (`jobs` + `seqs`: only the part of `seqs` the jobs refer to is uploaded) or positions in resident data (`pos`: expanded in HBM).
static int ksw_prepare_impl(wm_ctx_t *c, const wm_ksw_score_t *sc_in, int n_jobs, const wm_ksw_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes,
                            const wm_ksw_pos_t *pos, wm_ksw_dev_batch_t **out)
{
	*out = 0;
	if (!c) return set_err(WM_EINVAL, "null context");
	if (n_jobs < 0) return set_err(WM_EINVAL, "n_jobs < 0");
	wm_ksw_score_t sc = *sc_in;
	if (!wm_ksw_score_ok(&sc)) return set_err(WM_EINVAL, "unsupported scoring parameters (need match>0, mismatch<0, sc_ambi<=0, q,e,q2>0, e2>=0, (q+e)+(q2+e2)<=127 as src/options.c:166-176)");
	if (sc.q2 + sc.e2 < sc.q + sc.e) { int8_t t = sc.q; sc.q = sc.q2; sc.q2 = t; t = sc.e; sc.e = sc.e2; sc.e2 = t; }   // src/ksw2_extd2_sse.c:70
	if (pos && (!c->d_S || c->seq_len.empty())) return set_err(WM_EINVAL, "position jobs need wm_index_upload on this context");
	if (pos && !c->d_reads) return set_err(WM_EINVAL, "position jobs need wm_reads_upload on this context");
	HIPCHK(hipSetDevice(c->device));
	wm_ksw_dev_batch_t *b = new wm_ksw_dev_batch_t();
	b->n_jobs = n_jobs; b->sc = sc; b->sc_in = *sc_in; b->cells = b->tb_bytes = 0; b->dp_ms = b->bt_ms = 0; b->total_ops = 0; b->h_err = 0;
	memset(b->class_cells, 0, sizeof(b->class_cells));
	b->arena_mark = c->arena_used;
	b->jobs.resize(n_jobs);
	// where the operands go in the device slab: position jobs are laid out back to back; byte jobs keep their offsets, rebased to the
	// lowest one so that only [lo, hi) of the caller's buffer travels
	UBuf<wm_ksw_dsrc_t> dsrc(pos ? n_jobs : 0, c);             // (pinned: uploaded as it is)
	size_t slab_lo = 0, slab_bytes = 0;
	int bad0 = -1;
	if (pos) {
		uint64_t tot = 0;
		for (int i = 0; i < n_jobs; ++i) {                                        // slab offsets: a running sum
			const wm_ksw_pos_t &s = pos[i];
			wm_ksw_djob_t &d = b->jobs[i];
			const uint64_t ql = s.qlen > 0 ? s.qlen : 0, tl = s.tlen > 0 ? s.tlen : 0;
			d.q_off = (uint32_t)tot; d.t_off = (uint32_t)(tot + ql);
			tot += ql + tl;
		}
		std::atomic<int> badp(-1);
		WM_SITE("ksw.positions");
		wm::parallel_for(c->host_threads, (size_t)n_jobs, [&](size_t i) {
			const wm_ksw_pos_t &s = pos[i];
			memset(&dsrc[i], 0, sizeof(dsrc[i]));
			if (s.qlen <= 0 || s.tlen <= 0) return;                                // degenerate: never read
			const int64_t t_last = (int64_t)s.t_pos + (int64_t)(s.tlen - 1) * s.step;
			if ((s.step != 1 && s.step != -1) || s.rid < 0 || (size_t)s.rid >= c->seq_len.size() || s.t_pos < 0 || t_last < 0 ||
			    (uint32_t)s.t_pos >= c->seq_len[s.rid] || (uint64_t)t_last >= c->seq_len[s.rid] ||
			    s.qwin_off < 0 || s.qwin_len < 0 || (uint64_t)s.qwin_off + (uint64_t)s.qwin_len > c->reads_bytes) { badp = (int)i; return; }
			dsrc[i].qwin_off = s.qwin_off; dsrc[i].qwin_len = s.qwin_len; dsrc[i].q_pos = s.q_pos; dsrc[i].step = s.step; dsrc[i].pad = 0;
			dsrc[i].t_base = (int64_t)c->seq_off[s.rid] + s.t_pos;
		});
		if (badp >= 0) bad0 = badp;
		if (tot >= ((uint64_t)1 << 32)) { delete b; return set_err(WM_ENOMEM, "batch holds %.1f GB of sequence (limit 4 GB per batch)", tot / 1073741824.0); }
		slab_bytes = (size_t)tot;
	} else {
		size_t lo = seqs_bytes, hi = 0;
		for (int i = 0; i < n_jobs; ++i) {
			const wm_ksw_job_t &s = jobs[i];
			if (s.qlen <= 0 || s.tlen <= 0) continue;
			if ((size_t)s.q_off + s.qlen > seqs_bytes || (size_t)s.t_off + s.tlen > seqs_bytes) { if (bad0 < 0) bad0 = i; continue; }
			lo = std::min(lo, (size_t)std::min(s.q_off, s.t_off));
			hi = std::max(hi, std::max((size_t)s.q_off + s.qlen, (size_t)s.t_off + s.tlen));
		}
		if (hi > lo) { slab_lo = lo; slab_bytes = hi - lo; }
	}
	if (bad0 >= 0) { delete b; return set_err(WM_EINVAL, pos ? "job %d: operand positions outside the resident reads / reference" : "job %d: sequence offsets outside seqs", bad0); }
	// the reference returns before doing anything when a mismatch can never be seen (:92)
	const int n_sc = sc.sc_ambi == 0 ? -sc.e2 : sc.sc_ambi;
	int min_sc = sc.mismatch < n_sc ? sc.mismatch : n_sc;
	if (sc.sc_ambi < min_sc) min_sc = sc.sc_ambi;
	const bool never = -min_sc > 2 * (sc.q + sc.e);
	uint64_t tb_off = 0, cig_off = 0;
	std::vector<uint64_t> cells(n_jobs, 0), bands(n_jobs, 0);
	std::atomic<int> bad(-1);
	WM_SITE("ksw.classify");
	wm::parallel_for(c->host_threads, (size_t)n_jobs, [&](size_t i) {          // per-job classification (byte jobs: scans both sequences for N)
		wm_ksw_djob_t &d = b->jobs[i];
		int32_t qlen, tlen, w, zdrop, end_bonus, flag;
		uint32_t q_off = d.q_off, t_off = d.t_off;
		if (pos) { const wm_ksw_pos_t &s = pos[i]; qlen = s.qlen; tlen = s.tlen; w = s.w; zdrop = s.zdrop; end_bonus = s.end_bonus; flag = s.flag; }
		else { const wm_ksw_job_t &s = jobs[i]; qlen = s.qlen; tlen = s.tlen; w = s.w; zdrop = s.zdrop; end_bonus = s.end_bonus; flag = s.flag; q_off = s.q_off; t_off = s.t_off; }
		memset(&d, 0, sizeof(d));
		d.qlen = qlen; d.tlen = tlen; d.w = w; d.zdrop = zdrop; d.end_bonus = end_bonus; d.flag = flag;
		if (flag & (0x01 | 0x04 | 0x10 | 0x100 | 0x200 | 0x400)) { bad = (int)i; d.klass = -1; return; }
		if (qlen <= 0 || tlen <= 0 || never) { d.klass = -1; return; }                                                 // :68,:92
		const int has_n = pos ? (pos[i].has_n != 0) : (wm_ksw_has_n(seqs + q_off, qlen) | wm_ksw_has_n(seqs + t_off, tlen));
		d.q_off = pos ? q_off : (uint32_t)(q_off - slab_lo); d.t_off = pos ? t_off : (uint32_t)(t_off - slab_lo);
		int n_col;
		d.klass = wm_ksw_classify(qlen, tlen, w, has_n, flag, &n_col);
		if (stripe_min_rows(0)) d.klass = wm_ksw_route(d.klass, n_col, qlen, tlen, w, has_n, stripe_min_rows(4), stripe_min_rows(8), g_stripe_wide16.load(std::memory_order_relaxed));
		d.klass = wm_ksw_route_chain(d.klass, qlen, tlen, w, has_n, flag, chain_mode(), g_chain_rows.load(std::memory_order_relaxed), g_chain_geom.load(std::memory_order_relaxed));
		d.n_col = n_col;
		cells[i] = wm_ksw_cells(qlen, tlen, w, &bands[i]);
	});
	if (bad >= 0) {
		const int i = bad;
		delete b;
		return set_err(WM_EINVAL, "job %d: KSW_EZ_SCORE_ONLY/GENERIC_SC/APPROX_DROP/SPLICE flags are not used by the mapper (src/align.c) and not supported", i);
	}
	for (int i = 0; i < n_jobs; ++i) {
		wm_ksw_djob_t &d = b->jobs[i];
		if (d.klass < 0) { b->degenerate.push_back(i); continue; }
		d.tb_off = tb_off;
		const uint64_t rows = (uint64_t)d.qlen + d.tlen - 1;
		tb_off += (rows * d.n_col + 15) & ~(uint64_t)15;
		d.cig_off = (uint32_t)cig_off; d.cig_cap = d.qlen + d.tlen + 2;
		cig_off += d.cig_cap;
		b->cells += bands[i]; b->tb_bytes += cells[i];
		b->class_cells[d.klass] += bands[i];
		b->order[d.klass].push_back(i);
	}
	// inside a class the largest jobs go first (they bound the kernel's duration). Only the coarse order matters, so this is a
	// stable counting sort on a 7-bit logarithmic size key (exponent + 1 mantissa bit), not a comparison sort of ~10^6 jobs
	{
		auto size_key = [&](int j) { const uint64_t v = cells[j] | 1; const int e = 63 - __builtin_clzll(v); return 2 * e + (int)(e > 0 ? (v >> (e - 1)) & 1 : 0); };
		std::vector<int> tmp;
		for (int k = 0; k < WM_KSW_NCLASS; ++k) {
			std::vector<int> &o = b->order[k];
			if (o.size() < 2) continue;
			int cnt[130];
			memset(cnt, 0, sizeof(cnt));
			for (int j : o) ++cnt[size_key(j)];
			int pos_[130], acc = 0;
			for (int kk = 129; kk >= 0; --kk) { pos_[kk] = acc; acc += cnt[kk]; }      // descending keys
			tmp.resize(o.size());
			for (int j : o) tmp[pos_[size_key(j)]++] = j;
			o.swap(tmp);
		}
	}
	// chained-workgroup classes: ticket -> (job, wavefront, mailbox) per class and kind; the mailboxes' sizes
	size_t chain_mw = 0;
	bool chain_any = false;
	for (int kc = 0; kc < WM_KSW_NCLASS - WM_KSW_CHAIN; ++kc) {
		const int sw = 128 * wm_ksw_chain_bp[kc >> 2];
		const std::vector<int> &o = b->order[WM_KSW_CHAIN + kc];
		b->d_cmap[kc][0] = b->d_cmap[kc][1] = 0;
		for (size_t jo = 0; jo < o.size(); ++jo) {
			const wm_ksw_djob_t &d = b->jobs[o[jo]];
			const int nwv = wm_chain_nwv(d.n_col, d.tlen, sw);
			std::vector<uint4> &cm = b->cmap[kc][(d.flag & KSW_F_APPROX_MAX) ? 0 : 1];
			for (int wv = 0; wv < nwv; ++wv) cm.push_back(make_uint4((unsigned)jo, (unsigned)wv, (unsigned)(chain_mw / 16), (unsigned)nwv));
			chain_mw += ((size_t)wm_chain_box::words(nwv) + 15) & ~(size_t)15;
			chain_any = true;
		}
	}
	// device buffers. The host-made tables — jobs | launch order | operand sources | ticket tables — are ONE block and travel in ONE copy (round 6: a
	// batched call made ~20 small copies, each a blit kernel of ~0.2 ms on the call's stream: profiles/r05_last_bench.txt, __amd_rocclr_copyBuffer)
	const size_t nj = n_jobs > 0 ? n_jobs : 1;
	auto al256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t o_ord = al256(nj * sizeof(wm_ksw_djob_t)), o_src = o_ord + al256(nj * sizeof(int));
	size_t o_cm[WM_KSW_NCLASS - WM_KSW_CHAIN][2], tab_bytes = o_src + (pos ? al256(nj * sizeof(wm_ksw_dsrc_t)) : 0);
	for (int kc = 0; kc < WM_KSW_NCLASS - WM_KSW_CHAIN; ++kc)
		for (int e = 0; e < 2; ++e) { o_cm[kc][e] = tab_bytes; tab_bytes += al256(b->cmap[kc][e].size() * sizeof(uint4)); }
	uint8_t *d_tab = (uint8_t*)arena_take(c, tab_bytes);
	b->d_jobs = (wm_ksw_djob_t*)d_tab;
	b->d_order = d_tab ? (int*)(d_tab + o_ord) : 0;
	for (int kc = 0; kc < WM_KSW_NCLASS - WM_KSW_CHAIN && d_tab; ++kc)
		for (int e = 0; e < 2; ++e) if (!b->cmap[kc][e].empty()) b->d_cmap[kc][e] = (uint4*)(d_tab + o_cm[kc][e]);
	b->d_res = (wm_ksw_dres_t*)arena_take(c, al256(nj * sizeof(wm_ksw_dres_t)) + nj * 4 + 64);      // results | CIGAR offsets: one copy back (wm_ksw_dev_fetch)
	b->d_off = b->d_res ? (uint32_t*)((uint8_t*)b->d_res + al256(nj * sizeof(wm_ksw_dres_t))) : 0;
	b->d_err = (int*)arena_take(c, 64);                          // [0] error flag, [1] total CIGAR ops: one 8-byte copy back
	b->d_total = b->d_err ? (uint32_t*)(b->d_err + 1) : 0;
	bool any_zd = false;
	for (int i = 0; i < n_jobs && !any_zd; ++i) any_zd = (b->jobs[i].flag & WM_KSW_F_ZDWALK) != 0;
	b->d_zd = any_zd ? (wm_zd_t*)arena_take(c, nj * sizeof(wm_zd_t)) : 0;
	b->d_seqs = (uint8_t*)arena_take(c, slab_bytes + 64);
	b->slab_bytes = slab_bytes;
	wm_ksw_dsrc_t *d_src = pos && d_tab ? (wm_ksw_dsrc_t*)(d_tab + o_src) : 0;
	b->d_cig = (uint32_t*)arena_take(c, (cig_off + 16) * 4);
	b->pool_cap = cig_off + 16;
	b->d_pool = (uint32_t*)arena_take(c, b->pool_cap * 4);
	b->d_tb = (uint8_t*)arena_take(c, tb_off + 64);
	{   // scratch slabs of the generic class: 8*T bytes (7 int8 arrays, padded) + 4*T for H
		uint64_t go = 0;
		for (int j : b->order[WM_KSW_GENERIC]) { const uint64_t T = ((uint64_t)b->jobs[j].tlen + 15) / 16 * 16; b->goff.push_back(go); go += 12 * T + 256; }
		b->d_gscratch = (uint8_t*)arena_take(c, go + 256);
		b->d_goff = (uint64_t*)arena_take(c, b->goff.size() * 8 + 64);
		if (!b->d_gscratch || !b->d_goff) b->d_tb = 0;
		uint64_t so = 0;                               // BLOCK3: 3 * WN ints per job
		for (int j : b->order[WM_KSW_BLOCK3]) { b->b3off.push_back(so); so += 3 * wm_ksw_blk3_wn(b->jobs[j].tlen); }
		b->d_b3state = (uint8_t*)arena_take(c, so * 4 + 256);
		b->d_b3off = (uint64_t*)arena_take(c, b->b3off.size() * 8 + 64);
		if (!b->d_b3state || !b->d_b3off) b->d_tb = 0;
	}
	if (chain_any) {   // chained-workgroup classes: the mailboxes (filled with 0xff before the launches) and the ticket counters
		b->mail_words = chain_mw;
		b->d_mail = (wm_mbox_t*)arena_take(c, chain_mw * 8 + 256);
		b->d_tickets = (int*)arena_take(c, 64 * sizeof(int));
		if (!b->d_mail || !b->d_tickets) b->d_tb = 0;
	}
	if (!b->d_jobs || !b->d_order || !b->d_res || !b->d_off || !b->d_total || !b->d_err || !b->d_seqs || (any_zd && !b->d_zd) || (pos && !d_src) || !b->d_cig || !b->d_pool || !b->d_tb) {
		c->arena_used = b->arena_mark;
		delete b;
		return set_err(WM_ENOMEM, "batch needs %.1f MB of traceback + buffers; arena is %.1f MB", (tb_off + cig_off * 8 + slab_bytes) / 1048576.0, c->arena_bytes / 1048576.0);
	}
	b->ord.clear();
	b->ord.reserve(nj);
	for (int k = 0; k < WM_KSW_NCLASS; ++k) b->ord.insert(b->ord.end(), b->order[k].begin(), b->order[k].end());
	// the job table and the launch order travel through the context's pinned slab: an asynchronous copy out of pageable memory makes the runtime pin
	// the pages for the duration of the copy (or bounce them through its own staging buffer, waiting in between) — driver calls per batched call
	UBuf<uint8_t> ptab(tab_bytes, c);
	if (n_jobs) memcpy(ptab.data(), b->jobs.data(), (size_t)n_jobs * sizeof(wm_ksw_djob_t));
	if (!b->ord.empty()) memcpy(ptab.data() + o_ord, b->ord.data(), b->ord.size() * sizeof(int));
	if (pos && n_jobs > 0) memcpy(ptab.data() + o_src, dsrc.data(), (size_t)n_jobs * sizeof(wm_ksw_dsrc_t));
	for (int kc = 0; kc < WM_KSW_NCLASS - WM_KSW_CHAIN; ++kc)
		for (int e = 0; e < 2; ++e)
			if (!b->cmap[kc][e].empty()) memcpy(ptab.data() + o_cm[kc][e], b->cmap[kc][e].data(), b->cmap[kc][e].size() * sizeof(uint4));
	HIPCHK(hipMemcpyAsync(d_tab, ptab.data(), tab_bytes, hipMemcpyHostToDevice, c->stream));
	if (pos) {
		if (n_jobs > 0) hipLaunchKernelGGL(ksw_expand_kernel, dim3(n_jobs), dim3(64), 0, c->stream, b->d_jobs, d_src, c->d_reads, c->d_reads_nm, c->d_S, b->d_seqs);
	} else if (slab_bytes) HIPCHK(hipMemcpyAsync(b->d_seqs, seqs + slab_lo, slab_bytes, hipMemcpyHostToDevice, c->stream));
	if (!b->goff.empty()) HIPCHK(hipMemcpyAsync(b->d_goff, b->goff.data(), b->goff.size() * 8, hipMemcpyHostToDevice, c->stream));
	if (!b->b3off.empty()) HIPCHK(hipMemcpyAsync(b->d_b3off, b->b3off.data(), b->b3off.size() * 8, hipMemcpyHostToDevice, c->stream));
	HIPCHK(ctx_sync(c));             // (the host tables above are read by the copies until here)
	*out = b;
	return WM_OK;
}

extern "C" int wm_ksw_dev_prepare(wm_ctx_t *c, const wm_ksw_score_t *sc_in, int n_jobs, const wm_ksw_job_t *jobs,
                                  const uint8_t *seqs, size_t seqs_bytes, wm_ksw_dev_batch_t **out)
{
	return ksw_prepare_impl(c, sc_in, n_jobs, jobs, seqs, seqs_bytes, 0, out);
}

// the allocation that holds `cap` bases of resident reads: pk words first, then the bitmap
static int reads_alloc(wm_ctx_t *c, size_t cap)
{
	const size_t pkw = cap / 32 + 2, nmw = cap / 64 + 2;
	c->d_reads = 0; c->d_reads_nm = 0; c->reads_cap = 0;
	if (hipMalloc((void**)&c->d_reads, (pkw + nmw) * 8) != hipSuccess) { (void)hipGetLastError(); c->d_reads = 0; return set_err(WM_ENOMEM, "cannot allocate %zu bytes for the resident reads", (pkw + nmw) * 8); }
	c->d_reads_nm = c->d_reads + pkw;
	c->reads_cap = cap;
	return WM_OK;
}
extern "C" int wm_reads_upload(wm_ctx_t *c, const uint8_t *codes, size_t n)
{
	if (!c || (n && !codes)) return set_err(WM_EINVAL, "null argument");
	HIPCHK(hipSetDevice(c->device));
	if (!c->owns_reads) { c->d_reads = 0; c->d_reads_nm = 0; c->reads_cap = 0; }
	if (n + 256 > c->reads_cap) {
		if (c->d_reads) HIPCHK(hipFree(c->d_reads));
		c->d_reads = 0; c->d_reads_nm = 0; c->reads_cap = 0;
		const size_t cap = (n + n / 8 + (1 << 20) + 255) & ~(size_t)255;
		const int rc = reads_alloc(c, cap);
		if (rc) return rc;
	}
	c->owns_reads = true;
	// packed on the host (reads2bit.h: 2 bits per base + 1 ambiguity bit), 0.375 B per base across PCIe and in HBM
	std::vector<uint64_t> pk(wm_pk_words(n)), nm(wm_nm_words(n));
	wm_pack_codes(codes, n, pk.data(), nm.data());
	HIPCHK(hipMemcpy(c->d_reads, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(c->d_reads_nm, nm.data(), nm.size() * 8, hipMemcpyHostToDevice));
	c->reads_bytes = n;
	return WM_OK;
}

// WM_KSW_PMULTI: 1 = only the BLOCK / BLOCK2 classes run on the packed multi-wave kernel (ksw_pmulti_kernel<4,8> / <8,8>); 2 (default:
// +7 % on BASELINE config 2, profiles/r03a_first_run.txt) = the 16-pair register classes run on ksw_pmulti_kernel<4,4> as well
// WM_KSW_PMULTI_LEAN=0: every 16-pair class on the <CLIP, HASN> = <true, true> machine (the code before this switch; A/B)
static bool pmulti_lean() { static const bool v = !(getenv("WM_KSW_PMULTI_LEAN") && atoi(getenv("WM_KSW_PMULTI_LEAN")) == 0); return v; }
static int ksw_pmulti_level() { const char *e = getenv("WM_KSW_PMULTI"); return e ? atoi(e) : 2; }

extern "C" int wm_ksw_dev_run(wm_ctx_t *c, wm_ksw_dev_batch_t *b)
{
	HIPCHK(hipSetDevice(c->device));
	const int n = b->n_jobs;
	if (n == 0) return WM_OK;
	// degenerate jobs get the result of ksw_reset_extz (src/ksw2.h:153-158)
	HIPCHK(hipMemsetAsync(b->d_err, 0, 4, c->stream));
	if (b->d_mail) {            // mailboxes of the chained-workgroup classes: stamps -1, progress -1, STOP "none"; ticket counters 0
		HIPCHK(hipMemsetAsync(b->d_mail, 0xff, b->mail_words * 8, c->stream));
		HIPCHK(hipMemsetAsync(b->d_tickets, 0, 64 * sizeof(int), c->stream));
	}
	if (!b->degenerate.empty()) hipLaunchKernelGGL(ksw_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, b->d_jobs, b->d_res);
	HIPCHK(hipEventRecord(c->ev[0], c->stream));
	int off = 0;
	static const bool trace_k = getenv("WM_TRACE_KSW") != 0;          // per-class timing (serialises the launches)
	auto class_done = [&](int k, double t0) {
		if (!trace_k) return;
		hipStreamSynchronize(c->stream);
		const std::vector<int> &o = b->order[k];
		const wm_ksw_djob_t &big = b->jobs[o[0]];
		fprintf(stderr, "[ksw class %2d] jobs %zu  %.2f ms  largest q=%d t=%d w=%d flag=0x%x n_col=%d\n", k, o.size(), now_ms() - t0, big.qlen, big.tlen, big.w, big.flag, big.n_col);
	};
	// the size classes are independent: spread them over the side streams so that one class's long jobs overlap the others
	int n_nonempty = 0, used_mask = 0, rr = 0;
	for (int k = 0; k < WM_KSW_NCLASS; ++k) n_nonempty += !b->order[k].empty();
	static const int n_side = std::max(0, std::min(4, getenv("WM_SIDE_STREAMS") ? atoi(getenv("WM_SIDE_STREAMS")) : 3));   // contexts x (1 + side streams) should not exceed the hardware queues
	const bool fan = n_nonempty > 1 && !trace_k && n_side > 0;
	if (fan) HIPCHK(hipEventRecord(c->kev[4], c->stream));
	hipStream_t ks = c->stream;
	hipStream_t side[4] = {0, 0, 0, 0};          // the side streams of this call: from the mapper's pool, else the context's own
	// The pool is shared by the contexts, i.e. by concurrent calls: a kernel waits for whatever sits in front of it on its stream. A call of short
	// alignments must never queue behind a 50-ms launch of another call, so the pool is split by the weight of the call (its longest job, in the
	// units of the hub's queues: rows x register pairs): light | heavy | huge calls draw from their own part (WM_SIDE_SPLIT=light,heavy; rest = huge;
	// 0,0 = one pool as before).
	int pool_lo = 0, pool_n = c->n_side_pool, wclass = 0;
	if (c->side_pool && c->n_side_pool >= 6) {
		static const long heavy_units = getenv("WM_KSW_HEAVY_UNITS") ? atol(getenv("WM_KSW_HEAVY_UNITS")) : 8192, huge_units = getenv("WM_KSW_HUGE_UNITS") ? atol(getenv("WM_KSW_HUGE_UNITS")) : 131072;
		static int split_l = -1, split_h = -1;
		if (split_l < 0) { int a, h; side_split(c->n_side_pool, &a, &h); split_h = h; split_l = a; }
		if (split_l > 0 && split_h > 0 && split_l + split_h < c->n_side_pool) {
			long mx = 0;
			for (const wm_ksw_djob_t &d : b->jobs) {
				if (d.klass < 0) continue;
				long w_ = d.w < 0 ? std::max(d.qlen, d.tlen) : d.w, nn = std::min(d.qlen, d.tlen);
				if (nn > w_ + 1) nn = w_ + 1;
				mx = std::max(mx, ((long)d.qlen + d.tlen) * ((nn + 127) / 128 + 1));
			}
			wclass = huge_units > 0 && mx > huge_units ? 2 : heavy_units > 0 && mx > heavy_units ? 1 : 0;
			pool_lo = wclass == 0 ? 0 : wclass == 1 ? split_l : split_l + split_h;
			pool_n = wclass == 0 ? split_l : wclass == 1 ? split_h : c->n_side_pool - split_l - split_h;
		}
	}
	const int n_use = c->side_pool ? std::min(n_side, pool_n) : n_side;
	if (fan && n_use > 0) {
		if (c->side_pool) { const unsigned b0 = c->side_next[wclass].fetch_add((unsigned)n_use); for (int i = 0; i < n_use; ++i) side[i] = c->side_pool[pool_lo + (int)((b0 + (unsigned)i) % (unsigned)pool_n)]; }
		else for (int i = 0; i < n_use; ++i) { if (!c->kstream[i]) HIPCHK(hipStreamCreateWithFlags(&c->kstream[i], hipStreamNonBlocking)); side[i] = c->kstream[i]; }
	}
	auto next_stream = [&]() {
		if (!fan || n_use <= 0) return;
		const int si = rr++ % n_use;
		ks = side[si];
		if (!(used_mask >> si & 1)) { hipStreamWaitEvent(ks, c->kev[4], 0); used_mask |= 1 << si; }
	};
	// WM_KSW_CLASS_EVENTS=0: no start / stop events around the classes' launches (wm_mapper_kernel_stats then reports nothing; A/B of what the events cost)
	static const bool class_events = !(getenv("WM_KSW_CLASS_EVENTS") && atoi(getenv("WM_KSW_CLASS_EVENTS")) == 0);
	int offs[WM_KSW_NCLASS + 1];
	offs[0] = 0;
	for (int k = 0; k < WM_KSW_NCLASS; ++k) offs[k + 1] = offs[k] + (int)b->order[k].size();
	// launch order: the classes with the longest single jobs first
	int lorder[WM_KSW_NCLASS], nl = 0;
	lorder[nl++] = WM_KSW_GENERIC; lorder[nl++] = WM_KSW_BLOCK3; lorder[nl++] = WM_KSW_BLOCK2; lorder[nl++] = WM_KSW_BLOCK;
	for (int k = WM_KSW_NCLASS - 1; k >= WM_KSW_STRIPE; --k) lorder[nl++] = k;          // (the chained-workgroup classes, then the stripe classes)
	for (int k = WM_KSW_BLOCK - 1; k >= 0; --k) lorder[nl++] = k;
	for (int li = 0; li < nl; ++li) {
		const int k = lorder[li];
		const int nk = (int)b->order[k].size();
		if (nk == 0) continue;
		off = offs[k];
		next_stream();
		const double tk0 = trace_k ? now_ms() : 0;
		if (class_events) hipEventRecord(c->cev[k][0], ks);
		struct Done { decltype(class_done) &f; int k; double t; hipEvent_t e; hipStream_t s; bool on; ~Done() { if (on) hipEventRecord(e, s); f(k, t); } } done_guard{ class_done, k, tk0, c->cev[k][1], ks, class_events };
		if (k == WM_KSW_BLOCK || k == WM_KSW_BLOCK2 || k == WM_KSW_BLOCK3) {
			const size_t fixed = (size_t)WM_KSW_BLK_PUB * 4;
			if (k == WM_KSW_BLOCK || k == WM_KSW_BLOCK2) {
				const int seq_cap = 64 * 1024;
				if (k == WM_KSW_BLOCK) {
					const size_t lds = (size_t)wmk::ksw_pmulti_lds<4, 8>::INTS * 4 + seq_cap;
					HIPCHK(hipFuncSetAttribute((const void*)ksw_pmulti_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
					hipLaunchKernelGGL((ksw_pmulti_kernel<4, 8>), dim3(nk), dim3(64 * 8), lds, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res, seq_cap);
				} else {
					const size_t lds = (size_t)wmk::ksw_pmulti_lds<8, 8>::INTS * 4 + seq_cap;
					HIPCHK(hipFuncSetAttribute((const void*)ksw_pmulti_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
					hipLaunchKernelGGL((ksw_pmulti_kernel<8, 8>), dim3(nk), dim3(64 * 8), lds, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res, seq_cap);
				}
			} else {
				const size_t lds = fixed + WM_KSW_BLK3_SEQ_LDS;
				HIPCHK(hipFuncSetAttribute((const void*)ksw_block_kernel<WM_KSW_BLK2_K, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
				hipLaunchKernelGGL((ksw_block_kernel<WM_KSW_BLK2_K, 0>), dim3(nk), dim3(64 * WM_KSW_BLK_NWV), lds, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res, (int)WM_KSW_BLK3_SEQ_LDS, (int*)b->d_b3state, b->d_b3off);
			}
			continue;
		}
		if (k >= WM_KSW_CHAIN) {
			const int kc = k - WM_KSW_CHAIN, var = kc & 3;
			for (int e = 1; e >= 0; --e) {            // (the exact-maximum jobs first: the longer rows)
				const int nw = (int)b->cmap[kc][e].size();
				if (!nw) continue;
				if (wm_ksw_chain_bp[kc >> 2] == 2) {
					if (e) launch_chain<2, true>(var, nw, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res, b->d_cmap[kc][e], b->d_tickets + 2 * kc + e, b->d_mail);
					else launch_chain<2, false>(var, nw, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res, b->d_cmap[kc][e], b->d_tickets + 2 * kc + e, b->d_mail);
				} else {
					if (e) launch_chain<4, true>(var, nw, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res, b->d_cmap[kc][e], b->d_tickets + 2 * kc + e, b->d_mail);
					else launch_chain<4, false>(var, nw, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res, b->d_cmap[kc][e], b->d_tickets + 2 * kc + e, b->d_mail);
				}
			}
			continue;
		}
		if (k >= WM_KSW_STRIPE) {
			const int var = (k - WM_KSW_STRIPE) & 3;
			// WM_KSW_DUMP=<dir> (diagnostics): the jobs of a stripe launch and their operands go to <dir>/stripe_<pid>_<n>.bin before the launch and the
			// file is removed when the whole call has come back — what is left after a hang is the launch that hung, replayable on the emulator
			// (tools/replay_stripe_dump.py)
			static const char *dump_dir = getenv("WM_KSW_DUMP");
			if (dump_dir) {
				static std::atomic<int> seq(0);
				std::vector<uint8_t> hs(b->slab_bytes + 64);
				hipStreamSynchronize(c->stream);
				hipMemcpy(hs.data(), b->d_seqs, b->slab_bytes, hipMemcpyDeviceToHost);
				char path[512];
				snprintf(path, sizeof(path), "%s/stripe_%d_%d.bin", dump_dir, (int)getpid(), seq++);
				if (FILE *fp = fopen(path, "wb")) {
					const int32_t hdr[4] = { k, nk, (int32_t)sizeof(wm_ksw_djob_t), (int32_t)sizeof(wm_ksw_score_t) };
					fwrite(hdr, 4, 4, fp); fwrite(&b->sc, sizeof(b->sc), 1, fp);
					for (int j : b->order[k]) {
						const wm_ksw_djob_t &d = b->jobs[j];
						fwrite(&d, sizeof(d), 1, fp); fwrite(hs.data() + d.q_off, 1, d.qlen, fp); fwrite(hs.data() + d.t_off, 1, d.tlen, fp);
					}
					fclose(fp);
					b->dumped.push_back(path);
				}
			}
			switch ((k - WM_KSW_STRIPE) >> 2) {
			case 0: launch_stripe<2, 4>(var, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
			case 1: launch_stripe<2, 8>(var, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
			case 2: launch_stripe<4, 8>(var, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
			case 4: launch_stripe<1, 16>(var, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
			case 5: launch_stripe<2, 16>(var, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
			default: launch_stripe<8, 8>(var, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
			}
			continue;
		}
		if (k == WM_KSW_GENERIC) {
			hipLaunchKernelGGL(ksw_generic_kernel, dim3(nk), dim3(64), 0, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_gscratch, b->d_goff, b->d_res);
			continue;
		}
		switch (k & ~7) {
		case WM_KSW_P4: launch_dpp<4>(k & 7, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
		case WM_KSW_P8: launch_dpp<8>(k & 7, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
		default:
			if (ksw_pmulti_level() >= 2) {      // 4 wavefronts per alignment for the 16-pair classes too (shorter batch tails)
				const int seq_cap = 32 * 1024;
				const size_t lds = (size_t)wmk::ksw_pmulti_lds<4, 4>::INTS * 4 + seq_cap;
				auto go = [&](auto kern) -> int {
					HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
					hipLaunchKernelGGL(kern, dim3(nk), dim3(64 * 4), lds, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res, seq_cap);
					return WM_OK;
				};
				int rc_;
				if (!pmulti_lean()) rc_ = go(ksw_pmulti_kernel<4, 4, true, true>);
				else if ((k & 2) && (k & 1)) rc_ = go(ksw_pmulti_kernel<4, 4, true, true>);
				else if (k & 2) rc_ = go(ksw_pmulti_kernel<4, 4, true, false>);
				else if (k & 1) rc_ = go(ksw_pmulti_kernel<4, 4, false, true>);
				else rc_ = go(ksw_pmulti_kernel<4, 4, false, false>);
				if (rc_) return rc_;
			} else
				launch_dpp<16>(k & 7, nk, ks, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res);
			break;
		}
	}
	for (int si = 0; si < 4; ++si)
		if (used_mask >> si & 1) { HIPCHK(hipEventRecord(c->kev[si], side[si])); HIPCHK(hipStreamWaitEvent(c->stream, c->kev[si], 0)); }
	HIPCHK(hipEventRecord(c->ev[1], c->stream));
	if (getenv("WM_KSW_COOP_BT") && atoi(getenv("WM_KSW_COOP_BT")) > 0)
		hipLaunchKernelGGL(ksw_backtrack_coop_kernel, dim3(n), dim3(64), 0, c->stream, n, b->d_jobs, b->d_tb, b->d_res, b->d_cig, b->d_err);
	else
		hipLaunchKernelGGL(ksw_backtrack_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, b->d_jobs, b->d_tb, b->d_res, b->d_cig, b->d_err);
	hipLaunchKernelGGL(ksw_scan_kernel, dim3(1), dim3(1024), 0, c->stream, n, b->d_res, b->d_off, b->d_total);
	hipLaunchKernelGGL(ksw_gather_kernel, dim3(n), dim3(64), 0, c->stream, b->d_jobs, b->d_res, b->d_off, b->d_cig, b->d_pool, (uint32_t)b->pool_cap);
	if (b->d_zd) hipLaunchKernelGGL(ksw_zdwalk_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, b->sc_in, b->d_jobs, b->d_res, b->d_off, b->d_pool, b->d_seqs, b->d_zd);
	HIPCHK(hipEventRecord(c->ev[2], c->stream));
	HIPCHK(hipGetLastError());
	int *h_small = c->pin_small ? c->pin_small : &b->h_err;            // [0] error flag, [1] total ops
	uint32_t *h_total = c->pin_small ? (uint32_t*)(c->pin_small + 1) : &b->total_ops;
	if (c->pin_small) HIPCHK(hipMemcpyAsync(h_small, b->d_err, 8, hipMemcpyDeviceToHost, c->stream));      // (flag and total are neighbours on both sides)
	else { HIPCHK(hipMemcpyAsync(h_small, b->d_err, 4, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipMemcpyAsync(h_total, b->d_total, 4, hipMemcpyDeviceToHost, c->stream)); }
	HIPCHK(ctx_sync(c));
	b->h_err = *h_small; b->total_ops = *h_total;
	for (const std::string &f : b->dumped) unlink(f.c_str());
	b->dumped.clear();
	HIPCHK(hipEventElapsedTime(&b->dp_ms, c->ev[0], c->ev[1]));
	HIPCHK(hipEventElapsedTime(&b->bt_ms, c->ev[1], c->ev[2]));
	c->last_ms = b->dp_ms + b->bt_ms;
	for (int k = 0; k < WM_KSW_NCLASS && class_events; ++k)
		if (!b->order[k].empty()) {
			float ms = 0;
			if (hipEventElapsedTime(&ms, c->cev[k][0], c->cev[k][1]) == hipSuccess) {
				c->k_ms[k] += ms; c->k_cells[k] += b->class_cells[k]; c->k_launches[k] += 1;
				float t0 = 0;
				if (hipEvent_t base = device_base_event(c->device)) if (hipEventElapsedTime(&t0, base, c->cev[k][0]) == hipSuccess) {
					std::lock_guard<std::mutex> lk(c->iv_mu);
					if (c->k_iv[k].size() > 100000) c->k_iv[k].erase(c->k_iv[k].begin(), c->k_iv[k].begin() + 50000);      // (a file of any size: keep the recent past)
					c->k_iv[k].push_back(std::make_pair(t0, t0 + ms));
				}
			}
		}
	if (b->h_err == 2) return set_err(WM_EINTERNAL, "a stripe-pipelined / chained-workgroup alignment kernel gave up waiting for a neighbouring wavefront (watchdog, ksw_stripe_kernel.h, ksw_chain_kernel.h); WM_KSW_CHAIN=0 / WM_KSW_STRIPE=0 route around them");
	if (b->h_err) return set_err(WM_EINTERNAL, "cigar slot overflow in backtrack");
	return WM_OK;
}

extern "C" int wm_ksw_dev_fetch(wm_ctx_t *c, wm_ksw_dev_batch_t *b, wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used)
try {
	HIPCHK(hipSetDevice(c->device));
	const int n = b->n_jobs;
	if (cigar_used) *cigar_used = b->total_ops;
	if (n == 0) return WM_OK;
	const size_t off_at = (size_t)((uint8_t*)b->d_off - (uint8_t*)b->d_res);
	UBuf<uint8_t> ro(off_at + (size_t)n * 4, c);
	const wm_ksw_dres_t *res = (const wm_ksw_dres_t*)ro.data();
	const uint32_t *off = (const uint32_t*)(ro.data() + off_at);
	if (b->total_ops > cigar_cap) return set_err(WM_ENOMEM, "cigar_pool too small: need %u ops", b->total_ops);
	HIPCHK(hipMemcpyAsync(ro.data(), b->d_res, off_at + (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
	if (b->total_ops) HIPCHK(hipMemcpyAsync(cigar_pool, b->d_pool, (size_t)b->total_ops * 4, hipMemcpyDeviceToHost, c->stream));      // (the mapper hands a pinned buffer: GpuOps)
	HIPCHK(ctx_sync(c));
	WM_SITE("ksw.results");
	wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
		wm_ksw_result_t &o = results[i];
		const wm_ksw_dres_t &r = res[i];
		o.max = r.max; o.zdropped = r.zdropped; o.max_q = r.max_q; o.max_t = r.max_t; o.mqe = r.mqe; o.mqe_t = r.mqe_t;
		o.mte = r.mte; o.mte_q = r.mte_q; o.score = r.score; o.reach_end = r.reach_end; o.n_cigar = r.n_cigar; o.cig_off = off[i];
	});
	return WM_OK;
}
catch (const std::bad_alloc &) { return set_err(WM_ENOMEM, "out of host memory"); }

extern "C" int wm_ksw_dev_stats(const wm_ksw_dev_batch_t *b, uint64_t *cells, uint64_t *tb_bytes, float *dp_ms, float *bt_ms)
{
	if (cells) *cells = b->cells;
	if (tb_bytes) *tb_bytes = b->tb_bytes;
	if (dp_ms) *dp_ms = b->dp_ms;
	if (bt_ms) *bt_ms = b->bt_ms;
	return WM_OK;
}

extern "C" void wm_ksw_dev_free(wm_ctx_t *c, wm_ksw_dev_batch_t *b)
{
	if (!b) return;
	if (c && c->arena_used >= b->arena_mark) c->arena_used = b->arena_mark;   // batches are released in LIFO order
	delete b;
}

// the z-drop scans of a run batch (jobs flagged WM_KSW_F_ZDWALK; the others: no drop)
static int ksw_fetch_zd(wm_ctx_t *c, wm_ksw_dev_batch_t *b, wm_zd_t *out)
{
	const int n = b->n_jobs;
	if (!b->d_zd) { for (int i = 0; i < n; ++i) out[i] = wm_zd_t{ 0, -1, -1, -1, -1 }; return WM_OK; }
	if (n == 0) return WM_OK;
	HIPCHK(hipSetDevice(c->device));
	UBuf<wm_zd_t> z(n, c);
	HIPCHK(hipMemcpyAsync(z.data(), b->d_zd, (size_t)n * sizeof(wm_zd_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(ctx_sync(c));
	memcpy(out, z.data(), (size_t)n * sizeof(wm_zd_t));
	return WM_OK;
}

static int ksw_batch_impl(wm_ctx_t *c, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes, const wm_ksw_pos_t *pos,
                          wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used, wm_zd_t *zd = 0)
{
	// process in chunks whose traceback (and operands) fit the arena
	if (!c) return set_err(WM_EINVAL, "null context");
	size_t used = 0;
	int i0 = 0;
	float kms = 0;
	const size_t budget = (size_t)(c->arena_bytes * 0.8);
	while (i0 < n_jobs || (n_jobs == 0 && i0 == 0)) {
		int i1 = i0;
		size_t need = 0;
		while (i1 < n_jobs) {
			const int ql = pos ? pos[i1].qlen : jobs[i1].qlen, tl = pos ? pos[i1].tlen : jobs[i1].tlen, w = pos ? pos[i1].w : jobs[i1].w;
			size_t t = 128;
			if (ql > 0 && tl > 0) t = ((size_t)ql + tl) * ((size_t)wm_ksw_ncol(ql, tl, w) + 10) + 512;
			if (i1 > i0 && need + t > budget) break;
			need += t; ++i1;
		}
		wm_ksw_dev_batch_t *b = 0;
		const double ta = now_ms();
		int rc = ksw_prepare_impl(c, sc, i1 - i0, pos ? 0 : jobs + i0, seqs, seqs_bytes, pos ? pos + i0 : 0, &b);
		if (rc) return rc;
		const double tb_ = now_ms();
		rc = wm_ksw_dev_run(c, b);
		const double tc = now_ms();
		size_t u = 0;
		if (!rc) rc = wm_ksw_dev_fetch(c, b, results + i0, cigar_pool + used, cigar_cap - used, &u);
		if (!rc && zd) rc = ksw_fetch_zd(c, b, zd + i0);
		c->acc_cells += b->cells; c->t_prep += tb_ - ta; c->t_run += tc - tb_; c->t_fetch += now_ms() - tc; kms += b->dp_ms + b->bt_ms;
		wm_ksw_dev_free(c, b);
		if (rc) { if (cigar_used) *cigar_used = used + u; return rc; }
		for (int i = i0; i < i1; ++i) results[i].cig_off += (uint32_t)used;
		used += u;
		i0 = i1;
		if (n_jobs == 0) break;
	}
	if (cigar_used) *cigar_used = used;
	c->last_ms = kms;
	return WM_OK;
}

extern "C" int wm_ksw_batch(wm_ctx_t *c, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes,
                            wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used)
{
	return ksw_batch_impl(c, sc, n_jobs, jobs, seqs, seqs_bytes, 0, results, cigar_pool, cigar_cap, cigar_used);
}

extern "C" int wm_ksw_batch_pos(wm_ctx_t *c, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_pos_t *jobs,
                                wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used)
{
	if (n_jobs > 0 && !jobs) return set_err(WM_EINVAL, "null jobs");
	return ksw_batch_impl(c, sc, n_jobs, 0, 0, 0, jobs, results, cigar_pool, cigar_cap, cigar_used);
}

extern "C" int wm_ksw_batch_pos_zd(wm_ctx_t *c, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_pos_t *jobs,
                                   wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used, wm_zd_t *zd)
{
	if (n_jobs > 0 && (!jobs || !zd)) return set_err(WM_EINVAL, "null jobs / zd");
	return ksw_batch_impl(c, sc, n_jobs, 0, 0, 0, jobs, results, cigar_pool, cigar_cap, cigar_used, zd);
}

// diagnostic: the per-phase cycle table of the stripe-pipelined kernel (a library built with WM_KERNEL_DEFINES="WM_STRIPE_TIMING=1"; ksw_stripe_kernel.h).
// out[0..6] = shader-clock cycles summed over all wavefronts: scan / epoch set-up / cells / waiting for the left message / bookkeeping / waiting for the
// right neighbour's progress / publishing; out[7] rows, out[8] epochs, out[9] cycles inside the kernel, out[10] wavefronts. reset: clear afterwards.
extern "C" int wm_debug_stripe_timing(uint64_t *out16, int reset)
{
#ifdef WM_STRIPE_TIMING
	unsigned long long h[16];
	if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wm_stripe_timing), sizeof(h)) != hipSuccess) return set_err(WM_EINTERNAL, "hipMemcpyFromSymbol: %s", hipGetErrorString(hipGetLastError()));
	for (int i = 0; i < 16; ++i) out16[i] = h[i];
	if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_wm_stripe_timing), h, sizeof(h)) != hipSuccess) return set_err(WM_EINTERNAL, "hipMemcpyToSymbol failed"); }
	return WM_OK;
#else
	(void)out16; (void)reset;
	return set_err(WM_EINVAL, "this library was not built with WM_STRIPE_TIMING");
#endif
}

extern "C" int wm_ksw_extd2(wm_ctx_t *c, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                            int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, wm_ksw_result_t *ez, uint32_t **cigar_out)
{
	if (m != 5) return set_err(WM_EINVAL, "only the 5-letter alphabet of src/align.c:9 is supported");
	wm_ksw_score_t sc = { mat[0], mat[1], mat[24], q, e, q2, e2 };
	wm_ksw_job_t jb = { 0, (uint32_t)(qlen > 0 ? qlen : 0), qlen, tlen, w, zdrop, end_bonus, flag };
	std::vector<uint8_t> seqs((qlen > 0 ? qlen : 0) + (tlen > 0 ? tlen : 0) + 1);
	if (qlen > 0) memcpy(seqs.data(), query, qlen);
	if (tlen > 0) memcpy(seqs.data() + (qlen > 0 ? qlen : 0), target, tlen);
	const size_t cap = (size_t)(qlen > 0 ? qlen : 0) + (tlen > 0 ? tlen : 0) + 4;
	uint32_t *cig = (uint32_t*)malloc(cap * 4);
	size_t used = 0;
	int rc = wm_ksw_batch(c, &sc, 1, &jb, seqs.data(), seqs.size(), ez, cig, cap, &used);
	if (rc) { free(cig); *cigar_out = 0; return rc; }
	*cigar_out = cig;
	return WM_OK;
}

// ---- ksw_exts2_sse (src/ksw2.h:63-64): the splice-aware extension as a batch. One wavefront per alignment, state in a global scratch
// slab (ksw_exts2_kernel.h). Every alignment of splice mode goes through here (GpuOpsCtx::exts2_batch).
__global__ __launch_bounds__(64) void ksw_exts2_kernel(wm_ksw_score_t sc, int noncan, int junc_bonus, const wm_ksw_djob_t *__restrict__ jobs,
                                                        const uint8_t *__restrict__ seqs, const uint8_t *__restrict__ junc, uint8_t *__restrict__ tb,
                                                        uint8_t *scratch, const uint64_t *__restrict__ scratch_off, wm_ksw_dres_t *__restrict__ res)
{
	const int j = blockIdx.x;
	const wm_ksw_djob_t jb = jobs[j];
	const uint64_t T = ((uint64_t)jb.tlen + 15) / 16 * 16;
	signed char *mem = (signed char*)(scratch + scratch_off[j]);
	wmk::ksw_dp_exts2<true>(sc, noncan, junc_bonus, jb, seqs, junc, tb, mem, (int*)(mem + 8 * T), res + j);
}
__global__ __launch_bounds__(64) void ksw_exts2_backtrack_kernel(wm_ksw_score_t sc, int n, const wm_ksw_djob_t *__restrict__ jobs, const uint8_t *__restrict__ tb,
                                                                  wm_ksw_dres_t *__restrict__ res, uint32_t *__restrict__ cig_scratch, int *__restrict__ err)
{
	const int j = blockIdx.x * 64 + threadIdx.x;
	if (j >= n) return;
	wm_ksw_dres_t r = res[j];
	int nc = 0;
	if (r.bt_i == KSW_BT_WATCHDOG) atomicMax(err, 2);
	if (r.bt_i >= 0) {
		nc = wmk::ksw_exts2_backtrack_thread(sc, jobs[j], tb, r.bt_i, r.bt_j, cig_scratch + jobs[j].cig_off, jobs[j].cig_cap);
		if (nc < 0) { atomicMax(err, 1); nc = 0; }
	}
	res[j].n_cigar = nc;
}

extern "C" int wm_ksw_exts2_batch(wm_ctx_t *c, const wm_ksw_score_t *sc_in, int noncan, int junc_bonus, int n_jobs, const wm_ksw_job_t *jobs,
                                  const uint8_t *seqs, size_t seqs_bytes, const uint8_t *junc,
                                  wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used)
try {
	if (!c) return set_err(WM_EINVAL, "null context");
	if (cigar_used) *cigar_used = 0;
	if (n_jobs < 0) return set_err(WM_EINVAL, "n_jobs < 0");
	if (n_jobs == 0) return WM_OK;
	if (!sc_in || !jobs || !seqs || !results) return set_err(WM_EINVAL, "null argument");
	const wm_ksw_score_t sc = *sc_in;
	if (sc.e <= 0 || sc.q2 <= sc.q + sc.e) return set_err(WM_EINVAL, "ksw_exts2 needs e > 0 and q2 > q + e (src/ksw2_exts2_sse.c:66)");
	if (-(int)sc.mismatch > 2 * (sc.q + sc.e)) return set_err(WM_EINVAL, "mismatch penalty above 2 (q + e): the reference returns without aligning (src/ksw2_exts2_sse.c:84)");
	if (noncan < -127 || noncan > 127 || junc_bonus < -127 || junc_bonus > 127) return set_err(WM_EINVAL, "noncan / junc_bonus are int8 in the reference");
	HIPCHK(hipSetDevice(c->device));
	ArenaMark mark(c);
	std::vector<wm_ksw_djob_t> dj(n_jobs);
	std::vector<uint64_t> soff(n_jobs);
	uint64_t tb_off = 0, cig_off = 0, sc_off = 0;
	for (int i = 0; i < n_jobs; ++i) {
		const wm_ksw_job_t &jb = jobs[i];
		if (jb.qlen <= 0 || jb.tlen <= 0) return set_err(WM_EINVAL, "job %d: empty operand", i);
		if ((uint64_t)jb.q_off + jb.qlen > seqs_bytes || (uint64_t)jb.t_off + jb.tlen > seqs_bytes) return set_err(WM_EINVAL, "job %d: operands outside seqs", i);
		if (jb.flag & (0x01 | 0x04 | 0x10)) return set_err(WM_EINVAL, "job %d: KSW_EZ_SCORE_ONLY / GENERIC_SC / APPROX_DROP are not supported", i);
		wm_ksw_djob_t &d = dj[i];
		memset(&d, 0, sizeof(d));
		d.q_off = jb.q_off; d.t_off = jb.t_off; d.qlen = jb.qlen; d.tlen = jb.tlen; d.w = -1; d.zdrop = jb.zdrop; d.end_bonus = 0; d.flag = jb.flag;
		d.n_col = (((jb.qlen < jb.tlen ? jb.qlen : jb.tlen) + 15) / 16 + 1) * 16;               // src/ksw2_exts2_sse.c:78
		d.tb_off = tb_off;
		tb_off += ((uint64_t)(jb.qlen + jb.tlen - 1) * d.n_col + 15) & ~(uint64_t)15;
		d.cig_off = (uint32_t)cig_off; d.cig_cap = jb.qlen + jb.tlen + 2;
		cig_off += d.cig_cap;
		soff[i] = sc_off;
		sc_off += (12 * (((uint64_t)jb.tlen + 15) / 16 * 16) + 256 + 255) & ~(uint64_t)255;
	}
	if (cig_off >= ((uint64_t)1 << 32)) return set_err(WM_ENOMEM, "ksw_exts2: batch too large (split it)");
	const size_t nj = (size_t)n_jobs;
	wm_ksw_djob_t *d_jobs = (wm_ksw_djob_t*)arena_take(c, nj * sizeof(wm_ksw_djob_t));
	uint64_t *d_soff = (uint64_t*)arena_take(c, nj * 8);
	wm_ksw_dres_t *d_res = (wm_ksw_dres_t*)arena_take(c, nj * sizeof(wm_ksw_dres_t));
	uint32_t *d_off = (uint32_t*)arena_take(c, nj * 4 + 64);
	uint32_t *d_total = (uint32_t*)arena_take(c, 64);
	int *d_err = (int*)arena_take(c, 64);
	uint8_t *d_seqs = (uint8_t*)arena_take(c, seqs_bytes + 64);
	uint8_t *d_junc = junc ? (uint8_t*)arena_take(c, seqs_bytes + 64) : 0;
	uint32_t *d_cig = (uint32_t*)arena_take(c, (cig_off + 16) * 4);
	uint32_t *d_pool = (uint32_t*)arena_take(c, (cig_off + 16) * 4);
	uint8_t *d_scratch = (uint8_t*)arena_take(c, sc_off + 256);
	uint8_t *d_tb = (uint8_t*)arena_take(c, tb_off + 64);
	if (!d_jobs || !d_soff || !d_res || !d_off || !d_total || !d_err || !d_seqs || (junc && !d_junc) || !d_cig || !d_pool || !d_scratch || !d_tb)
		return set_err(WM_ENOMEM, "ksw_exts2: batch does not fit the arena (%llu traceback bytes); split it", (unsigned long long)tb_off);
	HIPCHK(hipMemcpyAsync(d_jobs, dj.data(), nj * sizeof(wm_ksw_djob_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d_soff, soff.data(), nj * 8, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d_seqs, seqs, seqs_bytes, hipMemcpyHostToDevice, c->stream));
	if (junc) HIPCHK(hipMemcpyAsync(d_junc, junc, seqs_bytes, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemsetAsync(d_err, 0, 4, c->stream));
	hipLaunchKernelGGL(ksw_exts2_kernel, dim3(n_jobs), dim3(64), 0, c->stream, sc, noncan, junc_bonus, d_jobs, d_seqs, d_junc, d_tb, d_scratch, d_soff, d_res);
	hipLaunchKernelGGL(ksw_exts2_backtrack_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, c->stream, sc, n_jobs, d_jobs, d_tb, d_res, d_cig, d_err);
	hipLaunchKernelGGL(ksw_scan_kernel, dim3(1), dim3(1024), 0, c->stream, n_jobs, d_res, d_off, d_total);
	hipLaunchKernelGGL(ksw_gather_kernel, dim3(n_jobs), dim3(64), 0, c->stream, d_jobs, d_res, d_off, d_cig, d_pool, (uint32_t)(cig_off + 16));
	HIPCHK(hipGetLastError());
	UBuf<wm_ksw_dres_t> res(nj, c);
	UBuf<uint32_t> off(nj, c);
	UBuf<uint32_t> small(4, c);
	HIPCHK(hipMemcpyAsync(res.data(), d_res, nj * sizeof(wm_ksw_dres_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(off.data(), d_off, nj * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(small.data(), d_total, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(small.data() + 1, d_err, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(ctx_sync(c));
	if (small[1]) return set_err(WM_EINTERNAL, "cigar slot overflow in backtrack");
	const uint32_t total = small[0];
	if (cigar_used) *cigar_used = total;
	if (total > cigar_cap) return set_err(WM_ENOMEM, "cigar_pool too small: need %u ops", total);
	if (total) {
		if (!cigar_pool) return set_err(WM_EINVAL, "null cigar_pool");
		HIPCHK(hipMemcpyAsync(cigar_pool, d_pool, (size_t)total * 4, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(ctx_sync(c));
	}
	for (int i = 0; i < n_jobs; ++i) {
		wm_ksw_result_t &o = results[i];
		const wm_ksw_dres_t &r = res[i];
		o.max = r.max; o.zdropped = r.zdropped; o.max_q = r.max_q; o.max_t = r.max_t; o.mqe = r.mqe; o.mqe_t = r.mqe_t;
		o.mte = r.mte; o.mte_q = r.mte_q; o.score = r.score; o.reach_end = r.reach_end; o.n_cigar = r.n_cigar; o.cig_off = off[i];
	}
	return WM_OK;
}
catch (const std::bad_alloc &) { return set_err(WM_ENOMEM, "out of host memory"); }

// ======================================================================================================
// sketch / seed / chain kernels and their batched entry points
// ======================================================================================================
#include "host/wm_core.cpp"
#include "host/wm_index.cpp"
#include "host/wm_seqio.cpp"
#include "host/wm_hit.cpp"
#include "host/wm_chain.cpp"
#include "host/wm_ops.cpp"
#include "host/wm_align.cpp"
#include "host/wm_mapper.cpp"
#include "host/wm_format.cpp"
#include "host/wm_kmers.cpp"
#include "host/wm_pipeline.cpp"

struct wm_index_s { wm::Index ix; };

// (seqs: the call's staged bytes; rpk / rnm: the resident packed reads, for jobs whose seq_off carries WM_RD_PACKED_BIT — reads2bit.h)
__global__ __launch_bounds__(64) void sketch_kernel(wm_sketch_params_t P, const wm_sketch_job_t *jobs, int n_jobs, const uint8_t *seqs, const uint64_t *rpk, const uint64_t *rnm,
                                                     const uint8_t *bloom, wm128_t *out, int *counts)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	double *ring_o = (double*)smem;
	uint32_t *ring_y = (uint32_t*)(smem + (size_t)P.w * 64 * sizeof(double));
	wmk::sketch_wave(P, jobs, n_jobs, blockIdx.x, seqs, rpk, rnm, bloom, ring_o, ring_y, out, counts);
}

// one wavefront per sequence (sketch_coop, odd k): order[] lists the jobs longest first; so / sx / sy / sl = per-position scratch
__global__ __launch_bounds__(64) void sketch_coop_kernel(wm_sketch_params_t P, const wm_sketch_job_t *jobs, const int *order, const uint8_t *seqs, const uint64_t *rpk, const uint64_t *rnm,
                                                          const uint8_t *bloom, double *so, uint64_t *sx, uint32_t *sy, uint32_t *sl, wm128_t *out, int *counts, int long_thr,
                                                          uint8_t *hc, uint32_t *he)
{
	WM_SETPRIO(2);
	const int j = order[blockIdx.x];
	const wm_sketch_job_t jb = jobs[j];
	if (long_thr > 0 && jb.len >= long_thr) return;           // sketched chunk by chunk (sketch_long_* kernels)
	wmk::sketch_coop(P, jb, seqs, rpk, rnm, bloom, so + jb.scratch_off, sx + jb.scratch_off, sy + jb.scratch_off, sl + jb.scratch_off, out, counts + j,
	                 hc ? hc + jb.scratch_off : 0, he ? he + jb.scratch_off : 0);        // (P.hpc: the job's compacted sequence)
}

// ---- long sequences (contigs of the reference at index time, query contigs, stage-2 passes of very long reads): one wavefront per CHUNK of the
// sequence instead of one per sequence (sketch_kernel.h: sketch_p1_range / sketch_find_sync / sketch_p2_range explain why that is exact). Four launches:
// phase 1 of every chunk | the first sync position of every chunk | phase 2 from sync to sync into chunk-local slots | per job: the chunks' minimizers
// concatenated into the job's output slot. sketch_coop_kernel leaves these jobs alone (long_thr).
struct wm_sk_chunk_t { int32_t job, begin, end, first; uint64_t out_off; int32_t cap, pad; };
__global__ __launch_bounds__(64) void sketch_long_p1_kernel(wm_sketch_params_t P, const wm_sketch_job_t *jobs, const wm_sk_chunk_t *chunks, const uint8_t *seqs, const uint64_t *rpk,
                                                             const uint64_t *rnm, const uint8_t *bloom, double *so, uint64_t *sx, uint32_t *sy, uint32_t *sl)
{
	const wm_sk_chunk_t ch = chunks[blockIdx.x];
	const wm_sketch_job_t jb = jobs[ch.job];
	wmk::sketch_p1_range(P, (long long)jb.seq_off, jb.len, seqs, rpk, rnm, bloom, so + jb.scratch_off, sx + jb.scratch_off, sy + jb.scratch_off, sl + jb.scratch_off, ch.begin, ch.end);
}
__global__ __launch_bounds__(64) void sketch_long_sync_kernel(wm_sketch_params_t P, const wm_sketch_job_t *jobs, const wm_sk_chunk_t *chunks, const double *so, int *sync)
{
	const wm_sk_chunk_t ch = chunks[blockIdx.x];
	const int t = ch.first ? 0 : wmk::sketch_find_sync(P.w, so + jobs[ch.job].scratch_off, ch.begin, ch.end);
	if (threadIdx.x == 0) sync[blockIdx.x] = t;
}
__global__ __launch_bounds__(64) void sketch_long_p2_kernel(wm_sketch_params_t P, const wm_sketch_job_t *jobs, const wm_sk_chunk_t *chunks, int n_chunks, const double *so, const uint64_t *sx,
                                                             const uint32_t *sy, const uint32_t *sl, const int *sync, wm128_t *cout, int *ccount)
{
	WM_SETPRIO(2);
	const int b = blockIdx.x;
	const wm_sk_chunk_t ch = chunks[b];
	int n = 0;
	if (ch.first || sync[b] >= 0) {                           // (a chunk without a sync position is covered by the wavefront of the chunk before it)
		int t_stop = -1;
		for (int d = b + 1; d < n_chunks && chunks[d].job == ch.job && t_stop < 0; ++d) t_stop = sync[d];
		const wm_sketch_job_t jb = jobs[ch.job];
		n = wmk::sketch_p2_range(P, jb.len, so + jb.scratch_off, sx + jb.scratch_off, sy + jb.scratch_off, sl + jb.scratch_off, ch.first ? 0 : sync[b], !ch.first, t_stop, cout + ch.out_off, ch.cap);
	}
	if (threadIdx.x == 0) ccount[b] = n;
}
// long_jobs[3 i ..]: job, its first chunk, its chunk count
__global__ __launch_bounds__(64) void sketch_long_gather_kernel(const wm_sketch_job_t *jobs, const int *long_jobs, const wm_sk_chunk_t *chunks, const wm128_t *cout, const int *ccount,
                                                                 wm128_t *out, int *counts)
{
	const int j = long_jobs[3 * blockIdx.x], c0 = long_jobs[3 * blockIdx.x + 1], nc = long_jobs[3 * blockIdx.x + 2];
	const wm_sketch_job_t jb = jobs[j];
	long long total = 0;
	bool over = false;
	for (int c = c0; c < c0 + nc; ++c) {
		const int m = ccount[c];
		if (m > chunks[c].cap) over = true;
		if (!over && total + m <= jb.cap) {
			const uint64_t *src = (const uint64_t*)(cout + chunks[c].out_off);
			uint64_t *dst = (uint64_t*)(out + jb.out_off + total);
			for (int i = threadIdx.x; i < 2 * m; i += 64) dst[i] = src[i];
		}
		total += m;
	}
	if (threadIdx.x == 0) counts[j] = over || total > jb.cap ? jb.cap + 1 : (int)total;      // (more than the slot holds: the caller repeats the job with a full-size slot)
}

__global__ __launch_bounds__(64) void seed_kernel(wm_index_view_t ix, const wm_seed_job_t *jobs, const wm128_t *mini, wm128_t *anchors,
                                                   int *occ_scratch, const uint64_t *occ_off, wm_seed_res_t *res)
{
	const int j = blockIdx.x;
	wmk::seed_wave(ix, jobs[j], mini, anchors, occ_scratch + occ_off[j], res + j);
}

// anchors of a seed batch as keys (x) / values (y) for the device sort; and back, with a per-job flag "two anchors share a key" (their
// relative order is then decided by the reference's unstable radix sort, src/ksort.h:101-151: such jobs are re-sorted on the host)
__global__ __launch_bounds__(256) void seed_split_kernel(const wm128_t *__restrict__ a, uint64_t n, uint64_t *__restrict__ k, uint64_t *__restrict__ v)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { const wm128_t e = a[i]; k[i] = e.x; v[i] = e.y; }
}
__global__ __launch_bounds__(64) void seed_merge_kernel(const uint32_t *__restrict__ beg, const uint32_t *__restrict__ end, const uint64_t *__restrict__ k, const uint64_t *__restrict__ v,
                                                         wm128_t *__restrict__ out, int *__restrict__ tie)
{
	const int j = blockIdx.x;
	const uint32_t b = beg[j], e = end[j];
	bool t = false;
	for (uint32_t i = b + threadIdx.x; i < e; i += 64) {
		wm128_t o; o.x = k[i]; o.y = v[i];
		out[i] = o;
		t |= i > b && k[i - 1] == o.x;
	}
	if (__any(t) && threadIdx.x == 0) tie[j] = 1;
}

// chain DP fill: one wave per anchor set, LDS window of W anchors (28 B each: x, y, f, p, t); f and p go to fpvt
__global__ __launch_bounds__(64) void chain_kernel(const wm_chain_job_t *jobs, const int *order, const wm128_t *anchors, int *fpvt, int W)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int j = order[blockIdx.x];
	const wm_chain_job_t jb = jobs[j];
	uint64_t *sx = (uint64_t*)smem, *sy = sx + W;
	int *sf = (int*)(sy + W), *sp = sf + W, *st = sp + W;
	int *gf = fpvt + jb.a_off * 4, *gp = gf + jb.n, *gt = gp + 2 * (size_t)jb.n;      // slab per job: f | p | v (host) | t
	wmk::chain_wave(jb, anchors, W, sx, sy, sf, sp, st, gf, gp, gt);
}

// large anchor sets: NWV waves cooperate on one job (chain_block); LDS = 28 B * W window + publish area
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void chain_kernel_block(const wm_chain_job_t *jobs, const int *order, const wm128_t *anchors, int *fpvt, int W)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int j = order[blockIdx.x];
	const wm_chain_job_t jb = jobs[j];
	uint64_t *sx = (uint64_t*)smem, *sy = sx + W;
	int *sf = (int*)(sy + W), *sp = sf + W, *st = sp + W, *pub = st + W;
	int *gf = fpvt + jb.a_off * 4, *gp = gf + jb.n, *gt = gp + 2 * (size_t)jb.n;
	wmk::chain_block(jb, anchors, NWV, W, sx, sy, sf, sp, st, pub, gf, gp, gt);
}

extern "C" float wm_last_aux_ms(const wm_ctx_t *c) { return c ? c->aux_ms : 0.f; }

extern "C" int wm_index_build(const char *fasta, const char *kmer_file, int k, int w, int n_threads, wm_index_t **out)
{
	return wm_index_build_flag(fasta, kmer_file, k, w, 0, n_threads, out);
}
extern "C" int wm_index_build_flag(const char *fasta, const char *kmer_file, int k, int w, int idx_flag, int n_threads, wm_index_t **out)
{
	*out = 0;
	if (idx_flag & ~1) return set_err(WM_EINVAL, "index flag %d: only MM_I_HPC (1) is known here", idx_flag);
	wm::IdxOpt io; io.k = k; io.w = w; io.flag = idx_flag;
	wm::MapOpt mo; std::string err;
	if (wm::check_opt(io, mo, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	wm_index_t *h = new wm_index_t();
	if (wm::index_build_from_fasta(io, fasta, kmer_file ? kmer_file : "", n_threads, h->ix, err) < 0) { delete h; return set_err(WM_EINVAL, "%s", err.c_str()); }
	*out = h;
	return WM_OK;
}
extern "C" void wm_index_destroy(wm_index_t *h) { delete h; }

// the reference's index file ("MMI\2", winnowmap -d; src/index.c:515-608): interchangeable in both directions
extern "C" int wm_index_save(const wm_index_t *h, const char *path)
{
	std::string err;
	if (!h) return set_err(WM_EINVAL, "null index");
	if (wm::index_save_mmi(h->ix, path, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	return WM_OK;
}
extern "C" int wm_index_load(const char *path, const char *kmer_file, wm_index_t **out)
{
	*out = 0;
	std::string err;
	wm_index_t *h = new wm_index_t();
	if (wm::index_load_mmi(path, kmer_file ? kmer_file : "", h->ix, err) < 0) { delete h; return set_err(WM_EINVAL, "%s", err.c_str()); }
	*out = h;
	return WM_OK;
}

namespace wm { int write_repetitive_kmers(const std::vector<std::string> &seqs, int k, double distinct, const std::string &out_path, uint64_t *n_out, std::string &err); }
// the -W list of a FASTA file (what `meryl count k=15` + `meryl print greater-than distinct=0.9998` would give)
extern "C" int wm_write_repetitive_kmers(const char *fasta, int k, double distinct, const char *out_path, uint64_t *n_out)
{
	std::vector<std::string> names, seqs; std::string err;
	if (wm::read_fastx(fasta, names, seqs, 0, 0, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	if (wm::write_repetitive_kmers(seqs, k, distinct, out_path, n_out, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	return WM_OK;
}

// ---- the -W list on the device: canonical k-mers of the whole reference -> radix sort -> run lengths -> count histogram ->
//      meryl's threshold (ext/meryl/src/meryl/merylOp-nextMer.C:103-115) -> the k-mers above it. Same output as wm_write_repetitive_kmers.
// codes: all contigs back to back, one code-4 byte between them; key of position i = canonical k-mer ending there, or `inv` (= 4^k, sorts last)
__global__ __launch_bounds__(256) void kmer_key_kernel(const uint8_t *__restrict__ codes, uint64_t n, int k, uint64_t inv, uint64_t *__restrict__ keys)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint64_t fw = 0, rc = 0;
	bool ok = i + 1 >= (uint64_t)k;
	if (ok)
		for (int j = 0; j < k; ++j) {              // base j steps back: digit j of the forward k-mer, digit k-1-j of the reverse complement
			const uint64_t c = codes[i - j];
			ok &= c < 4;
			fw |= (c & 3) << (2 * j);
			rc |= ((c & 3) ^ 3) << (2 * (k - 1 - j));
		}
	keys[i] = ok ? (fw < rc ? fw : rc) : inv;
}
__global__ __launch_bounds__(256) void kmer_hist_kernel(const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ n_runs, uint64_t inv,
                                                        unsigned long long *__restrict__ hist, uint32_t hcap)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= *n_runs || uniq[i] == inv) return;
	const uint32_t c = cnt[i] < hcap - 1 ? cnt[i] : hcap - 1;
	atomicAdd(&hist[c], 1ULL);
}
__global__ __launch_bounds__(256) void kmer_select_kernel(const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ n_runs, uint64_t inv,
                                                          uint32_t thr, uint64_t *__restrict__ out_key, uint32_t *__restrict__ out_cnt, unsigned long long *__restrict__ n_sel, uint64_t cap)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= *n_runs || uniq[i] == inv || cnt[i] <= thr) return;
	const unsigned long long o = atomicAdd(n_sel, 1ULL);
	if (o < cap) { out_key[o] = uniq[i]; out_cnt[o] = cnt[i]; }
}

extern "C" int wm_write_repetitive_kmers_gpu(wm_ctx_t *c, const char *fasta, int k, double distinct, const char *out_path, uint64_t *n_out, double *stats)
{
	if (!c) return set_err(WM_EINVAL, "null context");
	if (k < 1 || k > 28) return set_err(WM_EINVAL, "k out of range");
	HIPCHK(hipSetDevice(c->device));
	const double t0 = now_ms();
	std::vector<std::string> names, seqs; std::string err;
	if (wm::read_fastx(fasta, names, seqs, 0, 0, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	uint64_t n = 0;
	for (const std::string &sq : seqs) n += sq.size() + 1;
	if (n >= ((uint64_t)1 << 32)) return set_err(WM_EINVAL, "references of 4 Gbase and more must be counted in parts");   // (run counts and indices are 32 bits)
	std::unique_ptr<uint8_t[]> codes(new uint8_t[n + 1]);
	{
		std::vector<uint64_t> off(seqs.size());
		uint64_t o = 0;
		for (size_t i = 0; i < seqs.size(); ++i) { off[i] = o; o += seqs[i].size() + 1; }
		wm::parallel_for(16, seqs.size(), [&](size_t i) { uint8_t *d = codes.get() + off[i]; const std::string &sq = seqs[i]; for (size_t j = 0; j < sq.size(); ++j) d[j] = wm::nt4_table[(uint8_t)sq[j]]; d[sq.size()] = 4; });
	}
	const double t1 = now_ms();
	const uint64_t inv = 1ULL << 2 * k;
	const uint32_t hcap = 1u << 20;
	uint8_t *d_codes = 0; uint64_t *d_keys = 0, *d_sorted = 0, *d_uniq = 0, *d_okey = 0; uint32_t *d_cnt = 0, *d_nruns = 0, *d_ocnt = 0; unsigned long long *d_hist = 0, *d_nsel = 0; void *d_tmp = 0;
	auto cleanup = [&]() { hipFree(d_codes); hipFree(d_keys); hipFree(d_sorted); hipFree(d_uniq); hipFree(d_cnt); hipFree(d_nruns); hipFree(d_hist); hipFree(d_nsel); hipFree(d_tmp); hipFree(d_okey); hipFree(d_ocnt); };
#define KM_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return set_err(WM_ENOMEM, "%s failed: %s", #x, hipGetErrorString(e_)); } } while (0)
	KM_CHK(hipMalloc((void**)&d_codes, n + 8));
	KM_CHK(hipMalloc((void**)&d_keys, n * 8 + 8)); KM_CHK(hipMalloc((void**)&d_sorted, n * 8 + 8));
	KM_CHK(hipMalloc((void**)&d_nruns, 8)); KM_CHK(hipMalloc((void**)&d_hist, (size_t)hcap * 8)); KM_CHK(hipMalloc((void**)&d_nsel, 8));
	KM_CHK(hipMemcpy(d_codes, codes.get(), n, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(kmer_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_codes, n, k, inv, d_keys);
	size_t tmp_bytes = 0;
	KM_CHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys, d_sorted, (size_t)n, 0, 2 * k + 1, c->stream));
	KM_CHK(hipMalloc(&d_tmp, tmp_bytes + 8));
	KM_CHK(rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_keys, d_sorted, (size_t)n, 0, 2 * k + 1, c->stream));
	KM_CHK(hipStreamSynchronize(c->stream));
	hipFree(d_tmp); d_tmp = 0; hipFree(d_keys); d_keys = 0;
	KM_CHK(hipMalloc((void**)&d_uniq, n * 8 + 8)); KM_CHK(hipMalloc((void**)&d_cnt, n * 4 + 8));
	tmp_bytes = 0;
	KM_CHK(rocprim::run_length_encode(nullptr, tmp_bytes, d_sorted, (unsigned int)n, d_uniq, d_cnt, d_nruns, c->stream));
	KM_CHK(hipMalloc(&d_tmp, tmp_bytes + 8));
	KM_CHK(rocprim::run_length_encode(d_tmp, tmp_bytes, d_sorted, (unsigned int)n, d_uniq, d_cnt, d_nruns, c->stream));
	KM_CHK(hipMemsetAsync(d_hist, 0, (size_t)hcap * 8, c->stream));
	uint32_t n_runs = 0;
	KM_CHK(hipMemcpyAsync(&n_runs, d_nruns, 4, hipMemcpyDeviceToHost, c->stream));
	KM_CHK(hipStreamSynchronize(c->stream));
	hipLaunchKernelGGL(kmer_hist_kernel, dim3((n_runs + 255) / 256 + 1), dim3(256), 0, c->stream, d_uniq, d_cnt, d_nruns, inv, d_hist, hcap);
	std::vector<unsigned long long> hist(hcap);
	KM_CHK(hipMemcpyAsync(hist.data(), d_hist, (size_t)hcap * 8, hipMemcpyDeviceToHost, c->stream));
	KM_CHK(hipStreamSynchronize(c->stream));
	// threshold exactly as merylOp-nextMer.C:103-115 (and host/wm_kmers.cpp): truncated target, only count values that occur
	uint64_t n_distinct = 0, cum = 0, thr = 0, n_sel = 0;
	for (uint32_t cc = 1; cc < hcap; ++cc) n_distinct += hist[cc];
	const uint64_t target = (uint64_t)(distinct * (double)n_distinct);
	bool found = false;
	for (uint32_t cc = 1; cc < hcap; ++cc) {
		if (hist[cc] == 0) continue;
		cum += hist[cc];
		if (cum >= target) { thr = cc; found = true; break; }
	}
	if (found && thr == hcap - 1) { cleanup(); return set_err(WM_EINTERNAL, "count threshold beyond the device histogram (%u): use wm_write_repetitive_kmers", hcap); }
	if (!found) thr = 0;
	for (uint32_t cc = (uint32_t)thr + 1; cc < hcap; ++cc) n_sel += hist[cc];
	KM_CHK(hipMalloc((void**)&d_okey, n_sel * 8 + 8)); KM_CHK(hipMalloc((void**)&d_ocnt, n_sel * 4 + 8));
	KM_CHK(hipMemsetAsync(d_nsel, 0, 8, c->stream));
	hipLaunchKernelGGL(kmer_select_kernel, dim3((n_runs + 255) / 256 + 1), dim3(256), 0, c->stream, d_uniq, d_cnt, d_nruns, inv, (uint32_t)thr, d_okey, d_ocnt, d_nsel, n_sel);
	std::vector<uint64_t> okey(n_sel); std::vector<uint32_t> ocnt(n_sel);
	if (n_sel) { KM_CHK(hipMemcpyAsync(okey.data(), d_okey, n_sel * 8, hipMemcpyDeviceToHost, c->stream)); KM_CHK(hipMemcpyAsync(ocnt.data(), d_ocnt, n_sel * 4, hipMemcpyDeviceToHost, c->stream)); }
	KM_CHK(hipStreamSynchronize(c->stream));
	KM_CHK(hipGetLastError());
#undef KM_CHK
	cleanup();
	const double t2 = now_ms();
	std::vector<uint32_t> ord(n_sel);
	for (uint32_t i = 0; i < n_sel; ++i) ord[i] = i;
	std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return okey[a] < okey[b]; });      // (the appends arrive in any order)
	FILE *fp = fopen(out_path, "w");
	if (!fp) return set_err(WM_EINVAL, "cannot write %s", out_path);
	char buf[40];
	for (uint32_t oi : ord) {
		const uint64_t km = okey[oi];
		for (int i = 0; i < k; ++i) buf[i] = "ACGT"[km >> (2 * (k - 1 - i)) & 3];
		buf[k] = 0;
		fprintf(fp, "%s\t%u\n", buf, ocnt[oi]);
	}
	if (fclose(fp) != 0) return set_err(WM_EINVAL, "write error on %s", out_path);
	if (n_out) *n_out = n_sel;
	if (stats) { stats[0] = (t1 - t0) * 1e-3; stats[1] = (t2 - t1) * 1e-3; stats[2] = (now_ms() - t2) * 1e-3; stats[3] = (double)n_distinct; }
	return WM_OK;
}

// flat-array export / import of an index: what travels over RCCL when one rank builds and the others receive
// (sizes first, then the caller allocates and calls again with the buffers).
extern "C" int wm_index_export(const wm_index_t *h, uint64_t *sizes9, uint32_t *S, uint64_t *hkey, uint64_t *hval, uint64_t *P, uint8_t *bloom, uint64_t *seq_meta, char *names)
{
	const wm::Index &ix = h->ix;
	size_t name_bytes = 0;
	for (auto &r : ix.seq) name_bytes += r.name.size() + 1;
	sizes9[0] = ix.S.size(); sizes9[1] = ix.hkey.size(); sizes9[2] = ix.P.size(); sizes9[3] = ix.bloom.bits.size(); sizes9[4] = ix.seq.size(); sizes9[5] = name_bytes;
	sizes9[6] = (uint64_t)ix.k | (uint64_t)ix.w << 8 | (uint64_t)ix.hbits << 16 | (uint64_t)ix.flag << 24;
	sizes9[7] = ix.bloom.table_bits; sizes9[8] = (uint64_t)ix.bloom.salt[0] | (uint64_t)ix.bloom.salt[1] << 32;
	if (!S) return WM_OK;
	memcpy(S, ix.S.data(), ix.S.size() * 4); memcpy(hkey, ix.hkey.data(), ix.hkey.size() * 8); memcpy(hval, ix.hval.data(), ix.hval.size() * 8);
	if (!ix.P.empty()) memcpy(P, ix.P.data(), ix.P.size() * 8);
	memcpy(bloom, ix.bloom.bits.data(), ix.bloom.bits.size());
	char *q = names;
	for (size_t i = 0; i < ix.seq.size(); ++i) { seq_meta[2 * i] = ix.seq[i].offset; seq_meta[2 * i + 1] = ix.seq[i].len; memcpy(q, ix.seq[i].name.c_str(), ix.seq[i].name.size() + 1); q += ix.seq[i].name.size() + 1; }
	return WM_OK;
}
extern "C" int wm_index_import(const uint64_t *sizes9, const uint32_t *S, const uint64_t *hkey, const uint64_t *hval, const uint64_t *P, const uint8_t *bloom,
                               const uint64_t *seq_meta, const char *names, wm_index_t **out)
{
	wm_index_t *h = new wm_index_t();
	wm::Index &ix = h->ix;
	ix.k = (int)(sizes9[6] & 0xff); ix.w = (int)(sizes9[6] >> 8 & 0xff); ix.hbits = (int)(sizes9[6] >> 16 & 0xff); ix.flag = (int)(sizes9[6] >> 24);
	ix.S.assign(S, S + sizes9[0]); ix.hkey.assign(hkey, hkey + sizes9[1]); ix.hval.assign(hval, hval + sizes9[1]); ix.P.assign(P, P + sizes9[2]);
	ix.bloom.table_bits = sizes9[7]; ix.bloom.salt[0] = (uint32_t)sizes9[8]; ix.bloom.salt[1] = (uint32_t)(sizes9[8] >> 32); ix.bloom.bits.assign(bloom, bloom + sizes9[3]);
	const char *q = names;
	ix.total_len = 0;
	for (uint64_t i = 0; i < sizes9[4]; ++i) { wm::RefSeq r; r.offset = seq_meta[2 * i]; r.len = (uint32_t)seq_meta[2 * i + 1]; r.name = q; q += r.name.size() + 1; ix.total_len += r.len; ix.seq.push_back(r); }
	ix.n_minimizers = ix.P.size();
	ix.n_keys = 0;
	for (uint64_t kk : ix.hkey) ix.n_keys += kk != ~0ULL;
	ix.scan_n_runs();
	*out = h;
	return WM_OK;
}
extern "C" int wm_index_n_seq(const wm_index_t *h) { return (int)h->ix.seq.size(); }
extern "C" const char *wm_index_seq_name(const wm_index_t *h, int rid) { return h->ix.seq[rid].name.c_str(); }
extern "C" int wm_index_seq_len(const wm_index_t *h, int rid) { return (int)h->ix.seq[rid].len; }
extern "C" uint64_t wm_index_n_minimizers(const wm_index_t *h) { return h->ix.n_minimizers; }
extern "C" const uint64_t *wm_index_get(const wm_index_t *h, uint64_t minier, int *n) { return h->ix.get(minier, n); }

extern "C" int wm_index_upload(wm_ctx_t *c, const wm_index_t *h)
{
	if (!c || !h) return set_err(WM_EINVAL, "null argument");
	HIPCHK(hipSetDevice(c->device));
	const wm::Index &ix = h->ix;
	if (ix.bloom.table_bits >= ((uint64_t)1 << 32)) return set_err(WM_EINVAL, "bloom table of %llu bits not supported on device", (unsigned long long)ix.bloom.table_bits);
	// (new arrays first; the old index goes only when the new one is complete — as wm_index_upload_dev)
	uint64_t *n_hkey = 0, *n_hval = 0, *n_P = 0; uint8_t *n_bloom = 0; uint32_t *n_S = 0;
	auto drop = [&]() { hipFree(n_hkey); hipFree(n_hval); hipFree(n_P); hipFree(n_bloom); hipFree(n_S); (void)hipGetLastError(); };
#define UP_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { drop(); return set_err(e_ == hipErrorOutOfMemory ? WM_ENOMEM : WM_ENODEV, "%s failed: %s", #x, hipGetErrorString(e_)); } } while (0)
	UP_CHK(hipMalloc((void**)&n_hkey, ix.hkey.size() * 8 + 8));
	UP_CHK(hipMalloc((void**)&n_hval, ix.hval.size() * 8 + 8));
	UP_CHK(hipMalloc((void**)&n_P, ix.P.size() * 8 + 8));
	UP_CHK(hipMalloc((void**)&n_bloom, ix.bloom.bits.size() + 8));
	UP_CHK(hipMalloc((void**)&n_S, ix.S.size() * 4 + 8));
	UP_CHK(hipMemcpy(n_S, ix.S.data(), ix.S.size() * 4, hipMemcpyHostToDevice));
	UP_CHK(hipMemcpy(n_hkey, ix.hkey.data(), ix.hkey.size() * 8, hipMemcpyHostToDevice));
	UP_CHK(hipMemcpy(n_hval, ix.hval.data(), ix.hval.size() * 8, hipMemcpyHostToDevice));
	if (!ix.P.empty()) UP_CHK(hipMemcpy(n_P, ix.P.data(), ix.P.size() * 8, hipMemcpyHostToDevice));
	UP_CHK(hipMemcpy(n_bloom, ix.bloom.bits.data(), ix.bloom.bits.size(), hipMemcpyHostToDevice));
#undef UP_CHK
	if (c->have_index && c->owns_index) { hipFree(c->d_hkey); hipFree(c->d_hval); hipFree(c->d_P); hipFree(c->d_bloom); hipFree(c->d_S); }
	if (!c->have_index && c->owns_filter && c->d_bloom) hipFree(c->d_bloom);
	c->owns_filter = false;
	c->d_hkey = n_hkey; c->d_hval = n_hval; c->d_P = n_P; c->d_bloom = n_bloom; c->d_S = n_S;
	c->seq_off.clear(); c->seq_len.clear();
	for (const wm::RefSeq &r : ix.seq) { c->seq_off.push_back(r.offset); c->seq_len.push_back(r.len); }
	c->hbits = ix.hbits;
	c->skp.w = ix.w; c->skp.k = ix.k; c->skp.table_bits = (uint32_t)ix.bloom.table_bits; c->skp.salt0 = ix.bloom.salt[0]; c->skp.salt1 = ix.bloom.salt[1]; c->skp.hpc = ix.flag & 1;
	c->have_index = true; c->owns_index = true;
	return WM_OK;
}

// The index from DEVICE memory: the five flat arrays of `h` (S, hkey, hval, P, bloom bits; sizes and contig table from h) are taken from device pointers on
// device src_device — the receive buffers of an RCCL broadcast (winnowmap_amd/dist.py), or another context's copy (wm_index_upload_peer): one
// hipMemcpyPeer per array, device to device over xGMI, no host staging. SURVEY §8(b): "wm_index_bcast(rank, nranks) next to wm_index_upload".
extern "C" int wm_index_upload_dev(wm_ctx_t *c, const wm_index_t *h, const void *d_S, const void *d_hkey, const void *d_hval, const void *d_P, const void *d_bloom, int src_device)
{
	if (!c || !h || !d_S || !d_hkey || !d_hval || !d_bloom) return set_err(WM_EINVAL, "null argument");
	HIPCHK(hipSetDevice(c->device));
	const wm::Index &ix = h->ix;
	if (!ix.P.empty() && !d_P) return set_err(WM_EINVAL, "null argument");
	if (ix.bloom.table_bits >= ((uint64_t)1 << 32)) return set_err(WM_EINVAL, "bloom table of %llu bits not supported on device", (unsigned long long)ix.bloom.table_bits);
	if (src_device != c->device) {
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, c->device, src_device) == hipSuccess && can) { const hipError_t e = hipDeviceEnablePeerAccess(src_device, 0); if (e != hipSuccess) (void)hipGetLastError(); }   // (already enabled is fine; hipMemcpyPeer stages through the host otherwise)
	}
	// new arrays first, the context's old index is released only when all of them are there and filled (a failed allocation used to leave
	// the context without any index, and the arrays already allocated leaked)
	uint64_t *n_hkey = 0, *n_hval = 0, *n_P = 0; uint8_t *n_bloom = 0; uint32_t *n_S = 0;
	auto drop = [&]() { hipFree(n_hkey); hipFree(n_hval); hipFree(n_P); hipFree(n_bloom); hipFree(n_S); (void)hipGetLastError(); };
#define UP_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { drop(); return set_err(e_ == hipErrorOutOfMemory ? WM_ENOMEM : WM_ENODEV, "%s failed: %s", #x, hipGetErrorString(e_)); } } while (0)
	UP_CHK(hipMalloc((void**)&n_hkey, ix.hkey.size() * 8 + 8));
	UP_CHK(hipMalloc((void**)&n_hval, ix.hval.size() * 8 + 8));
	UP_CHK(hipMalloc((void**)&n_P, ix.P.size() * 8 + 8));
	UP_CHK(hipMalloc((void**)&n_bloom, ix.bloom.bits.size() + 8));
	UP_CHK(hipMalloc((void**)&n_S, ix.S.size() * 4 + 8));
	UP_CHK(hipMemcpyPeer(n_S, c->device, d_S, src_device, ix.S.size() * 4));
	UP_CHK(hipMemcpyPeer(n_hkey, c->device, d_hkey, src_device, ix.hkey.size() * 8));
	UP_CHK(hipMemcpyPeer(n_hval, c->device, d_hval, src_device, ix.hval.size() * 8));
	if (!ix.P.empty()) UP_CHK(hipMemcpyPeer(n_P, c->device, d_P, src_device, ix.P.size() * 8));
	UP_CHK(hipMemcpyPeer(n_bloom, c->device, d_bloom, src_device, ix.bloom.bits.size()));
	UP_CHK(hipDeviceSynchronize());
#undef UP_CHK
	if (c->have_index && c->owns_index) { hipFree(c->d_hkey); hipFree(c->d_hval); hipFree(c->d_P); hipFree(c->d_bloom); hipFree(c->d_S); }
	if (!c->have_index && c->owns_filter && c->d_bloom) hipFree(c->d_bloom);
	c->owns_filter = false;
	c->d_hkey = n_hkey; c->d_hval = n_hval; c->d_P = n_P; c->d_bloom = n_bloom; c->d_S = n_S;
	c->seq_off.clear(); c->seq_len.clear();
	for (const wm::RefSeq &r : ix.seq) { c->seq_off.push_back(r.offset); c->seq_len.push_back(r.len); }
	c->hbits = ix.hbits;
	c->skp.w = ix.w; c->skp.k = ix.k; c->skp.table_bits = (uint32_t)ix.bloom.table_bits; c->skp.salt0 = ix.bloom.salt[0]; c->skp.salt1 = ix.bloom.salt[1]; c->skp.hpc = ix.flag & 1;
	c->have_index = true; c->owns_index = true;
	return WM_OK;
}
// the index of context `src` (wm_index_upload / _dev of the same wm_index_t) copied into context `dst`, which may live on another GPU of the node: the
// one-process form of the index broadcast (a C host that drives N GPUs builds once, uploads once and hands the arrays on over xGMI)
extern "C" int wm_index_upload_peer(wm_ctx_t *dst, const wm_index_t *h, const wm_ctx_t *src)
{
	if (!dst || !h || !src) return set_err(WM_EINVAL, "null argument");
	if (!src->have_index) return set_err(WM_EINVAL, "the source context holds no index");
	if (dst == src) return set_err(WM_EINVAL, "source and destination are the same context");
	if (src->hbits != h->ix.hbits || src->seq_len.size() != h->ix.seq.size()) return set_err(WM_EINVAL, "the source context holds a different index");
	return wm_index_upload_dev(dst, h, src->d_S, src->d_hkey, src->d_hval, src->d_P, src->d_bloom, src->device);
}

static int sketch_batch_impl(wm_ctx_t *c, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len, const uint8_t *resident,
                             wm128_t *out, size_t out_cap, uint64_t *out_off, int32_t *counts);

// ---- the index TABLE on the device (worker_post + mm_idx_post, src/index.c:200-254): (key, position) records -> P (positions grouped by key, ascending) and
// the open-addressing table hkey / hval. Two stable LSD radix sorts (by position, then by key) give the reference's order inside a bucket (src/index.c:213,
// radix_sort_128x by x then the run's positions in y order); run-length encoding gives the distinct keys and their counts; the table layout is the canonical one
// of host/wm_index.cpp (keys enter in (home slot, key) order), whose linear probing is a prefix maximum: slot_j = j + max_{i <= j}(home_i - i).
__global__ __launch_bounds__(256) void idx_split_kernel(const wm128_t *__restrict__ a, uint64_t n, uint64_t *__restrict__ x, uint64_t *__restrict__ y)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { x[i] = a[i].x >> 8; y[i] = a[i].y; }
}
__global__ __launch_bounds__(256) void idx_home_kernel(const uint64_t *__restrict__ uniq, uint32_t nk, int hbits, uint32_t *__restrict__ home, uint32_t *__restrict__ idx)
{
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j < nk) { home[j] = (uint32_t)((uniq[j] * 0x9E3779B97F4A7C15ULL) >> (64 - hbits)); idx[j] = j; }       // Index::slot_of
}
__global__ __launch_bounds__(256) void idx_rel_kernel(const uint32_t *__restrict__ home, uint32_t nk, long long *__restrict__ t)
{
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j < nk) t[j] = (long long)home[j] - (long long)j;
}
__global__ __launch_bounds__(256) void idx_place_kernel(const long long *__restrict__ m, const uint32_t *__restrict__ idx, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ first,
                                                         const uint32_t *__restrict__ cnt, uint32_t nk, uint64_t size, uint64_t *__restrict__ hkey, uint64_t *__restrict__ hval,
                                                         uint32_t *__restrict__ n_over)
{
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j >= nk) return;
	const uint64_t p = (uint64_t)((long long)j + m[j]);
	if (p < size) { const uint32_t g = idx[j]; hkey[p] = uniq[g]; hval[p] = (uint64_t)first[g] << 32 | cnt[g]; }
	else atomicAdd(n_over, 1u);              // (runs past the last slot: the tail of the (home, key) order — placed by the host, wrapping around)
}

// returns WM_OK, 1 = not applicable here (too large for the arena or for 32-bit P offsets: the host builds the table), < 0 = error
static int index_table_on_device(wm_ctx_t *c, wm::Index &ix, const std::vector<wm::m128> &all, double *t_dev_s)
{
	const uint64_t n = all.size();
	if (n == 0 || n >= ((uint64_t)1 << 32)) return 1;                     // (P is indexed with 32 bits in hval: as the host build)
	ArenaMark mark(c);
	const double t0 = now_ms();
	wm128_t *d_a = (wm128_t*)arena_take(c, n * 16);
	uint64_t *d_x = (uint64_t*)arena_take(c, n * 8), *d_y = (uint64_t*)arena_take(c, n * 8), *d_x2 = (uint64_t*)arena_take(c, n * 8), *d_y2 = (uint64_t*)arena_take(c, n * 8);
	uint64_t *d_uniq = (uint64_t*)arena_take(c, n * 8);
	uint32_t *d_cnt = (uint32_t*)arena_take(c, n * 4), *d_first = (uint32_t*)arena_take(c, n * 4), *d_small = (uint32_t*)arena_take(c, 64);
	if (!d_a || !d_x || !d_y || !d_x2 || !d_y2 || !d_uniq || !d_cnt || !d_first || !d_small) return 1;            // does not fit the arena: the host builds the table
#define IX_CHK(call) do { if ((call) != hipSuccess) return set_err(WM_EINTERNAL, "device index table: %s", #call); } while (0)
	IX_CHK(hipMemcpyAsync(d_a, all.data(), n * 16, hipMemcpyHostToDevice, c->stream));
	hipLaunchKernelGGL(idx_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_a, n, d_x, d_y);
	size_t tmp_bytes = 0, need = 0;
	IX_CHK(rocprim::radix_sort_pairs(nullptr, need, d_y, d_y2, d_x, d_x2, (size_t)n, 0, 64, c->stream)); tmp_bytes = need;
	IX_CHK(rocprim::radix_sort_pairs(nullptr, need, d_x2, d_x, d_y2, d_y, (size_t)n, 0, 56, c->stream)); tmp_bytes = std::max(tmp_bytes, need);
	IX_CHK(rocprim::run_length_encode(nullptr, need, d_x, (unsigned int)n, d_uniq, d_cnt, d_small, c->stream)); tmp_bytes = std::max(tmp_bytes, need);
	IX_CHK(rocprim::exclusive_scan(nullptr, need, d_cnt, d_first, 0u, (size_t)n, rocprim::plus<uint32_t>(), c->stream)); tmp_bytes = std::max(tmp_bytes, need);
	void *d_tmp = arena_take(c, tmp_bytes + 256);
	if (!d_tmp) return 1;
	need = tmp_bytes;
	IX_CHK(rocprim::radix_sort_pairs(d_tmp, need, d_y, d_y2, d_x, d_x2, (size_t)n, 0, 64, c->stream));            // by position
	need = tmp_bytes;
	IX_CHK(rocprim::radix_sort_pairs(d_tmp, need, d_x2, d_x, d_y2, d_y, (size_t)n, 0, 56, c->stream));            // then, stable, by key: (key, position) order
	need = tmp_bytes;
	IX_CHK(rocprim::run_length_encode(d_tmp, need, d_x, (unsigned int)n, d_uniq, d_cnt, d_small, c->stream));
	uint32_t nk = 0;
	IX_CHK(hipMemcpyAsync(&nk, d_small, 4, hipMemcpyDeviceToHost, c->stream));
	IX_CHK(ctx_sync(c));
	need = tmp_bytes;
	IX_CHK(rocprim::exclusive_scan(d_tmp, need, d_cnt, d_first, 0u, (size_t)nk, rocprim::plus<uint32_t>(), c->stream));
	ix.scan_n_runs();
	ix.n_minimizers = n; ix.n_keys = nk;
	ix.hbits = 4;
	while (((uint64_t)1 << ix.hbits) < 2 * (uint64_t)nk + 2) ++ix.hbits;
	const uint64_t size = (uint64_t)1 << ix.hbits;
	uint64_t *d_hkey = (uint64_t*)arena_take(c, size * 8), *d_hval = (uint64_t*)arena_take(c, size * 8);
	uint32_t *d_home = (uint32_t*)arena_take(c, (size_t)nk * 4), *d_idx = (uint32_t*)arena_take(c, (size_t)nk * 4), *d_home2 = (uint32_t*)arena_take(c, (size_t)nk * 4), *d_idx2 = (uint32_t*)arena_take(c, (size_t)nk * 4);
	long long *d_t = (long long*)arena_take(c, (size_t)nk * 8), *d_m = (long long*)arena_take(c, (size_t)nk * 8);
	size_t need2 = 0, tmp2 = 0;
	if (!d_hkey || !d_hval || !d_home || !d_idx || !d_home2 || !d_idx2 || !d_t || !d_m) return 1;
	IX_CHK(rocprim::radix_sort_pairs(nullptr, need2, d_home, d_home2, d_idx, d_idx2, (size_t)nk, 0, ix.hbits, c->stream)); tmp2 = need2;
	IX_CHK(rocprim::inclusive_scan(nullptr, need2, d_t, d_m, (size_t)nk, rocprim::maximum<long long>(), c->stream)); tmp2 = std::max(tmp2, need2);
	void *d_tmp2 = tmp2 <= tmp_bytes ? d_tmp : arena_take(c, tmp2 + 256);
	if (!d_tmp2) return 1;
	const unsigned gk = (unsigned)((nk + 255) / 256);
	IX_CHK(hipMemsetAsync(d_hkey, 0xff, size * 8, c->stream));
	IX_CHK(hipMemsetAsync(d_hval, 0, size * 8, c->stream));
	IX_CHK(hipMemsetAsync(d_small, 0, 8, c->stream));
	hipLaunchKernelGGL(idx_home_kernel, dim3(gk), dim3(256), 0, c->stream, d_uniq, nk, ix.hbits, d_home, d_idx);
	need2 = tmp2;
	IX_CHK(rocprim::radix_sort_pairs(d_tmp2, need2, d_home, d_home2, d_idx, d_idx2, (size_t)nk, 0, ix.hbits, c->stream));   // stable: ties stay in key order
	hipLaunchKernelGGL(idx_rel_kernel, dim3(gk), dim3(256), 0, c->stream, d_home2, nk, d_t);
	need2 = tmp2;
	IX_CHK(rocprim::inclusive_scan(d_tmp2, need2, d_t, d_m, (size_t)nk, rocprim::maximum<long long>(), c->stream));
	hipLaunchKernelGGL(idx_place_kernel, dim3(gk), dim3(256), 0, c->stream, d_m, d_idx2, d_uniq, d_first, d_cnt, nk, size, d_hkey, d_hval, d_small);
	ix.hkey.resize(size); ix.hval.resize(size); ix.P.resize(n);
	uint32_t n_over = 0;
	IX_CHK(hipMemcpyAsync(ix.hkey.data(), d_hkey, size * 8, hipMemcpyDeviceToHost, c->stream));
	IX_CHK(hipMemcpyAsync(ix.hval.data(), d_hval, size * 8, hipMemcpyDeviceToHost, c->stream));
	IX_CHK(hipMemcpyAsync(ix.P.data(), d_y, n * 8, hipMemcpyDeviceToHost, c->stream));
	IX_CHK(hipMemcpyAsync(&n_over, d_small, 4, hipMemcpyDeviceToHost, c->stream));
	IX_CHK(ctx_sync(c));
	if (n_over) {                             // the last n_over keys of the (home, key) order wrap around: sequential probing from their home slots
		std::vector<uint32_t> idx2(n_over);
		std::vector<uint64_t> uq(nk); std::vector<uint32_t> fi(nk), cn(nk);
		IX_CHK(hipMemcpy(idx2.data(), d_idx2 + (nk - n_over), (size_t)n_over * 4, hipMemcpyDeviceToHost));
		IX_CHK(hipMemcpy(uq.data(), d_uniq, (size_t)nk * 8, hipMemcpyDeviceToHost));
		IX_CHK(hipMemcpy(fi.data(), d_first, (size_t)nk * 4, hipMemcpyDeviceToHost));
		IX_CHK(hipMemcpy(cn.data(), d_cnt, (size_t)nk * 4, hipMemcpyDeviceToHost));
		wm::index_table_insert(ix, n_over, [&](size_t t, uint64_t *key, uint64_t *val) { const uint32_t g = idx2[t]; *key = uq[g]; *val = (uint64_t)fi[g] << 32 | cn[g]; return wm::Index::slot_of(uq[g], ix.hbits); });
	}
#undef IX_CHK
	if (t_dev_s) *t_dev_s = (now_ms() - t0) * 1e-3;
	return WM_OK;
}

static int wm_index_build_seqs_dev(wm_ctx_t *c, const wm::IdxOpt &io, std::vector<std::string> &names, std::vector<std::string> &seqs, const std::string &kmer_file, int n_threads,
                                   wm_index_t **out, double *stats = 0, bool replace_ok = true, double t0 = -1);
extern "C" int wm_index_build_gpu(wm_ctx_t *c, const char *fasta, const char *kmer_file, int k, int w, int n_threads, wm_index_t **out, double *stats)
{
	return wm_index_build_gpu_flag(c, fasta, kmer_file, k, w, 0, n_threads, out, stats);
}
extern "C" int wm_index_build_gpu_flag(wm_ctx_t *c, const char *fasta, const char *kmer_file, int k, int w, int idx_flag, int n_threads, wm_index_t **out, double *stats)
{
	*out = 0;
	if (!c) return set_err(WM_EINVAL, "null context");
	if (idx_flag & ~1) return set_err(WM_EINVAL, "index flag %d: only MM_I_HPC (1) is known here", idx_flag);
	wm::IdxOpt io; io.k = k; io.w = w; io.flag = idx_flag;
	wm::MapOpt mo; std::string err;
	if (wm::check_opt(io, mo, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	const double t0 = now_ms();
	std::vector<std::string> names, seqs;
	if (wm::read_fastx(fasta, names, seqs, 0, 0, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	if (seqs.empty()) return set_err(WM_EINVAL, "no sequences in %s", fasta);
	return wm_index_build_seqs_dev(c, io, names, seqs, kmer_file ? kmer_file : "", n_threads, out, stats, false, t0);
}
// the same from sequences in memory. replace_ok: the context may hold an uploaded index that no mapper is using (wm_map_file_split_fasta between two
// parts): its filter and sketch parameters are put back when the build is done.
static int wm_index_build_seqs_dev(wm_ctx_t *c, const wm::IdxOpt &io, std::vector<std::string> &names, std::vector<std::string> &seqs, const std::string &kmer_file_s, int n_threads,
                                   wm_index_t **out, double *stats, bool replace_ok, double t0)
{
	*out = 0;
	const int k = io.k, w = io.w;
	const char *kmer_file = kmer_file_s.c_str();
	std::string err;
	if (!(k & 1) || k < 2) return set_err(WM_EINVAL, "the device index build needs an odd k (got %d): use wm_index_build", k);
	if (c->have_index && !replace_ok) return set_err(WM_EINVAL, "the context already holds an index: build on a fresh context, then wm_index_upload");
	HIPCHK(hipSetDevice(c->device));
	if (t0 < 0) t0 = now_ms();
	struct Keep { wm_ctx_t *c; uint8_t *bloom; bool owns; wm_sketch_params_t skp; ~Keep() { c->d_bloom = bloom; c->owns_filter = owns; c->skp = skp; } } keep{ c, c->d_bloom, c->owns_filter, c->skp };
	if (c->have_index) { c->d_bloom = 0; c->owns_filter = false; }      // (the resident index's filter: untouched, back in place on return)
	wm_index_t *h = new wm_index_t();
	wm::Index &ix = h->ix;
	if (wm::index_begin(io, names, seqs, kmer_file ? kmer_file : "", n_threads, ix, err) < 0) { delete h; return set_err(WM_EINVAL, "%s", err.c_str()); }
	if (ix.bloom.table_bits >= ((uint64_t)1 << 32)) { delete h; return set_err(WM_EINVAL, "bloom table of %llu bits not supported on device", (unsigned long long)ix.bloom.table_bits); }
	const double t1 = now_ms();
	// the bloom bit table goes first (the sketch kernel probes it); it is replaced by wm_index_upload later
	uint8_t *d_bloom = 0;
	if (hipMalloc((void**)&d_bloom, ix.bloom.bits.size() + 8) != hipSuccess || hipMemcpy(d_bloom, ix.bloom.bits.data(), ix.bloom.bits.size(), hipMemcpyHostToDevice) != hipSuccess) {
		delete h; if (d_bloom) hipFree(d_bloom); return set_err(WM_ENOMEM, "cannot place the bloom filter on the device");
	}
	if (c->owns_filter && c->d_bloom) hipFree(c->d_bloom);
	c->owns_filter = false;
	c->d_bloom = d_bloom;
	c->skp.w = w; c->skp.k = k; c->skp.table_bits = (uint32_t)ix.bloom.table_bits; c->skp.salt0 = ix.bloom.salt[0]; c->skp.salt1 = ix.bloom.salt[1]; c->skp.hpc = io.flag & 1;
	std::vector<wm::m128> all;
	int rc = WM_OK;
	const size_t budget = (size_t)(c->arena_bytes * 0.85);
	for (size_t g0 = 0; g0 < seqs.size() && rc == WM_OK;) {             // groups of contigs that fit the arena: 1 B codes + 24 B scratch + 2 B output + 4 B chunk-local output (+ tables) per base
		size_t g1 = g0, bases = 0;
		const size_t per_base = 34 + ((io.flag & 1) ? 5 : 0);             // (+ the homopolymer-compressed copy: a code and an end position per base)
		while (g1 < seqs.size() && (g1 == g0 || (bases + seqs[g1].size()) * per_base + 4096 * (g1 - g0 + 1) <= budget)) { bases += seqs[g1].size(); ++g1; }
		if (bases * per_base > budget) { rc = set_err(WM_ENOMEM, "contig %zu (%zu bases) needs %.1f GB of arena for the device sketch", g0, seqs[g0].size(), seqs[g0].size() * per_base / 1073741824.0); break; }
		const int n = (int)(g1 - g0);
		std::vector<uint64_t> off(n), ooff(n);
		std::vector<int32_t> len(n), cnt(n);
		std::unique_ptr<uint8_t[]> codes(new uint8_t[bases + 1]);
		size_t tot = 0;
		for (int i = 0; i < n; ++i) { off[i] = tot; len[i] = (int32_t)seqs[g0 + i].size(); tot += seqs[g0 + i].size(); }
		wm::parallel_for(n_threads, (size_t)n, [&](size_t i) { const std::string &sq = seqs[g0 + i]; uint8_t *d = codes.get() + off[i]; for (size_t j = 0; j < sq.size(); ++j) d[j] = wm::nt4_table[(uint8_t)sq[j]]; });
		std::vector<wm128_t> mv(bases / 8 + (size_t)17 * n + 64);
		rc = sketch_batch_impl(c, n, codes.get(), tot, off.data(), len.data(), 0, mv.data(), mv.size(), ooff.data(), cnt.data());
		if (rc == WM_ENOMEM && strstr(g_err, "minimizer output pool")) { mv.resize(bases + n + 1); rc = sketch_batch_impl(c, n, codes.get(), tot, off.data(), len.data(), 0, mv.data(), mv.size(), ooff.data(), cnt.data()); }
		if (rc) break;
		for (int i = 0; i < n; ++i)
			for (int t = 0; t < cnt[i]; ++t) { wm::m128 e; e.x = mv[ooff[i] + t].x; e.y = mv[ooff[i] + t].y | (uint64_t)(g0 + i) << 32; all.push_back(e); }     // rid (src/sketch.c:172)
		g0 = g1;
	}
	c->d_bloom = 0;
	hipFree(d_bloom);
	if (!c->have_index) { keep.bloom = 0; keep.owns = false; keep.skp = c->skp; }      // (as before on a fresh context: no filter left behind)
	if (rc) { delete h; return rc; }
	const double t2 = now_ms();
	const double n_mini = (double)all.size();
	// the table: on the device as well (WM_INDEX_TABLE_HOST=1: the host's sort + probing, A/B); a table that does not fit the arena falls to the host
	double t_tab_dev = -1;
	const int trc = getenv("WM_INDEX_TABLE_HOST") ? 1 : index_table_on_device(c, ix, all, &t_tab_dev);
	if (trc < 0) { delete h; return trc; }
	if (trc > 0) { t_tab_dev = -1; wm::index_table_from_minimizers(ix, all); }
	if (stats) { stats[0] = (t1 - t0) * 1e-3; stats[1] = (t2 - t1) * 1e-3; stats[2] = (now_ms() - t2) * 1e-3; stats[3] = n_mini; }
	c->aux_ms = (float)(t_tab_dev * 1e3);          // (wm_last_aux_ms: the device table build of this call, < 0 = built on the host)
	*out = h;
	return WM_OK;
}


// resident (optional, n flags): sequence i starts at code seq_off[i] of the resident read codes (wm_reads_upload) instead of `seqs`
static int sketch_batch_impl(wm_ctx_t *c, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len, const uint8_t *resident,
                             wm128_t *out, size_t out_cap, uint64_t *out_off, int32_t *counts);
extern "C" int wm_sketch_set_filter(wm_ctx_t *c, const uint8_t *bits, size_t n_bytes, uint64_t table_bits, uint32_t salt0, uint32_t salt1, int k, int w)
{
	if (!c) return set_err(WM_EINVAL, "null context");
	if (c->have_index) return set_err(WM_EINVAL, "the context holds an index (its filter is in use)");
	if (k < 1 || k > 28 || w < 1 || w > 255) return set_err(WM_EINVAL, "need 0 < k <= 28 and 0 < w < 256 (src/sketch.c:140)");
	if (table_bits >= ((uint64_t)1 << 32)) return set_err(WM_EINVAL, "bloom table of %llu bits not supported on device", (unsigned long long)table_bits);
	HIPCHK(hipSetDevice(c->device));
	static const uint8_t none[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // no filter: a table of 8 zero bits never matches
	if (!bits || table_bits == 0) { bits = none; n_bytes = 1; table_bits = 8; }
	if (c->d_bloom) { hipFree(c->d_bloom); c->d_bloom = 0; }
	HIPCHK(hipMalloc((void**)&c->d_bloom, n_bytes + 8));
	HIPCHK(hipMemcpy(c->d_bloom, bits, n_bytes, hipMemcpyHostToDevice));
	c->skp.w = w; c->skp.k = k; c->skp.table_bits = (uint32_t)table_bits; c->skp.salt0 = salt0; c->skp.salt1 = salt1; c->skp.hpc = 0;
	c->owns_filter = true;
	return WM_OK;
}

extern "C" int wm_sketch_batch(wm_ctx_t *c, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len,
                               wm128_t *out, size_t out_cap, uint64_t *out_off, int32_t *counts)
{
	return sketch_batch_impl(c, n, seqs, seqs_bytes, seq_off, len, 0, out, out_cap, out_off, counts);
}
// The one-wavefront-per-sequence sketch of `n` jobs (h_jobs = the host copy of d_jobs) and, for sequences of WM_SKETCH_LONG (65 536) codes and more, the
// chunked form: WM_SKETCH_CHUNK (16 384) positions per wavefront. allow_long = false (a repeat with full-size slots): everything on one wavefront each.
// Everything is queued on the context's stream; with long jobs the call waits once (its chunk tables are staged in the pinned slab).
static int sketch_long_thr(bool allow_long, int *chunk_out, bool hpc = false)
{
	if (hpc) { *chunk_out = 16384; return 0; }                 // homopolymer compression: every sequence on one wavefront (the chunks would have to be cut in run space)
	static const int long_env = getenv("WM_SKETCH_LONG") ? atoi(getenv("WM_SKETCH_LONG")) : 65536;
	static const int chunk = std::max(1024, getenv("WM_SKETCH_CHUNK") ? atoi(getenv("WM_SKETCH_CHUNK")) : 16384);
	*chunk_out = chunk;
	return allow_long && long_env > 0 ? std::max(long_env, 2 * chunk) : 0;
}
// device bytes sketch_launch needs on top of the caller's buffers (chunk tables + chunk-local output slots): a caller that hands the rest of the arena to
// something else (window_launch: the anchor pool) reserves them first and passes the block in
static size_t sketch_long_bytes(int n, const wm_sketch_job_t *h_jobs, bool allow_long, bool hpc = false)
{
	int chunk = 0;
	const int long_thr = sketch_long_thr(allow_long, &chunk, hpc);
	size_t bytes = 0;
	if (hpc) {                                                  // the compacted sequences: a code and an end position per base at most
		uint64_t slots = 0;
		for (int i = 0; i < n; ++i) if (h_jobs[i].len > 0) slots = std::max<uint64_t>(slots, h_jobs[i].scratch_off + (uint64_t)h_jobs[i].len);
		return (size_t)(slots + 1) * 5 + 4096;
	}
	if (long_thr > 0)
		for (int i = 0; i < n; ++i)
			if (h_jobs[i].len >= long_thr) {
				const size_t k = ((size_t)h_jobs[i].len + chunk - 1) / chunk;
				bytes += k * (sizeof(wm_sk_chunk_t) + 8 + ((size_t)chunk / 4 + 64 + 1) * sizeof(wm128_t)) + 12 + 1024;
			}
	return bytes ? bytes + 4096 : 0;
}
static int sketch_launch(wm_ctx_t *c, int n, const wm_sketch_job_t *h_jobs, const wm_sketch_job_t *d_jobs, const int *d_ord, const uint8_t *d_seqs,
                         double *d_so, uint64_t *d_sx, uint32_t *d_sy, uint32_t *d_sl, wm128_t *d_out, int *d_cnt, bool allow_long, uint8_t *mem = 0, size_t mem_bytes = 0)
{
	int chunk = 0;
	const bool hpc = c->skp.hpc != 0;
	const int long_thr = sketch_long_thr(allow_long, &chunk, hpc);
	size_t mem_used = 0;
	auto take = [&](size_t bytes) -> void* {                   // from the caller's block if there is one, else from the arena
		if (!mem) return arena_take(c, bytes);
		const size_t a = (mem_used + 255) & ~(size_t)255;
		if (a + bytes > mem_bytes) return (void*)0;
		mem_used = a + bytes;
		return mem + a;
	};
	std::vector<int> lj;                                       // job, first chunk, chunks
	size_t n_ch = 0;
	if (long_thr > 0)
		for (int i = 0; i < n; ++i)
			if (h_jobs[i].len >= long_thr) { const int k = (h_jobs[i].len + chunk - 1) / chunk; lj.push_back(i); lj.push_back((int)n_ch); lj.push_back(k); n_ch += (size_t)k; }
	uint8_t *d_hc = 0; uint32_t *d_he = 0;
	if (hpc) {
		uint64_t slots = 0;
		for (int i = 0; i < n; ++i) if (h_jobs[i].len > 0) slots = std::max<uint64_t>(slots, h_jobs[i].scratch_off + (uint64_t)h_jobs[i].len);
		d_he = (uint32_t*)take((size_t)(slots + 1) * 4); d_hc = (uint8_t*)take((size_t)slots + 1);
		if (!d_he || !d_hc) return set_err(WM_ENOMEM, "sketch batch does not fit the arena (homopolymer-compressed copies)");
	}
	hipLaunchKernelGGL(sketch_coop_kernel, dim3(n), dim3(64), 0, c->stream, c->skp, d_jobs, d_ord, d_seqs, c->d_reads, c->d_reads_nm, c->d_bloom, d_so, d_sx, d_sy, d_sl, d_out, d_cnt, long_thr,
	                   d_hc, d_he);
	if (lj.empty()) return WM_OK;
	UBuf<wm_sk_chunk_t> ch(n_ch, c);
	UBuf<int> plj(lj.size(), c);
	memcpy(plj.data(), lj.data(), lj.size() * sizeof(int));
	uint64_t co = 0;
	for (size_t q = 0; q < lj.size(); q += 3) {
		const int i = lj[q], c0 = lj[q + 1], k = lj[q + 2];
		for (int t = 0; t < k; ++t) {
			wm_sk_chunk_t &x = ch[(size_t)c0 + t];
			x.job = i; x.begin = t * chunk; x.end = std::min(h_jobs[i].len, (t + 1) * chunk); x.first = t == 0; x.pad = 0;
			x.cap = (x.end - x.begin) / 4 + 64;                   // (a chunk's wavefront also covers the chunks it absorbs: twice the job slot's density; beyond that the job is repeated)
			x.out_off = co; co += (uint64_t)x.cap;
		}
	}
	wm_sk_chunk_t *d_ch = (wm_sk_chunk_t*)take(n_ch * sizeof(wm_sk_chunk_t));
	int *d_lj = (int*)take(lj.size() * 4 + 64), *d_sync = (int*)take(n_ch * 4 + 64), *d_cc = (int*)take(n_ch * 4 + 64);
	wm128_t *d_cout = (wm128_t*)take((co + 1) * sizeof(wm128_t));
	if (!d_ch || !d_lj || !d_sync || !d_cc || !d_cout) return set_err(WM_ENOMEM, "sketch batch does not fit the arena");
	HIPCHK(hipMemcpyAsync(d_ch, ch.data(), n_ch * sizeof(wm_sk_chunk_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d_lj, plj.data(), lj.size() * 4, hipMemcpyHostToDevice, c->stream));
	hipLaunchKernelGGL(sketch_long_p1_kernel, dim3((unsigned)n_ch), dim3(64), 0, c->stream, c->skp, d_jobs, d_ch, d_seqs, c->d_reads, c->d_reads_nm, c->d_bloom, d_so, d_sx, d_sy, d_sl);
	hipLaunchKernelGGL(sketch_long_sync_kernel, dim3((unsigned)n_ch), dim3(64), 0, c->stream, c->skp, d_jobs, d_ch, d_so, d_sync);
	hipLaunchKernelGGL(sketch_long_p2_kernel, dim3((unsigned)n_ch), dim3(64), 0, c->stream, c->skp, d_jobs, d_ch, (int)n_ch, d_so, d_sx, d_sy, d_sl, d_sync, d_cout, d_cc);
	hipLaunchKernelGGL(sketch_long_gather_kernel, dim3((unsigned)(lj.size() / 3)), dim3(64), 0, c->stream, d_jobs, d_lj, d_ch, d_cout, d_cc, d_out, d_cnt);
	HIPCHK(ctx_sync(c));                 // (the staged tables above are read by the copies until here)
	return WM_OK;
}

static int sketch_batch_impl(wm_ctx_t *c, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len, const uint8_t *resident,
                             wm128_t *out, size_t out_cap, uint64_t *out_off, int32_t *counts)
try {
	if (!c || !c->d_bloom) return set_err(WM_EINVAL, "wm_index_upload has not been called on this context");
	if (n <= 0) return WM_OK;
	HIPCHK(hipSetDevice(c->device));
	const int w = c->skp.w;
	const size_t lds = (size_t)w * 64 * 12;
	if (lds > 160 * 1024) return set_err(WM_EINVAL, "window w=%d needs %zu B of LDS per wave (max 160 KB)", w, lds);
	HIPCHK(hipFuncSetAttribute((const void*)sketch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	// first try a slot of len/8+16 minimizers per sequence (typical density is 2/(w+1)); retry the rare overflow at full size
	std::vector<wm_sketch_job_t> jobs(n);
	std::vector<int> todo(n);
	for (int i = 0; i < n; ++i) todo[i] = i;
	size_t used = 0;
	float ms_total = 0;
	for (int round = 0; round < 2 && !todo.empty(); ++round) {
		ArenaMark mark(c);
		std::vector<wm_sketch_job_t> jb(todo.size());
		uint64_t tot = 0;
		wm_sketch_job_t *d_jobs = (wm_sketch_job_t*)arena_take(c, jb.size() * sizeof(wm_sketch_job_t));
		uint8_t *d_seqs = (uint8_t*)arena_take(c, seqs_bytes + 64);
		// odd k (every preset): one wavefront per sequence walking the chain of window minima (sketch_coop); even k: the palindrome rule
		// of src/sketch.c:166 makes the slot stream data dependent -> the one-lane-per-sequence automaton (sketch_wave). WM_SKETCH_LANE=1 forces the latter.
		static const bool force_lane = getenv("WM_SKETCH_LANE") != 0;
		const bool coop = (c->skp.k & 1) && c->skp.k >= 2 && !force_lane;
		if (c->skp.hpc && !coop) return set_err(WM_EINVAL, "homopolymer compression on the device needs an odd k (got %d)", c->skp.k);
		uint64_t slots = 0;
		for (size_t t = 0; t < todo.size(); ++t) {
			const int i = todo[t];
			const bool res = resident && resident[i];
			if (res ? (seq_off[i] + (uint64_t)len[i] > c->reads_bytes || !c->d_reads) : (seq_off[i] + (uint64_t)len[i] > seqs_bytes)) return set_err(WM_EINVAL, "sequence %d outside its buffer", i);
			jb[t].seq_off = res ? (WM_RD_PACKED_BIT | seq_off[i]) : seq_off[i]; jb[t].len = len[i];      // (resident: a base index into the packed reads, reads2bit.h)
			jb[t].cap = round == 0 ? len[i] / 8 + 16 : len[i] + 1;
			jb[t].out_off = tot; tot += jb[t].cap;
			jb[t].scratch_off = slots; slots += (uint64_t)(len[i] > 0 ? len[i] : 0);
		}
		wm128_t *d_out = (wm128_t*)arena_take(c, (tot + 1) * sizeof(wm128_t));
		int *d_cnt = (int*)arena_take(c, jb.size() * 4 + 64);
		if (!d_jobs || !d_seqs || !d_out || !d_cnt) return set_err(WM_ENOMEM, "sketch batch does not fit the arena");
		HIPCHK(hipMemcpyAsync(d_jobs, jb.data(), jb.size() * sizeof(wm_sketch_job_t), hipMemcpyHostToDevice, c->stream));
		if (seqs_bytes) HIPCHK(hipMemcpyAsync(d_seqs, seqs, seqs_bytes, hipMemcpyHostToDevice, c->stream));
		std::vector<int> ord;
		if (coop) {
			double *d_so = (double*)arena_take(c, (slots + 1) * 8);
			uint64_t *d_sx = (uint64_t*)arena_take(c, (slots + 1) * 8);
			uint32_t *d_sy = (uint32_t*)arena_take(c, (slots + 1) * 4), *d_sl = (uint32_t*)arena_take(c, (slots + 1) * 4);
			int *d_ord = (int*)arena_take(c, jb.size() * 4 + 64);
			if (!d_so || !d_sx || !d_sy || !d_sl || !d_ord) return set_err(WM_ENOMEM, "sketch batch does not fit the arena");
			ord.resize(jb.size());
			for (size_t t = 0; t < jb.size(); ++t) ord[t] = (int)t;
			std::sort(ord.begin(), ord.end(), [&](int a, int b) { return jb[a].len != jb[b].len ? jb[a].len > jb[b].len : a < b; });     // longest first
			HIPCHK(hipMemcpyAsync(d_ord, ord.data(), ord.size() * 4, hipMemcpyHostToDevice, c->stream));
			HIPCHK(hipEventRecord(c->ev[0], c->stream));
			if (const int rc = sketch_launch(c, (int)jb.size(), jb.data(), d_jobs, d_ord, d_seqs, d_so, d_sx, d_sy, d_sl, d_out, d_cnt, round == 0)) return rc;
		} else {
			HIPCHK(hipEventRecord(c->ev[0], c->stream));
			hipLaunchKernelGGL(sketch_kernel, dim3(((int)jb.size() + 63) / 64), dim3(64), lds, c->stream, c->skp, d_jobs, (int)jb.size(), d_seqs, c->d_reads, c->d_reads_nm, c->d_bloom, d_out, d_cnt);
		}
		HIPCHK(hipEventRecord(c->ev[1], c->stream));
		UBuf<int> cnt(jb.size() + 1, c);
		UBuf<wm128_t> tmp(tot + 1, c);
		HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt, jb.size() * 4, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipMemcpyAsync(tmp.data(), d_out, tot * sizeof(wm128_t), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(ctx_sync(c));
		HIPCHK(hipGetLastError());
		float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); ms_total += ms;
		std::vector<int> again;
		for (size_t t = 0; t < todo.size(); ++t) {
			const int i = todo[t];
			if (cnt[t] > jb[t].cap) { again.push_back(i); continue; }
			if (used + cnt[t] > out_cap) return set_err(WM_ENOMEM, "minimizer output pool too small");
			out_off[i] = used; counts[i] = cnt[t];
			memcpy(out + used, tmp.data() + jb[t].out_off, (size_t)cnt[t] * sizeof(wm128_t));
			used += cnt[t];
		}
		todo.swap(again);
	}
	c->aux_ms = ms_total;
	return WM_OK;
}
catch (const std::bad_alloc &) { return set_err(WM_ENOMEM, "out of host memory"); }

extern "C" int wm_seed_batch(wm_ctx_t *c, int n, const wm128_t *mini, const uint64_t *mini_off, const int32_t *n_mini, const int32_t *qlen,
                             int max_occ, int64_t flag, wm128_t *out, size_t out_cap, uint64_t *out_off, int32_t *n_anchors, int32_t *rep_len)
try {
	if (!c || !c->have_index) return set_err(WM_EINVAL, "wm_index_upload has not been called on this context");
	if (n <= 0) return WM_OK;
	HIPCHK(hipSetDevice(c->device));
	// pass 1 on the host side of the boundary: how many anchors each job can produce is unknown until the lookup,
	// so run with a generous slot and retry the overflowing jobs with the exact size the kernel reports
	std::vector<int> todo(n);
	for (int i = 0; i < n; ++i) todo[i] = i;
	std::vector<int> want(n);
	for (int i = 0; i < n; ++i) want[i] = n_mini[i] * 2 + 32;
	size_t used = 0;
	float ms_total = 0;
	uint64_t mini_total = 0;
	for (int i = 0; i < n; ++i) mini_total = std::max<uint64_t>(mini_total, mini_off[i] + n_mini[i]);
	for (int round = 0; round < 3 && !todo.empty(); ++round) {
		ArenaMark mark(c);
		std::vector<wm_seed_job_t> jb(todo.size());
		std::vector<uint64_t> occ_off(todo.size());
		uint64_t tot = 0, occ_tot = 0;
		for (size_t t = 0; t < todo.size(); ++t) {
			const int i = todo[t];
			jb[t].mini_off = mini_off[i]; jb[t].n_mini = n_mini[i]; jb[t].qlen = qlen[i]; jb[t].max_occ = max_occ; jb[t].cap = want[i];
			jb[t].flag = (int32_t)(flag & (0x100000 | 0x200000)); jb[t].pad = 0;
			jb[t].out_off = tot; tot += want[i];
			occ_off[t] = occ_tot; occ_tot += n_mini[i];
		}
		wm_seed_job_t *d_jobs = (wm_seed_job_t*)arena_take(c, jb.size() * sizeof(wm_seed_job_t));
		uint64_t *d_occ_off = (uint64_t*)arena_take(c, jb.size() * 8 + 64);
		wm128_t *d_mini = (wm128_t*)arena_take(c, (mini_total + 1) * sizeof(wm128_t));
		wm128_t *d_out = (wm128_t*)arena_take(c, (tot + 1) * sizeof(wm128_t));
		int *d_occ = (int*)arena_take(c, (occ_tot + 1) * 4);
		wm_seed_res_t *d_res = (wm_seed_res_t*)arena_take(c, jb.size() * sizeof(wm_seed_res_t) + 64);
		if (!d_jobs || !d_occ_off || !d_mini || !d_out || !d_occ || !d_res) return set_err(WM_ENOMEM, "seed batch does not fit the arena");
		HIPCHK(hipMemcpyAsync(d_jobs, jb.data(), jb.size() * sizeof(wm_seed_job_t), hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_occ_off, occ_off.data(), jb.size() * 8, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_mini, mini, mini_total * sizeof(wm128_t), hipMemcpyHostToDevice, c->stream));
		wm_index_view_t ix = { c->d_hkey, c->d_hval, c->d_P, c->hbits, 0 };
		HIPCHK(hipEventRecord(c->ev[0], c->stream));
		hipLaunchKernelGGL(seed_kernel, dim3((int)jb.size()), dim3(64), 0, c->stream, ix, d_jobs, d_mini, d_out, d_occ, d_occ_off, d_res);
		HIPCHK(hipEventRecord(c->ev[1], c->stream));
		UBuf<wm_seed_res_t> res(jb.size() + 1, c);
		HIPCHK(hipMemcpyAsync(res.data(), d_res, jb.size() * sizeof(wm_seed_res_t), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(ctx_sync(c));
		HIPCHK(hipGetLastError());
		float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); ms_total += ms;
		std::vector<int> again;
		std::vector<size_t> done_t;
		for (size_t t = 0; t < todo.size(); ++t) {
			const int i = todo[t];
			if (res[t].n_anchors > jb[t].cap) { want[i] = res[t].n_anchors; again.push_back(i); continue; }
			if (used + res[t].n_anchors > out_cap) return set_err(WM_ENOMEM, "anchor output pool too small");
			out_off[i] = used; n_anchors[i] = res[t].n_anchors; rep_len[i] = res[t].rep_len;
			used += res[t].n_anchors;
			done_t.push_back(t);
		}
		// sort by x (src/map.c:252). On the device: a segmented radix sort of all jobs at once gives THE order whenever the keys of a job are
		// distinct; a job in which two anchors share a key (the same reference position reached from two query positions) gets the tie
		// permutation of the reference's in-place unstable radix sort, which is sequential by nature -> those jobs are re-sorted on the host.
		// Opt-in (WM_SEED_DEVICE_SORT=1): on BASELINE config 2 the extra device pass + synchronisation costs about what the host sort on idle
		// workers costs (0.165 vs 0.174 Gbp/s, profiles/r02x_bench_device_seed_sort.json); it pays when single jobs hold 10^5..10^6 anchors.
		const bool host_sort = !(getenv("WM_SEED_DEVICE_SORT") && atoi(getenv("WM_SEED_DEVICE_SORT")) != 0);
		UBuf<wm128_t> tmp(tot + 1, c);
		UBuf<int> tie(jb.size() + 1, c);
		bool dev_sorted = false;
		if (!host_sort && tot > 0 && tot < ((uint64_t)1 << 32) && !done_t.empty()) {
			uint64_t *d_k = (uint64_t*)arena_take(c, (tot + 1) * 8), *d_v = (uint64_t*)arena_take(c, (tot + 1) * 8);
			uint64_t *d_k2 = (uint64_t*)arena_take(c, (tot + 1) * 8), *d_v2 = (uint64_t*)arena_take(c, (tot + 1) * 8);
			wm128_t *d_sorted = (wm128_t*)arena_take(c, (tot + 1) * sizeof(wm128_t));
			uint32_t *d_beg = (uint32_t*)arena_take(c, jb.size() * 4 + 64), *d_end = (uint32_t*)arena_take(c, jb.size() * 4 + 64);
			int *d_tie = (int*)arena_take(c, jb.size() * 4 + 64);
			size_t tmp_bytes = 0;
			if (d_k && d_v && d_k2 && d_v2 && d_sorted && d_beg && d_end && d_tie &&
			    rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, d_k, d_k2, d_v, d_v2, (unsigned)tot, (unsigned)jb.size(), d_beg, d_end, 0, 64, c->stream) == hipSuccess) {
				void *d_tmp = arena_take(c, tmp_bytes + 256);
				if (d_tmp) {
					UBuf<uint32_t> hb(2 * jb.size() + 2, c);
					uint32_t *hbeg = hb.data(), *hend = hb.data() + jb.size();
					for (size_t t = 0; t < jb.size(); ++t) { hbeg[t] = (uint32_t)jb[t].out_off; hend[t] = (uint32_t)jb[t].out_off; }       // (jobs to be retried: empty segments)
					for (size_t t : done_t) hend[t] = (uint32_t)(jb[t].out_off + (uint64_t)res[t].n_anchors);
					HIPCHK(hipMemcpyAsync(d_beg, hbeg, jb.size() * 4, hipMemcpyHostToDevice, c->stream));
					HIPCHK(hipMemcpyAsync(d_end, hend, jb.size() * 4, hipMemcpyHostToDevice, c->stream));
					HIPCHK(hipMemsetAsync(d_tie, 0, jb.size() * 4, c->stream));
					hipLaunchKernelGGL(seed_split_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, d_out, tot, d_k, d_v);
					HIPCHK(rocprim::segmented_radix_sort_pairs(d_tmp, tmp_bytes, d_k, d_k2, d_v, d_v2, (unsigned)tot, (unsigned)jb.size(), d_beg, d_end, 0, 64, c->stream));
					hipLaunchKernelGGL(seed_merge_kernel, dim3((unsigned)jb.size()), dim3(64), 0, c->stream, d_beg, d_end, d_k2, d_v2, d_sorted, d_tie);
					HIPCHK(hipMemcpyAsync(tie.data(), d_tie, jb.size() * 4, hipMemcpyDeviceToHost, c->stream));
					HIPCHK(hipMemcpyAsync(tmp.data(), d_sorted, tot * sizeof(wm128_t), hipMemcpyDeviceToHost, c->stream));
					HIPCHK(ctx_sync(c));
					HIPCHK(hipGetLastError());
					dev_sorted = true;
				}
			}
		}
		bool any_tie = false;
		if (dev_sorted) for (size_t t : done_t) any_tie |= tie[t] != 0;
		UBuf<wm128_t> raw(!dev_sorted || any_tie ? tot + 1 : 1, dev_sorted ? 0 : c);       // the unsorted anchors: for the host sort
		if (!dev_sorted || any_tie) {
			HIPCHK(hipMemcpyAsync(raw.data(), d_out, tot * sizeof(wm128_t), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(ctx_sync(c));
		}
		// jobs with very many anchors (reads inside a repeat family: 10^5..10^6 hits) one at a time, each spread over the threads: sorted by one
		// thread such a job alone would keep the whole call — and its device context — waiting
		std::vector<size_t> rest_t;
		rest_t.reserve(done_t.size());
		for (size_t t : done_t) {
			if (res[t].n_anchors < (1 << 16) || (dev_sorted && !tie[t])) { rest_t.push_back(t); continue; }
			wm128_t *dst = out + out_off[todo[t]];
			memcpy(dst, raw.data() + jb[t].out_off, (size_t)res[t].n_anchors * sizeof(wm128_t));
			WM_SITE("seed.giant_radix_sort");
			wm::radix_sort_128x_parallel(dst, dst + res[t].n_anchors, c->host_threads);
		}
		WM_SITE("seed.copy+radix_sort");
		wm::parallel_for(c->host_threads, rest_t.size(), [&](size_t k) {
			const size_t t = rest_t[k];
			wm128_t *dst = out + out_off[todo[t]];
			if (dev_sorted && !tie[t]) { memcpy(dst, tmp.data() + jb[t].out_off, (size_t)res[t].n_anchors * sizeof(wm128_t)); return; }
			memcpy(dst, raw.data() + jb[t].out_off, (size_t)res[t].n_anchors * sizeof(wm128_t));
			// the reference's in-place unstable radix sort (src/map.c:252); its tie permutation is sequential by nature
			wm::radix_sort_128x(dst, dst + res[t].n_anchors);
		});
		todo.swap(again);
	}
	if (!todo.empty()) return set_err(WM_EINTERNAL, "seed retry did not converge");
	c->aux_ms = ms_total;
	return WM_OK;
}
catch (const std::bad_alloc &) { return set_err(WM_ENOMEM, "out of host memory"); }

extern "C" int wm_chain_batch(wm_ctx_t *c, int n, wm128_t *a, const uint64_t *a_off, const int32_t *n_a, const wm_chain_par_t *par,
                              uint64_t *u, uint64_t *u_off, int32_t *n_u, int32_t *n_v)
try {
	if (!c) return set_err(WM_EINVAL, "null context");
	if (n <= 0) return WM_OK;
	HIPCHK(hipSetDevice(c->device));
	ArenaMark mark(c);
	static const bool trace = getenv("WM_TRACE") != 0;
	const double tt0 = trace ? now_ms() : 0;
	uint64_t tot = 0;
	for (int i = 0; i < n; ++i) tot = std::max<uint64_t>(tot, a_off[i] + n_a[i]);
	std::vector<wm_chain_job_t> jb(n);
	std::vector<int> order(n);
	WM_SITE("chain.jobs+avg_qspan");
	wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
		jb[i].a_off = a_off[i]; jb[i].n = n_a[i];
		jb[i].max_dist_x = par[i].max_dist_x; jb[i].min_dist_x = par[i].min_dist_x; jb[i].max_dist_y = par[i].max_dist_y; jb[i].bw = par[i].bw;
		jb[i].max_skip = par[i].max_skip; jb[i].max_iter = par[i].max_iter; jb[i].gap_scale = par[i].gap_scale; jb[i].is_cdna = par[i].is_cdna != 0;
		jb[i].avg_qspan = n_a[i] > 0 ? wm::chain_avg_qspan(n_a[i], a + a_off[i]) : 0.f;
		order[i] = (int)i;
	});
	// jobs larger than the small windows: DENSE if a sample of anchors has more than ~900 predecessors within max_dist_x (satellite
	// arrays, no -W list) -> multi-wave kernel with the 4096-anchor window; otherwise one wave with a 1024-anchor window
	std::vector<uint8_t> dense(n, 0);
	WM_SITE("chain.dense_probe");
	wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
		const int m = n_a[i];
		if (m <= 1024) return;
		const wm128_t *aa = a + a_off[i];
		int64_t worst = 0;
		for (int s = 1; s <= 32; ++s) {
			const int64_t k = (int64_t)m * s / 33;
			const uint64_t lim = aa[k].x > (uint64_t)par[i].max_dist_x ? aa[k].x - (uint64_t)par[i].max_dist_x : 0;
			int64_t lo = 0, hi = k;                        // first anchor with x >= lim
			while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (aa[mid].x < lim) lo = mid + 1; else hi = mid; }
			if (k - lo > worst) worst = k - lo;
		}
		dense[i] = worst > 900;
	});
	auto klass_of = [&](int i) { return n_a[i] > 1024 ? (dense[i] ? 0 : 1) : n_a[i] > 256 ? 2 : 3; };
	std::sort(order.begin(), order.end(), [&](int x, int y) { const int kx = klass_of(x), ky = klass_of(y); return kx != ky ? kx < ky : n_a[x] != n_a[y] ? n_a[x] > n_a[y] : x < y; });
	wm_chain_job_t *d_jobs = (wm_chain_job_t*)arena_take(c, (size_t)n * sizeof(wm_chain_job_t));
	int *d_order = (int*)arena_take(c, (size_t)n * 4 + 64);
	wm128_t *d_a = (wm128_t*)arena_take(c, (tot + 1) * sizeof(wm128_t));
	int *d_fpvt = (int*)arena_take(c, (tot + 1) * 16);
	if (!d_jobs || !d_order || !d_a || !d_fpvt) return set_err(WM_ENOMEM, "chain batch does not fit the arena");
	HIPCHK(hipMemcpyAsync(d_jobs, jb.data(), (size_t)n * sizeof(wm_chain_job_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d_a, a, tot * sizeof(wm128_t), hipMemcpyHostToDevice, c->stream));
	const double tt1 = trace ? now_ms() : 0;
	HIPCHK(hipEventRecord(c->ev[0], c->stream));
	{   // classes (order is grouped by class, largest jobs first inside a class): LDS footprint = 28 B * W
		HIPCHK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		constexpr int NWV = 8;
		HIPCHK(hipFuncSetAttribute((const void*)chain_kernel_block<NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		int b = 0;
		for (int k = 0; k < 4; ++k) {                        // 0 dense (8 waves, W 4096) | 1 large sparse (1 wave, W 1024) | 2 n <= 1024 | 3 n <= 256
			int e = b;
			while (e < n && klass_of(order[e]) == k) ++e;
			if (e > b) {
				if (k == 0) hipLaunchKernelGGL(chain_kernel_block<NWV>, dim3(e - b), dim3(64 * NWV), (size_t)4096 * 28 + NWV * 69 * 4 + 64, c->stream, d_jobs, d_order + b, d_a, d_fpvt, 4096);
				else hipLaunchKernelGGL(chain_kernel, dim3(e - b), dim3(64), (size_t)(k == 3 ? 256 : 1024) * 28, c->stream, d_jobs, d_order + b, d_a, d_fpvt, k == 3 ? 256 : 1024);
			}
			b = e;
		}
	}
	HIPCHK(hipEventRecord(c->ev[1], c->stream));
	const double tt2 = trace ? now_ms() : 0;
	UBuf<int> fpvt((tot + 1) * 4, c);
	HIPCHK(hipMemcpyAsync(fpvt.data(), d_fpvt, tot * 16, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(ctx_sync(c));
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventElapsedTime(&c->aux_ms, c->ev[0], c->ev[1]));
	const double tt3 = trace ? now_ms() : 0;
	// chain extraction (src/chain.c:93-165): O(n) bookkeeping on the fill's f/p/v
	std::vector<std::vector<uint64_t>> uus(n);
	WM_SITE("chain.extract");
	wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
		const int *f = fpvt.data() + a_off[i] * 4, *p = f + n_a[i];
		int *v = fpvt.data() + a_off[i] * 4 + 2 * (size_t)n_a[i];
		for (int k = 0; k < n_a[i]; ++k) v[k] = p[k] >= 0 && v[p[k]] > f[k] ? v[p[k]] : f[k];          // peak score, src/chain.c:89
		std::vector<wm::m128> bb;
		wm::chain_extract(n_a[i], a + a_off[i], f, p, v, par[i].min_cnt, par[i].min_sc, uus[i], bb);
		n_u[i] = (int)uus[i].size(); n_v[i] = (int)bb.size();
		if (!bb.empty()) memcpy(a + a_off[i], bb.data(), bb.size() * sizeof(wm128_t));
	});
	uint64_t uo = 0;
	for (int i = 0; i < n; ++i) {
		u_off[i] = uo;
		for (size_t k = 0; k < uus[i].size(); ++k) u[uo + k] = uus[i][k];
		uo += uus[i].size();
	}
	if (trace) fprintf(stderr, "[chain_batch] n=%d anchors=%llu prep+h2d %.2f launch %.2f wait %.2f (kernel %.2f) extract %.2f ms\n", n, (unsigned long long)tot,
	                   tt1 - tt0, tt2 - tt1, tt3 - tt2, c->aux_ms, now_ms() - tt3);
	return WM_OK;
}
catch (const std::bad_alloc &) { return set_err(WM_ENOMEM, "out of host memory"); }

// ======================================================================================================
// wm_window_batch: sketch → seed → sort → chain fill → chain extraction of n jobs, resident in HBM (window_kernel.h)
// ======================================================================================================
__global__ __launch_bounds__(64) void win_seed_kernel(wm_index_view_t ix, const wm_win_job_t *__restrict__ jobs, const wm_sketch_job_t *__restrict__ sj, const int *__restrict__ mcnt,
                                                       const wm128_t *__restrict__ mini_pool, const wm128_t *__restrict__ pre_pool, int *occ, uint32_t *first, int *emit,
                                                       wm128_t *anchors, uint64_t *used, uint64_t cap, wm_win_res_t *res, int *worst_err)
{
	WM_SETPRIO(2);
	const int j = blockIdx.x;
	const wm_win_job_t jb = jobs[j];
	const wm_sketch_job_t s = sj[j];
	int n_mini = jb.seq_off >= 0 ? mcnt[j] : 0;
	const bool over = n_mini > s.cap;                       // the minimizer slot was too small: the caller retries with full-size slots
	if (over) n_mini = 0;
	wmk::win_seed_wave(ix, jb, mini_pool + s.out_off, n_mini, pre_pool + jb.pre_off, occ + s.out_off, first + s.out_off, emit + s.out_off, anchors, used, cap, res + j);
	if (threadIdx.x == 0) {
		if (over) res[j].err = 1;
		const int e = res[j].err;
		if (e) atomicMax(worst_err, e);
	}
}

// jobs of at most WIN_SMALL anchors: sorts, fill and extraction by one wavefront in LDS (win_small_wave); larger ones take the kernels below
__global__ __launch_bounds__(64) void win_small_kernel(const wm_win_job_t *__restrict__ jobs, wm_win_res_t *res, const wm128_t *__restrict__ anchors,
                                                        uint64_t *u_pool, wm128_t *v_pool, uint64_t *pool_ctr)
{
	WM_SETPRIO(2);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int j = blockIdx.x;
	const wm_win_res_t r = res[j];
	if (r.n_a <= 0 || r.n_a > wmk::WIN_SMALL || r.err) return;
	wmk::win_small_wave(jobs[j], r.n_a, anchors + r.a_off, smem, res + j, u_pool, v_pool, pool_ctr);
}

// radix_sort_128x of the seeded anchors (src/map.c:252), of the union with the handed-in ones (src/map.c:833), then avg_qspan + the fill's class.
// lds_cap = anchors that fit the dynamic LDS of this launch; a job runs in the launch whose range (lo, lds_cap] holds its size, the last launch
// (lds_cap = 0) takes the rest in global memory
__global__ __launch_bounds__(64) void win_sort_kernel(const wm_win_job_t *__restrict__ jobs, const wm_win_res_t *res, wm128_t *anchors, wm_chain_job_t *cj,
                                                       int *lists, int *counts, int n_jobs, int lo, int lds_cap)
{
	WM_SETPRIO(2);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	int *ws = (int*)smem;
	wm128_t *stage = (wm128_t*)(ws + ((wmk::WIN_WS_INTS + 3) & ~3));
	const int j = blockIdx.x;
	const wm_win_res_t r = res[j];
	const int n = r.n_a;
	if (n <= lo || (lds_cap > 0 && n > lds_cap) || r.err) return;
	const wm_win_job_t jb = jobs[j];
	wm128_t *a = anchors + r.a_off;
	const bool seeded = jb.seq_off >= 0;
	const int n_pre = jb.n_pre < n ? jb.n_pre : n;
	if (lds_cap > 0) {
		uint64_t *ga = (uint64_t*)a, *la = (uint64_t*)stage;
		if (seeded) {
			for (int i = threadIdx.x; i < 2 * n; i += 64) la[i] = ga[i];
			simt::lds_sync();
			wmk::win_sort_wave<false>(stage + n_pre, n - n_pre, ws);
			if (n_pre > 0) wmk::win_sort_wave<false>(stage, n, ws);
			simt::lds_sync();
			for (int i = threadIdx.x; i < 2 * n; i += 64) ga[i] = la[i];
			wmk::win_plan_wave(jb, j, r.a_off, n, stage, cj, lists, counts, n_jobs);
		} else wmk::win_plan_wave(jb, j, r.a_off, n, a, cj, lists, counts, n_jobs);
	} else {
		if (seeded) {
			wmk::win_sort_wave<true>(a + n_pre, n - n_pre, ws);
			if (n_pre > 0) wmk::win_sort_wave<true>(a, n, ws);
			wmk::win_fence();
		}
		wmk::win_plan_wave(jb, j, r.a_off, n, a, cj, lists, counts, n_jobs);
	}
}

// anchor sets beyond the LDS classes: a whole workgroup sorts (win_bigsort_block: stable, exact whenever the keys are distinct); if two keys tie the
// job falls back to the literal replay of the reference's permutation by one wavefront (win_sort_wave<true>) on the untouched input
// tie_list != 0 (round 5): a job whose keys tie is not replayed here — one lane walking the permutation through global memory costs ~1.5 us per anchor and
// digit level (14 s for the 10^6 anchors of a 5-Mb contig's stage-2 pass, profiles/r05_config5.txt) — but handed to the HOST, where the same serial
// algorithm runs a thousand times faster (window_launch: the ranges of the listed jobs travel down, are sorted by host/wm_core.cpp's radix_sort_128x,
// travel back, and win_plan_list_kernel finishes the job). Entry i of the list (8 ints from tie_list + 8 + 8 i): job, round, a_off lo / hi, n, n_pre.
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void win_bigsort_kernel(const wm_win_job_t *__restrict__ jobs, const wm_win_res_t *res, wm128_t *anchors, wm128_t *buf0, wm128_t *buf1,
                                                               wm_chain_job_t *cj, int *lists, int *counts, int n_jobs, int lo, int *tie_list)
{
	WM_SETPRIO(2);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	int *big = (int*)smem, *ws = big + ((WIN_BIG_INTS(NWV) + 3) & ~3);
	const int j = blockIdx.x;
	const wm_win_res_t r = res[j];
	const int n = r.n_a;
	if (n <= lo || r.err) return;
	const wm_win_job_t jb = jobs[j];
	wm128_t *a = anchors + r.a_off;
	const int wv = simt::wave_in_block();
	if (jb.seq_off >= 0) {
		const int n_pre = jb.n_pre < n ? jb.n_pre : n;
		for (int round = 0; round < (n_pre > 0 ? 2 : 1); ++round) {           // the seeded anchors (src/map.c:252), then the union with the handed-in ones (:833)
			wm128_t *rng = round == 0 ? a + n_pre : a;
			const int m = round == 0 ? n - n_pre : n;
			int tie = 0;
			const int cur = wmk::win_bigsort_block(NWV, rng, buf0 + r.a_off, buf1 + r.a_off, m, big, &tie);
			if (tie && tie_list) {                 // (uniform over the workgroup) the host finishes this job: this round and what follows it
				if (threadIdx.x == 0) {
					int *e = tie_list + 8 + 8 * atomicAdd(tie_list, 1);
					e[0] = j; e[1] = round; e[2] = (int)(uint32_t)((uint64_t)r.a_off & 0xffffffffu); e[3] = (int)(uint32_t)((uint64_t)r.a_off >> 32); e[4] = n; e[5] = n_pre;
				}
				return;
			}
			if (tie) { if (wv == 0) wmk::win_sort_wave<true>(rng, m, ws); }
			else if (cur >= 0) {
				const uint64_t *src = (const uint64_t*)((cur ? buf1 : buf0) + r.a_off);
				uint64_t *dst = (uint64_t*)rng;
				for (long long i = threadIdx.x; i < 2LL * m; i += 64 * NWV) dst[i] = src[i];
			}
			wmk::win_fence();
			__syncthreads();
		}
	}
	if (wv == 0) wmk::win_plan_wave(jb, j, r.a_off, n, a, cj, lists, counts, n_jobs);
}

// the jobs the host sorted (win_bigsort_kernel's tie list): their fill is planned here
__global__ __launch_bounds__(64) void win_plan_list_kernel(const wm_win_job_t *__restrict__ jobs, const wm_win_res_t *res, const wm128_t *anchors, wm_chain_job_t *cj, int *lists, int *counts,
                                                           int n_jobs, const int *tie_list)
{
	if ((int)blockIdx.x >= tie_list[0]) return;
	const int j = tie_list[8 + 8 * blockIdx.x];
	const wm_win_res_t r = res[j];
	wmk::win_plan_wave(jobs[j], j, r.a_off, r.n_a, anchors + r.a_off, cj, lists, counts, n_jobs);
}

// the fills of seedchain_kernel.h over a device-side job list (block b serves list[b]; blocks beyond *count leave)
__global__ __launch_bounds__(64) void win_chain_kernel(const wm_chain_job_t *jobs, const int *list, const int *count, const wm128_t *anchors, int *fpvt, int W)
{
	WM_SETPRIO(2);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	if ((int)blockIdx.x >= *count) return;
	const wm_chain_job_t jb = jobs[list[blockIdx.x]];
	uint64_t *sx = (uint64_t*)smem, *sy = sx + W;
	int *sf = (int*)(sy + W), *sp = sf + W, *st = sp + W;
	int *gf = fpvt + jb.a_off * 4, *gp = gf + jb.n, *gt = gp + 2 * (size_t)jb.n;
	wmk::chain_wave(jb, anchors, W, sx, sy, sf, sp, st, gf, gp, gt);
}
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void win_chain_kernel_block(const wm_chain_job_t *jobs, const int *list, const int *count, const wm128_t *anchors, int *fpvt, int W)
{
	WM_SETPRIO(2);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	if ((int)blockIdx.x >= *count) return;
	const wm_chain_job_t jb = jobs[list[blockIdx.x]];
	uint64_t *sx = (uint64_t*)smem, *sy = sx + W;
	int *sf = (int*)(sy + W), *sp = sf + W, *st = sp + W, *pub = st + W;
	int *gf = fpvt + jb.a_off * 4, *gp = gf + jb.n, *gt = gp + 2 * (size_t)jb.n;
	wmk::chain_block(jb, anchors, NWV, W, sx, sy, sf, sp, st, pub, gf, gp, gt);
}

// the dense fill with the whole predecessor window of an anchor per step (seedchain_kernel.h: chain_block_wide): 16 wavefronts x KT tiles
template <int KT>
__global__ __launch_bounds__(1024) void win_chain_kernel_wide(const wm_chain_job_t *jobs, const int *list, const int *count, const wm128_t *anchors, int *fpvt, int W)
{
	WM_SETPRIO(2);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	if ((int)blockIdx.x >= *count) return;
	const wm_chain_job_t jb = jobs[list[blockIdx.x]];
	uint64_t *sx = (uint64_t*)smem, *sy = sx + W;
	int *sf = (int*)(sy + W), *sp = sf + W, *st = sp + W, *pub = st + W;
	int *gf = fpvt + jb.a_off * 4, *gp = gf + jb.n, *gt = gp + 2 * (size_t)jb.n;
	wmk::chain_block_wide<KT>(jb, anchors, 16, W, sx, sy, sf, sp, st, pub, gf, gp, gt);
}

// src/chain.c:89-165 per job; f, p staged in LDS when the job fits (lo, lds_cap], global slab otherwise (lds_cap = 0)
__global__ __launch_bounds__(64) void win_extract_kernel(const wm_win_job_t *__restrict__ jobs, wm_win_res_t *res, wm128_t *anchors, int *fpvt, uint64_t *zu, wm128_t *bbuf, wm128_t *wbuf,
                                                          int lo, int lds_cap, uint64_t *u_pool, wm128_t *v_pool, uint64_t *pool_ctr)
{
	WM_SETPRIO(2);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	int *ws = (int*)smem;
	const int j = blockIdx.x;
	const wm_win_res_t r = res[j];
	const int n = r.n_a;
	if (n <= lo || (lds_cap > 0 && n > lds_cap) || r.err) return;
	const wm_win_job_t jb = jobs[j];
	int *gf = fpvt + r.a_off * 4, *gp = gf + n, *gv = gp + n, *gt = gv + n;
	if (lds_cap > 0) {
		int *lf = ws + ((wmk::WIN_WS_INTS + 3) & ~3), *lp = lf + lds_cap, *lv = lp + lds_cap, *lt = lv + lds_cap;
		for (int i = threadIdx.x; i < n; i += 64) { lf[i] = gf[i]; lp[i] = gp[i]; }
		simt::lds_sync();
		wmk::win_extract_wave<false>(n, jb.min_cnt, jb.min_sc, anchors + r.a_off, lf, lp, lv, lt, zu + r.a_off, bbuf + r.a_off, wbuf + r.a_off, ws, res + j, u_pool, v_pool, pool_ctr);
	} else
		wmk::win_extract_wave<true>(n, jb.min_cnt, jb.min_sc, anchors + r.a_off, gf, gp, gv, gt, zu + r.a_off, bbuf + r.a_off, wbuf + r.a_off, ws, res + j, u_pool, v_pool, pool_ctr);
}

// device side of one call: everything up to the dense result pools; the caller copies them out. slot_full: full-size minimizer slots (retry)
struct WinDev { wm_win_res_t *d_res; uint64_t *d_upool; wm128_t *d_vpool; uint64_t *d_ctr; uint64_t ctr[4]; uint32_t tot[3]; };
static int window_launch(wm_ctx_t *c, int n, const wm_window_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes, const wm128_t *pre, size_t n_pre_total,
                         int max_occ, int64_t flag, bool slot_full, WinDev &D)
{
	const int w = c->skp.w;
	(void)w;
	// host tables
	UBuf<wm_win_job_t> jb(n, c);
	UBuf<wm_sketch_job_t> sj(n, c);
	UBuf<int> ord(n, c);
	uint64_t slots = 0, mtot = 0, stage_hi = 0, pre_hi = 0;
	int bad = -1;
	for (int i = 0; i < n; ++i) {
		const wm_window_job_t &s = jobs[i];
		wm_win_job_t &d = jb[i];
		const bool has_seq = s.seq_off >= -1 && s.len > 0;
		if (s.len < 0 || s.n_pre < 0 || (s.n_pre > 0 && s.pre_off + (uint64_t)s.n_pre > n_pre_total) || s.seq_off < -2 ||
		    (s.seq_off >= 0 && (!c->d_reads || (uint64_t)s.seq_off + (uint64_t)s.len > c->reads_bytes)) ||
		    (s.seq_off == -1 && s.stage_off + (uint64_t)s.len > seqs_bytes)) { if (bad < 0) bad = i; }
		d.seq_off = has_seq ? 0 : -1; d.pre_off = s.pre_off; d.len = s.len; d.n_pre = s.n_pre; d.max_occ = max_occ; d.seed_flag = (int32_t)(flag & (0x100000 | 0x200000));
		d.max_dist_x = s.par.max_dist_x; d.min_dist_x = s.par.min_dist_x; d.max_dist_y = s.par.max_dist_y; d.bw = s.par.bw; d.max_skip = s.par.max_skip; d.max_iter = s.par.max_iter;
		d.min_cnt = s.par.min_cnt; d.min_sc = s.par.min_sc; d.gap_scale = s.par.gap_scale; d.is_cdna = s.par.is_cdna != 0;
		wm_sketch_job_t &k = sj[i];
		k.len = has_seq ? s.len : 0;
		k.cap = has_seq ? (slot_full ? s.len + 1 : s.len / 8 + 16) : 0;
		k.out_off = mtot; mtot += (uint64_t)k.cap;
		k.scratch_off = slots; slots += (uint64_t)k.len;
		k.seq_off = 0;
		if (s.seq_off == -1 && has_seq) stage_hi = std::max<uint64_t>(stage_hi, s.stage_off + (uint64_t)s.len);
		if (s.n_pre > 0) pre_hi = std::max<uint64_t>(pre_hi, s.pre_off + (uint64_t)s.n_pre);
		ord[i] = i;
	}
	if (bad >= 0) return set_err(WM_EINVAL, "window job %d: sequence / anchors outside their buffers (or wm_reads_upload missing)", bad);
	if (!(c->skp.k & 1) || c->skp.k < 2) return set_err(WM_EINVAL, "wm_window_batch needs an odd k (got %d)", c->skp.k);
	std::sort(ord.begin(), ord.end(), [&](int x, int y) { return sj[x].len != sj[y].len ? sj[x].len > sj[y].len : x < y; });     // sketch: longest first
	// device buffers
	wm_win_job_t *d_jobs = (wm_win_job_t*)arena_take(c, (size_t)n * sizeof(wm_win_job_t));
	wm_sketch_job_t *d_sj = (wm_sketch_job_t*)arena_take(c, (size_t)n * sizeof(wm_sketch_job_t));
	int *d_ord = (int*)arena_take(c, (size_t)n * 4 + 64);
	uint8_t *d_seqs = (uint8_t*)arena_take(c, stage_hi + 64);
	wm128_t *d_pre = (wm128_t*)arena_take(c, (pre_hi + 1) * sizeof(wm128_t));
	double *d_so = (double*)arena_take(c, (slots + 1) * 8);
	uint64_t *d_sx = (uint64_t*)arena_take(c, (slots + 1) * 8);
	uint32_t *d_sy = (uint32_t*)arena_take(c, (slots + 1) * 4), *d_sl = (uint32_t*)arena_take(c, (slots + 1) * 4);
	wm128_t *d_mini = (wm128_t*)arena_take(c, (mtot + 1) * sizeof(wm128_t));
	int *d_mcnt = (int*)arena_take(c, (size_t)n * 4 + 64);
	int *d_occ = (int*)arena_take(c, (mtot + 1) * 4), *d_emit = (int*)arena_take(c, (mtot + 1) * 4);
	uint32_t *d_first = (uint32_t*)arena_take(c, (mtot + 1) * 4);
	D.d_res = (wm_win_res_t*)arena_take(c, (size_t)n * sizeof(wm_win_res_t) + 64);
	wm_chain_job_t *d_cj = (wm_chain_job_t*)arena_take(c, (size_t)n * sizeof(wm_chain_job_t) + 64);
	int *d_lists = (int*)arena_take(c, (size_t)n * 4 * 4 + 64);
	static const bool ties_on_host = !(getenv("WM_WINDOW_TIES_HOST") && atoi(getenv("WM_WINDOW_TIES_HOST")) == 0);      // (0: the literal replay on the device, as until round 4; A/B)
	int *d_tie = ties_on_host ? (int*)arena_take(c, (size_t)(8 + 8 * (size_t)n) * 4) : 0;
	if (ties_on_host && !d_tie) return set_err(WM_ENOMEM, "window batch does not fit the arena");
	uint64_t *d_ctr = (uint64_t*)arena_take(c, 64);          // [0] anchors used, [1] chains in the result pool, [2] anchors in the result pool, [3] worst err (int), [4..5] the four class counts (ints)
	if (!d_jobs || !d_sj || !d_ord || !d_seqs || !d_pre || !d_so || !d_sx || !d_sy || !d_sl || !d_mini || !d_mcnt || !d_occ || !d_emit || !d_first || !D.d_res || !d_cj || !d_lists || !d_ctr)
		return set_err(WM_ENOMEM, "window batch does not fit the arena");
	int *d_counts = (int*)(d_ctr + 4);
	D.d_ctr = d_ctr;
	const size_t long_bytes = sketch_long_bytes(n, sj.data(), !slot_full, c->skp.hpc != 0);       // chunked sketch of long sequences: its tables come before the pool takes the rest
	uint8_t *d_long = long_bytes ? (uint8_t*)arena_take(c, long_bytes) : 0;
	if (long_bytes && !d_long) return set_err(WM_ENOMEM, "window batch does not fit the arena");
	// the rest of the arena is the anchor pool: 72 B per anchor (anchors 16, f|p|v|t 16, z/u 8, b 16, w 16) + the two dense result pools (24)
	const size_t left = c->arena_bytes - ((c->arena_used + 255) & ~(size_t)255);
	const uint64_t cap = left > 4096 ? (left - 4096) / 96 : 0;
	wm128_t *d_a = (wm128_t*)arena_take(c, (cap + 1) * 16);
	int *d_fpvt = (int*)arena_take(c, (cap + 1) * 16);
	uint64_t *d_zu = (uint64_t*)arena_take(c, (cap + 1) * 8);
	wm128_t *d_b = (wm128_t*)arena_take(c, (cap + 1) * 16), *d_w = (wm128_t*)arena_take(c, (cap + 1) * 16);
	D.d_upool = (uint64_t*)arena_take(c, (cap + 1) * 8);
	D.d_vpool = (wm128_t*)arena_take(c, (cap + 1) * 16);
	if (cap < 1024 || !d_a || !d_fpvt || !d_zu || !d_b || !d_w || !D.d_upool || !D.d_vpool) return set_err(WM_ENOMEM, "window batch does not fit the arena");
	if (cap >= ((uint64_t)1 << 31)) return set_err(WM_EINTERNAL, "anchor pool beyond 2^31 entries");      // (32-bit offsets in the result table)
	HIPCHK(hipMemcpyAsync(d_jobs, jb.data(), (size_t)n * sizeof(wm_win_job_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d_ord, ord.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
	if (stage_hi) HIPCHK(hipMemcpyAsync(d_seqs, seqs, stage_hi, hipMemcpyHostToDevice, c->stream));
	if (pre_hi) HIPCHK(hipMemcpyAsync(d_pre, pre, pre_hi * sizeof(wm128_t), hipMemcpyHostToDevice, c->stream));
	// staged sequences: byte offsets into d_seqs; resident ones: base indices into the packed reads, flagged (reads2bit.h)
	for (int i = 0; i < n; ++i) if (sj[i].len > 0) sj[i].seq_off = jobs[i].seq_off >= 0 ? (WM_RD_PACKED_BIT | (uint64_t)jobs[i].seq_off) : jobs[i].stage_off;
	HIPCHK(hipMemcpyAsync(d_sj, sj.data(), (size_t)n * sizeof(wm_sketch_job_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemsetAsync(D.d_res, 0, (size_t)n * sizeof(wm_win_res_t), c->stream));
	HIPCHK(hipMemsetAsync(d_ctr, 0, 64, c->stream));
	if (d_tie) HIPCHK(hipMemsetAsync(d_tie, 0, 32, c->stream));
	HIPCHK(hipEventRecord(c->ev[0], c->stream));
	if (const int rc = sketch_launch(c, n, sj.data(), d_sj, d_ord, d_seqs, d_so, d_sx, d_sy, d_sl, d_mini, d_mcnt, !slot_full, d_long, long_bytes)) return rc;
	wm_index_view_t ix = { c->d_hkey, c->d_hval, c->d_P, c->hbits, 0 };
	hipLaunchKernelGGL(win_seed_kernel, dim3(n), dim3(64), 0, c->stream, ix, d_jobs, d_sj, d_mcnt, d_mini, d_pre, d_occ, d_first, d_emit, d_a, d_ctr, cap, D.d_res, (int*)(d_ctr + 3));
	const size_t ws_bytes = (size_t)wmk::WIN_WS_PAD * 4;
	static const int kSmall = wmk::WIN_SMALL, kLarge = 4096;
	// the bulk (jobs of at most WIN_SMALL anchors: one MCAS window yields ~100) finishes in one kernel; the rest goes class by class
	HIPCHK(hipFuncSetAttribute((const void*)win_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	hipLaunchKernelGGL(win_small_kernel, dim3(n), dim3(64), (size_t)wmk::WIN_SMALL_LDS, c->stream, d_jobs, D.d_res, d_a, D.d_upool, D.d_vpool, d_ctr + 1);
	HIPCHK(hipFuncSetAttribute((const void*)win_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	hipLaunchKernelGGL(win_sort_kernel, dim3(n), dim3(64), ws_bytes + (size_t)kLarge * 16, c->stream, d_jobs, D.d_res, d_a, d_cj, d_lists, d_counts, n, kSmall, kLarge);
	{
		constexpr int NWV = 8;
		const size_t big_bytes = (size_t)((WIN_BIG_INTS(NWV) + 3) & ~3) * 4 + ws_bytes;
		hipLaunchKernelGGL(win_bigsort_kernel<NWV>, dim3(n), dim3(64 * NWV), big_bytes, c->stream, d_jobs, D.d_res, d_a, d_b, d_w, d_cj, d_lists, d_counts, n, kLarge, d_tie);
	}
	if (d_tie) {
		// large anchor sets whose keys tie: the exact order of equal keys is the reference's unstable sort's (src/ksort.h:101-151), a serial algorithm —
		// serial work belongs on the host. One small read-back per window call; the ranges travel only when there are such jobs.
		int *h_cnt = c->pin_small ? c->pin_small + 16 : 0;
		int cnt_pageable = 0;
		HIPCHK(hipMemcpyAsync(h_cnt ? h_cnt : &cnt_pageable, d_tie, 4, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(ctx_sync(c));
		const int n_tie = h_cnt ? *h_cnt : cnt_pageable;
		if (n_tie > 0) {
			UBuf<int> tl((size_t)8 * n_tie, c);
			HIPCHK(hipMemcpyAsync(tl.data(), d_tie + 8, (size_t)8 * n_tie * 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(ctx_sync(c));
			std::vector<uint64_t> off((size_t)n_tie + 1, 0);
			for (int i = 0; i < n_tie; ++i) off[i + 1] = off[i] + (uint64_t)tl[8 * i + 4];
			UBuf<wm128_t> ha((size_t)off[n_tie] + 1, c);
			for (int i = 0; i < n_tie; ++i) {
				const uint64_t a_off = (uint64_t)(uint32_t)tl[8 * i + 2] | (uint64_t)(uint32_t)tl[8 * i + 3] << 32;
				HIPCHK(hipMemcpyAsync(ha.data() + off[i], d_a + a_off, (size_t)tl[8 * i + 4] * 16, hipMemcpyDeviceToHost, c->stream));
			}
			HIPCHK(ctx_sync(c));
			{
				WM_SITE("window.tie_sort");
				wm::parallel_for(c->host_threads, (size_t)n_tie, [&](size_t i) {
					wm::m128 *a0 = (wm::m128*)(ha.data() + off[i]);
					const int nn = tl[8 * i + 4], round = tl[8 * i + 1], n_pre = std::min(tl[8 * i + 5], nn);
					if (round == 0) wm::radix_sort_128x(a0 + n_pre, a0 + nn);                    // the seeded anchors (src/map.c:252) ...
					if (round == 1 || n_pre > 0) wm::radix_sort_128x(a0, a0 + nn);               // ... then the union with the handed-in ones (:833)
				});
			}
			for (int i = 0; i < n_tie; ++i) {
				const uint64_t a_off = (uint64_t)(uint32_t)tl[8 * i + 2] | (uint64_t)(uint32_t)tl[8 * i + 3] << 32;
				HIPCHK(hipMemcpyAsync(d_a + a_off, ha.data() + off[i], (size_t)tl[8 * i + 4] * 16, hipMemcpyHostToDevice, c->stream));
			}
			hipLaunchKernelGGL(win_plan_list_kernel, dim3(n_tie), dim3(64), 0, c->stream, d_jobs, D.d_res, d_a, d_cj, d_lists, d_counts, n, d_tie);
			HIPCHK(ctx_sync(c));                 // (the staging buffers above are released at the end of this block)
		}
	}
	{   // the fill, per class list: 0 dense (8 waves, W 4096) | 1 large sparse (1 wave, W 1024) | 2 n <= 1024 | 3 n <= 256
		HIPCHK(hipFuncSetAttribute((const void*)win_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		constexpr int NWV = 8;
		// WM_CHAIN_WIDE=0: the round-5 dense fill (8 wavefronts, 512 predecessors per step); default: 16 wavefronts x 5 tiles = a whole max_iter window per step
		static const bool wide = !(getenv("WM_CHAIN_WIDE") && atoi(getenv("WM_CHAIN_WIDE")) == 0);
		if (wide) {
			HIPCHK(hipFuncSetAttribute((const void*)win_chain_kernel_wide<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
			hipLaunchKernelGGL(win_chain_kernel_wide<5>, dim3(n), dim3(1024), (size_t)4096 * 28 + 80 * 69 * 4 + 64, c->stream, d_cj, d_lists, d_counts, d_a, d_fpvt, 4096);
		} else {
			HIPCHK(hipFuncSetAttribute((const void*)win_chain_kernel_block<NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
			hipLaunchKernelGGL(win_chain_kernel_block<NWV>, dim3(n), dim3(64 * NWV), (size_t)4096 * 28 + NWV * 69 * 4 + 64, c->stream, d_cj, d_lists, d_counts, d_a, d_fpvt, 4096);
		}
		hipLaunchKernelGGL(win_chain_kernel, dim3(n), dim3(64), (size_t)1024 * 28, c->stream, d_cj, d_lists + n, d_counts + 1, d_a, d_fpvt, 1024);
		hipLaunchKernelGGL(win_chain_kernel, dim3(n), dim3(64), (size_t)1024 * 28, c->stream, d_cj, d_lists + 2 * (size_t)n, d_counts + 2, d_a, d_fpvt, 1024);
		// (class 3, at most 256 anchors: served by win_small_kernel)
	}
	HIPCHK(hipFuncSetAttribute((const void*)win_extract_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	hipLaunchKernelGGL(win_extract_kernel, dim3(n), dim3(64), ws_bytes + (size_t)kLarge * 16, c->stream, d_jobs, D.d_res, d_a, d_fpvt, d_zu, d_b, d_w, kSmall, kLarge, D.d_upool, D.d_vpool, d_ctr + 1);
	hipLaunchKernelGGL(win_extract_kernel, dim3(n), dim3(64), ws_bytes, c->stream, d_jobs, D.d_res, d_a, d_fpvt, d_zu, d_b, d_w, kLarge, 0, D.d_upool, D.d_vpool, d_ctr + 1);
	HIPCHK(hipEventRecord(c->ev[1], c->stream));
	HIPCHK(hipGetLastError());
	uint64_t *h_ctr = c->pin_small ? (uint64_t*)(c->pin_small + 8) : D.ctr;
	HIPCHK(hipMemcpyAsync(h_ctr, d_ctr, 32, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(ctx_sync(c));                  // (also: the host tables above are read by the copies until here)
	memcpy(D.ctr, h_ctr, 32);
	D.tot[0] = (uint32_t)D.ctr[1]; D.tot[1] = (uint32_t)D.ctr[2]; D.tot[2] = (uint32_t)(D.ctr[3] & 0xffffffffu);
	float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->aux_ms += ms;
	return WM_OK;
}

// copies the result table and the two dense pools of a launched call to the host (pools sized by the caller from D.tot)
static int window_fetch(wm_ctx_t *c, const WinDev &D, int n, wm_window_res_t *res, uint64_t *u_pool, wm128_t *a_pool)
{
	UBuf<wm_win_res_t> hr(n, c);
	HIPCHK(hipMemcpyAsync(hr.data(), D.d_res, (size_t)n * sizeof(wm_win_res_t), hipMemcpyDeviceToHost, c->stream));
	if (D.tot[0]) HIPCHK(hipMemcpyAsync(u_pool, D.d_upool, (size_t)D.tot[0] * 8, hipMemcpyDeviceToHost, c->stream));
	if (D.tot[1]) HIPCHK(hipMemcpyAsync(a_pool, D.d_vpool, (size_t)D.tot[1] * 16, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(ctx_sync(c));
	for (int i = 0; i < n; ++i) {
		const wm_win_res_t &r = hr[i];
		res[i].n_anchors = r.n_a; res[i].rep_len = r.rep_len; res[i].n_mini = r.n_mini; res[i].n_u = r.n_u; res[i].n_v = r.n_v; res[i].u_off = r.u_out; res[i].a_off = r.v_out;
	}
	return WM_OK;
}
// verdict of a launched call: 0 = fetch, 1 = launch again with full-size minimizer slots, < 0 = error
static int window_verdict(const WinDev &D, int round)
{
	if (D.tot[2] == 2) return set_err(WM_ENOMEM, "window batch does not fit the arena (anchor pool)");
	if (D.tot[2] == 1) return round == 0 ? 1 : set_err(WM_EINTERNAL, "minimizer slot overflow at full size");
	return 0;
}

extern "C" int wm_window_batch(wm_ctx_t *c, int n, const wm_window_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes, const wm128_t *pre, size_t n_pre_total,
                               int max_occ, int64_t flag, wm_window_res_t *res, uint64_t *u_pool, size_t u_cap, size_t *u_used, wm128_t *a_pool, size_t a_cap, size_t *a_used)
try {
	if (u_used) *u_used = 0;
	if (a_used) *a_used = 0;
	if (!c || !c->have_index) return set_err(WM_EINVAL, "wm_index_upload has not been called on this context");
	if (n <= 0) return WM_OK;
	if (!jobs || !res) return set_err(WM_EINVAL, "null argument");
	HIPCHK(hipSetDevice(c->device));
	c->aux_ms = 0;
	for (int round = 0; round < 2; ++round) {
		ArenaMark mark(c);
		WinDev D;
		int rc = window_launch(c, n, jobs, seqs, seqs_bytes, pre, n_pre_total, max_occ, flag, round == 1, D);
		if (rc) return rc;
		rc = window_verdict(D, round);
		if (rc < 0) return rc;
		if (rc == 1) continue;
		if (u_used) *u_used = D.tot[0];
		if (a_used) *a_used = D.tot[1];
		if (D.tot[0] > u_cap || D.tot[1] > a_cap) return set_err(WM_ENOMEM, "result pools too small: need %u chains and %u anchors", D.tot[0], D.tot[1]);
		if ((D.tot[0] && !u_pool) || (D.tot[1] && !a_pool)) return set_err(WM_EINVAL, "null result pool");
		return window_fetch(c, D, n, res, u_pool, a_pool);
	}
	return set_err(WM_EINTERNAL, "window retry did not converge");
}
catch (const std::bad_alloc &) { return set_err(WM_ENOMEM, "out of host memory"); }

// ======================================================================================================
// GpuOps: the product implementation of the mapper's device operations
// ======================================================================================================
// one device context (stream + arena + staging slab) worth of batched operations; GpuOps below hands the contexts out
struct GpuOpsCtx {
	wm_ctx_t *c;
	uint64_t cells = 0;
	double ksw_us = 0, aux_us = 0;
	double t_pack = 0, t_prep = 0, t_run = 0, t_fetch = 0, t_unpack = 0, t_sketch = 0, t_seed = 0, t_chain = 0;
	std::string error;
	void fail(const char *what) { if (error.empty()) error = std::string(what) + ": " + g_err; }
	bool resident = false;                      // the mini-batch's read codes are on the device (load_reads)
	void sketch_batch(int, int, std::vector<wm::SketchReq*> &reqs)
	{
		const int n = (int)reqs.size();
		std::vector<uint64_t> off(n), ooff(n);
		std::vector<int32_t> len(n), cnt(n);
		std::vector<uint8_t> res(n, 0);
		size_t tot = 0, tot_all = 0;                // bytes to stage (sequences that are not resident); all bases
		for (int i = 0; i < n; ++i) {
			len[i] = reqs[i]->len; tot_all += reqs[i]->len;
			if (resident && reqs[i]->dev_off >= 0) { res[i] = 1; off[i] = (uint64_t)reqs[i]->dev_off; }
			else { off[i] = tot; tot += reqs[i]->len; }
		}
		UBuf<uint8_t> seqs(tot + 1, c);
		WM_SITE("sketch.stage");
		wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) { if (!res[i]) memcpy(seqs.data() + off[i], reqs[i]->seq, reqs[i]->len); });
		UBuf<wm128_t> out(tot_all / 8 + (size_t)17 * n + 64, c);          // the batch tries len/8 + 16 slots per sequence first
		const double ts = now_ms();
		int rc = sketch_batch_impl(c, n, seqs.data(), tot, off.data(), len.data(), res.data(), out.data(), out.size(), ooff.data(), cnt.data());
		if (rc == WM_ENOMEM && strstr(g_err, "minimizer output pool")) {   // pathological density: redo with one slot per base
			UBuf<wm128_t> big(tot_all + n + 1);
			rc = sketch_batch_impl(c, n, seqs.data(), tot, off.data(), len.data(), res.data(), big.data(), big.size(), ooff.data(), cnt.data());
			if (rc) { fail("sketch"); return; }
			t_sketch += now_ms() - ts;
			aux_us += c->aux_ms * 1e3;
			wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) { reqs[i]->mini.assign(big.begin() + ooff[i], big.begin() + ooff[i] + cnt[i]); });
			return;
		}
		if (rc) { fail("sketch"); return; }
		t_sketch += now_ms() - ts;
		aux_us += c->aux_ms * 1e3;
		WM_SITE("sketch.unpack");
		wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) { reqs[i]->mini.assign(out.begin() + ooff[i], out.begin() + ooff[i] + cnt[i]); });
	}
	void seed_batch(std::vector<wm::SeedReq*> &reqs)
	{
		const int n = (int)reqs.size();
		// one launch per (max_occ, flag) class; in practice a single class
		std::vector<uint64_t> moff(n), ooff(n);
		std::vector<int32_t> nm(n), ql(n), na(n), rl(n);
		size_t tot = 0;
		for (int i = 0; i < n; ++i) { moff[i] = tot; nm[i] = reqs[i]->n_mini; ql[i] = reqs[i]->qlen; tot += reqs[i]->n_mini; }
		UBuf<wm128_t> mini(tot + 1, c);
		WM_SITE("seed.pack");
		wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) { memcpy(mini.data() + moff[i], reqs[i]->mini, (size_t)reqs[i]->n_mini * sizeof(wm128_t)); });
		size_t cap = tot * 3 + 4096;               // anchors per minimizer: ~1.15 on the bench reference; repeats are retried at 8x
		for (int attempt = 0; attempt < 6; ++attempt) {
			UBuf<wm128_t> out(cap, c);
			const double ts = now_ms();
			const int rc = wm_seed_batch(c, n, mini.data(), moff.data(), nm.data(), ql.data(), reqs[0]->max_occ, reqs[0]->flag, out.data(), out.size(), ooff.data(), na.data(), rl.data());
			if (rc == WM_ENOMEM && strstr(g_err, "anchor output pool")) { cap *= 8; continue; }
			if (rc) { fail("seed"); return; }
			t_seed += now_ms() - ts;
			aux_us += c->aux_ms * 1e3;
			wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) { reqs[i]->a.assign(out.begin() + ooff[i], out.begin() + ooff[i] + na[i]); reqs[i]->rep_len = rl[i]; });
			return;
		}
		fail("seed (anchor pool)");
	}
	void chain_batch(std::vector<wm::ChainReq*> &reqs)
	{
		const int n = (int)reqs.size();
		std::vector<uint64_t> aoff(n), uoff(n);
		std::vector<int32_t> na(n), nu(n), nv(n);
		std::vector<wm_chain_par_t> par(n);
		size_t tot = 0;
		for (int i = 0; i < n; ++i) {
			aoff[i] = tot; na[i] = (int)reqs[i]->a.size(); tot += reqs[i]->a.size();
			wm::ChainReq &r = *reqs[i];
			par[i] = { r.max_dist_x, r.min_dist_x, r.max_dist_y, r.bw, r.max_skip, r.max_iter, r.min_cnt, r.min_sc, r.gap_scale, r.is_cdna ? 1 : 0 };
		}
		UBuf<wm128_t> a(tot + 1, c);
		UBuf<uint64_t> u(tot + 1, c);
		WM_SITE("chain.pack");
		wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) { memcpy(a.data() + aoff[i], reqs[i]->a.data(), reqs[i]->a.size() * sizeof(wm128_t)); });
		const double ts = now_ms();
		if (wm_chain_batch(c, n, a.data(), aoff.data(), na.data(), par.data(), u.data(), uoff.data(), nu.data(), nv.data())) { fail("chain"); return; }
		t_chain += now_ms() - ts;
		aux_us += c->aux_ms * 1e3;
		WM_SITE("chain.unpack");
		wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
			reqs[i]->u.assign(u.begin() + uoff[i], u.begin() + uoff[i] + nu[i]);
			reqs[i]->a.assign(a.begin() + aoff[i], a.begin() + aoff[i] + nv[i]);
		});
	}
	// the whole window on the device (wm_window_batch's machinery; the result pools are sized after the launch, in the pinned slab)
	void window_batch(std::vector<wm::WindowReq*> &reqs)
	{
		const int n = (int)reqs.size();
		if (n == 0) return;
		for (int i = 1; i < n; ++i)             // collect_seed_hits takes one (max_occ, flag) per call: requests that differ go in their own call
			if (reqs[i]->max_occ != reqs[0]->max_occ || reqs[i]->flag != reqs[0]->flag) {
				std::vector<wm::WindowReq*> same, rest;
				for (wm::WindowReq *r : reqs) (r->max_occ == reqs[0]->max_occ && r->flag == reqs[0]->flag ? same : rest).push_back(r);
				window_batch(same);
				if (error.empty()) window_batch(rest);
				return;
			}
		const double ts = now_ms();
		UBuf<wm_window_job_t> jobs(n, c);
		size_t stage = 0, npre = 0;
		for (int i = 0; i < n; ++i) {
			const wm::WindowReq &r = *reqs[i];
			wm_window_job_t &j = jobs[i];
			j.len = r.len; j.n_pre = (int32_t)r.pre.size(); j.pre_off = npre; npre += r.pre.size();
			j.stage_off = 0;
			if (r.len <= 0) j.seq_off = -2;
			else if (resident && r.dev_off >= 0) j.seq_off = r.dev_off;
			else { j.seq_off = -1; j.stage_off = stage; stage += (size_t)r.len; }
			j.par = { r.max_dist_x, r.min_dist_x, r.max_dist_y, r.bw, r.max_skip, r.max_iter, r.min_cnt, r.min_sc, r.gap_scale, r.is_cdna ? 1 : 0 };
		}
		UBuf<uint8_t> seqs(stage + 1, c);
		UBuf<wm128_t> pre(npre + 1, c);
		UBuf<wm_window_res_t> res(n, c);
		WM_SITE("window.stage");
		wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
			const wm::WindowReq &r = *reqs[i];
			if (jobs[i].seq_off == -1) memcpy(seqs.data() + jobs[i].stage_off, r.seq, (size_t)r.len);
			if (!r.pre.empty()) memcpy(pre.data() + jobs[i].pre_off, r.pre.data(), r.pre.size() * sizeof(wm128_t));
		});
		if (hipSetDevice(c->device) != hipSuccess) { error = "hipSetDevice failed"; return; }
		c->aux_ms = 0;
		for (int round = 0; round < 2; ++round) {
			ArenaMark mark(c);
			WinDev D;
			int rc = window_launch(c, n, jobs.data(), seqs.data(), stage, pre.data(), npre, reqs[0]->max_occ, reqs[0]->flag, round == 1, D);
			if (!rc) rc = window_verdict(D, round);
			if (rc < 0) { fail("window"); return; }
			if (rc == 1) continue;
			UBuf<uint64_t> up((size_t)D.tot[0] + 1, c);
			UBuf<wm128_t> ap((size_t)D.tot[1] + 1, c);
			if (window_fetch(c, D, n, res.data(), up.data(), ap.data())) { fail("window"); return; }
			t_sketch += now_ms() - ts;
			aux_us += c->aux_ms * 1e3;
			WM_SITE("window.unpack");
			wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
				wm::WindowReq &r = *reqs[i];
				const wm_window_res_t &o = res[i];
				r.rep_len = o.rep_len; r.n_anchors = o.n_anchors;
				r.u.assign(up.begin() + o.u_off, up.begin() + o.u_off + o.n_u);
				r.a.resize((size_t)o.n_v);
				if (o.n_v) memcpy(r.a.data(), ap.data() + o.a_off, (size_t)o.n_v * sizeof(wm128_t));
			});
			return;
		}
		error = "window retry did not converge";
	}
	void ksw_batch(const wm_ksw_score_t &sc, std::vector<wm::KswReq*> &reqs)
	{
		const double t0 = now_ms();
		const int n = (int)reqs.size();
		size_t cap = 16;
		bool all_res = resident;
		for (int i = 0; i < n && all_res; ++i) all_res = reqs[i]->resident();
		for (int i = 0; i < n; ++i) cap += (size_t)reqs[i]->ql + reqs[i]->tl + 2;
		std::vector<wm_ksw_result_t> res(n);
		std::vector<wm_zd_t> zd;
		UBuf<uint32_t> pool(cap, c);
		size_t used = 0;
		c->acc_cells = 0; c->t_prep = c->t_run = c->t_fetch = 0;
		double t1;
		if (all_res) {          // operands as positions in the resident reads / packed reference: nothing is copied or shipped per alignment
			UBuf<wm_ksw_pos_t> jobs(n + 1, c);
			WM_SITE("ksw.pos_jobs");
			wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
				const wm::KswReq &r = *reqs[i];
				wm_ksw_pos_t &j = jobs[i];
				j.qwin_off = r.qwin_off; j.qwin_len = r.qwin_len; j.q_pos = r.q_pos; j.rid = r.rid; j.t_pos = r.t_pos; j.qlen = r.ql; j.tlen = r.tl;
				j.w = r.w; j.zdrop = r.zdrop; j.end_bonus = r.end_bonus; j.flag = r.flag | (r.want_zd && r.step == 1 ? WM_KSW_F_ZDWALK : 0); j.step = (int8_t)r.step; j.has_n = r.has_n; memset(j.pad, 0, sizeof(j.pad));
			});
			t1 = now_ms();
			bool any_zd = false;
			for (int i = 0; i < n && !any_zd; ++i) any_zd = reqs[i]->want_zd && reqs[i]->step == 1;
			if (any_zd) {           // the z-drop scans of the gap fills come back with the alignments (ksw_zdwalk_kernel)
				zd.resize(n);
				if (wm_ksw_batch_pos_zd(c, &sc, n, jobs.data(), res.data(), pool.data(), cap, &used, zd.data())) { fail("ksw"); return; }
			} else if (wm_ksw_batch_pos(c, &sc, n, jobs.data(), res.data(), pool.data(), cap, &used)) { fail("ksw"); return; }
		} else {                // host views -> one byte slab
			std::vector<wm_ksw_job_t> jobs(n);
			size_t tot = 0;
			for (int i = 0; i < n; ++i) {
				wm::KswReq &r = *reqs[i];
				jobs[i].q_off = (uint32_t)tot; tot += (size_t)r.ql;
				jobs[i].t_off = (uint32_t)tot; tot += (size_t)r.tl;
				jobs[i].qlen = r.ql; jobs[i].tlen = r.tl;
				jobs[i].w = r.w; jobs[i].zdrop = r.zdrop; jobs[i].end_bonus = r.end_bonus; jobs[i].flag = r.flag;
			}
			if (tot >= ((size_t)1 << 32)) { error = "ksw batch exceeds 4 GB of sequence"; return; }
			UBuf<uint8_t> seqs(tot + 1, c);
			wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) { reqs[i]->copy_query(seqs.data() + jobs[i].q_off); reqs[i]->copy_target(seqs.data() + jobs[i].t_off); });
			t1 = now_ms();
			if (wm_ksw_batch(c, &sc, n, jobs.data(), seqs.data(), tot, res.data(), pool.data(), cap, &used)) { fail("ksw"); return; }
		}
		const double t2 = now_ms();
		ksw_us += c->last_ms * 1e3;
		cells += c->acc_cells;
		WM_SITE("ksw.unpack");
		wm::parallel_for(c->host_threads, (size_t)n, [&](size_t i) {
			reqs[i]->ez = res[i];
			reqs[i]->cigar.assign(pool.begin() + res[i].cig_off, pool.begin() + res[i].cig_off + res[i].n_cigar);
			reqs[i]->has_zd = !zd.empty() && reqs[i]->want_zd && reqs[i]->step == 1;
			if (reqs[i]->has_zd) reqs[i]->zd = zd[i];
		});
		t_pack += t1 - t0; t_unpack += now_ms() - t2; t_prep += c->t_prep; t_run += c->t_run; t_fetch += c->t_fetch;
	}
	// splice mode (src/align.c:326-327): the requests through wm_ksw_exts2_batch, in groups whose unbanded traceback matrices fit the arena
	void exts2_batch(const wm_ksw_score_t &sc, int noncan, int junc_bonus, std::vector<wm::KswReq*> &reqs)
	{
		const size_t budget = (size_t)(c->arena_bytes * 0.6);
		const int n = (int)reqs.size();
		for (int i0 = 0; i0 < n;) {
			size_t need = 0, tot = 0, cap = 16;
			int i1 = i0;
			for (; i1 < n; ++i1) {
				const wm::KswReq &r = *reqs[i1];
				const size_t n_col = (size_t)((((r.ql < r.tl ? r.ql : r.tl) + 15) / 16 + 1) * 16);
				const size_t b = ((size_t)r.ql + r.tl) * (n_col + 10) + 16 * (size_t)r.tl + 1024;      // traceback + CIGAR slots + operands + row state
				if (i1 > i0 && (need + b > budget || tot + r.ql + r.tl >= ((size_t)1 << 31))) break;
				need += b; tot += (size_t)r.ql + r.tl; cap += (size_t)r.ql + r.tl + 2;
			}
			const int m = i1 - i0;
			std::vector<wm_ksw_job_t> jobs(m);
			std::vector<wm_ksw_result_t> res(m);
			std::vector<uint8_t> seqs(tot + 1), junc;
			std::vector<uint32_t> pool(cap);
			size_t off = 0, used = 0;
			bool any_junc = false;
			for (int i = 0; i < m; ++i) any_junc |= !reqs[i0 + i]->junc.empty();
			if (any_junc) junc.assign(tot + 1, 0);                     // parallel to seqs: junction bits at the targets (mm_idx_bed_junc, src/index.c:768-803)
			for (int i = 0; i < m; ++i) {
				wm::KswReq &r = *reqs[i0 + i];
				jobs[i].q_off = (uint32_t)off; off += (size_t)r.ql;
				jobs[i].t_off = (uint32_t)off; off += (size_t)r.tl;
				jobs[i].qlen = r.ql; jobs[i].tlen = r.tl; jobs[i].w = -1; jobs[i].zdrop = r.zdrop; jobs[i].end_bonus = 0; jobs[i].flag = r.flag;
			}
			wm::parallel_for(c->host_threads, (size_t)m, [&](size_t i) {
				const wm::KswReq &r = *reqs[i0 + i];
				r.copy_query(seqs.data() + jobs[i].q_off); r.copy_target(seqs.data() + jobs[i].t_off);
				if (!r.junc.empty()) memcpy(junc.data() + jobs[i].t_off, r.junc.data(), r.junc.size());
			});
			if (wm_ksw_exts2_batch(c, &sc, noncan, junc_bonus, m, jobs.data(), seqs.data(), tot, any_junc ? junc.data() : 0, res.data(), pool.data(), cap, &used)) { fail("ksw_exts2"); return; }
			for (int i = 0; i < m; ++i) {
				reqs[i0 + i]->ez = res[i];
				reqs[i0 + i]->cigar.assign(pool.begin() + res[i].cig_off, pool.begin() + res[i].cig_off + res[i].n_cigar);
			}
			i0 = i1;
		}
	}
};

// The product's DeviceOps: a pool of device contexts. Every batched call borrows a free context (its own HIP stream, arena and pinned
// slab), so up to max_inflight() batches — of the same or of different operations — are on the device at once, issued by different
// host threads (wm_fiber.h).
struct GpuOps {                          // the device contexts of a mapper, shared by its (at most WM_MAX_SLOTS concurrent) mapping calls: see CallOps
	std::vector<GpuOpsCtx> ctxs;
	std::vector<int> free_;
	std::mutex mu;
	std::condition_variable cv;
	void init(const std::vector<wm_ctx_t*> &cs) { ctxs.resize(cs.size()); for (size_t i = 0; i < cs.size(); ++i) { ctxs[i].c = cs[i]; free_.push_back((int)i); } }
	int max_inflight() const { return (int)ctxs.size(); }
	bool waits_asleep() const { return getenv("WM_SPIN_SYNC") == 0; }
	// The read codes of a mini-batch go to the device once. Up to WM_MAX_SLOTS mini-batches can be in flight (concurrent mapping calls, one slot each): one
	// allocation of WM_MAX_SLOTS slabs, owned by the first context and aliased by the others (one device); a call's offsets start at slot * slab.
	std::mutex reads_mu;
	hipStream_t up_stream = 0;                  // uploads of the mini-batches' read codes
	uint64_t *stage[WM_MAX_SLOTS] = { 0 }; size_t stage_words[WM_MAX_SLOTS] = { 0 }; bool stage_pinned[WM_MAX_SLOTS] = { false };      // host staging of the packed codes, per slot
	~GpuOps()
	{
		if (up_stream) hipStreamDestroy(up_stream);
		for (int i = 0; i < WM_MAX_SLOTS; ++i) if (stage[i]) { if (stage_pinned[i]) hipHostFree(stage[i]); else free(stage[i]); }
	}
	size_t slab = 0;
	int n_slabs = 0;
	std::atomic<int> slots_hint{0};             // mini-batches the caller keeps in flight: wm_mapper_set_slots, the lanes of wm_map_file[_multi], or the highest slot seen + 1
	bool slot_busy[WM_MAX_SLOTS] = { false };
	bool load_reads(const uint8_t *codes, size_t n, int slot, int64_t *base, std::string &err)
	{
		*base = 0;
		const bool off = getenv("WM_NO_RESIDENT") != 0;            // A/B switch: per-request staging as before
		if (off || ctxs.empty() || slot < 0 || slot >= WM_MAX_SLOTS) return false;
		wm_ctx_t *c0 = ctxs[0].c;
		std::lock_guard<std::mutex> lk(reads_mu);
		if (hipSetDevice(c0->device) != hipSuccess) return false;
		if (slot + 1 > slots_hint.load()) slots_hint = slot + 1;
		if (n + 256 > slab || slot >= n_slabs || !c0->d_reads || !c0->owns_reads) {
			for (int o = 0; o < WM_MAX_SLOTS; ++o) if (o != slot && slot_busy[o]) {      // another mini-batch lives in the allocation: this one is served from its host views
				static std::atomic<bool> told(false);
				if (!told.exchange(true)) fprintf(stderr, "[wmgpu] mini-batch on slot %d is served from host views (no resident slab: %d slab(s) of %zu bases, another slot busy); "
				                                  "tell the mapper how many mini-batches are in flight (wm_mapper_set_slots / WM_READ_SLABS)\n", slot, n_slabs, slab);
				return false;
			}
			if (c0->d_reads && c0->owns_reads) hipFree(c0->d_reads);
			c0->d_reads = 0; c0->owns_reads = false; slab = 0;
			const size_t want = (n + n / 8 + (1 << 20) + 255) & ~(size_t)255;
			// slabs for the mini-batches that can be in flight: WM_READ_SLABS, else the lanes of wm_map_file (WM_MAP_LANES), at least 2 (four were
			// allocated whatever the caller used — 4.5 GB for 1-Gbase mini-batches); a call on a slot beyond them is served from its host views
			n_slabs = std::max(2, std::min((int)WM_MAX_SLOTS, getenv("WM_READ_SLABS") ? atoi(getenv("WM_READ_SLABS")) : std::max(slots_hint.load(), getenv("WM_MAP_LANES") ? atoi(getenv("WM_MAP_LANES")) : 2)));
			if (slot >= n_slabs) n_slabs = slot + 1;
			if (reads_alloc(c0, (size_t)n_slabs * want) != WM_OK) return false;              // (bases: 2 bits + 1 ambiguity bit each, reads2bit.h)
			c0->owns_reads = true; c0->reads_bytes = (size_t)n_slabs * want; slab = want;
			for (size_t i = 1; i < ctxs.size(); ++i) {
				wm_ctx_t *c = ctxs[i].c;
				if (c->owns_reads && c->d_reads) hipFree(c->d_reads);
				c->d_reads = c0->d_reads; c->d_reads_nm = c0->d_reads_nm; c->reads_bytes = c0->reads_bytes; c->reads_cap = 0; c->owns_reads = false;
			}
		}
		// the mini-batch's codes are packed on the host — 2 bits per base + 1 ambiguity bit, 0.375 B per base across PCIe instead of one byte — into this slot's
		// pinned staging buffer, by a few threads over disjoint 64-base-aligned ranges (slab is a multiple of 256 bases: the slot's words are its own)
		const size_t pkw = wm_pk_words(n), nmw = wm_nm_words(n);
		if (stage_words[slot] < pkw + nmw) {
			if (stage[slot]) { if (stage_pinned[slot]) hipHostFree(stage[slot]); else free(stage[slot]); }
			stage[slot] = 0; stage_words[slot] = 0;
			const size_t want_w = pkw + nmw + (pkw + nmw) / 8 + 1024;
			stage_pinned[slot] = hipHostMalloc((void**)&stage[slot], want_w * 8, hipHostMallocDefault) == hipSuccess;
			if (!stage_pinned[slot]) { (void)hipGetLastError(); stage[slot] = (uint64_t*)malloc(want_w * 8); }
			if (!stage[slot]) { err = "no host memory for the packed reads"; return false; }
			stage_words[slot] = want_w;
		}
		uint64_t *h_pk = stage[slot], *h_nm = stage[slot] + pkw;
		{
			const size_t CH = (size_t)1 << 22;                     // bases per range (a multiple of 64)
			const size_t n_ch = (n + CH - 1) / CH;
			const int nt = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(8, (size_t)wm::usable_cores()), n_ch));
			wm::parallel_for(nt, n_ch, [&](size_t ci) { const size_t at = ci * CH, len = std::min(CH, n - at); wm_pack_blocks(codes + at, len, h_pk + at / 32, h_nm + at / 64); });
			for (size_t i = 2 * ((n + 63) / 64); i < pkw; ++i) h_pk[i] = 0;
			for (size_t i = (n + 63) / 64; i < nmw; ++i) h_nm[i] = 0;
		}
		// (a stream of its own, not the NULL stream: a NULL-stream copy waits for every blocking stream of the device and makes them wait for it)
		if (!up_stream && hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking) != hipSuccess) { up_stream = 0; (void)hipGetLastError(); }
		uint64_t *d_pk = c0->d_reads + (size_t)slot * slab / 32, *d_nm = c0->d_reads_nm + (size_t)slot * slab / 64;
		if (n && (up_stream ? (hipMemcpyAsync(d_pk, h_pk, pkw * 8, hipMemcpyHostToDevice, up_stream) != hipSuccess || hipMemcpyAsync(d_nm, h_nm, nmw * 8, hipMemcpyHostToDevice, up_stream) != hipSuccess ||
		                       hipStreamSynchronize(up_stream) != hipSuccess)
		                    : (hipMemcpy(d_pk, h_pk, pkw * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_nm, h_nm, nmw * 8, hipMemcpyHostToDevice) != hipSuccess))) {
			err = std::string("reads upload: ") + hipGetErrorString(hipGetLastError()); return false;
		}
		for (GpuOpsCtx &x : ctxs) x.resident = true;
		slot_busy[slot] = true;
		*base = (int64_t)((size_t)slot * slab);
		return true;
	}
	void release_reads(int slot) { std::lock_guard<std::mutex> lk(reads_mu); if (slot >= 0 && slot < WM_MAX_SLOTS) slot_busy[slot] = false; }
	// One batched call on a free context. A context belongs to exactly one call while it is out of the free list, so whatever the call leaves in
	// the context's `error` is ITS error: it moves into the sink of the mapping call that issued the batch before the context is handed back
	// (two mapping calls share the contexts, wm_map_reads_slot). A mapping call that has failed issues nothing more.
	template <class F> void with(wm::ErrorSink &sink, F f)
	{
		{ std::lock_guard<std::mutex> lk(sink.mu); if (!sink.msg.empty()) return; }
		int i;
		{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !free_.empty(); }); i = free_.back(); free_.pop_back(); }
		ctxs[i].error.clear();
		try { f(ctxs[i]); }
		catch (const std::exception &e) { if (ctxs[i].error.empty()) ctxs[i].error = std::string("batched device call: ") + e.what(); }
		if (!ctxs[i].error.empty()) { sink.put(ctxs[i].error); ctxs[i].error.clear(); }
		{ std::lock_guard<std::mutex> lk(mu); free_.push_back(i); }
		cv.notify_one();
	}
	// a batch whose buffers do not fit the context's arena is served in halves (recursively): the hub sizes batches by demand, not by HBM
	template <class R, class F> static void run_split(GpuOpsCtx &x, std::vector<R*> &reqs, F f)
	{
		if (!x.error.empty()) return;          // an earlier part of this call failed for good: nothing more is attempted (and nothing is cleared)
		f(reqs);
		if (x.error.empty() || reqs.size() < 2 || x.error.find("does not fit the arena") == std::string::npos) return;
		x.error.clear();                       // (set by THIS call: the context was clean on entry)
		std::vector<R*> a(reqs.begin(), reqs.begin() + reqs.size() / 2), b(reqs.begin() + reqs.size() / 2, reqs.end());
		run_split(x, a, f);
		if (x.error.empty()) run_split(x, b, f);
	}
	void sketch_batch(wm::ErrorSink &e, int w, int k, std::vector<wm::SketchReq*> &reqs) { with(e, [&](GpuOpsCtx &x) { run_split(x, reqs, [&](std::vector<wm::SketchReq*> &part) { x.sketch_batch(w, k, part); }); }); }
	void seed_batch(wm::ErrorSink &e, std::vector<wm::SeedReq*> &reqs) { with(e, [&](GpuOpsCtx &x) { run_split(x, reqs, [&](std::vector<wm::SeedReq*> &part) { x.seed_batch(part); }); }); }
	void chain_batch(wm::ErrorSink &e, std::vector<wm::ChainReq*> &reqs) { with(e, [&](GpuOpsCtx &x) { run_split(x, reqs, [&](std::vector<wm::ChainReq*> &part) { x.chain_batch(part); }); }); }
	void ksw_batch(wm::ErrorSink &e, const wm_ksw_score_t &sc, std::vector<wm::KswReq*> &reqs) { with(e, [&](GpuOpsCtx &x) { x.ksw_batch(sc, reqs); }); }
	void exts2_batch(wm::ErrorSink &e, const wm_ksw_score_t &sc, int noncan, int junc_bonus, std::vector<wm::KswReq*> &reqs) { with(e, [&](GpuOpsCtx &x) { x.exts2_batch(sc, noncan, junc_bonus, reqs); }); }
	// collect_seed_hits takes one (max_occ, flag) per call: the mapper's requests of one mapping call all share them
	void window_batch(wm::ErrorSink &e, std::vector<wm::WindowReq*> &reqs) { with(e, [&](GpuOpsCtx &x) { run_split(x, reqs, [&](std::vector<wm::WindowReq*> &part) { x.window_batch(part); }); }); }
};

// What ONE mapping call hands to wm::map_batch: the shared contexts behind it, and the call's own error sink — the first failed batch of this
// call fails this call and no other (a per-context error field let the call that finished first take, and clear, its neighbour's).
struct CallOps : wm::DeviceOps {
	GpuOps &g;
	wm::ErrorSink err;
	explicit CallOps(GpuOps &g_) : g(g_) {}
	int max_inflight() const override { return g.max_inflight(); }
	bool waits_asleep() const override { return g.waits_asleep(); }
	bool load_reads(const uint8_t *codes, size_t n, int slot, int64_t *base) override
	{
		std::string e;
		const bool ok = g.load_reads(codes, n, slot, base, e);
		if (!e.empty()) err.put(e);
		return ok;
	}
	void release_reads(int slot) override { g.release_reads(slot); }
	void sketch_batch(int w, int k, std::vector<wm::SketchReq*> &reqs) override { g.sketch_batch(err, w, k, reqs); }
	void seed_batch(std::vector<wm::SeedReq*> &reqs) override { g.seed_batch(err, reqs); }
	void chain_batch(std::vector<wm::ChainReq*> &reqs) override { g.chain_batch(err, reqs); }
	void ksw_batch(const wm_ksw_score_t &sc, std::vector<wm::KswReq*> &reqs) override { g.ksw_batch(err, sc, reqs); }
	void exts2_batch(const wm_ksw_score_t &sc, int noncan, int junc_bonus, std::vector<wm::KswReq*> &reqs) override { g.exts2_batch(err, sc, noncan, junc_bonus, reqs); }
	void window_batch(int, int, std::vector<wm::WindowReq*> &reqs) override { g.window_batch(err, reqs); }
};

struct wm_mapper_s {
	wm_ctx_t *c; const wm_index_t *idx;
	std::vector<wm_ctx_t*> workers;        // extra contexts (own stream + arena slice) for groups 1..G-1
	int n_threads = 1;
	int n_threads_cap = 0;                 // > 0: a file loop over several mappers has divided the host's cores among its mapping calls (wm_map_file_multi)
	int call_threads() const { return n_threads_cap > 0 && n_threads_cap < n_threads ? n_threads_cap : n_threads; }
	wm::IdxOpt io; wm::MapOpt mo;
	// results of the last mapping call per slot (wm_map_reads = slot 0; wm_map_reads_slot: two calls may run concurrently)
	struct Result { std::string text; std::vector<int32_t> hits; std::vector<uint32_t> cigars; std::vector<int64_t> first; std::vector<uint8_t> rl_defined; } res[WM_MAX_SLOTS];
	uint64_t stats[9];
	double host_stats[24] = {0};
	std::mutex stats_mu;
	std::unique_ptr<GpuOps> ops;           // the device contexts as a pool shared by the mapping calls (created on first use, rebuilt by wm_mapper_set_threads)
	int slots_hint = 0;                    // wm_mapper_set_slots
	bool sam_header = true;                // wm_map_file writes the @SQ / @PG lines (wm_mapper_set_sam_header)
	std::vector<std::string> cmdline;      // argv of the front end, for the @PG line of SAM files (wm_mapper_set_cmdline)
};

// what the mapper's device path cannot serve is refused when the mapper is made, not in the middle of a mapping call: an even k (the fused
// window call sketches with sketch_coop, which relies on k-mer != reverse complement, src/sketch.c:189). An index built with homopolymer compression
// (MM_I_HPC, -H) is served since round 5: sketch_coop compacts every sequence into its runs first (src/sketch.c:152-163), mm_adjust_minier's HPC branch
// (src/align.c:352-361) runs on the host.
static int mapper_index_ok(const wm_index_t *idx)
{
	if (!(idx->ix.k & 1)) return set_err(WM_EINVAL, "k = %d: the mapper's device path needs an odd k (every preset of the reference has one)", idx->ix.k);
	return WM_OK;
}
extern "C" int wm_mapper_create(wm_ctx_t *c, const wm_index_t *idx, const char *preset, int64_t flag, wm_mapper_t **out)
{
	*out = 0;
	if (!c || !idx) return set_err(WM_EINVAL, "null argument");
	if (!c->have_index) return set_err(WM_EINVAL, "wm_index_upload has not been called on this context");
	if (mapper_index_ok(idx)) return WM_EINVAL;
	wm_default_malloc();
	wm_mapper_t *m = new wm_mapper_t();
	m->c = c; m->idx = idx;
	wm::set_preset(0, m->io, m->mo);
	if (preset && preset[0] && wm::set_preset(preset, m->io, m->mo) < 0) { delete m; return set_err(WM_EINVAL, "unknown preset '%s'", preset); }
	m->mo.flag |= flag;
	m->io.k = idx->ix.k; m->io.w = idx->ix.w;
	wm::mapopt_update(m->mo, idx->ix);
	std::string err;
	if (wm::check_opt(m->io, m->mo, err) < 0) { delete m; return set_err(WM_EINVAL, "%s", err.c_str()); }
	memset(m->stats, 0, sizeof(m->stats));
	*out = m;
	return WM_OK;
}
#define WM_MAPOPT_FIELDS(X) X(flag) X(seed) X(sdust_thres) X(max_qlen) X(bw) X(max_gap) X(max_gap_ref) X(min_gap_ref) X(max_frag_len) \
	X(max_chain_skip) X(max_chain_iter) X(min_cnt) X(min_chain_score) X(chain_gap_scale) X(SVawareMinReadLength) X(suffixSampleOffset) X(min_mapq) \
	X(min_qcov) X(minPrefixLength) X(maxPrefixLength) X(prefixIncrementFactor) X(stage2_bw) X(stage2_zdrop_inv) X(stage2_max_gap) X(mask_level) \
	X(mask_len) X(pri_ratio) X(best_n) X(max_join_long) X(max_join_short) X(min_join_flank_sc) X(min_join_flank_ratio) X(alt_drop) X(a) X(b) X(q) X(e) \
	X(q2) X(e2) X(sc_ambi) X(zdrop) X(zdrop_inv) X(end_bonus) X(min_dp_max) X(min_ksw_len) X(max_clip_ratio) X(mid_occ_frac) X(min_mid_occ) X(mid_occ) \
	X(max_occ) X(mini_batch_size) X(max_sw_mat) X(noncan) X(junc_bonus) X(anchor_ext_len) X(anchor_ext_shift)
static void mapopt_to_c(const wm::MapOpt &o, wm_mapopt_t *c)
{
	memset(c, 0, sizeof(*c));
#define X(f) c->f = (decltype(c->f))o.f;
	WM_MAPOPT_FIELDS(X)
#undef X
	c->SVaware = o.SVaware ? 1 : 0;
}
static void mapopt_from_c(const wm_mapopt_t *c, wm::MapOpt &o)
{
#define X(f) o.f = (decltype(o.f))c->f;
	WM_MAPOPT_FIELDS(X)
#undef X
	o.SVaware = c->SVaware != 0;
}
extern "C" int wm_mapopt_preset(const char *preset, wm_mapopt_t *out, int *k, int *w)
{
	wm::IdxOpt io; wm::MapOpt mo;
	wm::set_preset(0, io, mo);
	if (preset && preset[0] && wm::set_preset(preset, io, mo) < 0) return set_err(WM_EINVAL, "unknown preset '%s'", preset);
	mapopt_to_c(mo, out);
	if (k) *k = io.k;
	if (w) *w = io.w;
	return WM_OK;
}
extern "C" int wm_mapper_create_opt(wm_ctx_t *c, const wm_index_t *idx, const wm_mapopt_t *opt, wm_mapper_t **out)
{
	*out = 0;
	if (!c || !idx || !opt) return set_err(WM_EINVAL, "null argument");
	if (!c->have_index) return set_err(WM_EINVAL, "wm_index_upload has not been called on this context");
	if (mapper_index_ok(idx)) return WM_EINVAL;
	wm_default_malloc();
	wm_mapper_t *m = new wm_mapper_t();
	m->c = c; m->idx = idx;
	wm::set_preset(0, m->io, m->mo);
	mapopt_from_c(opt, m->mo);
	m->io.k = idx->ix.k; m->io.w = idx->ix.w;
	wm::mapopt_update(m->mo, idx->ix);
	std::string err;
	if (wm::check_opt(m->io, m->mo, err) < 0) { delete m; return set_err(WM_EINVAL, "%s", err.c_str()); }
	memset(m->stats, 0, sizeof(m->stats));
	*out = m;
	return WM_OK;
}
extern "C" int wm_mapper_set_sam_header(wm_mapper_t *m, int on) { if (!m) return set_err(WM_EINVAL, "null mapper"); m->sam_header = on != 0; return WM_OK; }

extern "C" void wm_mapper_destroy(wm_mapper_t *m)
{
	if (!m) return;
	if (wm::prof_on()) wm::prof_report(stderr);                // WM_PROF=1: the host glue's time per named region (accumulated over the process)
	for (wm_ctx_t *w : m->workers) wm_ctx_destroy(w);
	delete m;
}

// Host parallelism: n_threads worker threads run the host glue of the reads (fibers, wm_fiber.h) and take turns issuing the batched
// device calls; C device contexts (own HIP stream + arena + pinned slab each; WM_CONTEXTS, default 4) let C batches be in flight at once.
extern "C" int wm_mapper_set_threads(wm_mapper_t *m, int n_threads, size_t arena_bytes_per_context)
{
	if (n_threads < 1) return set_err(WM_EINVAL, "n_threads < 1");
	for (wm_ctx_t *w : m->workers) wm_ctx_destroy(w);
	m->workers.clear();
	m->ops.reset();
	int C = getenv("WM_CONTEXTS") ? atoi(getenv("WM_CONTEXTS")) : (getenv("WM_GROUPS") ? atoi(getenv("WM_GROUPS")) : (n_threads >= 8 ? 6 : n_threads >= 2 ? 2 : 1));
	if (C < 1) C = 1;
	m->n_threads = n_threads;
	const int bt = getenv("WM_BATCH_THREADS") ? atoi(getenv("WM_BATCH_THREADS")) : 1;     // extra threads a batched call may spawn for its own packing
	m->c->host_threads = std::max(1, bt);
	for (int g = 1; g < C; ++g) {
		wm_ctx_t *w = 0;
		const int rc = wm_ctx_create(m->c->device, arena_bytes_per_context ? arena_bytes_per_context : m->c->arena_bytes, &w);
		if (rc) return rc;
		w->d_hkey = m->c->d_hkey; w->d_hval = m->c->d_hval; w->d_P = m->c->d_P; w->d_bloom = m->c->d_bloom; w->hbits = m->c->hbits; w->skp = m->c->skp;
		w->d_S = m->c->d_S; w->seq_off = m->c->seq_off; w->seq_len = m->c->seq_len;
		w->have_index = true; w->owns_index = false;
		w->host_threads = m->c->host_threads;
		m->workers.push_back(w);
	}
	// side streams: one pool for all contexts, so that main streams + pool = the hardware queues (WM_SIDE_POOL overrides the pool size)
	{
		const int hwq = getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
		int P = getenv("WM_SIDE_POOL") ? atoi(getenv("WM_SIDE_POOL")) : std::max(0, hwq - C);
		if (C == 1) P = 0;                                   // a single context keeps its own side streams
		wm_ctx_t *c0 = m->c;                                  // (the pool lives and dies with the mapper's first context)
		HIPCHK(hipStreamSynchronize(c0->stream));
		// (compute units of their own for the heavy / huge calls' side streams — hipExtStreamCreateWithCUMask, VERDICT r4 item 2 iii — were measured in round 5:
		// 0.075 / 0.118 / 0.174 Gbp/s with 32 / 64 / 96 CUs against 0.257 without; profiles/r05_sched.txt. The few hundred latency-bound wavefronts of a
		// stripe launch need the whole chip's SIMDs.)
		while ((int)c0->owned_pool.size() > P) { hipStreamDestroy(c0->owned_pool.back()); c0->owned_pool.pop_back(); }
		while ((int)c0->owned_pool.size() < P) { hipStream_t st; HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); c0->owned_pool.push_back(st); }
		c0->side_pool = P > 0 ? c0->owned_pool.data() : 0; c0->n_side_pool = P; c0->side_next = c0->owned_next;
		for (wm_ctx_t *w : m->workers) { w->side_pool = c0->side_pool; w->n_side_pool = P; w->side_next = c0->owned_next; }
	}
	// the pinned staging slabs are allocated now, not inside the first mapping call (page-locking a few GB takes a noticeable fraction of a second)
	{ size_t mark = 0; if (pin_take(m->c, 1, &mark)) pin_release(m->c, mark); }
	for (wm_ctx_t *w : m->workers) { size_t mark = 0; if (pin_take(w, 1, &mark)) pin_release(w, mark); }
	return WM_OK;
}

static int map_reads_impl(wm_mapper_t *m, std::vector<wm::ReadIn> &reads, double tm0, int slot = 0);

extern "C" int wm_map_reads(wm_mapper_t *m, int n, const char *const *names, const char *const *seqs, const int32_t *lens,
                            const char **text, size_t *text_len, const int32_t **hits, const uint32_t **cigars, const int64_t **hit_first)
{
	const double tm0 = now_ms();
	std::vector<wm::ReadIn> reads(n);
	wm::parallel_for(m->n_threads, (size_t)n, [&](size_t i) { reads[i].name = names[i]; reads[i].seq.assign(seqs[i], lens[i]); });
	const int rc = map_reads_impl(m, reads, tm0, 0);
	if (rc) return rc;
	if (text) *text = m->res[0].text.data();
	if (text_len) *text_len = m->res[0].text.size();
	if (hits) *hits = m->res[0].hits.data();
	if (cigars) *cigars = m->res[0].cigars.data();
	if (hit_first) *hit_first = m->res[0].first.data();
	return WM_OK;
}

// wm_map_reads with its own result buffers and its own slab of resident read codes: calls with different slots (0 and 1) may run concurrently from two
// host threads. A mapping call spends its first and last few hundred milliseconds filling and draining its pipeline of dependent device calls (window
// -> align -> align -> window -> align ...): with two mini-batches in flight those phases of one hide behind the steady state of the other.
extern "C" int wm_map_reads_slot(wm_mapper_t *m, int slot, int n, const char *const *names, const char *const *seqs, const int32_t *lens,
                                 const char **text, size_t *text_len, const int32_t **hits, const uint32_t **cigars, const int64_t **hit_first)
{
	if (!m || slot < 0 || slot >= WM_MAX_SLOTS) return set_err(WM_EINVAL, "slot must be 0 .. WM_MAX_SLOTS - 1");
	const double tm0 = now_ms();
	std::vector<wm::ReadIn> reads(n);
	wm::parallel_for(m->n_threads, (size_t)n, [&](size_t i) { reads[i].name = names[i]; reads[i].seq.assign(seqs[i], lens[i]); });
	const int rc = map_reads_impl(m, reads, tm0, slot);
	if (rc) return rc;
	wm_mapper_t::Result &R = m->res[slot];
	if (text) *text = R.text.data();
	if (text_len) *text_len = R.text.size();
	if (hits) *hits = R.hits.data();
	if (cigars) *cigars = R.cigars.data();
	if (hit_first) *hit_first = R.first.data();
	return WM_OK;
}

// the device contexts of a mapper as the pool its mapping calls share (caller holds m->stats_mu)
static void ensure_ops(wm_mapper_t *m)
{
	if (m->ops) return;
	m->ops.reset(new GpuOps());
	std::vector<wm_ctx_t*> cs; cs.push_back(m->c); cs.insert(cs.end(), m->workers.begin(), m->workers.end());
	m->ops->init(cs);
	m->ops->slots_hint = m->slots_hint;
}

// how many mini-batches the caller keeps in flight on this mapper (wm_map_reads_slot on slots 0 .. n - 1): the resident-reads allocation gets that many slabs
// the next time it is (re)made (a caller of slots 2..3 was silently served from host views)
extern "C" int wm_mapper_set_slots(wm_mapper_t *m, int n)
{
	if (!m || n < 1 || n > WM_MAX_SLOTS) return set_err(WM_EINVAL, "slots must be 1 .. WM_MAX_SLOTS");
	std::lock_guard<std::mutex> lk(m->stats_mu);
	if (n > m->slots_hint) m->slots_hint = n;
	ensure_ops(m);
	if (n > m->ops->slots_hint.load()) m->ops->slots_hint = n;
	return WM_OK;
}

// per read of the slot's last mapping call: 1 = the mapper assigned rep_len where the reference assigns it (src/map.c:808-813 rescan, :859-861 fallback),
// 0 = the pure-MCAS path, where the reference feeds mm_set_mapq an uninitialised word (src/map.c:281,933) and MAPQ / rl:i are not comparable
extern "C" int wm_map_reads_rep_len_defined(const wm_mapper_t *m, int slot, const uint8_t **flags, size_t *n)
{
	if (!m || slot < 0 || slot >= WM_MAX_SLOTS || !flags) return set_err(WM_EINVAL, "bad argument");
	*flags = m->res[slot].rl_defined.data();
	if (n) *n = m->res[slot].rl_defined.size();
	return WM_OK;
}

// maps `reads` in the given order; results land in m->text / hits / cigars / first / stats
static int map_reads_impl(wm_mapper_t *m, std::vector<wm::ReadIn> &reads, double tm0, int slot)
{
	wm_mapper_t::Result &R = m->res[slot];
	static const bool trace_m = getenv("WM_TRACE") != 0;
	const int n = (int)reads.size();
	uint64_t bases = 0;
	for (int i = 0; i < n; ++i) bases += reads[i].seq.size();
	std::vector<wm::ReadOut> out(n);
	const double tm1 = now_ms();
	{
		std::lock_guard<std::mutex> lk(m->stats_mu);
		ensure_ops(m);
	}
	GpuOps &ops = *m->ops;
	uint64_t cells0 = 0; double ksw_us0 = 0, aux_us0 = 0;
	for (GpuOpsCtx &x : ops.ctxs) { cells0 += x.cells; ksw_us0 += x.ksw_us; aux_us0 += x.aux_us; }      // (contexts are shared: this call's share = the difference; approximate when two calls overlap)
	wm::MapStats st;
	hipSetDevice(m->c->device);
	// records are formatted by the worker that finishes a read, while the other reads are still being mapped
	std::vector<std::string> texts(n);
	const std::function<void(size_t)> fmt = [&](size_t i) { wm::write_read(texts[i], m->idx->ix, reads[i], out[i], m->mo.flag); };
	CallOps call(ops);                       // this call's view of the shared contexts: its failed batches fail this call, nobody else's
	wm::map_batch(m->idx->ix, m->mo, &call, reads, out, &st, m->call_threads(), &fmt, slot);
	if (!call.err.msg.empty()) return set_err(WM_ENODEV, "%s", call.err.msg.c_str());
	if (!st.internal_error.empty()) return set_err(WM_EINTERNAL, "%s", st.internal_error.c_str());
	{ std::string ie; if (wm::take_internal_error(ie)) return set_err(WM_EINTERNAL, "%s", ie.c_str()); }      // (recorded by a thread outside any call's team)
	GpuOpsCtx tot; tot.c = m->c;
	for (GpuOpsCtx &x : ops.ctxs) { tot.cells += x.cells; tot.ksw_us += x.ksw_us; tot.aux_us += x.aux_us; }
	tot.cells -= cells0; tot.ksw_us -= ksw_us0; tot.aux_us -= aux_us0;
	GpuOpsCtx &opsr = tot;
	if (getenv("WM_TRACE")) {
		double a[8] = {0};
		for (GpuOpsCtx &x : ops.ctxs) { a[0] += x.t_pack; a[1] += x.t_prep; a[2] += x.t_run; a[3] += x.t_fetch; a[4] += x.t_unpack; a[5] += x.t_sketch; a[6] += x.t_seed; a[7] += x.t_chain; }
		fprintf(stderr, "[ops, sum over %zu contexts, ms] ksw: pack %.0f prepare %.0f run %.0f fetch %.0f unpack %.0f | sketch %.0f seed %.0f chain %.0f | batches %llu\n", ops.ctxs.size(), a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], (unsigned long long)st.n_flush);
	}
	const double tm2 = now_ms();
	// output records (formatted above, per read), laid out in input order
	std::vector<size_t> toff(n + 1, 0), coff(n + 1, 0);
	R.first.assign(n + 1, 0);
	R.rl_defined.resize(n);
	for (int i = 0; i < n; ++i) {
		R.rl_defined[i] = out[i].rep_len_defined ? 1 : 0;
		toff[i + 1] = toff[i] + texts[i].size();
		R.first[i + 1] = R.first[i] + (int64_t)out[i].regs.size();
		size_t nc = 0;
		for (const wm::Reg &r : out[i].regs) nc += r.cigar.size();
		coff[i + 1] = coff[i] + nc;
	}
	R.text.resize(toff[n]); R.hits.resize((size_t)R.first[n] * 16); R.cigars.resize(coff[n]);
	wm::parallel_for(m->n_threads, (size_t)n, [&](size_t i) {
		if (!texts[i].empty()) memcpy(&R.text[toff[i]], texts[i].data(), texts[i].size());
		int32_t *ho = R.hits.data() + (size_t)R.first[i] * 16;
		uint32_t *co = R.cigars.data() + coff[i];
		for (const wm::Reg &r : out[i].regs) {
			const int32_t o[16] = { r.rid, r.rs, r.re, r.qs, r.qe, (int32_t)r.rev, (int32_t)r.mapq, r.has_p ? (int32_t)r.cigar.size() : 0, r.score, r.cnt, r.mlen, r.blen,
			                        r.dp_score, r.dp_max, r.dp_max2, (int32_t)((r.parent == r.id) | r.inv << 1 | r.sam_pri << 2 | r.split << 3 | (r.has_p ? r.trans_strand << 5 : 0)) };
			memcpy(ho, o, sizeof(o)); ho += 16;
			if (!r.cigar.empty()) { memcpy(co, r.cigar.data(), r.cigar.size() * 4); co += r.cigar.size(); }
		}
	});
	if (trace_m) fprintf(stderr, "[map_reads] n=%d ingest %.1f ms, map %.1f ms, format %.1f ms\n", n, tm1 - tm0, tm2 - tm1, now_ms() - tm2);
	std::lock_guard<std::mutex> stats_lk(m->stats_mu);
	m->stats[0] = st.n_flush; m->stats[1] = st.n_ksw; m->stats[2] = st.n_chain; m->stats[3] = st.n_seed; m->stats[4] = st.n_sketch;
	m->host_stats[0] += st.cpu_fiber; m->host_stats[1] += st.wall_idle;
	for (int op = 0; op < 4; ++op) { m->host_stats[2 + op] += st.cpu_op[op]; m->host_stats[6 + op] += st.wall_op[op]; m->host_stats[10 + op] += (double)st.n_batches[op]; }
	m->host_stats[14] += (tm2 - tm1) * 1e-3; m->host_stats[15] += (now_ms() - tm2) * 1e-3; m->host_stats[16] = m->call_threads(); m->host_stats[17] += st.cpu_help;
	m->host_stats[18] += st.wall_fiber; m->host_stats[19] += st.wall_lock; m->host_stats[20] += st.wall_total;
	if (trace_m) fprintf(stderr, "[host] fibers cpu %.2f s | idle wall %.2f s | batched calls cpu/wall/n: sketch %.2f/%.2f/%llu seed %.2f/%.2f/%llu chain %.2f/%.2f/%llu ksw %.2f/%.2f/%llu\n", st.cpu_fiber, st.wall_idle,
	                     st.cpu_op[0], st.wall_op[0], (unsigned long long)st.n_batches[0], st.cpu_op[1], st.wall_op[1], (unsigned long long)st.n_batches[1],
	                     st.cpu_op[2], st.wall_op[2], (unsigned long long)st.n_batches[2], st.cpu_op[3], st.wall_op[3], (unsigned long long)st.n_batches[3]);
	m->stats[5] = opsr.cells; m->stats[6] = (uint64_t)opsr.ksw_us; m->stats[7] = (uint64_t)opsr.aux_us; m->stats[8] = bases;
	return WM_OK;
}

// The file-level loop (mm_map_file, src/map.c:1226-1268): reads FASTA/FASTQ(.gz) mini-batches of `mini_batch_bases` (0 = the
// reference's default 1 Gbase), maps them and writes the records to out_path ("-" = stdout), reader / mapper / writer
// overlapped. Every mini-batch is ordered like the reference orders it, so the file equals the reference's output.
// stats (optional, 6 doubles): reads, bases, batches, seconds spent reading / mapping / writing.
// the command line for the @PG line of wm_map_file_split (no mapper object outlives its parts); wm_mapper_set_cmdline stores it here as well
static std::mutex g_cmdline_mu;
static std::vector<std::string> g_cmdline;
static bool g_split_pg = true;              // wm_map_file_split prints the @PG line itself
extern "C" int wm_set_cmdline(int argc, const char *const *argv)
{
	if (argc > 0 && !argv) return set_err(WM_EINVAL, "bad argument");
	std::lock_guard<std::mutex> lk(g_cmdline_mu);
	g_split_pg = argc >= 0;                  // argc < 0: the front end has printed @PG already (the reference's main does, src/main.c:395)
	g_cmdline.clear();
	if (argc > 0) g_cmdline.assign(argv, argv + argc);
	return WM_OK;
}
extern "C" int wm_mapper_set_cmdline(wm_mapper_t *m, int argc, const char *const *argv)
{
	if (!m || argc < 0 || (argc > 0 && !argv)) return set_err(WM_EINVAL, "bad argument");
	m->cmdline.assign(argv, argv + argc);
	return WM_OK;
}

// wm_last_error is per thread and the second mapping lane of the file loops is a thread of its own: a lane keeps the code and message of its
// failed call here and the entry point re-issues them on the caller's thread (ENODEV / ENOMEM of lane 1 used to surface as "mapping failed")
struct LaneError {
	std::mutex mu; int code = 0; std::string msg;
	void keep(int rc) { std::lock_guard<std::mutex> lk(mu); if (!code) { code = rc; msg = g_err; } }
};

extern "C" int wm_map_file(wm_mapper_t *m, const char *reads_path, const char *out_path, int64_t mini_batch_bases, double *stats)
{
	g_err[0] = 0;
	FILE *out = strcmp(out_path, "-") == 0 ? stdout : fopen(out_path, "wb");
	if (!out) return set_err(WM_EINVAL, "cannot open '%s' for writing", out_path);
	std::string err;
	if ((m->mo.flag & 0x8) && m->sam_header) {                         // MM_F_OUT_SAM: @SQ / @PG lines first (mm_write_sam_hdr, src/main.c:393)
		std::string hdr;
		std::vector<const char*> av;
		for (const std::string &a : m->cmdline) av.push_back(a.c_str());
		wm::write_sam_header(hdr, m->idx->ix, (int)av.size(), av.data());
		if (fwrite(hdr.data(), 1, hdr.size(), out) != hdr.size()) { if (out != stdout) fclose(out); return set_err(WM_EINVAL, "write error on '%s'", out_path); }
	}
	wm::FileStats fs;
	const bool with_qual = (m->mo.flag & 0x8) != 0;                    // SAM output prints QUAL
	LaneError le;
	const int rc = wm::map_file(reads_path, mini_batch_bases, with_qual, [&](std::vector<wm::ReadIn> &batch, std::string &text, int lane) {
		const int r = map_reads_impl(m, batch, now_ms(), lane);             // (WM_MAP_LANES mini-batches in flight, default 2: lane = result slot = slab of resident read codes)
		if (r == 0) text.swap(m->res[lane].text);
		else le.keep(r);
		return r;
	}, out, &fs, err);
	if (out != stdout) fclose(out);
	if (stats) { stats[0] = (double)fs.n_reads; stats[1] = (double)fs.n_bases; stats[2] = (double)fs.n_batches; stats[3] = fs.t_read; stats[4] = fs.t_map; stats[5] = fs.t_write; }
	if (rc) return le.code ? set_err(le.code, "%s", le.msg.c_str()) : set_err(WM_EINVAL, "%s", err.c_str());
	return WM_OK;
}

// The file loop over N mappers — one per GPU of the node (each with its own context, index copy and host threads: wm_ctx_create(device i),
// wm_index_upload_peer, wm_mapper_create, wm_mapper_set_threads) — inside ONE process: the C twin of `one rank per GPU`. The reader hands mini-batches
// to WM_MAP_LANES (default 2) lanes per mapper (lane l -> mapper l % n, result slot l / n), reads shard by mini-batch, nothing is exchanged between the devices, and the
// ordered writer puts the records back into input order: the output file equals wm_map_file's (and the reference's). The SAM header, if wanted, is
// written once from the first mapper's index and command line.
extern "C" int wm_map_file_multi(wm_mapper_t *const *ms, int n, const char *reads_path, const char *out_path, int64_t mini_batch_bases, double *stats)
{
	g_err[0] = 0;
	if (!ms || n < 1 || !reads_path || !out_path) return set_err(WM_EINVAL, "bad argument");
	for (int i = 0; i < n; ++i) {
		if (!ms[i]) return set_err(WM_EINVAL, "null mapper");
		if (ms[i]->mo.flag != ms[0]->mo.flag || ms[i]->idx->ix.seq.size() != ms[0]->idx->ix.seq.size() || ms[i]->idx->ix.hbits != ms[0]->idx->ix.hbits)
			return set_err(WM_EINVAL, "mapper %d differs from mapper 0 (options or index)", i);
	}
	FILE *out = strcmp(out_path, "-") == 0 ? stdout : fopen(out_path, "wb");
	if (!out) return set_err(WM_EINVAL, "cannot open '%s' for writing", out_path);
	wm_mapper_t *m0 = ms[0];
	std::string err;
	if ((m0->mo.flag & 0x8) && m0->sam_header) {
		std::string hdr;
		std::vector<const char*> av;
		for (const std::string &a : m0->cmdline) av.push_back(a.c_str());
		wm::write_sam_header(hdr, m0->idx->ix, (int)av.size(), av.data());
		if (fwrite(hdr.data(), 1, hdr.size(), out) != hdr.size()) { if (out != stdout) fclose(out); return set_err(WM_EINVAL, "write error on '%s'", out_path); }
	}
	wm::FileStats fs;
	const bool with_qual = (m0->mo.flag & 0x8) != 0;
	LaneError le;
	// n mappers x lanes mapping calls run at once, each with its mapper's worker threads: on a host whose usable cores (affinity, cgroup quota) are fewer
	// than that product the calls are given an equal share each for the duration of the loop (eight mappers of sixteen threads under a 16-CPU quota were
	// 256 runnable threads: the collapse of profiles/r04j). WM_MULTI_THREADS=<n> sets the share per call, 0 leaves the mappers' own counts.
	{
		const int lanes = wm::default_lanes() * n;
		int share = wm::usable_cores() / lanes;
		if (getenv("WM_MULTI_THREADS")) share = atoi(getenv("WM_MULTI_THREADS"));
		else if (share < 2) share = 2;
		for (int i = 0; i < n; ++i) ms[i]->n_threads_cap = n > 1 ? share : 0;
	}
	struct Uncap { wm_mapper_t *const *ms; int n; ~Uncap() { for (int i = 0; i < n; ++i) ms[i]->n_threads_cap = 0; } } uncap{ ms, n };
	const int rc = wm::map_file(reads_path, mini_batch_bases, with_qual, [&](std::vector<wm::ReadIn> &batch, std::string &text, int lane) {
		wm_mapper_t *m = ms[lane % n];
		const int slot = lane / n;
		const int r = map_reads_impl(m, batch, now_ms(), slot);
		if (r == 0) text.swap(m->res[slot].text);
		else le.keep(r);
		return r;
	}, out, &fs, err, wm::default_lanes() * n);
	if (out != stdout) fclose(out);
	if (stats) { stats[0] = (double)fs.n_reads; stats[1] = (double)fs.n_bases; stats[2] = (double)fs.n_batches; stats[3] = fs.t_read; stats[4] = fs.t_map; stats[5] = fs.t_write; }
	if (rc) return le.code ? set_err(le.code, "%s", le.msg.c_str()) : set_err(WM_EINVAL, "%s", err.c_str());
	return WM_OK;
}

// ---- a reference indexed in parts (-I, --split-prefix; src/main.c:398-429, src/map.c:1050-1105, src/splitidx.c) ----
extern "C" int wm_index_build_parts(const char *fasta, const char *kmer_file, int k, int w, int n_threads, uint64_t batch_bases, wm_index_t **out, int cap, int *n_parts)
{
	if (!fasta || !out || !n_parts || cap < 1 || batch_bases == 0) return set_err(WM_EINVAL, "bad argument");
	*n_parts = 0;
	wm::IdxOpt io; io.k = k; io.w = w;
	wm::MapOpt mo; std::string err;
	if (wm::check_opt(io, mo, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	std::vector<wm::Index> parts;
	const int n = wm::index_build_parts_from_fasta(io, fasta, kmer_file ? kmer_file : "", n_threads, batch_bases, parts, err);
	if (n < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	if (n > cap) return set_err(WM_ENOMEM, "the reference has %d index parts, room for %d", n, cap);
	for (int i = 0; i < n; ++i) { out[i] = new wm_index_t(); out[i]->ix = std::move(parts[i]); }
	*n_parts = n;
	return WM_OK;
}

// one mini-batch against the mapper's index, hits kept as they are (no records)
static int map_reads_raw(wm_mapper_t *m, std::vector<wm::ReadIn> &reads, std::vector<wm::ReadOut> &out, int slot)
{
	{
		std::lock_guard<std::mutex> lk(m->stats_mu);
		ensure_ops(m);
	}
	GpuOps &ops = *m->ops;
	wm::MapStats st;
	hipSetDevice(m->c->device);
	CallOps call(ops);
	wm::map_batch(m->idx->ix, m->mo, &call, reads, out, &st, m->n_threads, 0, slot);
	if (!call.err.msg.empty()) return set_err(WM_ENODEV, "%s", call.err.msg.c_str());
	if (!st.internal_error.empty()) return set_err(WM_EINTERNAL, "%s", st.internal_error.c_str());
	{ std::string ie; if (wm::take_internal_error(ie)) return set_err(WM_EINTERNAL, "%s", ie.c_str()); }
	return WM_OK;
}

// One part at a time (src/main.c:398-429: the reference's main holds one mm_idx_t, maps every read against it, destroys it, reads the next): wm_split_begin,
// wm_split_add_part per part — upload, mapper, the whole reads file against it, hits spilled; the part may be destroyed on return —, wm_split_finish = header + merge.
struct wm_split_s {
	wm_ctx_t *c; wm_mapopt_t copt; wm::MapOpt mo; int n_threads; int k, w;
	wm::SplitRun *run;
};
extern "C" int wm_split_begin(wm_ctx_t *c, const wm_mapopt_t *opt, int k, int w, int n_threads, const char *reads_path, int64_t mini_batch_bases, wm_split_t **out)
{
	g_err[0] = 0;
	if (!c || !opt || !reads_path || !out) return set_err(WM_EINVAL, "bad argument");
	*out = 0;
	wm::MapOpt mo; wm::IdxOpt io;
	wm::set_preset(0, io, mo);
	mapopt_from_c(opt, mo);
	if (mo.flag & (wm::F_OUT_CS | wm::F_OUT_MD)) return set_err(WM_EINVAL, "--cs or --MD doesn't work with a reference indexed in parts");      // src/options.c:139-141
	wm_split_t *s = new wm_split_t();
	s->c = c; s->copt = *opt; s->mo = mo; s->n_threads = n_threads > 1 ? n_threads : 1; s->k = k; s->w = w;
	s->run = new wm::SplitRun(reads_path, mini_batch_bases, mo, k, w);
	*out = s;
	return WM_OK;
}
extern "C" void wm_split_abort(wm_split_t *s) { if (s) { delete s->run; delete s; } }
extern "C" int wm_split_add_part(wm_split_t *s, wm_index_t *part)
{
	g_err[0] = 0;
	if (!s || !part) return set_err(WM_EINVAL, "null argument");
	if (part->ix.k != s->k || part->ix.w != s->w) return set_err(WM_EINVAL, "index part built with k = %d, w = %d; the run was started for k = %d, w = %d", part->ix.k, part->ix.w, s->k, s->w);
	wm_mapper_t *m = 0;
	int rc0;
	if ((rc0 = wm_index_upload(s->c, part)) != WM_OK) return rc0;                       // (the message is the failing call's)
	if ((rc0 = wm_mapper_create_opt(s->c, part, &s->copt, &m)) != WM_OK) return rc0;
	if ((rc0 = wm_mapper_set_threads(m, s->n_threads, 0)) != WM_OK) { wm_mapper_destroy(m); return rc0; }
	LaneError le;
	std::string err;
	const int rc = s->run->add_part(part->ix.seq, [&](std::vector<wm::ReadIn> &batch, std::vector<wm::ReadOut> &o, int lane) -> int { const int r = map_reads_raw(m, batch, o, lane); if (r) le.keep(r); return r; }, err);
	wm_mapper_destroy(m);
	if (rc) return le.code ? set_err(le.code, "%s", le.msg.c_str()) : set_err(WM_EINVAL, "%s", err.c_str());
	return WM_OK;
}
extern "C" int wm_split_finish(wm_split_t *s, const char *out_path, double *stats)
{
	g_err[0] = 0;
	if (!s || !out_path) { wm_split_abort(s); return set_err(WM_EINVAL, "bad argument"); }
	FILE *out = strcmp(out_path, "-") == 0 ? stdout : fopen(out_path, "wb");
	if (!out) { wm_split_abort(s); return set_err(WM_EINVAL, "cannot open '%s' for writing", out_path); }
	if (s->mo.flag & 0x8) {
		// SAM: the reference's main prints @PG (with CL:) when it sees the first of several parts (mm_write_sam_hdr(0, ...), src/main.c:395); the merge
		// pass then lists every part's contigs (src/map.c:1304-1306) — @PG first, @SQ after it
		std::string hdr, sq;
		wm::Index none;
		std::vector<const char*> av;
		bool with_pg;
		{ std::lock_guard<std::mutex> lk(g_cmdline_mu); for (const std::string &a : g_cmdline) av.push_back(a.c_str()); with_pg = g_split_pg; }
		if (with_pg) wm::write_sam_header(hdr, none, (int)av.size(), av.data());
		wm::write_sam_header(sq, s->run->dict(), 0, 0);
		hdr += sq.substr(0, sq.rfind("@PG"));
		if (fwrite(hdr.data(), 1, hdr.size(), out) != hdr.size()) { if (out != stdout) fclose(out); wm_split_abort(s); return set_err(WM_EINVAL, "write error on '%s'", out_path); }
	}
	wm::FileStats fs;
	std::string err;
	const int rc = s->run->finish(out, &fs, err);
	if (out != stdout) fclose(out);
	if (stats) { stats[0] = (double)fs.n_reads; stats[1] = (double)fs.n_bases; stats[2] = (double)fs.n_batches; stats[3] = fs.t_read; stats[4] = fs.t_map; stats[5] = fs.t_write; }
	wm_split_abort(s);
	if (rc) return set_err(WM_EINVAL, "%s", err.c_str());
	return WM_OK;
}

// every part given up front (rounds 3-4; the parts stay the caller's)
extern "C" int wm_map_file_split(wm_ctx_t *c, int n_parts, wm_index_t *const *parts, const wm_mapopt_t *opt, int n_threads, const char *reads_path, const char *out_path,
                                 int64_t mini_batch_bases, double *stats)
{
	g_err[0] = 0;
	if (!c || n_parts < 1 || !parts || !opt || !reads_path || !out_path) return set_err(WM_EINVAL, "bad argument");
	for (int j = 0; j < n_parts; ++j) if (!parts[j]) return set_err(WM_EINVAL, "null index part");
	wm_split_t *s = 0;
	int rc = wm_split_begin(c, opt, parts[0]->ix.k, parts[0]->ix.w, n_threads, reads_path, mini_batch_bases, &s);
	if (rc) return rc;
	for (int j = 0; j < n_parts; ++j)
		if ((rc = wm_split_add_part(s, parts[j])) != WM_OK) { wm_split_abort(s); return rc; }
	return wm_split_finish(s, out_path, stats);
}

// the reference FASTA indexed part by part AS THE RUN GOES: one part in host memory (and one on the device) at a time, like `winnowmap -I <batch_bases>
// --split-prefix` (src/main.c:417-419). on_device: the part's minimizers are sketched and its table built on the GPU (wm_index_build_dev's path).
extern "C" int wm_map_file_split_fasta(wm_ctx_t *c, const char *fasta, const char *kmer_file, int k, int w, int build_threads, uint64_t batch_bases, int on_device,
                                       const wm_mapopt_t *opt, int n_threads, const char *reads_path, const char *out_path, int64_t mini_batch_bases, double *stats, int *n_parts)
{
	g_err[0] = 0;
	if (!c || !fasta || !opt || !reads_path || !out_path || batch_bases == 0) return set_err(WM_EINVAL, "bad argument");
	if (n_parts) *n_parts = 0;
	wm::IdxOpt io; io.k = k; io.w = w;
	{ wm::MapOpt mo; std::string e; if (wm::check_opt(io, mo, e) < 0) return set_err(WM_EINVAL, "%s", e.c_str()); }
	wm::IndexPartReader rd;
	std::string err;
	if (rd.open(fasta, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	wm_split_t *s = 0;
	int rc = wm_split_begin(c, opt, k, w, n_threads, reads_path, mini_batch_bases, &s);
	if (rc) return rc;
	std::vector<std::string> names, seqs;
	int n = 0;
	while (rd.next(batch_bases, names, seqs) > 0) {
		wm_index_t *part = 0;
		if (on_device) rc = wm_index_build_seqs_dev(c, io, names, seqs, kmer_file ? kmer_file : "", build_threads, &part);
		else {
			part = new wm_index_t();
			if (wm::index_build(io, names, seqs, kmer_file ? kmer_file : "", build_threads, part->ix, err) < 0) { delete part; part = 0; rc = set_err(WM_EINVAL, "%s", err.c_str()); }
		}
		std::vector<std::string>().swap(seqs);
		if (rc == WM_OK) rc = wm_split_add_part(s, part);
		if (part) wm_index_destroy(part);
		if (rc) { wm_split_abort(s); return rc; }
		++n;
	}
	if (n == 0) { wm_split_abort(s); return set_err(WM_EINVAL, "no sequences in %s", fasta); }
	if (n_parts) *n_parts = n;
	return wm_split_finish(s, out_path, stats);
}

extern "C" int wm_index_read_junc_bed(wm_index_t *idx, const char *path)
{
	if (!idx || !path) return set_err(WM_EINVAL, "null argument");
	std::string err;
	if (wm::index_read_bed(idx->ix, path, true, err) < 0) return set_err(WM_EINVAL, "%s", err.c_str());
	return WM_OK;
}
extern "C" int wm_index_add_junc(wm_index_t *idx, int ctg, int n, const int32_t *st, const int32_t *en, const int32_t *strand)
{
	if (!idx || ctg < 0 || ctg >= (int)idx->ix.seq.size() || n < 0 || (n > 0 && (!st || !en || !strand))) return set_err(WM_EINVAL, "bad argument");
	if (idx->ix.I.empty()) idx->ix.I.resize(idx->ix.seq.size());
	std::vector<wm::JuncIntv> &r = idx->ix.I[ctg];
	for (int i = 0; i < n; ++i) r.push_back(wm::JuncIntv{ st[i], en[i], strand[i] });
	std::stable_sort(r.begin(), r.end(), [](const wm::JuncIntv &a, const wm::JuncIntv &b) { return a.st < b.st; });
	return WM_OK;
}

extern "C" int wm_ksw_ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat5x5, int gapo, int gape, int *qe, int *te)
{
	int q = -1, t = -1;
	const int sc = wm::ll_i16(qlen, query, tlen, target, mat5x5, gapo, gape, &q, &t);
	if (qe) *qe = q;
	if (te) *te = t;
	return sc;
}

extern "C" int wm_ksw_n_classes(void) { return WM_KSW_NCLASS; }      // kernel classes wm_mapper_kernel_stats / _union report on (ksw_plan.h)
extern "C" int wm_mapper_stats(const wm_mapper_t *m, uint64_t *out9) { memcpy(out9, m->stats, sizeof(m->stats)); return WM_OK; }
// per ksw kernel class (ksw_plan.h) since the mapper was created: out[3*k] = summed launch durations in ms (HIP events on the
// launching stream), out[3*k+1] = DP cells, out[3*k+2] = launches; n_classes receives WM_KSW_NCLASS
extern "C" int wm_mapper_kernel_stats(const wm_mapper_t *m, double *out, int cap, int *n_classes)
{
	if (n_classes) *n_classes = WM_KSW_NCLASS;
	if (cap < 3 * WM_KSW_NCLASS) return set_err(WM_EINVAL, "need room for %d doubles", 3 * WM_KSW_NCLASS);
	for (int k = 0; k < WM_KSW_NCLASS; ++k) {
		double ms = m->c->k_ms[k], cells = (double)m->c->k_cells[k], ln = (double)m->c->k_launches[k];
		for (const wm_ctx_t *w : m->workers) { ms += w->k_ms[k]; cells += (double)w->k_cells[k]; ln += (double)w->k_launches[k]; }
		out[3 * k] = ms; out[3 * k + 1] = cells; out[3 * k + 2] = ln;
	}
	return WM_OK;
}

// out[k] = milliseconds during which at least one launch of ksw class k was running (union of its launch intervals over all contexts), counting
// only what lies after `since_ms` on the device clock; returns through *now_ms the current reading of that clock (pass it as since_ms next time)
extern "C" int wm_mapper_kernel_union(const wm_mapper_t *m, double since_ms, double *out, int cap, double *now_ms_out)
{
	if (!m || cap < WM_KSW_NCLASS) return set_err(WM_EINVAL, "need room for %d doubles", WM_KSW_NCLASS);
	std::vector<const wm_ctx_t*> cs; cs.push_back(m->c); cs.insert(cs.end(), m->workers.begin(), m->workers.end());
	for (int k = 0; k < WM_KSW_NCLASS; ++k) {
		std::vector<std::pair<float, float>> iv;
		for (const wm_ctx_t *c : cs) {
			std::lock_guard<std::mutex> lk(const_cast<wm_ctx_t*>(c)->iv_mu);
			for (const auto &p : c->k_iv[k]) if (p.second > since_ms) iv.push_back(std::make_pair(std::max(p.first, (float)since_ms), p.second));
		}
		std::sort(iv.begin(), iv.end());
		double tot = 0, cur_s = 0, cur_e = -1;
		for (const auto &p : iv) {
			if (p.first > cur_e) { if (cur_e > cur_s) tot += cur_e - cur_s; cur_s = p.first; cur_e = p.second; }
			else if (p.second > cur_e) cur_e = p.second;
		}
		if (cur_e > cur_s) tot += cur_e - cur_s;
		out[k] = tot;
	}
	if (now_ms_out) {
		*now_ms_out = 0;
		hipEvent_t base = device_base_event(m->c->device), e;
		if (base && hipEventCreate(&e) == hipSuccess) {
			float t = 0;
			if (hipEventRecord(e, m->c->stream) == hipSuccess && hipEventSynchronize(e) == hipSuccess && hipEventElapsedTime(&t, base, e) == hipSuccess) *now_ms_out = t;
			hipEventDestroy(e);
		}
	}
	return WM_OK;
}

extern "C" int wm_mapper_host_stats(const wm_mapper_t *m, double *out, int cap)
{
	if (cap < 18) return set_err(WM_EINVAL, "need room for 18 doubles");
	memcpy(out, m->host_stats, (size_t)(cap < 24 ? cap : 24) * sizeof(double));
	return WM_OK;
}

extern "C" int wm_sam_header(const wm_index_t *idx, int argc, const char *const *argv, const char **text, size_t *text_len)
{
	static thread_local std::string s;
	s.clear();
	wm::write_sam_header(s, idx->ix, argc, argv);
	*text = s.data(); *text_len = s.size();
	return WM_OK;
}
