// wm_ksw.hip — libwmgpu.so, alignment unit: the gfx950 kernel entry points of ksw_extd2_sse / ksw_extz2_sse / ksw_exts2_sse (src/ksw2_extd2_sse.c, src/ksw2_exts2_sse.c),
// kernel routing, batch planning (class, traceback pitch, arena offsets), launch, traceback, result fetch — wm_ksw_* of include/wm_gpu.h.
#include "wm_rt.h"
#include "simt.h"
#include "ksw_kernel.h"
#include "ksw_packed_kernel.h"
#include "ksw_packed_multi_kernel.h"
#include "ksw_dual_kernel.h"
#include "ksw_stripe_kernel.h"
#include "ksw_chain_kernel.h"
#include "ksw_exts2_kernel.h"
#include "cigar_walk.h"
#include "reads2bit.h"

// ======================================================================================================
// kernels
// ======================================================================================================
// WM_OCC_HINT (build define, A/B): ask the register allocator for one more wavefront per SIMD where a kernel sits just above a step of the register file
// (512 / 4 = 128 registers: ksw_dpp_kernel<8, true, *, true> 130-132, ksw_chain_kernel<2, *, *, true> 122-138; tools/kernel_regs.py) at the price of 3-6 spilled values
#ifndef WM_OCC_HINT
#define WM_OCC_HINT 1          // (round 6, two A/B pairs at the full step size: +0.9 % each, gpurun_out/r06o; 0 = the allocator's own choice)
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

// two alignments per wavefront (ksw_dp_dual, ksw_dual_kernel.h): workgroup g takes jobs order[2g] and order[2g + 1] of a class's size-sorted list (the last one alone
// when the list is odd). Only the <CLIP, HASN, EXACT> = <false, false, false> classes — the gap fills, 60 % of all DP cells — run here.
template <int NC>
__global__ __launch_bounds__(64, (WM_OCC_HINT && NC == 8) ? 4 : 1) void ksw_dual_kernel(wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs, const int *__restrict__ order, int n,
                                                      const uint8_t *__restrict__ seqs, uint8_t *__restrict__ tb, wm_ksw_dres_t *__restrict__ res)
{
	const int g = blockIdx.x, jA = order[2 * g];
	const bool hasB = 2 * g + 1 < n;
	const int jB = hasB ? order[2 * g + 1] : jA;
	wmk::ksw_dp_dual<NC>(sc, jobs[jA], jobs[jB], hasB, seqs, tb, res + jA, res + jB);
}
// WM_KSW_DUAL=1 switches it on (wm_ksw_set_dual overrides). OFF by default: bit-exact (emulator fuzz, GPU suite) but SLOWER in the mapper — 0.286-0.291 against 0.317-0.319 Gbp/s
// at 32 768 reads per step in one call, the 4-pair gap fills 253 against 328 GCUPS, the 8-pair ones 86 against 176 (profiles/r06_dual.txt): the hull arithmetic,
// query fetch, boundary lane and traceback row of a row are still computed PER ALIGNMENT (scalar code that does not vectorise over two different hulls), so a wavefront
// retires the same instructions for its two alignments as two wavefronts did for one each — on one dependent chain instead of two, with 128 / 245 registers (4 / 2
// wavefronts per SIMD) instead of 54 / 76. Sharing a row loop only pays for alignments with the SAME (qlen, tlen), which a size-sorted list rarely offers.
static std::atomic<int> g_dual(-1);
static int ksw_dual_on()
{
	if (g_dual.load(std::memory_order_relaxed) < 0) g_dual = getenv("WM_KSW_DUAL") && atoi(getenv("WM_KSW_DUAL")) != 0;
	return g_dual.load(std::memory_order_relaxed);
}
extern "C" void wm_ksw_set_dual(int on) { g_dual = on ? 1 : 0; }
extern "C" int wm_ksw_dual_enabled(void) { return ksw_dual_on(); }

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
		else if (BP <= 8 && ksw_dual_on()) hipLaunchKernelGGL((ksw_dual_kernel<(BP <= 8 ? 2 * BP : 16)>), dim3((n + 1) / 2), b, 0, s, sc, jobs, order, n, seqs, tb, res);
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
		// exact extensions of the 8-pair window whose band does not clip are a few dozen long jobs per batched call (0.35 % of the cells): as a class of their own they are
		// ~140 near-empty launches per step, each as long as its longest job. The CLIP instantiation is a superset (it serves the jobs with an N the same way), so they join
		// the clipped exact 8-pair class (WM_KSW_MERGE_P8X=0: a class of their own, A/B)
		static const bool merge_p8x = !(getenv("WM_KSW_MERGE_P8X") && atoi(getenv("WM_KSW_MERGE_P8X")) == 0);
		if (merge_p8x && (d.klass & ~7) == WM_KSW_P8 && (d.klass & 4) && !(d.klass & 2)) d.klass |= 2;
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
int reads_alloc(wm_ctx_t *c, size_t cap)
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

