// wm_window.hip — libwmgpu.so, window unit: one MCAS window / stage-2 pass on the device in ONE call, nothing leaves HBM in between — sketch (mm_sketch)
// -> seed (collect_seed_hits) -> sort (radix_sort_128x) -> chain fill + extraction (mm_chain_dp): window_kernel.h, seedchain_kernel.h; wm_window_batch of include/wm_gpu.h.
#include "wm_rt.h"
#include "simt.h"
#include "reads2bit.h"
#include "sketch_kernel.h"
#include "seedchain_kernel.h"
#include "window_kernel.h"
#include "host/wm_core.h"

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
__global__ __launch_bounds__(1024) void win_chain_kernel_wide(const wm_chain_job_t *jobs, const int *list, const int *count, const wm128_t *anchors, int *fpvt, int W, int kt_first)
{
	WM_SETPRIO(2);
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	if ((int)blockIdx.x >= *count) return;
	const wm_chain_job_t jb = jobs[list[blockIdx.x]];
	uint64_t *sx = (uint64_t*)smem, *sy = sx + W;
	int *sf = (int*)(sy + W), *sp = sf + W, *st = sp + W, *pub = st + W;
	int *gf = fpvt + jb.a_off * 4, *gp = gf + jb.n, *gt = gp + 2 * (size_t)jb.n;
	wmk::chain_block_wide<KT>(jb, anchors, 16, kt_first, W, sx, sy, sf, sp, st, pub, gf, gp, gt);
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
int window_launch(wm_ctx_t *c, int n, const wm_window_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes, const wm128_t *pre, size_t n_pre_total,
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
			static const int kt_first = getenv("WM_CHAIN_WIDE_FIRST") ? std::min(5, std::max(1, atoi(getenv("WM_CHAIN_WIDE_FIRST")))) : 5;      // tiles per wavefront in an anchor's first step (seedchain_kernel.h)
			hipLaunchKernelGGL(win_chain_kernel_wide<5>, dim3(n), dim3(1024), (size_t)4096 * 28 + 80 * 69 * 4 + 64, c->stream, d_cj, d_lists, d_counts, d_a, d_fpvt, 4096, kt_first);
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
int window_fetch(wm_ctx_t *c, const WinDev &D, int n, wm_window_res_t *res, uint64_t *u_pool, wm128_t *a_pool)
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
int window_verdict(const WinDev &D, int round)
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

