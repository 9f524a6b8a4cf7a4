// wm_gpu.hip — libwmgpu.so: C-ABI shim (include/wm_gpu.h) + gfx950 kernel entry points.
// Host side: plans a batch (kernel class, traceback pitch, arena offsets), uploads, launches on the
// context's stream, measures kernel time with HIP events on that stream, and gathers results.
// There is no CPU compute path in this file: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include "simt.h"
#include "ksw_kernel.h"
#include "ksw_plan.h"

// ======================================================================================================
// kernels
// ======================================================================================================
template <int B, bool CLIP, bool HASN>
__global__ __launch_bounds__(64) void ksw_dp_kernel(wm_ksw_score_t sc, const wm_ksw_djob_t *__restrict__ jobs,
                                                     const int *__restrict__ order, const uint8_t *__restrict__ seqs,
                                                     uint8_t *__restrict__ tb, wm_ksw_dres_t *__restrict__ res)
{
	const int j = order[blockIdx.x];
	wmk::ksw_dp_wave<B, CLIP, HASN>(sc, jobs[j], seqs, tb, res + j);
}

// one thread per alignment: walk the traceback, write run-length ops (backtrack order) into the job's slot
__global__ __launch_bounds__(64) void ksw_backtrack_kernel(int n, const wm_ksw_djob_t *__restrict__ jobs, const uint8_t *__restrict__ tb,
                                                            wm_ksw_dres_t *__restrict__ res, uint32_t *__restrict__ cig_scratch, int *__restrict__ err)
{
	const int j = blockIdx.x * 64 + threadIdx.x;
	if (j >= n) return;
	wm_ksw_dres_t r = res[j];
	int nc = 0;
	if (r.bt_i >= 0) {
		nc = wmk::ksw_backtrack_thread(jobs[j], tb, r.bt_i, r.bt_j, cig_scratch + jobs[j].cig_off, jobs[j].cig_cap);
		if (nc < 0) { atomicExch(err, 1); nc = 0; }
	}
	res[j].n_cigar = nc;
}

// exclusive prefix sum of n_cigar (single block; n is at most a few hundred thousand)
__global__ __launch_bounds__(1024) void ksw_scan_kernel(int n, const wm_ksw_dres_t *__restrict__ res, uint32_t *__restrict__ off, uint32_t *__restrict__ total)
{
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
	const int j = blockIdx.x, n = res[j].n_cigar;
	const uint32_t *src = cig_scratch + jobs[j].cig_off;
	const bool rev = (jobs[j].flag & KSW_F_REV_CIGAR) != 0;
	for (int i = threadIdx.x; i < n; i += 64) {
		const uint32_t dst = off[j] + (uint32_t)i;
		if (dst < pool_cap) pool[dst] = src[rev ? i : n - 1 - i];
	}
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
	uint8_t *arena;
	size_t arena_bytes, arena_used;
	hipEvent_t ev[4];
	float last_ms;
};

struct wm_ksw_dev_batch_s {
	int n_jobs;
	wm_ksw_score_t sc;
	std::vector<wm_ksw_djob_t> jobs;            // host copy
	std::vector<int> order[WM_KSW_NCLASS];      // job indices per class, largest first
	std::vector<int> degenerate;                // jobs the reference returns from early (src/ksw2_extd2_sse.c:68,92)
	// device pointers (inside the arena)
	wm_ksw_djob_t *d_jobs; int *d_order; uint8_t *d_seqs, *d_tb; wm_ksw_dres_t *d_res; uint32_t *d_cig, *d_off, *d_total, *d_pool; int *d_err;
	size_t pool_cap, arena_mark;
	uint64_t cells, tb_bytes;
	float dp_ms, bt_ms;
	uint32_t total_ops;
};

extern "C" const char *wm_last_error(void) { return g_err; }

extern "C" int wm_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" int wm_ctx_create(int device, size_t arena_bytes, wm_ctx_t **out)
{
	int n = 0;
	*out = 0;
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
	for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&c->ev[i]));
	c->arena_used = 0; c->last_ms = 0;
	*out = c;
	return WM_OK;
}

extern "C" void wm_ctx_destroy(wm_ctx_t *c)
{
	if (!c) return;
	hipSetDevice(c->device);
	hipStreamSynchronize(c->stream);
	for (int i = 0; i < 4; ++i) hipEventDestroy(c->ev[i]);
	hipStreamDestroy(c->stream);
	hipFree(c->arena);
	delete c;
}

extern "C" float wm_last_kernel_ms(const wm_ctx_t *c) { return c ? c->last_ms : 0.f; }

static void *arena_take(wm_ctx_t *c, size_t bytes)
{
	size_t a = (c->arena_used + 255) & ~(size_t)255;
	if (a + bytes > c->arena_bytes) return 0;
	c->arena_used = a + bytes;
	return c->arena + a;
}

template <int B> static void launch_dp(int clip, int hasn, int n, hipStream_t s, const wm_ksw_score_t &sc, const wm_ksw_djob_t *jobs, const int *order,
                                       const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res)
{
	dim3 g(n), b(64);
	if (clip && hasn) hipLaunchKernelGGL((ksw_dp_kernel<B, true, true>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
	else if (clip) hipLaunchKernelGGL((ksw_dp_kernel<B, true, false>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
	else if (hasn) hipLaunchKernelGGL((ksw_dp_kernel<B, false, true>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
	else hipLaunchKernelGGL((ksw_dp_kernel<B, false, false>), g, b, 0, s, sc, jobs, order, seqs, tb, res);
}

extern "C" int wm_ksw_dev_prepare(wm_ctx_t *c, const wm_ksw_score_t *sc_in, int n_jobs, const wm_ksw_job_t *jobs,
                                  const uint8_t *seqs, size_t seqs_bytes, wm_ksw_dev_batch_t **out)
{
	*out = 0;
	if (!c) return set_err(WM_EINVAL, "null context");
	if (n_jobs < 0) return set_err(WM_EINVAL, "n_jobs < 0");
	wm_ksw_score_t sc = *sc_in;
	if (!wm_ksw_score_ok(&sc)) return set_err(WM_EINVAL, "unsupported scoring parameters (need match>0, mismatch<0, sc_ambi<=0, q,e,q2>0, e2>=0, (q+e)+(q2+e2)<=127 as src/options.c:166-176)");
	if (sc.q2 + sc.e2 < sc.q + sc.e) { int8_t t = sc.q; sc.q = sc.q2; sc.q2 = t; t = sc.e; sc.e = sc.e2; sc.e2 = t; }   // src/ksw2_extd2_sse.c:70
	HIPCHK(hipSetDevice(c->device));
	wm_ksw_dev_batch_t *b = new wm_ksw_dev_batch_t();
	b->n_jobs = n_jobs; b->sc = sc; b->cells = b->tb_bytes = 0; b->dp_ms = b->bt_ms = 0; b->total_ops = 0;
	b->arena_mark = c->arena_used;
	b->jobs.resize(n_jobs);
	// the reference returns before doing anything when a mismatch can never be seen (:92)
	const int n_sc = sc.sc_ambi == 0 ? -sc.e2 : sc.sc_ambi;
	int min_sc = sc.mismatch < n_sc ? sc.mismatch : n_sc;
	if (sc.sc_ambi < min_sc) min_sc = sc.sc_ambi;
	const bool never = -min_sc > 2 * (sc.q + sc.e);
	uint64_t tb_off = 0, cig_off = 0;
	std::vector<uint64_t> cells(n_jobs, 0);
	for (int i = 0; i < n_jobs; ++i) {
		const wm_ksw_job_t &s = jobs[i];
		wm_ksw_djob_t &d = b->jobs[i];
		memset(&d, 0, sizeof(d));
		d.q_off = s.q_off; d.t_off = s.t_off; d.qlen = s.qlen; d.tlen = s.tlen;
		d.w = s.w; d.zdrop = s.zdrop; d.end_bonus = s.end_bonus; d.flag = s.flag;
		if (s.flag & (0x01 | 0x04 | 0x10 | 0x100 | 0x200 | 0x400)) { delete b; return set_err(WM_EINVAL, "job %d: KSW_EZ_SCORE_ONLY/GENERIC_SC/APPROX_DROP/SPLICE flags are not used by the mapper (src/align.c) and not supported", i); }
		if (s.qlen <= 0 || s.tlen <= 0 || never) { d.klass = -1; b->degenerate.push_back(i); continue; }           // :68,:92
		if ((size_t)s.q_off + s.qlen > seqs_bytes || (size_t)s.t_off + s.tlen > seqs_bytes) { delete b; return set_err(WM_EINVAL, "job %d: sequence offsets outside seqs", i); }
		int n_col;
		d.klass = wm_ksw_classify(s.qlen, s.tlen, s.w, wm_ksw_has_n(seqs + s.q_off, s.qlen) | wm_ksw_has_n(seqs + s.t_off, s.tlen), &n_col);
		if (d.klass >= WM_KSW_GENERIC) { delete b; return set_err(WM_EINVAL, "job %d: band hull of %d lanes exceeds the register kernels (generic kernel not built yet)", i, n_col); }
		d.n_col = n_col;
		d.tb_off = tb_off;
		const uint64_t rows = (uint64_t)s.qlen + s.tlen - 1;
		tb_off += (rows * n_col + 15) & ~(uint64_t)15;
		d.cig_off = (uint32_t)cig_off; d.cig_cap = s.qlen + s.tlen + 2;
		cig_off += d.cig_cap;
		uint64_t band;
		cells[i] = wm_ksw_cells(s.qlen, s.tlen, s.w, &band);
		b->cells += band; b->tb_bytes += cells[i];
		b->order[d.klass].push_back(i);
	}
	for (int k = 0; k < WM_KSW_NCLASS; ++k)
		std::sort(b->order[k].begin(), b->order[k].end(), [&](int x, int y) { return cells[x] != cells[y] ? cells[x] > cells[y] : x < y; });
	// device buffers
	const size_t nj = n_jobs > 0 ? n_jobs : 1;
	b->d_jobs = (wm_ksw_djob_t*)arena_take(c, nj * sizeof(wm_ksw_djob_t));
	b->d_order = (int*)arena_take(c, nj * sizeof(int));
	b->d_res = (wm_ksw_dres_t*)arena_take(c, nj * sizeof(wm_ksw_dres_t));
	b->d_off = (uint32_t*)arena_take(c, nj * 4 + 64);
	b->d_total = (uint32_t*)arena_take(c, 64);
	b->d_err = (int*)arena_take(c, 64);
	b->d_seqs = (uint8_t*)arena_take(c, seqs_bytes + 64);
	b->d_cig = (uint32_t*)arena_take(c, (cig_off + 16) * 4);
	b->pool_cap = cig_off + 16;
	b->d_pool = (uint32_t*)arena_take(c, b->pool_cap * 4);
	b->d_tb = (uint8_t*)arena_take(c, tb_off + 64);
	if (!b->d_jobs || !b->d_order || !b->d_res || !b->d_off || !b->d_total || !b->d_err || !b->d_seqs || !b->d_cig || !b->d_pool || !b->d_tb) {
		c->arena_used = b->arena_mark;
		delete b;
		return set_err(WM_ENOMEM, "batch needs %.1f MB of traceback + buffers; arena is %.1f MB", (tb_off + cig_off * 8 + seqs_bytes) / 1048576.0, c->arena_bytes / 1048576.0);
	}
	std::vector<int> ord;
	ord.reserve(nj);
	for (int k = 0; k < WM_KSW_NCLASS; ++k) ord.insert(ord.end(), b->order[k].begin(), b->order[k].end());
	HIPCHK(hipMemcpyAsync(b->d_jobs, b->jobs.data(), n_jobs * sizeof(wm_ksw_djob_t), hipMemcpyHostToDevice, c->stream));
	if (!ord.empty()) HIPCHK(hipMemcpyAsync(b->d_order, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(b->d_seqs, seqs, seqs_bytes, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	*out = b;
	return WM_OK;
}

extern "C" int wm_ksw_dev_run(wm_ctx_t *c, wm_ksw_dev_batch_t *b)
{
	HIPCHK(hipSetDevice(c->device));
	const int n = b->n_jobs;
	if (n == 0) return WM_OK;
	// degenerate jobs get the result of ksw_reset_extz (src/ksw2.h:153-158)
	HIPCHK(hipMemsetAsync(b->d_err, 0, 4, c->stream));
	if (!b->degenerate.empty()) {
		wm_ksw_dres_t z;
		memset(&z, 0, sizeof(z));
		z.max_q = z.max_t = z.mqe_t = z.mte_q = -1; z.score = z.mqe = z.mte = KSW_NEG_INF; z.bt_i = z.bt_j = -1;
		for (int j : b->degenerate) HIPCHK(hipMemcpyAsync(b->d_res + j, &z, sizeof(z), hipMemcpyHostToDevice, c->stream));
	}
	HIPCHK(hipEventRecord(c->ev[0], c->stream));
	int off = 0;
	for (int k = 0; k < WM_KSW_GENERIC; ++k) {
		const int nk = (int)b->order[k].size();
		if (nk == 0) continue;
		const int clip = k >> 1 & 1, hasn = k & 1;
		switch (k & ~3) {
		case WM_KSW_B4: launch_dp<4>(clip, hasn, nk, c->stream, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
		case WM_KSW_B8: launch_dp<8>(clip, hasn, nk, c->stream, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
		default: launch_dp<16>(clip, hasn, nk, c->stream, b->sc, b->d_jobs, b->d_order + off, b->d_seqs, b->d_tb, b->d_res); break;
		}
		off += nk;
	}
	HIPCHK(hipEventRecord(c->ev[1], c->stream));
	hipLaunchKernelGGL(ksw_backtrack_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, b->d_jobs, b->d_tb, b->d_res, b->d_cig, b->d_err);
	hipLaunchKernelGGL(ksw_scan_kernel, dim3(1), dim3(1024), 0, c->stream, n, b->d_res, b->d_off, b->d_total);
	hipLaunchKernelGGL(ksw_gather_kernel, dim3(n), dim3(64), 0, c->stream, b->d_jobs, b->d_res, b->d_off, b->d_cig, b->d_pool, (uint32_t)b->pool_cap);
	HIPCHK(hipEventRecord(c->ev[2], c->stream));
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipEventElapsedTime(&b->dp_ms, c->ev[0], c->ev[1]));
	HIPCHK(hipEventElapsedTime(&b->bt_ms, c->ev[1], c->ev[2]));
	c->last_ms = b->dp_ms + b->bt_ms;
	int err = 0;
	HIPCHK(hipMemcpy(&err, b->d_err, 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(&b->total_ops, b->d_total, 4, hipMemcpyDeviceToHost));
	if (err) return set_err(WM_EINTERNAL, "cigar slot overflow in backtrack");
	return WM_OK;
}

extern "C" int wm_ksw_dev_fetch(wm_ctx_t *c, wm_ksw_dev_batch_t *b, wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used)
{
	HIPCHK(hipSetDevice(c->device));
	const int n = b->n_jobs;
	if (cigar_used) *cigar_used = b->total_ops;
	if (n == 0) return WM_OK;
	std::vector<wm_ksw_dres_t> res(n);
	std::vector<uint32_t> off(n);
	HIPCHK(hipMemcpy(res.data(), b->d_res, n * sizeof(wm_ksw_dres_t), hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(off.data(), b->d_off, n * 4, hipMemcpyDeviceToHost));
	for (int i = 0; i < n; ++i) {
		wm_ksw_result_t &o = results[i];
		const wm_ksw_dres_t &r = res[i];
		o.max = r.max; o.zdropped = r.zdropped; o.max_q = r.max_q; o.max_t = r.max_t; o.mqe = r.mqe; o.mqe_t = r.mqe_t;
		o.mte = r.mte; o.mte_q = r.mte_q; o.score = r.score; o.reach_end = r.reach_end; o.n_cigar = r.n_cigar; o.cig_off = off[i];
	}
	if (b->total_ops > cigar_cap) return set_err(WM_ENOMEM, "cigar_pool too small: need %u ops", b->total_ops);
	if (b->total_ops) HIPCHK(hipMemcpy(cigar_pool, b->d_pool, (size_t)b->total_ops * 4, hipMemcpyDeviceToHost));
	return WM_OK;
}

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

extern "C" int wm_ksw_batch(wm_ctx_t *c, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes,
                            wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used)
{
	// process in chunks whose traceback fits the arena
	if (!c) return set_err(WM_EINVAL, "null context");
	size_t used = 0;
	int i0 = 0;
	const size_t budget = (size_t)(c->arena_bytes * 0.8);
	while (i0 < n_jobs || (n_jobs == 0 && i0 == 0)) {
		int i1 = i0;
		size_t need = seqs_bytes;
		while (i1 < n_jobs) {
			const wm_ksw_job_t &s = jobs[i1];
			size_t t = 0;
			if (s.qlen > 0 && s.tlen > 0) t = ((size_t)s.qlen + s.tlen) * ((size_t)wm_ksw_ncol(s.qlen, s.tlen, s.w) + 8) + 256;
			if (i1 > i0 && need + t > budget) break;
			need += t; ++i1;
		}
		wm_ksw_dev_batch_t *b = 0;
		int rc = wm_ksw_dev_prepare(c, sc, i1 - i0, jobs + i0, seqs, seqs_bytes, &b);
		if (rc) return rc;
		rc = wm_ksw_dev_run(c, b);
		size_t u = 0;
		if (!rc) rc = wm_ksw_dev_fetch(c, b, results + i0, cigar_pool + used, cigar_cap - used, &u);
		wm_ksw_dev_free(c, b);
		if (rc) { if (cigar_used) *cigar_used = used + u; return rc; }
		for (int i = i0; i < i1; ++i) results[i].cig_off += (uint32_t)used;
		used += u;
		i0 = i1;
		if (n_jobs == 0) break;
	}
	if (cigar_used) *cigar_used = used;
	return WM_OK;
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
