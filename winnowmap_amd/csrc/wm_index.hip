// wm_index.hip — libwmgpu.so, index unit: the sketch / seed / chain batch operations, the index built on and uploaded to the device, .mmi load / save,
// the -W list counted on the device (rocPRIM sorts: the one library kernel family of the product) — wm_sketch_*, wm_seed_batch, wm_chain_batch, wm_index_*, wm_write_repetitive_kmers*.
#include "wm_rt.h"
#include <rocprim/rocprim.hpp>          // device radix sort + run-length encode
#include "simt.h"
#include "reads2bit.h"
#include "sketch_kernel.h"
#include "seedchain_kernel.h"
#include "host/wm_chain.h"


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
// ... with the whole predecessor window of an anchor per step (chain_block_wide, round 6): blockDim / 64 wavefronts x KT tiles
template <int KT>
__global__ __launch_bounds__(1024) void chain_kernel_wide(const wm_chain_job_t *jobs, const int *order, const wm128_t *anchors, int *fpvt, int W, int kt_first)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int j = order[blockIdx.x];
	const wm_chain_job_t jb = jobs[j];
	uint64_t *sx = (uint64_t*)smem, *sy = sx + W;
	int *sf = (int*)(sy + W), *sp = sf + W, *st = sp + W, *pub = st + W;
	int *gf = fpvt + jb.a_off * 4, *gp = gf + jb.n, *gt = gp + 2 * (size_t)jb.n;
	wmk::chain_block_wide<KT>(jb, anchors, (int)(blockDim.x >> 6), kt_first, W, sx, sy, sf, sp, st, pub, gf, gp, gt);
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

int sketch_batch_impl(wm_ctx_t *c, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len, const uint8_t *resident,
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

int wm_index_build_seqs_dev(wm_ctx_t *c, const wm::IdxOpt &io, std::vector<std::string> &names, std::vector<std::string> &seqs, const std::string &kmer_file, int n_threads,
                                   wm_index_t **out, double *stats, bool replace_ok, double t0);
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
int wm_index_build_seqs_dev(wm_ctx_t *c, const wm::IdxOpt &io, std::vector<std::string> &names, std::vector<std::string> &seqs, const std::string &kmer_file_s, int n_threads,
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
		if (rc == WM_ENOMEM && strstr(wm_err_text(), "minimizer output pool")) { mv.resize(bases + n + 1); rc = sketch_batch_impl(c, n, codes.get(), tot, off.data(), len.data(), 0, mv.data(), mv.size(), ooff.data(), cnt.data()); }
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
int sketch_batch_impl(wm_ctx_t *c, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len, const uint8_t *resident,
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
size_t sketch_long_bytes(int n, const wm_sketch_job_t *h_jobs, bool allow_long, bool hpc)
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
int sketch_launch(wm_ctx_t *c, int n, const wm_sketch_job_t *h_jobs, const wm_sketch_job_t *d_jobs, const int *d_ord, const uint8_t *d_seqs,
                         double *d_so, uint64_t *d_sx, uint32_t *d_sy, uint32_t *d_sl, wm128_t *d_out, int *d_cnt, bool allow_long, uint8_t *mem, size_t mem_bytes)
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

int sketch_batch_impl(wm_ctx_t *c, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len, const uint8_t *resident,
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
				// WM_CHAIN_WIDE=0: the round-5 dense fill; WM_CHAIN_WIDE_GEOM=<wavefronts>x<tiles per wavefront> (16x5 | 8x10 | 16x3 | 8x5 | 4x10; tools/chain_fill_probe.py)
				static const bool wide = !(getenv("WM_CHAIN_WIDE") && atoi(getenv("WM_CHAIN_WIDE")) == 0);
				int gw = 16, gk = 5;
				if (const char *g = getenv("WM_CHAIN_WIDE_GEOM")) sscanf(g, "%dx%d", &gw, &gk);
				if ((gk != 10 && gk != 3 && gk != 5) || gw < 1 || gw > 16 || gw * gk > 128) { gw = 16; gk = 5; }      // (the instantiated tile counts; chain_block_wide: NT <= 128)
				const int kt_first = getenv("WM_CHAIN_WIDE_FIRST") ? std::min(gk, std::max(1, atoi(getenv("WM_CHAIN_WIDE_FIRST")))) : gk;      // tiles per wavefront in an anchor's first step (seedchain_kernel.h)
				if (k == 0 && wide) {
					const size_t lds = (size_t)4096 * 28 + (size_t)gw * gk * 69 * 4 + 64;
					if (gk == 10) { HIPCHK(hipFuncSetAttribute((const void*)chain_kernel_wide<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(chain_kernel_wide<10>, dim3(e - b), dim3(64 * gw), lds, c->stream, d_jobs, d_order + b, d_a, d_fpvt, 4096, kt_first); }
					else if (gk == 3) { HIPCHK(hipFuncSetAttribute((const void*)chain_kernel_wide<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(chain_kernel_wide<3>, dim3(e - b), dim3(64 * gw), lds, c->stream, d_jobs, d_order + b, d_a, d_fpvt, 4096, kt_first); }
					else { HIPCHK(hipFuncSetAttribute((const void*)chain_kernel_wide<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(chain_kernel_wide<5>, dim3(e - b), dim3(64 * gw), lds, c->stream, d_jobs, d_order + b, d_a, d_fpvt, 4096, kt_first); }
				} else if (k == 0) hipLaunchKernelGGL(chain_kernel_block<NWV>, dim3(e - b), dim3(64 * NWV), (size_t)4096 * 28 + NWV * 69 * 4 + 64, c->stream, d_jobs, d_order + b, d_a, d_fpvt, 4096);
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

