// oracle/wm_binding.cpp — TEST INFRASTRUCTURE: the reference-side binding of INTEGRATION.md (Level 0 / 1), made real.
//
// This file is compiled TOGETHER WITH THE REFERENCE'S OWN SOURCES (oracle/Makefile target `wm`, against /root/reference/src/minimap.h)
// into oracle/_ref/winnowmap_wm: the reference's CLI, option parser, index builder and main() unchanged, with its per-file mapping entry
// point mm_map_file (src/map.c:1273, called from src/main.c:419) redirected to libwmgpu.so by the linker (-Wl,--wrap=mm_map_file) — no
// reference source is modified or copied. It is what a Winnowmap maintainer would write to adopt the library:
//   * the index the reference has just built (mm_idx_t) is handed over through the reference's own index file format (mm_idx_dump,
//     src/index.c:515 → wm_index_load), the -W list through opt->kmer_freq_filename (the reference does not persist its bloom filter);
//   * every field of mm_mapopt_t goes into wm_mapopt_t (same names), so presets AND individual command-line options carry over;
//   * records are written by wm_map_file to stdout exactly where the reference writes them (the SAM header was printed by main already);
//   * --split-prefix (a reference indexed in parts): mm_split_merge is wrapped as well and runs wm_map_file_split over the parts main presented.
// WM_BACKEND=cpu in the environment runs the reference's own mm_map_file instead (A/B inside one binary).
// tests/test_binding_gpu.py diffs `winnowmap_wm ...` against `winnowmap_ref ...`.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <unistd.h>
#include "minimap.h"
#include "../include/wm_gpu.h"

extern "C" int __real_mm_map_file(const mm_idx_t *idx, const char *fn, const mm_mapopt_t *opt, int n_threads);
extern "C" int __real_mm_split_merge(int n_segs, const char **fn, const mm_mapopt_t *opt, int n_split_idx);

namespace {
struct Backend {
	const mm_idx_t *for_idx = 0;
	wm_ctx_t *ctx = 0; wm_index_t *idx = 0; wm_mapper_t *mapper = 0;
	void close()
	{
		if (mapper) wm_mapper_destroy(mapper);
		if (idx) wm_index_destroy(idx);
		if (ctx) wm_ctx_destroy(ctx);
		mapper = 0; idx = 0; ctx = 0; for_idx = 0;
	}
} g_be;

void copy_opt(const mm_mapopt_t *o, wm_mapopt_t *w)
{
	memset(w, 0, sizeof(*w));
#define CP(f) w->f = o->f
	CP(flag); CP(seed); CP(sdust_thres); CP(max_qlen); CP(bw); CP(max_gap); CP(max_gap_ref); CP(min_gap_ref); CP(max_frag_len);
	CP(max_chain_skip); CP(max_chain_iter); CP(min_cnt); CP(min_chain_score); CP(chain_gap_scale);
	w->SVaware = o->SVaware ? 1 : 0;
	CP(SVawareMinReadLength); CP(suffixSampleOffset); CP(min_mapq); CP(min_qcov); CP(minPrefixLength); CP(maxPrefixLength); CP(prefixIncrementFactor);
	CP(stage2_bw); CP(stage2_zdrop_inv); CP(stage2_max_gap); CP(mask_level); CP(mask_len); CP(pri_ratio); CP(best_n);
	CP(max_join_long); CP(max_join_short); CP(min_join_flank_sc); CP(min_join_flank_ratio); CP(alt_drop);
	CP(a); CP(b); CP(q); CP(e); CP(q2); CP(e2); CP(sc_ambi); CP(zdrop); CP(zdrop_inv); CP(end_bonus); CP(min_dp_max); CP(min_ksw_len);
	CP(max_clip_ratio); CP(mid_occ_frac); CP(min_mid_occ); CP(mid_occ); CP(max_occ); CP(mini_batch_size); CP(max_sw_mat);
	CP(noncan); CP(junc_bonus); CP(anchor_ext_len); CP(anchor_ext_shift);
#undef CP
}

// the reference's index -> the library's, through the reference's own index file format (mm_idx_dump, src/index.c:515)
int take_index(const mm_idx_t *mi, const mm_mapopt_t *opt, wm_index_t **out)
{
	char tmpl[] = "/tmp/wm_binding_XXXXXX";
	const int fd = mkstemp(tmpl);
	if (fd < 0) return -1;
	FILE *fp = fdopen(fd, "wb");
	mm_idx_dump(fp, mi);
	fclose(fp);
	const int rc = wm_index_load(tmpl, opt->kmer_freq_filename, out);
	unlink(tmpl);
	return rc ? -1 : 0;
}

// --split-prefix (a reference indexed in parts, src/main.c:365-429): main calls mm_map_file once per index part and mm_split_merge at the end. Bound:
// every part's index is taken over when main presents it (nothing is mapped yet, no <prefix>.NNNN.tmp is written), and the wrapped mm_split_merge
// runs the library's twin of the whole flow, wm_map_file_split, over the parts.
struct SplitState { std::vector<wm_index_t*> parts; std::vector<int> seen; int n_threads = 1; } g_split;       // seen: mm_idx_t::index, the part's ordinal

int open_backend(const mm_idx_t *mi, const mm_mapopt_t *opt, int n_threads)
{
	g_be.close();
	if (wm_ctx_create(0, 0, &g_be.ctx)) return -1;                       // fails without a GPU: the library has no CPU path
	if (take_index(mi, opt, &g_be.idx)) return -1;
	if (wm_index_upload(g_be.ctx, g_be.idx)) return -1;
	if (mi->I) {                                                         // --junc-bed: main has read the annotation into the index (src/main.c:416)
		// (mm_idx_intv_s is private to src/index.c:40-48; in-tree this would be an accessor next to mm_idx_bed_junc)
		struct Intv1 { int32_t st, en, max; int32_t score:30, strand:2; };
		struct Intv { int32_t n, m; Intv1 *a; };
		const Intv *I = (const Intv*)mi->I;
		for (uint32_t c = 0; c < mi->n_seq; ++c) {
			const Intv &r = I[c];
			std::vector<int32_t> st(r.n), en(r.n), sd(r.n);
			for (int32_t i = 0; i < r.n; ++i) st[i] = r.a[i].st, en[i] = r.a[i].en, sd[i] = r.a[i].strand;
			if (r.n && wm_index_add_junc(g_be.idx, (int)c, r.n, st.data(), en.data(), sd.data())) return -1;
		}
	}
	wm_mapopt_t wo;
	copy_opt(opt, &wo);
	if (wm_mapper_create_opt(g_be.ctx, g_be.idx, &wo, &g_be.mapper)) return -1;
	if (wm_mapper_set_threads(g_be.mapper, n_threads > 1 ? n_threads : 1, 0)) return -1;
	wm_mapper_set_sam_header(g_be.mapper, 0);                            // main() has printed it (src/main.c:393)
	g_be.for_idx = mi;
	return 0;
}
} // namespace

extern "C" int __wrap_mm_map_file(const mm_idx_t *idx, const char *fn, const mm_mapopt_t *opt, int n_threads)
{
	const char *be = getenv("WM_BACKEND");
	if (be && strcmp(be, "cpu") == 0) return __real_mm_map_file(idx, fn, opt, n_threads);
	if (opt->flag & (MM_F_SR | MM_F_FRAG_MODE)) {
		fprintf(stderr, "[wm_gpu] short-read / fragment modes are outside the library's path (single-segment long reads): using the CPU path\n");
		return __real_mm_map_file(idx, fn, opt, n_threads);
	}
	if (opt->split_prefix) {                                             // one call per (index part, reads file): keep the part, map in mm_split_merge
		if (g_split.seen.empty() || g_split.seen.back() != idx->index) {
			wm_index_t *part = 0;
			if (take_index(idx, opt, &part)) { fprintf(stderr, "[wm_gpu] %s\n", wm_last_error()); return -1; }
			g_split.parts.push_back(part); g_split.seen.push_back(idx->index);
		}
		g_split.n_threads = n_threads;
		return 0;
	}
	if (g_be.for_idx != idx && open_backend(idx, opt, n_threads)) {
		fprintf(stderr, "[wm_gpu] %s\n", wm_last_error());
		return -1;
	}
	fflush(stdout);
	double st[6];
	const int rc = wm_map_file(g_be.mapper, fn, "-", opt->mini_batch_size, st);
	if (rc) { fprintf(stderr, "[wm_gpu] %s\n", wm_last_error()); return -1; }
	if (mm_verbose >= 3) fprintf(stderr, "[M::wm_gpu] mapped %.0f sequences (%.0f bases) in %.0f mini-batch(es) on the GPU\n", st[0], st[1], st[2]);
	return 0;
}

extern "C" int __wrap_mm_split_merge(int n_segs, const char **fn, const mm_mapopt_t *opt, int n_split_idx)
{
	const char *be = getenv("WM_BACKEND");
	if ((be && strcmp(be, "cpu") == 0) || g_split.parts.empty()) return __real_mm_split_merge(n_segs, fn, opt, n_split_idx);
	wm_ctx_t *ctx = 0;
	if (wm_ctx_create(0, 0, &ctx)) { fprintf(stderr, "[wm_gpu] %s\n", wm_last_error()); return -1; }
	wm_mapopt_t wo;
	copy_opt(opt, &wo);
	wm_set_cmdline(-1, 0);                                               // main has printed the @PG line (mm_write_sam_hdr(0, ...), src/main.c:395)
	fflush(stdout);
	int rc = 0;
	for (int i = 0; i < n_segs && !rc; ++i) {                            // (single-segment reads: one merged pass per reads file)
		double st[6];
		rc = wm_map_file_split(ctx, (int)g_split.parts.size(), g_split.parts.data(), &wo, g_split.n_threads > 1 ? g_split.n_threads : 1, fn[i], "-", opt->mini_batch_size, st);
		if (rc) fprintf(stderr, "[wm_gpu] %s\n", wm_last_error());
	}
	for (wm_index_t *p : g_split.parts) wm_index_destroy(p);
	g_split.parts.clear(); g_split.seen.clear();
	wm_ctx_destroy(ctx);
	return rc ? -1 : 0;
}

__attribute__((destructor)) static void wm_binding_fini() { g_be.close(); }
