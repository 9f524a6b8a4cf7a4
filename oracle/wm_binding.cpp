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
//   * --split-prefix (a reference indexed in parts): every (part, reads file) call maps and spills at once (wm_split_add_part); the wrapped mm_split_merge
//     runs the merge passes (wm_split_finish). One index part in memory at a time, as in the reference's own main.
// WM_BACKEND=cpu in the environment runs the reference's own mm_map_file instead (A/B inside one binary).
// tests/test_binding_gpu.py diffs `winnowmap_wm ...` against `winnowmap_ref ...`.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
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

// --split-prefix (a reference indexed in parts, src/main.c:365-429): main calls mm_map_file once per (index part, reads file) and mm_split_merge at the
// end. Bound one part at a time, like main itself holds them: the part main presents is taken over, the reads file is mapped against it at once and the
// hits are spilled by the library (wm_split_add_part — its twin of <prefix>.NNNN.tmp); the part is dropped when main presents the next one. The wrapped
// mm_split_merge runs the merge passes (wm_split_finish), one per reads file.
struct SplitState {
	wm_ctx_t *ctx = 0; wm_index_t *part = 0; int part_no = -1;
	std::vector<std::string> files; std::vector<wm_split_t*> runs;
	void drop() { for (wm_split_t *r : runs) if (r) wm_split_abort(r); runs.clear(); files.clear(); if (part) wm_index_destroy(part); part = 0; part_no = -1; if (ctx) wm_ctx_destroy(ctx); ctx = 0; }
} g_split;

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
	if (opt->split_prefix) {                                             // one call per (index part, reads file)
		if (!g_split.ctx && wm_ctx_create(0, 0, &g_split.ctx)) { fprintf(stderr, "[wm_gpu] %s\n", wm_last_error()); return -1; }
		if (g_split.part_no != idx->index) {                             // main has moved on to the next part: the one before is done with
			if (g_split.part) wm_index_destroy(g_split.part);
			g_split.part = 0; g_split.part_no = idx->index;
			if (take_index(idx, opt, &g_split.part)) { fprintf(stderr, "[wm_gpu] %s\n", wm_last_error()); return -1; }
		}
		size_t f = 0;
		while (f < g_split.files.size() && g_split.files[f] != fn) ++f;
		if (f == g_split.files.size()) {
			wm_mapopt_t wo;
			copy_opt(opt, &wo);
			wm_split_t *run = 0;
			if (wm_split_begin(g_split.ctx, &wo, idx->k, idx->w, n_threads > 1 ? n_threads : 1, fn, opt->mini_batch_size, &run)) { fprintf(stderr, "[wm_gpu] %s\n", wm_last_error()); return -1; }
			g_split.files.push_back(fn); g_split.runs.push_back(run);
		}
		if (wm_split_add_part(g_split.runs[f], g_split.part)) { fprintf(stderr, "[wm_gpu] %s\n", wm_last_error()); return -1; }
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
	if ((be && strcmp(be, "cpu") == 0) || g_split.runs.empty()) return __real_mm_split_merge(n_segs, fn, opt, n_split_idx);
	wm_set_cmdline(-1, 0);                                               // main has printed the @PG line (mm_write_sam_hdr(0, ...), src/main.c:395)
	fflush(stdout);
	int rc = 0;
	for (int i = 0; i < n_segs && !rc; ++i) {                            // (single-segment reads: one merged pass per reads file)
		size_t f = 0;
		while (f < g_split.files.size() && g_split.files[f] != fn[i]) ++f;
		if (f == g_split.files.size() || !g_split.runs[f]) { fprintf(stderr, "[wm_gpu] %s was not mapped against the index parts\n", fn[i]); rc = -1; break; }
		double st[6];
		rc = wm_split_finish(g_split.runs[f], "-", st);                  // (frees the run, also on error)
		g_split.runs[f] = 0;
		if (rc) fprintf(stderr, "[wm_gpu] %s\n", wm_last_error());
	}
	g_split.drop();
	return rc ? -1 : 0;
}

__attribute__((destructor)) static void wm_binding_fini() { g_split.drop(); g_be.close(); }
