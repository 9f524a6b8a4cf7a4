// oracle/wm_subst.cpp — TEST INFRASTRUCTURE: link-level substitutes for the hot functions of the reference.
//
// Compiled against the reference's own headers into oracle/_ref/winnowmap_subst (oracle/Makefile target `subst`): the reference's CLI with
// ALL of its objects, in which the C symbols
//     mm_sketch        src/mmpriv.h:61      ksw_extd2_sse    src/ksw2.h:60-61
//     mm_chain_dp      src/mmpriv.h:73      ksw_extz2_sse    src/ksw2.h:54-55
//     mm_idx_get       src/mmpriv.h:71      ksw_exts2_sse    src/ksw2.h:63-64
//                                           ksw_ll_qinit / ksw_ll_i16   src/ksw2.h:82-83
// have been renamed to ref_<name> (objcopy --redefine-sym on copies of sketch.o / chain.o / index.o / ksw2_dispatch.o / ksw2_ll_sse.o) and are DEFINED HERE with the
// exact signatures, ownership and kalloc conventions of the reference, each as a one-job call of the batched device operation behind it
// (wm_sketch_batch, wm_chain_batch, wm_ksw_batch, wm_ksw_exts2_batch). So mm_map_frag, mm_align_skeleton, the index builder … all run unchanged on top of the
// device kernels — slowly (one launch per call), which is the point of the batched entry points, but it proves the symbols are drop-ins.
// Cases the kernels do not cover (HPC sketching, multi-segment chaining, score-only DP) go to the renamed originals.
// WM_SUBST=off in the environment routes everything to the originals (A/B inside one binary). tests/test_binding_gpu.py diffs the output.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <mutex>
#include <vector>
#include "minimap.h"
#include "mmpriv.h"
#include "ksw2.h"
#include "kalloc.h"
#include "kvec.h"
#include "bloom_filter.hpp"
#include "../include/wm_gpu.h"

extern "C" {
void ref_mm_sketch(void *km, const char *str, int len, int w, int k, uint32_t rid, int is_hpc, mm128_v *p, const mm_idx_t *mi);
mm128_t *ref_mm_chain_dp(int max_dist_x, int min_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc, float gap_scale,
                         int is_cdna, int n_segs, int64_t n, mm128_t *a, int *n_u_, uint64_t **_u, void *km);
void ref_ksw_extd2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                       int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez);
void ref_ksw_extz2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                       int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez);
void ref_ksw_exts2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                       int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, ksw_extz_t *ez);
const uint64_t *ref_mm_idx_get(const mm_idx_t *mi, uint64_t minier, int *n);
void *ref_ksw_ll_qinit(void *km, int size, int qlen, const uint8_t *query, int m, const int8_t *mat);
int ref_ksw_ll_i16(void *q, int tlen, const uint8_t *target, int gapo, int gape, int *qe, int *te);
}

namespace {
std::mutex g_mu;                         // one device context, one stream: calls from the reference's worker threads take turns
wm_ctx_t *g_ctx = 0;
const void *g_filter_of = 0; int g_fk = 0, g_fw = 0;
bool off() { static const bool v = getenv("WM_SUBST") && strcmp(getenv("WM_SUBST"), "off") == 0; return v; }
wm_ctx_t *ctx()
{
	if (!g_ctx && wm_ctx_create(0, (size_t)2 << 30, &g_ctx)) { fprintf(stderr, "[wm_subst] %s\n", wm_last_error()); exit(1); }   // no GPU: no fallback
	return g_ctx;
}
void die(const char *what) { fprintf(stderr, "[wm_subst] %s: %s\n", what, wm_last_error()); exit(1); }
struct FilterView : bloom_filter { using bloom_filter::salt_; };             // the salts are protected members
}

extern "C" void mm_sketch(void *km, const char *str, int len, int w, int k, uint32_t rid, int is_hpc, mm128_v *p, const mm_idx_t *mi)
{
	if (off() || is_hpc) { ref_mm_sketch(km, str, len, w, k, rid, is_hpc, p, mi); return; }
	std::lock_guard<std::mutex> lk(g_mu);
	wm_ctx_t *c = ctx();
	if (g_filter_of != (const void*)mi->downFilter || g_fk != k || g_fw != w) {
		const FilterView *f = static_cast<const FilterView*>(mi->downFilter);
		uint32_t s0 = 0, s1 = 0;
		if (f && f->salt_.size() >= 2) s0 = f->salt_[0], s1 = f->salt_[1];
		if (f && f->salt_.size() != 2 && f->size() > 0) { ref_mm_sketch(km, str, len, w, k, rid, is_hpc, p, mi); return; }   // (the reference always builds 2 hashes, src/index.c:416-421)
		if (wm_sketch_set_filter(c, f ? f->table() : 0, f ? (size_t)(f->size() / 8) : 0, f ? f->size() : 0, s0, s1, k, w)) die("mm_sketch");
		g_filter_of = mi->downFilter; g_fk = k; g_fw = w;
	}
	std::vector<uint8_t> codes(len);
	for (int i = 0; i < len; ++i) codes[i] = seq_nt4_table[(uint8_t)str[i]];
	std::vector<wm128_t> out((size_t)len + 1);
	uint64_t off0 = 0, ooff = 0; int32_t ln = len, cnt = 0;
	if (wm_sketch_batch(c, 1, codes.data(), (size_t)len, &off0, &ln, out.data(), out.size(), &ooff, &cnt)) die("mm_sketch");
	for (int i = 0; i < cnt; ++i) {                                       // appended, grown with krealloc(km, ...) like src/sketch.c:188
		mm128_t e; e.x = out[ooff + i].x; e.y = out[ooff + i].y | (uint64_t)rid << 32;
		kv_push(mm128_t, km, *p, e);
	}
}

extern "C" mm128_t *mm_chain_dp(int max_dist_x, int min_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc, float gap_scale,
                                int is_cdna, int n_segs, int64_t n, mm128_t *a, int *n_u_, uint64_t **_u, void *km)
{
	if (off() || n_segs > 1 || n >= ((int64_t)1 << 31)) return ref_mm_chain_dp(max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, gap_scale, is_cdna, n_segs, n, a, n_u_, _u, km);
	if (_u) *_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) { kfree(km, a); return 0; }                     // src/chain.c:33-36
	std::vector<wm128_t> aa((size_t)n);
	memcpy(aa.data(), a, (size_t)n * sizeof(mm128_t));
	std::vector<uint64_t> u((size_t)n + 1);
	wm_chain_par_t par = { max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, gap_scale, is_cdna ? 1 : 0 };
	uint64_t a_off = 0, u_off = 0; int32_t na = (int32_t)n, nu = 0, nv = 0;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (wm_chain_batch(ctx(), 1, aa.data(), &a_off, &na, &par, u.data(), &u_off, &nu, &nv)) die("mm_chain_dp");
	}
	kfree(km, a);                                                          // the callee owns `a` (src/chain.c:166)
	if (nu == 0) return 0;
	uint64_t *uo = (uint64_t*)kmalloc(km, (size_t)nu * 8);
	memcpy(uo, u.data() + u_off, (size_t)nu * 8);
	mm128_t *b = (mm128_t*)kmalloc(km, (size_t)nv * sizeof(mm128_t));
	memcpy(b, aa.data(), (size_t)nv * sizeof(mm128_t));
	*n_u_ = nu; *_u = uo;
	return b;
}

extern "C" void ksw_extd2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                              int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez)
{
	if (off() || m != 5 || (flag & (KSW_EZ_SCORE_ONLY | KSW_EZ_GENERIC_SC | KSW_EZ_APPROX_DROP | KSW_EZ_SPLICE_FOR | KSW_EZ_SPLICE_REV | KSW_EZ_SPLICE_FLANK))) {
		ref_ksw_extd2_sse(km, qlen, query, tlen, target, m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
		return;
	}
	wm_ksw_result_t r;
	uint32_t *cig = 0;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (wm_ksw_extd2(ctx(), qlen, query, tlen, target, m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, &r, &cig)) die("ksw_extd2_sse");
	}
	ksw_reset_extz(ez);                                                    // src/ksw2.h:153 — keeps ez->cigar / m_cigar (reused across calls, freed by the caller)
	ez->max = r.max; ez->zdropped = r.zdropped; ez->max_q = r.max_q; ez->max_t = r.max_t; ez->mqe = r.mqe; ez->mqe_t = r.mqe_t;
	ez->mte = r.mte; ez->mte_q = r.mte_q; ez->score = r.score; ez->reach_end = r.reach_end;
	if (r.n_cigar > ez->m_cigar) {                                         // grow like ksw_push_cigar (src/ksw2.h:103-113)
		ez->m_cigar = r.n_cigar + (r.n_cigar >> 1) + 4;
		ez->cigar = (uint32_t*)krealloc(km, ez->cigar, (size_t)ez->m_cigar << 2);
	}
	if (r.n_cigar) memcpy(ez->cigar, cig, (size_t)r.n_cigar * 4);
	ez->n_cigar = r.n_cigar;
	free(cig);
}

extern "C" void ksw_extz2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                              int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez)
{   // single-affine = the dual-affine recursion with equal pieces (pinned against ksw_extz2_sse in tests/test_oracle_vs_ref.py)
	if (off()) { ref_ksw_extz2_sse(km, qlen, query, tlen, target, m, mat, q, e, w, zdrop, end_bonus, flag, ez); return; }
	ksw_extd2_sse(km, qlen, query, tlen, target, m, mat, q, e, q, e, w, zdrop, end_bonus, flag, ez);
}

extern "C" void ksw_exts2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                              int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, ksw_extz_t *ez)
{   // the splice-aware extension (selected at src/align.c:326-327): one job of wm_ksw_exts2_batch
	if (off() || m != 5 || qlen <= 0 || tlen <= 0 || q2 <= q + e || -(int)mat[1] > 2 * (q + e) || (flag & (KSW_EZ_SCORE_ONLY | KSW_EZ_GENERIC_SC | KSW_EZ_APPROX_DROP))) {
		ref_ksw_exts2_sse(km, qlen, query, tlen, target, m, mat, q, e, q2, noncan, zdrop, junc_bonus, flag, junc, ez);
		if (const char *dp = getenv("WM_DUMP_EXTS2")) {     // tests: the reference's own splice-mode jobs and results, for replays through the oracle / the kernel emulator
			std::lock_guard<std::mutex> lk(g_mu);
			if (FILE *f = fopen(dp, "ab")) {
				const int32_t hdr[24] = { qlen, tlen, m, mat[0], mat[1], mat[24], q, e, q2, noncan, zdrop, junc_bonus, flag, junc ? 1 : 0,
				                          ez->max, ez->zdropped, ez->max_q, ez->max_t, ez->mqe, ez->mqe_t, ez->mte, ez->mte_q, ez->score, ez->n_cigar };
				fwrite(hdr, 4, 24, f);
				if (qlen > 0) fwrite(query, 1, qlen, f);
				if (tlen > 0) fwrite(target, 1, tlen, f);
				if (junc && tlen > 0) fwrite(junc, 1, tlen, f);
				if (ez->n_cigar > 0) fwrite(ez->cigar, 4, ez->n_cigar, f);
				fclose(f);
			}
		}
		return;
	}
	const wm_ksw_score_t sc = { mat[0], mat[1], mat[24], q, e, q2, 0 };
	const wm_ksw_job_t jb = { 0, (uint32_t)qlen, qlen, tlen, -1, zdrop, 0, flag };
	std::vector<uint8_t> seqs((size_t)qlen + tlen), jn;
	memcpy(seqs.data(), query, qlen); memcpy(seqs.data() + qlen, target, tlen);
	if (junc) { jn.assign((size_t)qlen + tlen, 0); memcpy(jn.data() + qlen, junc, tlen); }
	std::vector<uint32_t> cig((size_t)qlen + tlen + 4);
	wm_ksw_result_t r;
	size_t used = 0;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (wm_ksw_exts2_batch(ctx(), &sc, noncan, junc_bonus, 1, &jb, seqs.data(), seqs.size(), junc ? jn.data() : 0, &r, cig.data(), cig.size(), &used)) die("ksw_exts2_sse");
	}
	ksw_reset_extz(ez);
	ez->max = r.max; ez->zdropped = r.zdropped; ez->max_q = r.max_q; ez->max_t = r.max_t; ez->mqe = r.mqe; ez->mqe_t = r.mqe_t;
	ez->mte = r.mte; ez->mte_q = r.mte_q; ez->score = r.score; ez->reach_end = r.reach_end;
	if (r.n_cigar > ez->m_cigar) {
		ez->m_cigar = r.n_cigar + (r.n_cigar >> 1) + 4;
		ez->cigar = (uint32_t*)krealloc(km, ez->cigar, (size_t)ez->m_cigar << 2);
	}
	if (r.n_cigar) memcpy(ez->cigar, cig.data() + r.cig_off, (size_t)r.n_cigar * 4);
	ez->n_cigar = r.n_cigar;
}

__attribute__((destructor)) static void wm_subst_fini() { if (g_ctx) wm_ctx_destroy(g_ctx); }


// ---- mm_idx_get (src/mmpriv.h:71, src/index.c:88-105): the library's flat table, built ONCE from the reference's finished index through the
// reference's own file format (mm_idx_dump, src/index.c:515 → wm_index_load). The returned pointer aims into the library's position array
// (same words as the reference's: rid<<32 | last position<<1 | strand), valid for the life of the process like the reference's.
namespace {
std::mutex g_idx_mu;
const mm_idx_t *g_idx_of = 0;
wm_index_t *g_idx = 0;
const wm_index_t *flat_index(const mm_idx_t *mi)
{
	std::lock_guard<std::mutex> lk(g_idx_mu);
	if (g_idx_of == mi) return g_idx;
	if (g_idx) { wm_index_destroy(g_idx); g_idx = 0; }                 // (a multi-part index: the previous part is gone by now, src/main.c:398-425)
	char tmpl[] = "/tmp/wm_subst_idx_XXXXXX";
	const int fd = mkstemp(tmpl);
	if (fd < 0) { fprintf(stderr, "[wm_subst] cannot create a temporary index file\n"); exit(1); }
	FILE *fp = fdopen(fd, "wb");
	mm_idx_dump(fp, mi);
	fclose(fp);
	const int rc = wm_index_load(tmpl, 0, &g_idx);
	unlink(tmpl);
	if (rc) die("mm_idx_get (wm_index_load)");
	g_idx_of = mi;
	return g_idx;
}
}

extern "C" const uint64_t *mm_idx_get(const mm_idx_t *mi, uint64_t minier, int *n)
{
	if (off() || !mi->B) return ref_mm_idx_get(mi, minier, n);
	return wm_index_get(flat_index(mi), minier, n);
}

// ---- ksw_ll_qinit / ksw_ll_i16 (src/ksw2.h:82-83): the query profile is opaque to its callers (src/align.c:78-79, 539-540, 824-825: create, one
// ksw_ll_i16, kfree) — ours is ONE kalloc block holding the operands, and the score comes from the library's wm_ksw_ll_i16.
namespace { struct LlProfile { int32_t magic, qlen, m, size; int8_t mat[25]; uint8_t query[1]; }; }

extern "C" void *ksw_ll_qinit(void *km, int size, int qlen, const uint8_t *query, int m, const int8_t *mat)
{
	if (off() || m != 5 || size != 2) return ref_ksw_ll_qinit(km, size, qlen, query, m, mat);
	LlProfile *p = (LlProfile*)kmalloc(km, sizeof(LlProfile) + (size_t)(qlen > 0 ? qlen : 0));
	p->magic = 0x4c4c5157; p->qlen = qlen; p->m = m; p->size = size;
	memcpy(p->mat, mat, 25);
	if (qlen > 0) memcpy(p->query, query, (size_t)qlen);
	return p;
}

extern "C" int ksw_ll_i16(void *q, int tlen, const uint8_t *target, int gapo, int gape, int *qe, int *te)
{
	const LlProfile *p = (const LlProfile*)q;
	if (off() || p->magic != 0x4c4c5157) return ref_ksw_ll_i16(q, tlen, target, gapo, gape, qe, te);   // (a profile made by the original)
	return wm_ksw_ll_i16(p->qlen, p->query, tlen, target, p->mat, gapo, gape, qe, te);
}
