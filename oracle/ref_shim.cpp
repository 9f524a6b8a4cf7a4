// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// Thin C-ABI wrappers (flat arrays, no reference structs in the signatures) around the REAL reference
// library built by oracle/Makefile from /root/reference/src. Linked into oracle/_ref/libwinnowmap_ref.so
// and driven from tests / the golden-fixture generator through ctypes. Nothing in the product path
// (winnowmap_amd/, include/) may include, link or call this file.
//
// Every wrapper names the reference entry point it forwards to.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "minimap.h"   // /root/reference/src/minimap.h
#include "mmpriv.h"    // /root/reference/src/mmpriv.h
#include "ksw2.h"      // /root/reference/src/ksw2.h

extern "C" {

// ---- index (src/index.c:634,660 mm_idx_reader_open/read → mm_idx_gen :378) ----
void *refshim_idx_build_flag(const char *fasta, const char *kmer_file, int k, int w, int idx_flag, int n_threads);
void *refshim_idx_build(const char *fasta, const char *kmer_file, int k, int w, int n_threads) { return refshim_idx_build_flag(fasta, kmer_file, k, w, 0, n_threads); }
void *refshim_idx_build_flag(const char *fasta, const char *kmer_file, int k, int w, int idx_flag, int n_threads)      // idx_flag: MM_I_HPC = 1 (the CLI's -H)
{
	mm_idxopt_t io;
	mm_idxopt_init(&io);
	io.k = k, io.w = w; io.flag |= idx_flag;
	mm_verbose = 1;
	mm_idx_reader_t *r = mm_idx_reader_open(fasta, &io, 0);
	if (!r) return 0;
	mm_idx_t *mi = mm_idx_reader_read(r, n_threads, kmer_file && kmer_file[0] ? kmer_file : NULL);
	mm_idx_reader_close(r);
	return mi;
}
void refshim_idx_destroy(void *mi) { mm_idx_destroy((mm_idx_t*)mi); }
// --junc-bed (src/main.c:416). NB: in this reference mm_idx_read_bed CRASHES on any input ("realloc(): invalid pointer" / SIGSEGV in the CLI):
// src/index.c sees kstring_t = { unsigned l, m; char *s } (src/mmpriv.h:41-44) while ks_getuntil2, compiled in src/bseq.c, uses
// { size_t l, m; char *s } (src/kseq.h:91-94) — 16 vs 24 bytes on the caller's stack. So the annotation is injected below instead of parsed.
int refshim_idx_bed_read(void *mi, const char *fn) { return mm_idx_bed_read((mm_idx_t*)mi, fn, 1); }
// the intervals of one contig straight into mi->I (layout of the private mm_idx_intv_s, src/index.c:40-48), sorted by start as mm_idx_bed_read leaves them
int refshim_idx_set_junc(void *mi_, int ctg, int n, const int32_t *st, const int32_t *en, const int32_t *strand)
{
	struct Intv1 { int32_t st, en, max; int32_t score:30, strand:2; };
	struct Intv { int32_t n, m; Intv1 *a; };
	mm_idx_t *mi = (mm_idx_t*)mi_;
	if (ctg < 0 || ctg >= (int)mi->n_seq) return -1;
	if (!mi->I) mi->I = (struct mm_idx_intv_s*)calloc(mi->n_seq, sizeof(Intv));
	Intv *r = (Intv*)mi->I + ctg;
	free(r->a);
	r->a = (Intv1*)calloc(n > 0 ? n : 1, sizeof(Intv1)); r->n = r->m = n;
	for (int i = 0; i < n; ++i) { r->a[i].st = st[i]; r->a[i].en = en[i]; r->a[i].max = -1; r->a[i].score = 0; r->a[i].strand = strand[i]; }
	for (int i = 1; i < n; ++i) for (int j = i; j > 0 && r->a[j].st < r->a[j - 1].st; --j) { Intv1 t = r->a[j]; r->a[j] = r->a[j - 1]; r->a[j - 1] = t; }
	return 0;
}
int refshim_idx_bed_junc(void *mi, int ctg, int st, int en, uint8_t *s) { return mm_idx_bed_junc((const mm_idx_t*)mi, ctg, st, en, s); }
// src/index.c:515-608 mm_idx_dump / mm_idx_load (the CLI's -d is disabled in this fork, the library functions are intact)
int refshim_idx_dump(void *mi, const char *path) { FILE *fp = fopen(path, "wb"); if (!fp) return -1; mm_idx_dump(fp, (mm_idx_t*)mi); fclose(fp); return 0; }
void *refshim_idx_load(const char *path) { FILE *fp = fopen(path, "rb"); if (!fp) return 0; mm_idx_t *mi = mm_idx_load(fp); fclose(fp); return mi; }
int refshim_idx_nseq(void *mi) { return ((mm_idx_t*)mi)->n_seq; }
int refshim_idx_seqlen(void *mi, int rid) { return ((mm_idx_t*)mi)->seq[rid].len; }
int refshim_idx_getseq(void *mi, uint32_t rid, uint32_t st, uint32_t en, uint8_t *out) { return mm_idx_getseq((mm_idx_t*)mi, rid, st, en, out); }

// src/index.c:88 mm_idx_get. Copies up to cap positions; returns the occurrence count.
int refshim_idx_get(void *mi, uint64_t minier, uint64_t *out, int cap)
{
	int n, i;
	const uint64_t *p = mm_idx_get((mm_idx_t*)mi, minier, &n);
	for (i = 0; i < n && i < cap; ++i) out[i] = p[i];
	return n;
}

// ext/bloom/bloom_filter.hpp:303 contains(); table geometry for pinning the restated filter.
int refshim_bloom_contains(void *mi, uint64_t kmer) { return ((mm_idx_t*)mi)->downFilter->contains(kmer) ? 1 : 0; }
uint64_t refshim_bloom_table_bits(void *mi) { return ((mm_idx_t*)mi)->downFilter->size(); }
uint64_t refshim_bloom_hash_count(void *mi) { return ((mm_idx_t*)mi)->downFilter->hash_count(); }
uint64_t refshim_bloom_table_bytes(void *mi, uint8_t *out, uint64_t cap)
{
	bloom_filter *f = ((mm_idx_t*)mi)->downFilter;
	uint64_t nb = f->size() / 8;
	if (out) memcpy(out, f->table(), nb < cap ? nb : cap);
	return nb;
}

// ---- src/sketch.c:128 mm_sketch ----
int64_t refshim_sketch(void *mi, const char *seq, int len, int w, int k, uint32_t rid, int is_hpc, uint64_t *ox, uint64_t *oy, int64_t cap)
{
	mm128_v v = {0, 0, 0};
	mm_sketch(0, seq, len, w, k, rid, is_hpc, &v, (mm_idx_t*)mi);
	int64_t n = v.n;
	for (int64_t i = 0; i < n && i < cap; ++i) ox[i] = v.a[i].x, oy[i] = v.a[i].y;
	free(v.a);
	return n;
}

// ---- src/misc.c:155 radix_sort_128x / radix_sort_64 ----
void refshim_radix_sort_128x(uint64_t *x, uint64_t *y, int64_t n)
{
	mm128_t *a = (mm128_t*)malloc((n ? n : 1) * sizeof(mm128_t));
	for (int64_t i = 0; i < n; ++i) a[i].x = x[i], a[i].y = y[i];
	radix_sort_128x(a, a + n);
	for (int64_t i = 0; i < n; ++i) x[i] = a[i].x, y[i] = a[i].y;
	free(a);
}
void refshim_radix_sort_64(uint64_t *x, int64_t n) { radix_sort_64(x, x + n); }

// ---- src/chain.c:22 mm_chain_dp ----
// in: anchors (ax, ay)[n]; out: u[n_u] (score<<32|cnt), chained anchors (bx,by)[n_v]. Returns n_v, *n_u_out set.
int64_t refshim_chain_dp(int max_dist_x, int min_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
						 float gap_scale, int is_cdna, int n_segs, int64_t n, const uint64_t *ax, const uint64_t *ay,
						 int *n_u_out, uint64_t *u_out, uint64_t *bx, uint64_t *by)
{
	mm128_t *a = (mm128_t*)malloc((n ? n : 1) * sizeof(mm128_t));
	for (int64_t i = 0; i < n; ++i) a[i].x = ax[i], a[i].y = ay[i];
	int n_u = 0;
	uint64_t *u = 0;
	mm128_t *b = mm_chain_dp(max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, gap_scale, is_cdna, n_segs, n, a, &n_u, &u, 0);
	int64_t n_v = 0;
	for (int i = 0; i < n_u; ++i) { u_out[i] = u[i]; n_v += (int32_t)u[i]; }
	for (int64_t i = 0; i < n_v; ++i) bx[i] = b[i].x, by[i] = b[i].y;
	*n_u_out = n_u;
	free(u); free(b);
	return n_v;
}

// ---- src/ksw2.h:60 ksw_extd2_sse / :54 ksw_extz2_sse ----
// ez_out[0..9] = max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end ; returns n_cigar.
static int pack_ez(ksw_extz_t *ez, int32_t *ez_out, uint32_t *cigar_out, int cigar_cap)
{
	ez_out[0] = ez->max; ez_out[1] = ez->zdropped; ez_out[2] = ez->max_q; ez_out[3] = ez->max_t;
	ez_out[4] = ez->mqe; ez_out[5] = ez->mqe_t; ez_out[6] = ez->mte; ez_out[7] = ez->mte_q;
	ez_out[8] = ez->score; ez_out[9] = ez->reach_end;
	int n = ez->n_cigar;
	for (int i = 0; i < n && i < cigar_cap; ++i) cigar_out[i] = ez->cigar[i];
	free(ez->cigar);
	return n;
}
int refshim_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
					  int q, int e, int q2, int e2, int w, int zdrop, int end_bonus, int flag,
					  int32_t *ez_out, uint32_t *cigar_out, int cigar_cap)
{
	ksw_extz_t ez;
	memset(&ez, 0, sizeof(ez));
	ksw_extd2_sse(0, qlen, query, tlen, target, 5, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, &ez);
	return pack_ez(&ez, ez_out, cigar_out, cigar_cap);
}
int refshim_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
					  int q, int e, int w, int zdrop, int end_bonus, int flag,
					  int32_t *ez_out, uint32_t *cigar_out, int cigar_cap)
{
	ksw_extz_t ez;
	memset(&ez, 0, sizeof(ez));
	ksw_extz2_sse(0, qlen, query, tlen, target, 5, mat, q, e, w, zdrop, end_bonus, flag, &ez);
	return pack_ez(&ez, ez_out, cigar_out, cigar_cap);
}
// ---- src/ksw2.h:63-64 ksw_exts2_sse (junc may be NULL) ----
int refshim_ksw_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
					  int q, int e, int q2, int noncan, int zdrop, int junc_bonus, int flag, const uint8_t *junc,
					  int32_t *ez_out, uint32_t *cigar_out, int cigar_cap)
{
	ksw_extz_t ez;
	memset(&ez, 0, sizeof(ez));
	ksw_exts2_sse(0, qlen, query, tlen, target, 5, mat, q, e, q2, noncan, zdrop, junc_bonus, flag, junc, &ez);
	return pack_ez(&ez, ez_out, cigar_out, cigar_cap);
}
// ---- src/ksw2.h:82-83 ksw_ll_qinit + ksw_ll_i16 ----
int refshim_ksw_ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int gapo, int gape, int *qe, int *te)
{
	void *qp = ksw_ll_qinit(0, 2, qlen, query, 5, mat);
	int s = ksw_ll_i16(qp, tlen, target, gapo, gape, qe, te);
	free(qp);
	return s;
}

// ---- end-to-end: src/minimap.h:358 mm_map with a preset (src/options.c:89) ----
// flag_extra is OR-ed into mm_mapopt_t::flag (MM_F_CIGAR=0x4 | MM_F_OUT_SAM=0x8 ...).
// Output per hit, 16 int32: rid rs re qs qe rev mapq n_cigar | score cnt mlen blen dp_score dp_max dp_max2 flags(parent==id | inv<<1 | sam_pri<<2 | split<<3 | trans_strand<<5)
// cigars are appended to cig_out. Returns n_regs.
static mm_mapopt_t g_mo; static mm_idxopt_t g_io;
void *refshim_mapopt(const char *preset, int64_t flag_extra, void *mi)
{
	mm_set_opt(0, &g_io, &g_mo);
	if (preset && preset[0] && mm_set_opt(preset, &g_io, &g_mo) < 0) return 0;
	g_mo.flag |= flag_extra;
	mm_mapopt_update(&g_mo, (mm_idx_t*)mi);
	return &g_mo;
}
void refshim_mapopt_clear_flag(void *opt, int64_t bits) { ((mm_mapopt_t*)opt)->flag &= ~bits; }     // e.g. -uf = MM_F_SPLICE_REV cleared (src/main.c)
void refshim_mapopt_set_max_sw_mat(void *opt, int64_t v) { ((mm_mapopt_t*)opt)->max_sw_mat = v; }     // (no command-line switch reaches it: the long-option table lacks --cap-sw-mat)
// every mm_mapopt_t field that wm_mapopt_t mirrors (include/wm_gpu.h), in that struct's order, after mm_set_opt(0) + mm_set_opt(preset)
int refshim_preset_fields(const char *preset, double *o, int cap)
{
	mm_idxopt_t io; mm_mapopt_t m;
	mm_set_opt(0, &io, &m);
	if (preset && preset[0] && mm_set_opt(preset, &io, &m) < 0) return -1;
	const double v[] = { (double)m.flag, (double)m.seed, (double)m.sdust_thres, (double)m.max_qlen, (double)m.bw, (double)m.max_gap, (double)m.max_gap_ref, (double)m.min_gap_ref, (double)m.max_frag_len,
		(double)m.max_chain_skip, (double)m.max_chain_iter, (double)m.min_cnt, (double)m.min_chain_score, (double)m.chain_gap_scale, (double)m.SVaware, (double)m.SVawareMinReadLength,
		(double)m.suffixSampleOffset, (double)m.min_mapq, (double)m.min_qcov, (double)m.minPrefixLength, (double)m.maxPrefixLength, (double)m.prefixIncrementFactor, (double)m.stage2_bw,
		(double)m.stage2_zdrop_inv, (double)m.stage2_max_gap, (double)m.mask_level, (double)m.mask_len, (double)m.pri_ratio, (double)m.best_n, (double)m.max_join_long, (double)m.max_join_short,
		(double)m.min_join_flank_sc, (double)m.min_join_flank_ratio, (double)m.alt_drop, (double)m.a, (double)m.b, (double)m.q, (double)m.e, (double)m.q2, (double)m.e2, (double)m.sc_ambi,
		(double)m.zdrop, (double)m.zdrop_inv, (double)m.end_bonus, (double)m.min_dp_max, (double)m.min_ksw_len, (double)m.max_clip_ratio, (double)m.mid_occ_frac, (double)m.min_mid_occ,
		(double)m.mid_occ, (double)m.max_occ, (double)m.mini_batch_size, (double)m.max_sw_mat,
		(double)m.noncan, (double)m.junc_bonus, (double)m.anchor_ext_len, (double)m.anchor_ext_shift, (double)io.k, (double)io.w };
	const int n = (int)(sizeof(v) / sizeof(v[0]));
	for (int i = 0; i < n && i < cap; ++i) o[i] = v[i];
	return n;
}
int refshim_preset_k(const char *preset) { mm_idxopt_t io; mm_mapopt_t mo; mm_set_opt(0, &io, &mo); mm_set_opt(preset, &io, &mo); return io.k; }
int refshim_preset_w(const char *preset) { mm_idxopt_t io; mm_mapopt_t mo; mm_set_opt(0, &io, &mo); mm_set_opt(preset, &io, &mo); return io.w; }

int refshim_map(void *mi, void *opt, const char *seq, int len, const char *name, int32_t *hit_out, int hit_cap, uint32_t *cig_out, int64_t cig_cap, int64_t *n_cig_total)
{
	mm_tbuf_t *b = mm_tbuf_init();
	int n_regs = 0;
	mm_reg1_t *regs = mm_map((mm_idx_t*)mi, len, seq, &n_regs, b, (mm_mapopt_t*)opt, name);
	int64_t nc = 0;
	for (int i = 0; i < n_regs; ++i) {
		mm_reg1_t *r = &regs[i];
		if (i < hit_cap) {
			int32_t *h = hit_out + 16 * i;
			h[0] = r->rid; h[1] = r->rs; h[2] = r->re; h[3] = r->qs; h[4] = r->qe; h[5] = r->rev; h[6] = r->mapq;
			h[7] = r->p ? r->p->n_cigar : 0;
			h[8] = r->score; h[9] = r->cnt; h[10] = r->mlen; h[11] = r->blen;
			h[12] = r->p ? r->p->dp_score : 0; h[13] = r->p ? r->p->dp_max : 0; h[14] = r->p ? r->p->dp_max2 : 0;
			h[15] = (r->parent == r->id) | r->inv << 1 | r->sam_pri << 2 | r->split << 3 | (r->p ? r->p->trans_strand << 5 : 0);
			if (r->p) for (uint32_t j = 0; j < r->p->n_cigar; ++j) { if (nc < cig_cap) cig_out[nc] = r->p->cigar[j]; ++nc; }
		}
		free(r->p);
	}
	free(regs);
	mm_tbuf_destroy(b);
	*n_cig_total = nc;
	return n_regs;
}

} // extern "C"
