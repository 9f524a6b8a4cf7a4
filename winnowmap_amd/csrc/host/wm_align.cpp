// wm_align.cpp — see wm_align.h.
#include "wm_align.h"
#include "../cigar_walk.h"
#include <math.h>
#include <algorithm>
#include <list>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <memory>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace wm {

enum { WM_WALK_PAD = 16 };           // readable bytes behind every sequence extra_walk_fast walks
enum { EZ_RIGHT = 0x02, EZ_APPROX_MAX = 0x08, EZ_EXTZ_ONLY = 0x40, EZ_REV_CIGAR = 0x80, EZ_SPLICE_FOR = 0x100, EZ_SPLICE_REV = 0x200, EZ_SPLICE_FLANK = 0x400 };   // src/ksw2.h:8-20

static void gen_simple_mat(int8_t *mat, int a, int b, int sc_ambi)
{   // ksw_gen_simple_mat, src/align.c:9-22 (m = 5)
	a = a < 0 ? -a : a; b = b > 0 ? -b : b; sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
	for (int i = 0; i < 5; ++i)
		for (int j = 0; j < 5; ++j)
			mat[i * 5 + j] = (i == 4 || j == 4) ? (int8_t)sc_ambi : i == j ? (int8_t)a : (int8_t)b;
}

// ------------------------------------------------------------------------------------------------------------
// ksw_ll_i16: striped local alignment score, emulated lane-exactly (8 x int16 per vector)
// ------------------------------------------------------------------------------------------------------------
namespace {
inline int16_t sat_add(int a, int b) { const int s = a + b; return (int16_t)(s > 32767 ? 32767 : s < -32768 ? -32768 : s); }
inline int16_t usub(int16_t a, int16_t b) { const uint16_t x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(x > y ? x - y : 0); }
}
#if defined(__SSE2__)
// the same machine on real 8 x int16 vectors (the host is x86-64 wherever the library runs today); the portable lane-by-lane form below is its
// specification and what other hosts compile
int ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int gapo, int gape, int *qe, int *te)
{
	const int V = 8, slen = (qlen + V - 1) / V, n = slen * V;
	if (slen <= 0) { *qe = *te = -1; return 0; }
	std::vector<__m128i> buf((size_t)9 * slen + 1);
	__m128i *profile = buf.data(), *Hprev = profile + (size_t)5 * slen, *Hcur = Hprev + slen, *E = Hcur + slen, *Hbest = E + slen;
	{
		int16_t *pr = (int16_t*)profile;
		for (int a = 0; a < 5; ++a)
			for (int j = 0; j < slen; ++j)
				for (int l = 0; l < V; ++l) {
					const int pos = j + l * slen;
					pr[((size_t)a * slen + j) * V + l] = pos < qlen ? mat[a * 5 + query[pos]] : 0;
				}
	}
	const __m128i zero = _mm_setzero_si128(), open_ext = _mm_set1_epi16((int16_t)(gapo + gape)), ext = _mm_set1_epi16((int16_t)gape);
	for (int j = 0; j < slen; ++j) Hprev[j] = Hcur[j] = E[j] = Hbest[j] = zero;
	int gmax = 0;
	*qe = *te = -1;
	for (int i = 0; i < tlen; ++i) {
		const __m128i *S = profile + (size_t)target[i] * slen;
		__m128i h = _mm_slli_si128(Hprev[slen - 1], 2), f = zero, best = zero;
		for (int j = 0; j < slen; ++j) {
			const __m128i e = E[j];
			__m128i v = _mm_adds_epi16(h, S[j]);
			v = _mm_max_epi16(v, e); v = _mm_max_epi16(v, f);
			best = _mm_max_epi16(best, v);
			Hcur[j] = v;
			v = _mm_subs_epu16(v, open_ext);
			E[j] = _mm_max_epi16(_mm_subs_epu16(e, ext), v);
			f = _mm_max_epi16(_mm_subs_epu16(f, ext), v);
			h = Hprev[j];
		}
		bool settled = false;
		for (int k = 0; k < V && !settled; ++k) {              // lazy-F correction rounds
			f = _mm_slli_si128(f, 2);
			for (int j = 0; j < slen && !settled; ++j) {
				__m128i v = _mm_max_epi16(Hcur[j], f);
				Hcur[j] = v;
				v = _mm_subs_epu16(v, open_ext);
				f = _mm_subs_epu16(f, ext);
				if (!_mm_movemask_epi8(_mm_cmpgt_epi16(f, v))) settled = true;
			}
		}
		int16_t b8[8];
		_mm_storeu_si128((__m128i*)b8, best);
		int imax = b8[0];
		for (int l = 1; l < V; ++l) imax = imax > b8[l] ? imax : b8[l];
		if (imax >= gmax) { gmax = imax; *te = i; memcpy(Hbest, Hcur, (size_t)slen * sizeof(__m128i)); }
		std::swap(Hprev, Hcur);
	}
	const int16_t *hb = (const int16_t*)Hbest;
	for (int i = 0; i < n; ++i)
		if ((int)(uint16_t)hb[i] == gmax) *qe = i / V + i % V * slen;
	return gmax;
}
int ll_i16_portable(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int gapo, int gape, int *qe, int *te)
#else
int ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int gapo, int gape, int *qe, int *te)
#endif
{
	const int V = 8, slen = (qlen + V - 1) / V, n = slen * V;
	std::vector<int16_t> profile((size_t)5 * n), Hprev(n, 0), Hcur(n, 0), E(n, 0), Hbest(n, 0);
	for (int a = 0; a < 5; ++a)
		for (int j = 0; j < slen; ++j)
			for (int l = 0; l < V; ++l) {
				const int pos = j + l * slen;
				profile[((size_t)a * slen + j) * V + l] = pos < qlen ? mat[a * 5 + query[pos]] : 0;
			}
	const int16_t open_ext = (int16_t)(gapo + gape), ext = (int16_t)gape;
	int gmax = 0;
	*qe = *te = -1;
	for (int i = 0; i < tlen; ++i) {
		const int16_t *S = &profile[(size_t)target[i] * n];
		int16_t h[V], f[V], best[V];
		h[0] = 0;
		for (int l = 1; l < V; ++l) h[l] = Hprev[(size_t)(slen - 1) * V + l - 1];
		for (int l = 0; l < V; ++l) f[l] = 0, best[l] = 0;
		for (int j = 0; j < slen; ++j)
			for (int l = 0; l < V; ++l) {
				const size_t o = (size_t)j * V + l;
				int16_t e = E[o], v = sat_add(h[l], S[o]);
				v = std::max(v, e); v = std::max(v, f[l]);
				best[l] = std::max(best[l], v);
				Hcur[o] = v;
				v = usub(v, open_ext);
				E[o] = std::max(usub(e, ext), v);
				f[l] = std::max(usub(f[l], ext), v);
				h[l] = Hprev[o];
			}
		bool settled = false;
		for (int k = 0; k < V && !settled; ++k) {              // lazy-F correction rounds
			for (int l = V - 1; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (int j = 0; j < slen && !settled; ++j) {
				bool any = false;
				for (int l = 0; l < V; ++l) {
					const size_t o = (size_t)j * V + l;
					int16_t v = std::max(Hcur[o], f[l]);
					Hcur[o] = v;
					v = usub(v, open_ext);
					f[l] = usub(f[l], ext);
					any |= f[l] > v;
				}
				if (!any) settled = true;
			}
		}
		int imax = best[0];
		for (int l = 1; l < V; ++l) imax = std::max<int>(imax, best[l]);
		if (imax >= gmax) { gmax = imax; *te = i; Hbest = Hcur; }
		Hprev.swap(Hcur);
	}
	for (int i = 0; i < n; ++i)
		if ((int)(uint16_t)Hbest[i] == gmax) *qe = i / V + i % V * slen;
	return gmax;
}

// ------------------------------------------------------------------------------------------------------------
// helpers of mm_align1
// ------------------------------------------------------------------------------------------------------------
struct AlnEnv {
	const MapOpt *opt; const Index *idx; int qlen; const uint8_t *qseq0[2]; int8_t mat[25];
	int64_t q_dev_off;      // where qcodes[0] lives in the resident read codes (-1: not resident), see KswReq
	bool q_has_n;
};

// one ksw request on query strand `rev`, strand coordinates [qs, qe), and reference [rs, re) of contig rid; `tb` = codes of the
// reference from base tb0 on. reversed: both operands back to front (left extension)
static KswReq make_job(const AlnEnv &E, int rev, int32_t qs, int32_t qe, int32_t rid, int32_t rs, int32_t re, const uint8_t *tb, int32_t tb0, bool t_has_n, bool reversed)
{
	KswReq j;
	const uint8_t *qstr = E.qseq0[rev];
	j.ql = qe - qs; j.tl = re - rs; j.step = reversed ? -1 : 1;
	j.qp = reversed ? qstr + qe - 1 : qstr + qs;
	j.tp = reversed ? tb + (re - 1 - tb0) : tb + (rs - tb0);
	j.qwin_off = E.q_dev_off; j.qwin_len = E.qlen;
	j.q_pos = (rev ? E.qlen : 0) + (reversed ? qe - 1 : qs);
	j.rid = rid; j.t_pos = reversed ? re - 1 : rs;
	j.has_n = E.q_has_n || t_has_n;
	return j;
}

// mm_idx_bed_junc for the target range of a splice-mode job (src/align.c:693, 723, 770)
static void attach_junc(const AlnEnv &E, KswReq &j, int32_t rid, int32_t rs, int32_t re, bool reversed)
{
	if (!(E.opt->flag & F_SPLICE) || !E.idx->has_junc() || re <= rs) return;
	j.junc.resize((size_t)(re - rs));
	E.idx->bed_junc(rid, rs, re, j.junc.data());
	if (reversed) std::reverse(j.junc.begin(), j.junc.end());
}

static inline void adjust_minier(const Index &idx, const uint8_t *const qseq0[2], const m128 &a, int32_t *r, int32_t *q)
{   // mm_adjust_minier (src/align.c:350-365)
	if (idx.flag & 1) {
		// homopolymer-compressed index (MM_I_HPC): an anchor sits on the LAST base of a run on either side — step back to the run's first base
		// (mm_get_hplen_back on the packed reference, src/align.c:340-348; the query strand by hand, :353-358)
		const uint8_t *qseq = qseq0[a.x >> 63];
		int i, c;
		*q = (int32_t)a.y;
		for (i = *q - 1, c = qseq[*q]; i > 0; --i)
			if (qseq[i] != c) break;
		*q = i + 1;
		const uint32_t rid = (uint32_t)(a.x << 1 >> 33), x = (uint32_t)(int32_t)a.x;
		const int64_t off0 = (int64_t)idx.seq[rid].offset, off = off0 + x;
		const int ct = (int)(idx.S[off >> 3] >> ((off & 7) << 2) & 0xf);
		int64_t t;
		for (t = off - 1; t >= off0; --t)
			if ((int)(idx.S[t >> 3] >> ((t & 7) << 2) & 0xf) != ct) break;
		*r = (int32_t)a.x + 1 - (int)(off - t);
	} else {
		*r = (int32_t)a.x - (idx.k >> 1);
		*q = (int32_t)a.y - (idx.k >> 1);
	}
}

static std::vector<int> collect_long_gaps(int as1, int cnt1, const m128 *a, int min_gap)
{   // src/align.c:367-384: positions whose diagonal jump exceeds min_gap; meaningful only if there are >= 2
	std::vector<int> K;
	for (int i = 1; i < cnt1; ++i) {
		const int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
		if (gap < -min_gap || gap > min_gap) K.push_back(i);
	}
	if (K.size() <= 1) K.clear();
	return K;
}

static void filter_bad_seeds(int as1, int cnt1, m128 *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{   // mm_filter_bad_seeds, src/align.c:386-421
	const std::vector<int> K = collect_long_gaps(as1, cnt1, a, min_gap);
	const int n = (int)K.size();
	if (n == 0) return;
	int max = 0, max_st = -1, max_en = -1;
	for (int k = 0;; ++k) {
		if (k == n || k >= max_en) {
			if (max_en > 0)
				for (int i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= SEED_IGNORE;
			max = 0, max_st = max_en = -1;
			if (k == n) break;
		}
		int i = K[k], n_ins = 0, n_del = 0, max_diff = 0, max_diff_l = -1;
		int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
		if (gap > 0) n_ins += gap; else n_del += -gap;
		const int qs = (int32_t)a[as1 + i - 1].y, rs = (int32_t)a[as1 + i - 1].x;
		for (int l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
			const int j = K[l];
			if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
			gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			if (gap > 0) n_ins += gap; else n_del += -gap;
			const int diff = n_ins + n_del - abs(n_ins - n_del);
			if (max_diff < diff) max_diff = diff, max_diff_l = l;
		}
		if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
	}
}

static void filter_bad_seeds_alt(int as1, int cnt1, m128 *a, int min_gap, int max_ext)
{   // mm_filter_bad_seeds_alt, src/align.c:423-457
	const std::vector<int> K = collect_long_gaps(as1, cnt1, a, min_gap);
	const int n = (int)K.size();
	for (int k = 0; k < n;) {
		const int i = K[k];
		int l;
		int gap1 = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
		int re1 = (int32_t)a[as1 + i].x, qe1 = (int32_t)a[as1 + i].y;
		gap1 = gap1 > 0 ? gap1 : -gap1;
		for (l = k + 1; l < n; ++l) {
			const int j = K[l];
			if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
			int gap2 = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			const int span_pre = (int)(a[as1 + j - 1].y >> 32 & 0xff);
			const int rs2 = (int32_t)a[as1 + j - 1].x + span_pre, qs2 = (int32_t)a[as1 + j - 1].y + span_pre;
			const int m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
			gap2 = gap2 > 0 ? gap2 : -gap2;
			if (m > gap1 + gap2) break;
			re1 = (int32_t)a[as1 + j].x; qe1 = (int32_t)a[as1 + j].y;
			gap1 = gap2;
		}
		if (l > k + 1) {
			const int end = K[l - 1];
			for (int j = K[k]; j < end; ++j) a[as1 + j].y |= SEED_IGNORE;
			a[as1 + end].y |= SEED_LONG_JOIN;
		}
		k = l;
	}
}

static void fix_bad_ends(const Reg &r, const m128 *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{   // mm_fix_bad_ends, src/align.c:459-493
	*as = r.as, *cnt = r.cnt;
	if (r.cnt < 3) return;
	int32_t m, l;
	m = l = (int32_t)(a[r.as].y >> 32 & 0xff);
	for (int32_t i = r.as + 1; i < r.as + r.cnt - 1; ++i) {
		const int32_t span = (int32_t)(a[i].y >> 32 & 0xff);
		if (a[i].y & SEED_LONG_JOIN) break;
		const int32_t lr = (int32_t)a[i].x - (int32_t)a[i - 1].x, lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
		const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
		if (mx - mn > l >> 1) *as = i;
		l += mn;
		m += mn < span ? mn : span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
	}
	*cnt = r.as + r.cnt - *as;
	m = l = (int32_t)(a[r.as + r.cnt - 1].y >> 32 & 0xff);
	for (int32_t i = r.as + r.cnt - 2; i > *as; --i) {
		const int32_t span = (int32_t)(a[i + 1].y >> 32 & 0xff);
		if (a[i + 1].y & SEED_LONG_JOIN) break;
		const int32_t lr = (int32_t)a[i + 1].x - (int32_t)a[i].x, lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
		const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
		if (mx - mn > l >> 1) *cnt = i + 1 - *as;
		l += mn;
		m += mn < span ? mn : span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
	}
}

// how many leading positions of t[0..n) and q[0..n) hold the same unambiguous base (codes < 4): eight at a time
// mm_seed_ext_score (src/align.c:523-543): local alignment score (ksw_ll_i16) of one anchor stretched by anchor_ext_len to both sides
static int seed_ext_score(const MapOpt &opt, const Index &mi, const int8_t *mat, int qlen, const uint8_t *const qseq0[2], const m128 &a)
{
	const int q_span = (int)(a.y >> 32 & 0xff), ext_len = opt.anchor_ext_len;
	const int rid = (int)(a.x << 1 >> 33);
	int re = (int32_t)a.x + 1, rs = re - q_span, qe = (int32_t)a.y + 1, qs = qe - q_span;
	rs = rs - ext_len > 0 ? rs - ext_len : 0;
	qs = qs - ext_len > 0 ? qs - ext_len : 0;
	re = re + ext_len < (int32_t)mi.seq[rid].len ? re + ext_len : (int32_t)mi.seq[rid].len;
	qe = qe + ext_len < qlen ? qe + ext_len : qlen;
	std::vector<uint8_t> tseq(re - rs);
	mi.getseq(rid, rs, re, tseq.data());
	int q_off, t_off;
	return ll_i16(qe - qs, qseq0[a.x >> 63] + qs, re - rs, tseq.data(), mat, opt.q, opt.e, &q_off, &t_off);
}

// mm_fix_bad_ends_splice (src/align.c:545-563): a boundary anchor that sits across a long gap must carry its own weight — its span, or
// the score of the sequence around it, has to exceed log(gap) + anchor_ext_shift; otherwise it is a tiny terminal exon placed by chance
static void fix_bad_ends_splice(const MapOpt &opt, const Index &mi, const Reg &r, const int8_t *mat, int qlen, const uint8_t *const qseq0[2], const m128 *a, int32_t *as1, int32_t *cnt1)
{
	*as1 = r.as, *cnt1 = r.cnt;
	if (r.cnt < 3) return;
	double log_gap = std::log((double)((int32_t)a[r.as + 1].x - (int32_t)a[r.as].x));
	if ((double)(int)(a[r.as].y >> 32 & 0xff) < log_gap + opt.anchor_ext_shift) {
		const int score = seed_ext_score(opt, mi, mat, qlen, qseq0, a[r.as]);
		if ((double)score / mat[0] < log_gap + opt.anchor_ext_shift) ++(*as1), --(*cnt1);
	}
	log_gap = std::log((double)((int32_t)a[r.as + r.cnt - 1].x - (int32_t)a[r.as + r.cnt - 2].x));
	if ((double)(int)(a[r.as + r.cnt - 1].y >> 32 & 0xff) < log_gap + opt.anchor_ext_shift) {
		const int score = seed_ext_score(opt, mi, mat, qlen, qseq0, a[r.as + r.cnt - 1]);
		if ((double)score / mat[0] < log_gap + opt.anchor_ext_shift) --(*cnt1);
	}
}

static void append_cigar(Reg &r, const std::vector<uint32_t> &c)
{   // mm_append_cigar, src/align.c:288-311
	if (c.empty()) return;
	if (!r.has_p) { r.has_p = true; r.dp_score = r.dp_max = r.dp_max2 = 0; r.n_ambi = 0; r.cigar.clear(); }
	size_t from = 0;
	if (!r.cigar.empty() && (r.cigar.back() & 0xf) == (c[0] & 0xf)) { r.cigar.back() += c[0] >> 4 << 4; from = 1; }
	r.cigar.insert(r.cigar.end(), c.begin() + from, c.end());
}

static bool zdwalk_on_host() { static const bool h = getenv("WM_ZDWALK_HOST") != 0; return h; }      // A/B switch: the host walks the fills' CIGARs as before

// the verdict of mm_test_zdrop (src/align.c:68-89) given the scan's result (wm_zdrop_walk, cigar_walk.h — run by the device over the CIGAR pool of the
// ksw call, or here): 2 = a candidate inversion (the local alignment of the reverse-complemented query stretch scores high enough), 1 = z-drop
static int zdrop_verdict(const MapOpt &opt, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, const wm_zd_t &z)
{
	const int q_len = z.q1 - z.q0, t_len = z.t1 - z.t0;
	if (!(opt.flag & (F_SPLICE | F_SR | F_FOR_ONLY | F_REV_ONLY)) && z.max_zdrop > opt.zdrop_inv && q_len < opt.max_gap && t_len < opt.max_gap) {
		std::vector<uint8_t> q2(q_len > 0 ? q_len : 0);
		for (int x = 0; x < q_len; ++x) { const int c = qseq[z.q1 - x - 1]; q2[x] = c >= 4 ? 4 : 3 - c; }
		int q_off, t_off;
		const int sc = ll_i16(q_len, q2.data(), t_len, tseq + z.t0, mat, opt.q, opt.e, &q_off, &t_off);
		if (sc >= opt.min_chain_score * opt.a && sc >= opt.min_dp_max) return 2;
	}
	return z.max_zdrop > opt.zdrop ? 1 : 0;
}

static int test_zdrop(const MapOpt &opt, const uint8_t *qseq, const uint8_t *tseq, const std::vector<uint32_t> &cigar, const int8_t *mat)
{   // mm_test_zdrop + update_max_zdrop, src/align.c:32-89
	WM_PROF("align.test_zdrop");
	wm_zd_t z;
	wm_zdrop_walk(qseq, tseq, cigar.data(), (int)cigar.size(), mat[0], mat[1], mat[24], opt.q, opt.e, &z);
	return zdrop_verdict(opt, qseq, tseq, mat, z);
}

static void fix_cigar(Reg &r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{   // mm_fix_cigar, src/align.c:91-167
	std::vector<uint32_t> &cg = r.cigar;
	int32_t toff = 0, qoff = 0, to_shrink = 0;
	*qshift = *tshift = 0;
	if (cg.size() <= 1) return;
	for (uint32_t k = 0; k < cg.size(); ++k) {                         // left-align indels
		const uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
		if (len == 0) to_shrink = 1;
		if (op == 0) toff += len, qoff += len;
		else if (op == 1 || op == 2) {
			if (k > 0 && k < cg.size() - 1 && (cg[k - 1] & 0xf) == 0 && (cg[k + 1] & 0xf) == 0) {
				int l;
				const int prev_len = cg[k - 1] >> 4;
				if (op == 1) { for (l = 0; l < prev_len; ++l) if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break; }
				else { for (l = 0; l < prev_len; ++l) if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break; }
				if (l > 0) cg[k - 1] -= l << 4, cg[k + 1] += l << 4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	WM_INVARIANT(qoff == r.qe - r.qs && toff == r.re - r.rs);
	for (uint32_t k = 0; k + 2 < cg.size(); ++k) {                     // 5I6D7I → one I and one D
		if ((cg[k] & 0xf) > 0 && (cg[k] & 0xf) + (cg[k + 1] & 0xf) == 3) {
			uint32_t l, s[3] = {0, 0, 0};
			for (l = k; l < cg.size(); ++l) {
				const uint32_t op = cg[l] & 0xf;
				if (op == 1 || op == 2 || cg[l] >> 4 == 0) s[op] += cg[l] >> 4;
				else break;
			}
			if (s[1] > 0 && s[2] > 0 && l - k > 2) {
				cg[k] = s[1] << 4 | 1; cg[k + 1] = s[2] << 4 | 2;
				for (k += 2; k < l; ++k) cg[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {
		size_t l = 0;
		for (size_t k = 0; k < cg.size(); ++k) if (cg[k] >> 4 != 0) cg[l++] = cg[k];
		cg.resize(l);
		l = 0;
		for (size_t k = 0; k < cg.size(); ++k)
			if (k == cg.size() - 1 || (cg[k] & 0xf) != (cg[k + 1] & 0xf)) cg[l++] = cg[k];
			else cg[k + 1] += cg[k] >> 4 << 4;
		cg.resize(l);
	}
	if ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2) {                    // no leading I/D
		const int32_t l = cg[0] >> 4;
		if ((cg[0] & 0xf) == 1) { if (r.rev) r.qe -= l; else r.qs += l; *qshift = l; }
		else r.rs += l, *tshift = l;
		cg.erase(cg.begin());
	}
}

// wm_extra_walk (cigar_walk.h) with the match stretches compared 16 bases per step. An ONT alignment alternates M runs of a dozen bases with short
// indels, so the eight-byte run finder of the generic walk spends its time in its byte tail; here one M run is usually one masked 16-byte
// compare. Both sequences must be readable WM_WALK_PAD bytes beyond the aligned stretch (ref_codes and the two-strand buffer pad for it).
// mm_update_extra was 15 % of the host's CPU samples in a bench run (gpurun_out/r05a/sprof_report.txt).
#if defined(__SSE2__)
static void extra_walk_fast(const uint8_t *qseq, const uint8_t *tseq, const uint32_t *cigar, int n_cigar, int match, int mismatch, int ambi, int q, int e, wm_extra_t *out)
{
	int32_t s = 0, max = 0, toff = 0, qoff = 0, blen = 0, mlen = 0, n_ambi_all = 0;
	const __m128i three = _mm_set1_epi8(3);
	for (int k = 0; k < n_cigar; ++k) {
		const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			const uint8_t *tp = tseq + toff, *qp = qseq + qoff;
			for (uint32_t l0 = 0; l0 < len; l0 += 16) {
				const uint32_t m = len - l0 < 16 ? len - l0 : 16;
				const __m128i a = _mm_loadu_si128((const __m128i*)(tp + l0)), b = _mm_loadu_si128((const __m128i*)(qp + l0));
				// a base is "plain" when both codes are equal and below 4 (codes are 0..4: x > 3 <=> x == 4)
				const __m128i plain = _mm_andnot_si128(_mm_cmpgt_epi8(_mm_or_si128(a, b), three), _mm_cmpeq_epi8(a, b));
				uint32_t bad = (uint32_t)(~_mm_movemask_epi8(plain)) & ((1u << m) - 1u);
				uint32_t done = 0;
				while (bad) {
					const uint32_t p = (uint32_t)__builtin_ctz(bad);
					bad &= bad - 1;
					if (p > done) { s += (int32_t)(p - done) * match; max = max > s ? max : s; }
					const int ct = tp[l0 + p], cq = qp[l0 + p];
					if (ct > 3 || cq > 3) { ++n_ambi; s += ambi; }
					else { ++n_diff; s += mismatch; }
					if (s < 0) s = 0; else max = max > s ? max : s;
					done = p + 1;
				}
				if (m > done) { s += (int32_t)(m - done) * match; max = max > s ? max : s; }
			}
			blen += (int32_t)len - n_ambi; mlen += (int32_t)len - (n_ambi + n_diff); n_ambi_all += n_ambi;
			toff += len; qoff += len;
		} else if (op == 1 || op == 2) {
			const uint8_t *p = op == 1 ? qseq + qoff : tseq + toff;
			int n_ambi = 0;
			for (uint32_t l = 0; l < len; ++l) n_ambi += p[l] > 3;
			blen += (int32_t)len - n_ambi; n_ambi_all += n_ambi;
			s -= q + e * (int32_t)len;
			if (s < 0) s = 0;
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	out->dp_max = max; out->mlen = mlen; out->blen = blen; out->n_ambi = n_ambi_all; out->qoff = qoff; out->toff = toff;
}
#endif
// test hook (tests/host_harness): the fast walk against the generic one
void extra_walk_both(const uint8_t *qseq, const uint8_t *tseq, const uint32_t *cigar, int n_cigar, int match, int mismatch, int ambi, int q, int e, int32_t *fast6, int32_t *generic6)
{
	wm_extra_t a, b;
	wm_extra_walk(qseq, tseq, cigar, n_cigar, match, mismatch, ambi, q, e, &b);
#if defined(__SSE2__)
	if (match > 0) extra_walk_fast(qseq, tseq, cigar, n_cigar, match, mismatch, ambi, q, e, &a); else a = b;
#else
	a = b;
#endif
	memcpy(fast6, &a, sizeof(a)); memcpy(generic6, &b, sizeof(b));
}

static void update_extra(Reg &r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int q, int e)
{
	WM_PROF("align.update_extra");   // mm_update_extra, src/align.c:240-286 (no =/X rewriting: MM_F_EQX is applied at output time if requested)
	if (!r.has_p) return;
	int qshift, tshift;
	fix_cigar(r, qseq, tseq, &qshift, &tshift);
	qseq += qshift, tseq += tshift;
	wm_extra_t x;
#if defined(__SSE2__)
	if (mat[0] > 0) extra_walk_fast(qseq, tseq, r.cigar.data(), (int)r.cigar.size(), mat[0], mat[1], mat[24], q, e, &x);
	else
#endif
	wm_extra_walk(qseq, tseq, r.cigar.data(), (int)r.cigar.size(), mat[0], mat[1], mat[24], q, e, &x);
	r.blen = x.blen; r.mlen = x.mlen; r.n_ambi += x.n_ambi; r.dp_max = x.dp_max;
	const int32_t qoff = x.qoff, toff = x.toff; (void)qoff; (void)toff;
	WM_INVARIANT(qoff == r.qe - r.qs && toff == r.re - r.rs);
}

// ------------------------------------------------------------------------------------------------------------
// one region = one mm_align1 call, split into plan → (batched ksw) → judge → (batched re-dos) → finish
// ------------------------------------------------------------------------------------------------------------
struct Fill { int idx; int32_t qs, qe, rs, re; int bw1; int job; int redo_job; };

struct RegAln {
	Reg r, r2;
	int n_a = 0;
	int32_t rid = 0, rev = 0, as1 = 0, cnt1 = 0, bw = 0;
	int32_t rs = 0, qs = 0, re = 0, qe = 0, rs0 = 0, qs0 = 0, re0 = 0, qe0 = 0;
	int left_job = -1, right_job = -1;
	std::vector<Fill> fills;
	std::vector<int> redo_code;
	bool empty = false;
	std::vector<uint8_t> tbuf;       // reference codes of [rs0, re0): every DP of the region and the final statistics read from here
	bool t_has_n = false;
};

static inline std::vector<uint8_t> ref_codes(const Index &idx, int rid, int st, int en)
{
	WM_PROF("align.ref_codes");
	std::vector<uint8_t> t((en > st ? en - st : 0) + WM_WALK_PAD, 4);      // (+ padding: extra_walk_fast loads 16 bases at a time)
	if (en > st) idx.getseq(rid, st, en, t.data());
	return t;
}

// splice_flag: F_SPLICE_FOR / F_SPLICE_REV bits of the transcript strand this pass assumes (src/align.c:565, 602-606)
static void plan_reg(const AlnEnv &E, RegAln &A, m128 *a, std::vector<KswReq> &jobs, int64_t splice_flag)
{
	WM_PROF("align.plan_reg");
	const MapOpt &opt = *E.opt;
	const Index &mi = *E.idx;
	Reg &r = A.r;
	const int qlen = E.qlen;
	A.r2 = Reg(); A.r2.cnt = 0;
	if (r.cnt == 0) { A.empty = true; return; }
	A.rid = (int32_t)(a[r.as].x << 1 >> 33); A.rev = (int32_t)(a[r.as].x >> 63);
	A.bw = (int)(opt.bw * 1.5 + 1.);
	const bool is_splice = (opt.flag & F_SPLICE) != 0;
	if (!(opt.flag & F_NO_END_FLT)) {
		if (is_splice) fix_bad_ends_splice(opt, mi, r, E.mat, qlen, E.qseq0, a, &A.as1, &A.cnt1);
		else fix_bad_ends(r, a, opt.bw, opt.min_chain_score * 2, &A.as1, &A.cnt1);
	} else A.as1 = r.as, A.cnt1 = r.cnt;
	int extra_flag = 0;
	if (is_splice) {                                                   // which strand carries GT..AG: in alignment (= reference) orientation
		if (splice_flag & F_SPLICE_FOR) extra_flag |= A.rev ? EZ_SPLICE_REV : EZ_SPLICE_FOR;
		if (splice_flag & F_SPLICE_REV) extra_flag |= A.rev ? EZ_SPLICE_FOR : EZ_SPLICE_REV;
		if (opt.flag & F_SPLICE_FLANK) extra_flag |= EZ_SPLICE_FLANK;
	}
	filter_bad_seeds(A.as1, A.cnt1, a, 10, 40, opt.max_gap >> 1, 10);
	filter_bad_seeds_alt(A.as1, A.cnt1, a, 30, opt.max_gap >> 1);
	adjust_minier(mi, E.qseq0, a[A.as1], &A.rs, &A.qs);
	adjust_minier(mi, E.qseq0, a[A.as1 + A.cnt1 - 1], &A.re, &A.qe);
	const int32_t as1 = A.as1, cnt1 = A.cnt1, rid = A.rid;
	int32_t rs = A.rs, qs = A.qs, re = A.re, qe = A.qe, rs0, qs0, re0, qe0, rs1, qs1, re1, qe1, l, i;
	const int32_t ref_len = (int32_t)mi.seq[rid].len;
	// region bounds (src/align.c:613-684)
	rs0 = (int32_t)a[r.as].x + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
	qs0 = (int32_t)a[r.as].y + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
	if (rs0 < 0) rs0 = 0;
	rs1 = qs1 = 0;
	for (i = r.as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r.as].x >> 32; --i) {
		const int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff), y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		if (x < rs0 && y < qs0) {
			if (++l > opt.min_cnt) {
				l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
				rs1 = rs0 - l, qs1 = qs0 - l;
				if (rs1 < 0) rs1 = 0;
				break;
			}
		}
	}
	if (qs > 0 && rs > 0) {
		l = qs < opt.max_gap ? qs : opt.max_gap;
		qs1 = qs1 > qs - l ? qs1 : qs - l;
		qs0 = qs0 < qs1 ? qs0 : qs1;
		l += l * opt.a > opt.q ? (l * opt.a - opt.q) / opt.e : 0;
		l = l < opt.max_gap ? l : opt.max_gap;
		l = l < rs ? l : rs;
		rs1 = rs1 > rs - l ? rs1 : rs - l;
		rs0 = rs0 < rs1 ? rs0 : rs1;
		rs0 = rs0 < rs ? rs0 : rs;
	} else rs0 = rs, qs0 = qs;
	re0 = (int32_t)a[r.as + r.cnt - 1].x + 1;
	qe0 = (int32_t)a[r.as + r.cnt - 1].y + 1;
	re1 = ref_len, qe1 = qlen;
	for (i = r.as + r.cnt, l = 0; i < A.n_a && a[i].x >> 32 == a[r.as].x >> 32; ++i) {
		const int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
		if (x > re0 && y > qe0) {
			if (++l > opt.min_cnt) {
				l = x - re0 > y - qe0 ? x - re0 : y - qe0;
				re1 = re0 + l, qe1 = qe0 + l;
				break;
			}
		}
	}
	if (qe < qlen && re < ref_len) {
		l = qlen - qe < opt.max_gap ? qlen - qe : opt.max_gap;
		qe1 = qe1 < qe + l ? qe1 : qe + l;
		qe0 = qe0 > qe1 ? qe0 : qe1;
		l += l * opt.a > opt.q ? (l * opt.a - opt.q) / opt.e : 0;
		l = l < opt.max_gap ? l : opt.max_gap;
		l = l < ref_len - re ? l : ref_len - re;
		re1 = re1 < re + l ? re1 : re + l;
		re0 = re0 > re1 ? re0 : re1;
	} else re0 = re, qe0 = qe;
	if (a[r.as].y & SEED_SELF) {
		int max_ext = r.qs > r.rs ? r.qs - r.rs : r.rs - r.qs;
		if (r.rs - rs0 > max_ext) rs0 = r.rs - max_ext;
		if (r.qs - qs0 > max_ext) qs0 = r.qs - max_ext;
		max_ext = r.qe > r.re ? r.qe - r.re : r.re - r.qe;
		if (re0 - r.re > max_ext) re0 = r.re + max_ext;
		if (qe0 - r.qe > max_ext) qe0 = r.qe + max_ext;
	}
	WM_INVARIANT(re0 > rs0);
	A.rs0 = rs0, A.qs0 = qs0, A.re0 = re0, A.qe0 = qe0;
	A.tbuf = ref_codes(mi, rid, rs0, re0);                             // (mm_idx_getseq per DP in the reference, src/align.c:699,724,773)
	A.t_has_n = mi.has_n(rid, rs0, re0);
	const uint8_t *tb = A.tbuf.data();

	if (qs > 0 && rs > 0) {                                            // left extension on reversed sequences (:690-705)
		KswReq j = make_job(E, A.rev, qs0, qs, rid, rs0, rs, tb, rs0, A.t_has_n, true);
		j.w = A.bw; j.end_bonus = opt.end_bonus; j.zdrop = r.split_inv ? opt.zdrop_inv : opt.zdrop;
		j.flag = extra_flag | EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR;
		attach_junc(E, j, rid, rs0, rs, true);
		A.left_job = (int)jobs.size(); jobs.push_back(std::move(j));
	}
	for (i = 1; i < cnt1; ++i) {                                       // gap filling (:709-765), first pass
		if ((a[as1 + i].y & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
		adjust_minier(mi, E.qseq0, a[as1 + i], &re, &qe);
		if (i == cnt1 - 1 || (a[as1 + i].y & SEED_LONG_JOIN) || (qe - qs >= opt.min_ksw_len && re - rs >= opt.min_ksw_len)) {
			Fill f; f.idx = i; f.qs = qs; f.qe = qe; f.rs = rs; f.re = re; f.bw1 = A.bw; f.redo_job = -1;
			if (a[as1 + i].y & SEED_LONG_JOIN) f.bw1 = qe - qs > re - rs ? qe - qs : re - rs;
			KswReq j = make_job(E, A.rev, qs, qe, rid, rs, re, tb, rs0, A.t_has_n, false);
			j.w = f.bw1; j.end_bonus = -1; j.zdrop = opt.zdrop; j.flag = extra_flag | EZ_APPROX_MAX;
			j.want_zd = !zdwalk_on_host();                                // judge_reg tests this alignment's z-drop (src/align.c:736)
			attach_junc(E, j, rid, rs, re, false);
			f.job = (int)jobs.size(); jobs.push_back(std::move(j));
			A.fills.push_back(f);
			rs = re, qs = qe;
		}
	}
	A.re = re, A.qe = qe;                                              // coordinates of the last anchor
	if (qe < qe0 && re < re0) {                                        // right extension (:767-778), used unless a fill z-drops
		KswReq j = make_job(E, A.rev, qe, qe0, rid, re, re0, tb, rs0, A.t_has_n, false);
		j.w = A.bw; j.end_bonus = opt.end_bonus; j.zdrop = opt.zdrop; j.flag = extra_flag | EZ_EXTZ_ONLY;
		attach_junc(E, j, rid, re, re0, false);
		A.right_job = (int)jobs.size(); jobs.push_back(std::move(j));
	}
}

// after the first pass: which fills need the exact second pass (src/align.c:736-737)
static void judge_reg(const AlnEnv &E, RegAln &A, const std::vector<KswReq> &jobs, std::vector<KswReq> &redo)
{
	WM_PROF("align.judge_reg");
	if (A.empty) return;
	A.redo_code.assign(A.fills.size(), 0);
	for (size_t k = 0; k < A.fills.size(); ++k) {
		Fill &f = A.fills[k];
		const KswReq &j = jobs[f.job];
		// (fills are never reversed) the scan comes back with the alignment when the device ran it (ksw_zdwalk_kernel), else the host walks the CIGAR
		const int code = j.has_zd ? zdrop_verdict(*E.opt, j.qp, j.tp, E.mat, j.zd) : test_zdrop(*E.opt, j.qp, j.tp, j.cigar, E.mat);
		A.redo_code[k] = code;
		if (code != 0) {
			KswReq d = j;                                                      // same operands, exact maximum this time
			d.cigar.clear(); d.want_zd = d.has_zd = false; d.w = f.bw1; d.end_bonus = -1; d.zdrop = code == 2 ? E.opt->zdrop_inv : E.opt->zdrop; d.flag = j.flag & ~EZ_APPROX_MAX;      // (keeps the splice bits)
			f.redo_job = (int)redo.size(); redo.push_back(std::move(d));
		}
	}
}

static void finish_reg(const AlnEnv &E, RegAln &A, m128 *a, const std::vector<KswReq> &jobs, const std::vector<KswReq> &redo)
{
	WM_PROF("align.finish_reg");
	if (A.empty) return;
	const MapOpt &opt = *E.opt;
	Reg &r = A.r;
	const int qlen = E.qlen;
	int32_t rs1, qs1, re1, qe1;
	bool dropped = false;
	{   // one allocation for the stitched CIGAR
		size_t tot = A.left_job >= 0 ? jobs[A.left_job].cigar.size() : 0;
		for (const Fill &f : A.fills) tot += (f.redo_job >= 0 ? redo[f.redo_job] : jobs[f.job]).cigar.size();
		if (A.right_job >= 0) tot += jobs[A.right_job].cigar.size();
		r.cigar.reserve(tot);
	}
	if (A.left_job >= 0) {
		const KswReq &j = jobs[A.left_job];
		if (j.ez.n_cigar > 0) { append_cigar(r, j.cigar); r.dp_score += j.ez.max; }
		rs1 = A.rs - (j.ez.reach_end ? j.ez.mqe_t + 1 : j.ez.max_t + 1);
		qs1 = A.qs - (j.ez.reach_end ? A.qs - A.qs0 : j.ez.max_q + 1);
	} else rs1 = A.rs, qs1 = A.qs;
	re1 = A.rs, qe1 = A.qs;
	int32_t last_re = A.rs, last_qe = A.qs;
	if (A.cnt1 > 1 || true) {
		// re1/qe1 follow the last inspected anchor even when no DP is run for it (src/align.c:715)
	}
	for (size_t k = 0; k < A.fills.size(); ++k) {
		const Fill &f = A.fills[k];
		const KswReq &j = f.redo_job >= 0 ? redo[f.redo_job] : jobs[f.job];
		const int zdrop_code = A.redo_code[k];
		re1 = f.re, qe1 = f.qe;
		if (j.ez.n_cigar > 0) append_cigar(r, j.cigar);
		if (j.ez.zdropped) {                                           // truncated: maybe split the chain here (:741-761)
			if (!r.has_p) { r.has_p = true; r.cigar.clear(); r.dp_score = r.dp_max = r.dp_max2 = 0; r.n_ambi = 0; }
			int jj;
			for (jj = f.idx - 1; jj >= 0; --jj)
				if ((int32_t)a[A.as1 + jj].x <= f.rs + j.ez.max_t) break;
			dropped = true;
			if (jj < 0) jj = 0;
			r.dp_score += j.ez.max;
			re1 = f.rs + (j.ez.max_t + 1);
			qe1 = f.qs + (j.ez.max_q + 1);
			if (A.cnt1 - (jj + 1) >= opt.min_cnt) {
				split_reg(r, A.r2, A.as1 + jj + 1 - r.as, qlen, a);
				if (zdrop_code == 2) A.r2.split_inv = 1;
			}
			break;
		} else r.dp_score += j.ez.score;
		last_re = f.re, last_qe = f.qe;
	}
	(void)last_re; (void)last_qe;
	if (!dropped) {
		// NB: without a z-drop the loop leaves re1/qe1 at the LAST anchor examined (src/align.c:715), which is A.re/A.qe
		if (A.cnt1 > 1) re1 = A.re, qe1 = A.qe;
		if (A.right_job >= 0) {
			const KswReq &j = jobs[A.right_job];
			if (j.ez.n_cigar > 0) { append_cigar(r, j.cigar); r.dp_score += j.ez.max; }
			re1 = A.re + (j.ez.reach_end ? j.ez.mqe_t + 1 : j.ez.max_t + 1);
			qe1 = A.qe + (j.ez.reach_end ? A.qe0 - A.qe : j.ez.max_q + 1);
		}
	}
	WM_INVARIANT(qe1 <= qlen);
	r.rs = rs1, r.re = re1;
	if (A.rev) r.qs = qlen - qe1, r.qe = qlen - qs1;
	else r.qs = qs1, r.qe = qe1;
	if (r.has_p) {
		WM_INVARIANT(rs1 >= A.rs0 && re1 <= A.re0);
		update_extra(r, E.qseq0[r.rev] + qs1, A.tbuf.data() + (rs1 - A.rs0), E.mat, opt.q, opt.e);
	}
}

// mm_align1_inv (src/align.c:797-852): try to align the reverse strand between two pieces split by an inversion z-drop
static bool align_inv(Scheduler &sch, const AlnEnv &E, const Reg &r1, const Reg &r2, Reg &r_inv)
{
	const MapOpt &opt = *E.opt;
	r_inv = Reg();
	if (!(r1.split & 1) || !(r2.split & 2)) return false;
	if (r1.id != r1.parent && r1.parent != PARENT_TMP_PRI) return false;
	if (r2.id != r2.parent && r2.parent != PARENT_TMP_PRI) return false;
	if (r1.rid != r2.rid || r1.rev != r2.rev) return false;
	const int ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe, tl = r2.rs - r1.re;
	if (ql < opt.min_chain_score || ql > opt.max_gap) return false;
	if (tl < opt.min_chain_score || tl > opt.max_gap) return false;
	std::vector<uint8_t> tseq = ref_codes(*E.idx, r1.rid, r1.re, r2.rs);
	const uint8_t *qsrc = r1.rev ? &E.qseq0[0][r2.qe] : &E.qseq0[1][E.qlen - r2.qs];
	std::vector<uint8_t> qr(ql), tr(tseq.rend() - tl, tseq.rend());      // (tseq carries padding behind its tl bases)
	for (int i = 0; i < ql; ++i) qr[i] = qsrc[ql - 1 - i];
	int q_off, t_off;
	const int score = ll_i16(ql, qr.data(), tl, tr.data(), E.mat, opt.q, opt.e, &q_off, &t_off);
	if (score < opt.min_dp_max) return false;
	// NB: the striped layout pads the query to a multiple of 8, so the best cell may sit on a pad slot and q_off can
	// be as low as -7: the reference then starts the extension a few bases BEFORE the gap (pointer arithmetic on its
	// strand buffer, src/align.c:826-828). Reproduced here through the contiguous two-strand buffer.
	q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
	const uint8_t *qstart = qsrc + q_off;
	std::vector<KswReq> jobs(1);
	jobs[0].qp = qstart; jobs[0].ql = ql - q_off; jobs[0].tp = tseq.data() + t_off; jobs[0].tl = tl - t_off; jobs[0].step = 1;
	jobs[0].qwin_off = E.q_dev_off; jobs[0].qwin_len = E.qlen; jobs[0].q_pos = (int32_t)(qstart - E.qseq0[0]);   // (two-strand space: the strands are contiguous)
	jobs[0].rid = r1.rid; jobs[0].t_pos = r1.re + t_off; jobs[0].has_n = E.q_has_n || q_off < 0 || E.idx->has_n(r1.rid, r1.re, r2.rs);
	jobs[0].w = (int)(opt.bw * 1.5); jobs[0].end_bonus = -1; jobs[0].zdrop = opt.zdrop; jobs[0].flag = EZ_EXTZ_ONLY;
	sch.ksw(jobs);
	const KswReq &j = jobs[0];
	if (j.ez.n_cigar == 0) return false;
	append_cigar(r_inv, j.cigar);
	r_inv.dp_score = j.ez.max;
	r_inv.id = -1; r_inv.parent = PARENT_UNSET; r_inv.inv = 1; r_inv.rev = !r1.rev; r_inv.rid = r1.rid; r_inv.div = -1.0f;
	if (r_inv.rev == 0) { r_inv.qs = r2.qe + q_off; r_inv.qe = r_inv.qs + j.ez.max_q + 1; }
	else { r_inv.qe = r2.qs - q_off; r_inv.qs = r_inv.qe - (j.ez.max_q + 1); }
	r_inv.rs = r1.re + t_off;
	r_inv.re = r_inv.rs + j.ez.max_t + 1;
	update_extra(r_inv, qstart, &tseq[t_off], E.mat, opt.q, opt.e);
	return true;
}

void align_skeleton(Scheduler &sch, const MapOpt &opt, const Index &idx, int qlen, const uint8_t *qcodes, int64_t q_dev_off, std::vector<Reg> &regs, m128 *a)
{
	AlnEnv E;
	E.opt = &opt; E.idx = &idx; E.qlen = qlen; E.q_dev_off = q_dev_off; E.q_has_n = false;
	// both strands in ONE buffer, reverse complement right behind the forward strand, exactly like the reference's
	// qseq0 (src/align.c:871-877): mm_align1_inv may step a few bases in front of a strand (see align_inv)
	// (three flat loops the compiler vectorises — a copy, a reversed complement, an OR-reduction — instead of one byte loop with two branches:
	// that loop was 10 % of the host glue's CPU samples, tests/host_harness/prof.sh)
	std::unique_ptr<uint8_t[]> both(new uint8_t[(size_t)2 * qlen + 8 + WM_WALK_PAD]);
	memset(both.get(), 4, 8);
	memset(both.get() + 8 + (size_t)2 * qlen, 4, WM_WALK_PAD);
	uint8_t *fw = both.get() + 8, *rc = fw + qlen;
	memcpy(fw, qcodes, (size_t)qlen);
	{
		const uint8_t *last = qcodes + qlen - 1;
		unsigned any_n = 0;
		for (int i = 0; i < qlen; ++i) { const uint8_t c = last[-i]; rc[i] = (uint8_t)(c < 4 ? 3 ^ c : 4); }      // (3 - c == 3 ^ c for c in 0..3)
		for (int i = 0; i < qlen; ++i) any_n |= qcodes[i];
		E.q_has_n = (any_n & 0xfc) != 0;
	}
	E.qseq0[0] = fw; E.qseq0[1] = rc;
	gen_simple_mat(E.mat, opt.a, opt.b, opt.sc_ambi);
	const int n_a = squeeze_a(regs, a);

	// The reference aligns regions one after another and inserts the pieces split off by a z-drop right behind
	// their origin (src/align.c:879-913). Regions are independent, so every round aligns all pending regions at
	// once; new pieces are aligned in the next round and the list order reproduces the sequential insertion order.
	struct Node { Reg r; bool pending; };
	std::list<Node> L;
	for (Reg &r : regs) L.push_back(Node{ std::move(r), true });
	for (;;) {
		std::vector<std::list<Node>::iterator> todo;
		for (auto it = L.begin(); it != L.end(); ++it) if (it->pending) todo.push_back(it);
		if (todo.empty()) break;
		// splice mode with both transcript strands allowed: every region is aligned twice, once per assumed strand, and the better
		// scoring pass is kept (src/align.c:884-900); the two passes are independent and run in the same batches
		const bool is_splice = (opt.flag & F_SPLICE) != 0, two_strands = is_splice && (opt.flag & F_SPLICE_FOR) && (opt.flag & F_SPLICE_REV);
		std::vector<RegAln> A(todo.size()), B(two_strands ? todo.size() : 0);
		std::vector<KswReq> jobs, redo;
		for (size_t k = 0; k < todo.size(); ++k) {
			if (two_strands) { B[k].r = todo[k]->r; B[k].n_a = n_a; }
			A[k].r = std::move(todo[k]->r); A[k].n_a = n_a;
			// job indices inside RegAln are relative to the shared vector
			plan_reg(E, A[k], a, jobs, two_strands ? (int64_t)F_SPLICE_FOR : opt.flag);
			if (two_strands) plan_reg(E, B[k], a, jobs, F_SPLICE_REV);
		}
		sch.ksw(jobs);
		for (size_t k = 0; k < todo.size(); ++k) { judge_reg(E, A[k], jobs, redo); if (two_strands) judge_reg(E, B[k], jobs, redo); }
		sch.ksw(redo);
		for (size_t k = 0; k < todo.size(); ++k) {
			finish_reg(E, A[k], a, jobs, redo);
			if (two_strands) {
				finish_reg(E, B[k], a, jobs, redo);
				// (the reference reads p->dp_score of both passes unconditionally; a pass without any CIGAR counts as 0 here)
				const int s0 = A[k].r.has_p ? A[k].r.dp_score : 0, s1 = B[k].r.has_p ? B[k].r.dp_score : 0;
				int which, trans_strand;
				if (s0 > s1) which = 0, trans_strand = 1;
				else if (s0 < s1) which = 1, trans_strand = 2;
				else trans_strand = 3, which = (qlen + s0) & 1;
				if (which == 1) std::swap(A[k], B[k]);
				if (A[k].r.has_p) A[k].r.trans_strand = trans_strand;
			} else if (is_splice && A[k].r.has_p) A[k].r.trans_strand = (opt.flag & F_SPLICE_FOR) ? 1 : 2;
			auto it = todo[k];
			it->r = std::move(A[k].r); it->pending = false;
			auto after = std::next(it);
			// inversion rescue between this piece and its predecessor (src/align.c:906-911)
			if (it != L.begin() && it->r.split_inv) {
				Reg inv;
				if (align_inv(sch, E, std::prev(it)->r, it->r, inv)) L.insert(after, Node{ std::move(inv), false });
			}
			if (A[k].r2.cnt > 0) L.insert(after, Node{ std::move(A[k].r2), true });
		}
	}
	regs.clear();
	for (Node &n : L) regs.push_back(std::move(n.r));
	filter_regs(opt, qlen, regs);
	hit_sort(regs);
}

} // namespace wm
