/* oracle/wm_oracle.c — TEST INFRASTRUCTURE ONLY (see wm_oracle.h).
 *
 * CPU restatement of the reference's hot path, written from its behaviour; every function cites the
 * reference file:line it follows (paths relative to /root/reference). Compiled with -ffp-contract=off:
 * the minimizer order and the chain gap cost are IEEE fp64/fp32 and must not be fused.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "wm_oracle.h"

/* ------------------------------------------------------------------------------------------------
 * Nucleotide code table: A/a=0 C/c=1 G/g=2 T/t/U/u=3, everything else 4 (src/sketch.c:19-36).
 * ---------------------------------------------------------------------------------------------- */
static uint8_t nt4(uint8_t ch)
{
	switch (ch) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return ch < 4 ? ch : 4; /* bytes 0..3 map to themselves in the reference table */
	}
}

/* src/sketch.c:53-63 — invertible integer mix confined to 2k bits. */
uint64_t wmo_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

/* src/sketch.c:43-51 — MurmurHash3 64-bit finaliser (mask = all ones at the only call site :72). */
uint64_t wmo_fmix64(uint64_t key)
{
	key ^= key >> 33;
	key *= 0xff51afd7ed558ccdULL;
	key ^= key >> 33;
	key *= 0xc4ceb9fe1a85ec53ULL;
	key ^= key >> 33;
	return key;
}

/* src/sketch.c:70-89 — order key in [-1,0]; x^8 by three squarings when the k-mer is down-weighted. */
double wmo_order(uint64_t kmer, int in_filter)
{
	double x = (double)wmo_fmix64(kmer) * 1.0 / 18446744073709551616.0; /* (double)UINT64_MAX == 2^64 */
	if (in_filter) {
		double p2 = x * x;
		double p4 = p2 * p2;
		return -1.0 * (p4 * p4);
	}
	return -1.0 * x;
}

/* ------------------------------------------------------------------------------------------------
 * Bloom filter (ext/bloom/bloom_filter.hpp). Sizing :108-160 with the parameters of src/index.c:411-414
 * (projected = max(n,1000), fpp = 0.001, at most 2 hashes); seed :186; salts :513-528; hash :551-608.
 * ---------------------------------------------------------------------------------------------- */
wmo_bloom_t *wmo_bloom_new(uint64_t n_kmers)
{
	wmo_bloom_t *f = (wmo_bloom_t*)calloc(1, sizeof(*f));
	double n = (double)(n_kmers > 1000 ? n_kmers : 1000), best = INFINITY, kk;
	for (kk = 1.0; kk < 1000.0; kk += 1.0) {
		double m = (-kk * n) / log(1.0 - pow(0.001, 1.0 / kk));
		if (m < best) best = m;
	}
	f->table_bits = (uint64_t)best;
	if (f->table_bits % 8) f->table_bits += 8 - f->table_bits % 8;
	{ /* two predefined salts mixed with random_seed_ = 0xA5A5A5A55A5A5A5A*0xA5A5A5A5+1, in place, in order */
		uint64_t seed = 0xA5A5A5A55A5A5A5AULL * 0xA5A5A5A5ULL + 1;
		uint32_t s[2] = { 0xAAAAAAAAu, 0x55555555u };
		int i;
		for (i = 0; i < 2; ++i) s[i] = s[i] * s[(i + 3) % 2] + (uint32_t)seed;
		f->salt[0] = s[0], f->salt[1] = s[1];
	}
	f->bits = (uint8_t*)calloc(f->table_bits / 8, 1);
	return f;
}
void wmo_bloom_free(wmo_bloom_t *f) { if (f) { free(f->bits); free(f); } }

uint32_t wmo_bloom_hash(uint64_t key, uint32_t h)
{ /* one 8-byte round of hash_ap: i1 = low word, i2 = high word (little endian POD insert, :276-280) */
	uint32_t i1 = (uint32_t)key, i2 = (uint32_t)(key >> 32);
	h ^= (h << 7) ^ i1 * (h >> 3) ^ (~((h << 11) + (i2 ^ (h >> 5))));
	return h;
}
void wmo_bloom_insert(wmo_bloom_t *f, uint64_t key)
{
	int i;
	for (i = 0; i < 2; ++i) {
		uint64_t bit = wmo_bloom_hash(key, f->salt[i]) % f->table_bits;
		f->bits[bit >> 3] |= (uint8_t)(1u << (bit & 7));
	}
	++f->n_inserted;
}
int wmo_bloom_contains(const wmo_bloom_t *f, uint64_t key)
{
	int i;
	if (f == 0) return 0;
	for (i = 0; i < 2; ++i) {
		uint64_t bit = wmo_bloom_hash(key, f->salt[i]) % f->table_bits;
		if (!(f->bits[bit >> 3] >> (bit & 7) & 1)) return 0;
	}
	return 1;
}
/* src/index.c:362-376 — canonical 2-bit encoding of a k-mer string (no masking: k <= 28 at the caller) */
uint64_t wmo_encode_kmer(const char *s, int k)
{
	uint64_t fw = 0, rc = 0;
	int i;
	for (i = 0; i < k; ++i) {
		uint64_t c = nt4((uint8_t)s[i]);
		fw = fw << 2 | c;
		rc = rc >> 2 | (3ULL ^ c) << (2 * (k - 1));
	}
	return fw < rc ? fw : rc;
}

/* ------------------------------------------------------------------------------------------------
 * mm_sketch (src/sketch.c:128-219). Weighted robust winnowing; see SURVEY.md Appendix D.
 * is_hpc (homopolymer compression, :152-163): a run of one unambiguous base is ONE step of the automaton — a k-mer is made of k consecutive runs, its
 * position is the LAST base of its last run (:159), its span (the low byte of x) the summed length of those runs (:161-162, a queue of the last k run
 * lengths that an ambiguous base empties, :175); k-mers that span 256 bases or more are not used (:168).
 * ---------------------------------------------------------------------------------------------- */
#define EMPTY64 UINT64_MAX
static int64_t sketch_impl(const char *seq, int len, int w, int k, uint32_t rid, const wmo_bloom_t *f, int is_hpc,
                           uint64_t *ox, uint64_t *oy, int64_t cap)
{
	const uint64_t mask = (1ULL << 2 * k) - 1, top = 2 * (uint64_t)(k - 1);
	uint64_t fw = 0, rc = 0, ring_x[256], ring_y[256], best_x = EMPTY64, best_y = EMPTY64;
	double ring_o[256], best_o = 2.0;
	int i, j, run = 0, slot = 0, best_slot = 0;
	int rl[32], rl_n = 0, rl_head = 0, span = 0;            /* HPC: lengths of the last (up to) k runs since the last ambiguous base, and their sum */
	int64_t n_out = 0;
	assert(len > 0 && w > 0 && w < 256 && k > 0 && k <= 28);
	for (j = 0; j < w; ++j) ring_x[j] = ring_y[j] = EMPTY64, ring_o[j] = 2.0;
#define EMIT() do { if (n_out < cap) ox[n_out] = best_x, oy[n_out] = best_y; ++n_out; } while (0)
	for (i = 0; i < len; ++i) {
		int c = nt4((uint8_t)seq[i]);
		uint64_t cx = EMPTY64, cy = EMPTY64;
		double co = 2.0;
		if (c < 4) {
			int strand;
			if (is_hpc) {                             /* the whole run is this step; i moves to its last base (:153-159) */
				int n = 1;
				while (i + n < len && nt4((uint8_t)seq[i + n]) == c) ++n;
				i += n - 1;
				rl[(rl_head + rl_n++) & 31] = n; span += n;
				if (rl_n > k) { span -= rl[rl_head & 31]; ++rl_head; --rl_n; }
			}
			fw = (fw << 2 | (uint64_t)c) & mask;
			rc = rc >> 2 | (3ULL ^ (uint64_t)c) << top;
			if (fw == rc) continue;                 /* palindrome: the whole step is skipped (:166) */
			strand = fw < rc ? 0 : 1;
			if (++run >= k && (!is_hpc || span < 256)) {
				uint64_t km = strand ? rc : fw;
				cx = wmo_hash64(km, mask) << 8 | (uint64_t)(is_hpc ? span : k);
				cy = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint64_t)strand;
				co = wmo_order(km, wmo_bloom_contains(f, km));
			}
		} else run = 0, rl_n = rl_head = 0, span = 0;
		ring_x[slot] = cx, ring_y[slot] = cy, ring_o[slot] = co;
		if (co < best_o) {                           /* strictly smaller: older of equal orders stays (:180) */
			if (run >= w + k && best_x != EMPTY64) EMIT();
			best_x = cx, best_y = cy, best_o = co, best_slot = slot;
		} else if (slot == best_slot) {              /* the minimum is being overwritten (:191) */
			if (run >= w + k - 1 && best_x != EMPTY64) EMIT();
			best_x = best_y = EMPTY64, best_o = 2.0;
			for (j = slot + 1; j < w; ++j)           /* oldest → newest, >= keeps the newest of equals */
				if (best_o >= ring_o[j]) best_x = ring_x[j], best_y = ring_y[j], best_o = ring_o[j], best_slot = j;
			for (j = 0; j <= slot; ++j)
				if (best_o >= ring_o[j]) best_x = ring_x[j], best_y = ring_y[j], best_o = ring_o[j], best_slot = j;
		}
		if (++slot == w) slot = 0;
	}
	if (best_x != EMPTY64) EMIT();
#undef EMIT
	return n_out;
}
int64_t wmo_sketch(const char *seq, int len, int w, int k, uint32_t rid, const wmo_bloom_t *f, uint64_t *ox, uint64_t *oy, int64_t cap)
{
	return sketch_impl(seq, len, w, k, rid, f, 0, ox, oy, cap);
}
int64_t wmo_sketch_hpc(const char *seq, int len, int w, int k, uint32_t rid, const wmo_bloom_t *f, uint64_t *ox, uint64_t *oy, int64_t cap)
{
	return sketch_impl(seq, len, w, k, rid, f, 1, ox, oy, cap);
}

/* ------------------------------------------------------------------------------------------------
 * radix_sort_128x / radix_sort_64 (src/ksort.h:101-151, src/misc.c:155-159): key = .x, in-place MSD
 * "American flag" byte sort, buckets <= 64 finished by insertion sort. Unstable; the exact permutation
 * matters (ties feed the chain DP) — SURVEY.md Appendix G.
 * ---------------------------------------------------------------------------------------------- */
#define DEF_RADIX(NAME, T, KEY) \
static void NAME##_ins(T *b, T *e) { \
	T *i, *j; \
	for (i = b + 1; i < e; ++i) if (KEY(*i) < KEY(*(i - 1))) { \
		T t = *i; \
		for (j = i; j > b && KEY(t) < KEY(*(j - 1)); --j) *j = *(j - 1); \
		*j = t; \
	} \
} \
static void NAME##_rec(T *beg, T *end, int shift) { \
	T *lo[256], *hi[256], *i; \
	int64_t cnt[256]; \
	int d, c; \
	memset(cnt, 0, sizeof(cnt)); \
	for (i = beg; i != end; ++i) ++cnt[KEY(*i) >> shift & 255]; \
	{ T *p = beg; for (d = 0; d < 256; ++d) { lo[d] = p; p += cnt[d]; hi[d] = p; } } \
	for (d = 0; d < 256;) { \
		if (lo[d] == hi[d]) { ++d; continue; } \
		c = (int)(KEY(*lo[d]) >> shift & 255); \
		if (c == d) { ++lo[d]; continue; } \
		{ T hand = *lo[d], sw; \
		  do { sw = hand; hand = *lo[c]; *lo[c]++ = sw; c = (int)(KEY(hand) >> shift & 255); } while (c != d); \
		  *lo[d]++ = hand; } \
	} \
	if (shift) { \
		int ns = shift > 8 ? shift - 8 : 0; \
		T *p = beg; \
		for (d = 0; d < 256; ++d) { \
			T *q = hi[d]; \
			if (q - p > 64) NAME##_rec(p, q, ns); \
			else if (q - p > 1) NAME##_ins(p, q); \
			p = q; \
		} \
	} \
} \
void NAME(T *beg, T *end) { if (end - beg <= 64) NAME##_ins(beg, end); else NAME##_rec(beg, end, 56); }
#define KEY128(a) ((a).x)
#define KEY64(a) (a)
DEF_RADIX(wmo_radix_sort_128x, wmo128_t, KEY128)
DEF_RADIX(wmo_radix_sort_64, uint64_t, KEY64)

/* ------------------------------------------------------------------------------------------------
 * mm_chain_dp (src/chain.c:22-167) for one query segment, is_cdna = 0.
 * ---------------------------------------------------------------------------------------------- */
static int ilog2_u32(uint32_t v) { int l = -1; while (v) { v >>= 1; ++l; } return l; } /* :15-20 (table form) */

int64_t wmo_chain_stat[4]; /* diagnostics only: predecessors visited, sum of (i-st), max (i-st), 64-wide tiles touched */

/* splice mode (is_cdna of src/chain.c:22): selected per thread before the call, so that the signature the tests bind stays as it is */
static __thread int wmo_chain_cdna = 0;
void wmo_chain_set_cdna(int is_cdna) { wmo_chain_cdna = is_cdna; }

int64_t wmo_chain_dp(int max_dist_x, int min_dist_x, int max_dist_y, int bw, int max_skip, int max_iter,
                     int min_cnt, int min_sc, float gap_scale, int64_t n, const wmo128_t *a,
                     int *n_u_, uint64_t *u, wmo128_t *b)
{
	int32_t *f, *p, *t, *v;
	int64_t i, j, st = 0, n_u = 0, n_v = 0, k;
	uint64_t sum_span = 0;
	float avg_span;
	*n_u_ = 0;
	if (n == 0) return 0;
	f = (int32_t*)malloc(n * 4); p = (int32_t*)malloc(n * 4);
	t = (int32_t*)calloc(n, 4);  v = (int32_t*)malloc(n * 4);
	for (i = 0; i < n; ++i) sum_span += a[i].y >> 32 & 0xff;
	avg_span = (float)sum_span / n;                                   /* :42-43 */
	for (i = 0; i < n; ++i) {                                          /* score fill :45-90 */
		uint64_t ri = a[i].x;
		int32_t qi = (int32_t)a[i].y, span = (int32_t)(a[i].y >> 32 & 0xff);
		int32_t best = span, n_skip = 0;
		int64_t best_j = -1;
		while (st < i && ri > a[st].x + (uint64_t)max_dist_x) ++st;   /* :50 */
		if (i - st > max_iter)                                         /* Winnowmap window relaxation :51-55 */
			while (i - st > max_iter && ri > a[st].x + (uint64_t)min_dist_x) ++st;
		wmo_chain_stat[1] += i - st; if (i - st > wmo_chain_stat[2]) wmo_chain_stat[2] = i - st;
		for (j = i - 1; j >= st; --j) {
			int64_t dr = (int64_t)(ri - a[j].x);
			++wmo_chain_stat[0]; if (((i - 1 - j) & 63) == 0) ++wmo_chain_stat[3];
			int32_t dq = qi - (int32_t)a[j].y, dd, sc, lg, gc;
			if (dr == 0 || dq <= 0) continue;                          /* :60 */
			if (dq > max_dist_y || dq > max_dist_x) continue;          /* :61 */
			dd = dr > dq ? (int32_t)(dr - dq) : (int32_t)(dq - dr);
			if (dd > bw) continue;                                      /* :63 */
			{ int32_t md = dq < dr ? dq : (int32_t)dr; sc = md > span ? span : md; } /* :65-66 */
			lg = dd ? ilog2_u32((uint32_t)dd) : 0;
			gc = (int)(dd * .01 * avg_span) + (lg >> 1);               /* :76 */
			if (wmo_chain_cdna && dr > dq) { int c_lin = (int)(dd * .01 * avg_span); gc = c_lin < lg ? c_lin : lg; }   /* :69-74, one segment */
			sc -= (int)((double)gc * gap_scale + .499);                /* :77 */
			sc += f[j];
			if (sc > best) {
				best = sc, best_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {
				if (++n_skip > max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = (int32_t)i;
		}
		f[i] = best, p[i] = (int32_t)best_j;
		v[i] = best_j >= 0 && v[best_j] > best ? v[best_j] : best;
	}
	/* chain ends :93-116 */
	memset(t, 0, n * 4);
	for (i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
	for (i = 0; i < n; ++i)
		if (t[i] == 0 && v[i] >= min_sc) {
			j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];
			if (j < 0) j = i;
			u[n_u++] = (uint64_t)f[j] << 32 | (uint64_t)j;
		}
	if (n_u == 0) { free(f); free(p); free(t); free(v); return 0; }
	wmo_radix_sort_64(u, u + n_u);
	for (i = 0; i < n_u >> 1; ++i) { uint64_t x = u[i]; u[i] = u[n_u - 1 - i]; u[n_u - 1 - i] = x; }
	/* backtrack :119-135 */
	memset(t, 0, n * 4);
	for (i = 0, k = 0; i < n_u; ++i) {
		int64_t nv0 = n_v, k0 = k;
		j = (int32_t)u[i];
		do { v[n_v++] = (int32_t)j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
		if (j < 0) {
			if (n_v - nv0 >= min_cnt) u[k++] = u[i] >> 32 << 32 | (uint64_t)(n_v - nv0);
		} else if ((int32_t)(u[i] >> 32) - f[j] >= min_sc) {
			if (n_v - nv0 >= min_cnt) u[k++] = ((u[i] >> 32) - (uint64_t)f[j]) << 32 | (uint64_t)(n_v - nv0);
		}
		if (k0 == k) n_v = nv0;
	}
	n_u = k;
	/* gather chains, then order them by the x of their first anchor :141-165 */
	{
		wmo128_t *tmp = (wmo128_t*)malloc((n_v ? n_v : 1) * sizeof(wmo128_t));
		wmo128_t *w = (wmo128_t*)malloc((n_u ? n_u : 1) * sizeof(wmo128_t));
		uint64_t *u2 = (uint64_t*)malloc((n_u ? n_u : 1) * 8);
		for (i = 0, k = 0; i < n_u; ++i) {
			int64_t k0 = k, ni = (int32_t)u[i];
			for (j = 0; j < ni; ++j) tmp[k++] = a[v[k0 + (ni - j - 1)]];
		}
		for (i = k = 0; i < n_u; ++i) {
			w[i].x = tmp[k].x, w[i].y = (uint64_t)k << 32 | (uint64_t)i;
			k += (int32_t)u[i];
		}
		wmo_radix_sort_128x(w, w + n_u);
		for (i = k = 0; i < n_u; ++i) {
			int32_t src = (int32_t)w[i].y, cnt = (int32_t)u[src];
			u2[i] = u[src];
			memcpy(&b[k], &tmp[w[i].y >> 32], (size_t)cnt * sizeof(wmo128_t));
			k += cnt;
		}
		memcpy(u, u2, (size_t)n_u * 8);
		free(tmp); free(w); free(u2);
	}
	free(f); free(p); free(t); free(v);
	*n_u_ = (int)n_u;
	return n_v;
}

/* ------------------------------------------------------------------------------------------------
 * ksw_extd2_sse (src/ksw2_extd2_sse.c:26-393) — lane-exact scalar emulation, wrapping int8 state,
 * 16-lane hull, stale lanes included; ksw_backtrack / ksw_apply_zdrop from src/ksw2.h:119-176.
 * Memory layout of the reference's single calloc'd block is reproduced (u v x y x2 y2 s sf qr) because
 * the score pass may read sf past its end (into qr) and write s past its end (into sf) — SURVEY §7.
 * ---------------------------------------------------------------------------------------------- */
#define I8(x) ((int8_t)(x))
static int push_op(uint32_t *cig, int n, uint32_t op, int len)
{ /* src/ksw2.h:103-113 */
	if (n == 0 || op != (cig[n - 1] & 0xf)) cig[n++] = (uint32_t)len << 4 | op;
	else cig[n - 1] += (uint32_t)len << 4;
	return n;
}
static int backtrack(int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0, uint32_t *cig)
{ /* src/ksw2.h:119-151 with is_rot = 1, min_intron_len = 0 */
	int n = 0, i = i0, j = j0, state = 0, k;
	while (i >= 0 && j >= 0) {
		int r = i + j, force = -1;
		uint32_t d;
		if (i < off[r]) force = 2;
		if (i > off_end[r]) force = 1;
		d = force < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = d & 7;
		else if (!(d >> (state + 2) & 1)) state = 0;
		if (state == 0) state = d & 7;
		if (force >= 0) state = force;
		if (state == 0) n = push_op(cig, n, 0, 1), --i, --j;
		else if (state == 1 || state == 3) n = push_op(cig, n, 2, 1), --i;
		else n = push_op(cig, n, 1, 1), --j;
	}
	if (i >= 0) n = push_op(cig, n, 2, i + 1);
	if (j >= 0) n = push_op(cig, n, 1, j + 1);
	if (!is_rev)
		for (k = 0; k < n >> 1; ++k) { uint32_t x = cig[k]; cig[k] = cig[n - 1 - k]; cig[n - 1 - k] = x; }
	return n;
}
static int apply_zdrop(wmo_ez_t *ez, int32_t H, int r, int t, int zdrop, int e)
{ /* src/ksw2.h:160-176, is_rot = 1 */
	if (H > ez->max) {
		ez->max = H, ez->max_t = t, ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && ez->max - H > zdrop + l * e) { ez->zdropped = 1; return 1; }
	}
	return 0;
}
static void reset_ez(wmo_ez_t *ez)
{ /* src/ksw2.h:153-158 */
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = WMO_NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
}

void wmo_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int q_, int e_, int q2_, int e2_, int w, int zdrop, int end_bonus, int flag,
                   wmo_ez_t *ez, uint32_t *cigar_out, int *stats)
{
	int8_t q = (int8_t)q_, e = (int8_t)e_, q2 = (int8_t)q2_, e2 = (int8_t)e2_;
	const int approx = !!(flag & 0x08), right = !!(flag & 0x02), extz_only = !!(flag & 0x40), rev_cigar = !!(flag & 0x80);
	const int generic = !!(flag & 0x04), with_cigar = !(flag & 0x01), approx_drop = !!(flag & 0x10);
	int r, t, T, n_col, last_st = -1, last_en = -1, long_thres, long_diff, max_sc, min_sc, qe, qe2;
	int8_t *blk, *u, *v, *x, *y, *x2, *y2, *s, sc_mch, sc_mis, sc_N;
	uint8_t *sf, *qr, *p = 0;
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int *off = 0, *off_end = 0, st_out = 0, st_in = 0;

	reset_ez(ez);
	if (stats) stats[0] = stats[1] = 0;
	if (m <= 1 || qlen <= 0 || tlen <= 0) return;                                     /* :68 */
	if (q2 + e2 < q + e) { int8_t z; z = q, q = q2, q2 = z; z = e, e = e2, e2 = z; } /* :70 */
	qe = q + e, qe2 = q2 + e2;
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? (int8_t)-e2 : mat[m * m - 1];                         /* :79 */
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	T = (tlen + 15) / 16 * 16;
	n_col = qlen < tlen ? qlen : tlen;
	n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;                   /* :85-86, in bytes */
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		if (mat[t] > max_sc) max_sc = mat[t];
		if (mat[t] < min_sc) min_sc = mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;                                                  /* :92 */
	long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;                                 /* :94-97 */
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	blk = (int8_t*)calloc((size_t)8 * T + ((size_t)(qlen + 15) / 16 + 1) * 16, 1);    /* :99 */
	u = blk, v = u + T, x = v + T, y = x + T, x2 = y + T, y2 = x2 + T, s = y2 + T;
	sf = (uint8_t*)(s + T), qr = sf + T;
	memset(u, -qe, T); memset(v, -qe, T); memset(x, -qe, T); memset(y, -qe, T);
	memset(x2, -qe2, T); memset(y2, -qe2, T);
	if (!approx) {
		H = (int32_t*)malloc((size_t)T * 4);
		for (t = 0; t < T; ++t) H[t] = WMO_NEG_INF;
	}
	if (with_cigar) {
		p = (uint8_t*)malloc((size_t)(qlen + tlen - 1) * n_col + 16);
		off = (int*)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);

	for (r = 0; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1;
		const uint8_t *qrr = qr + (qlen - 1 - r);
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
		if (en > (r + w) >> 1) en = (r + w) >> 1;
		if (st > en) { ez->zdropped = 1; break; }                                       /* :134-137 */
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) {                                                                    /* :141-151 */
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			else x1 = I8(-qe), x21 = I8(-qe2), v1 = I8(-qe);
		} else {
			x1 = I8(-qe), x21 = I8(-qe2);
			v1 = r == 0 ? I8(-qe) : r < long_thres ? I8(-e) : r == long_thres ? I8(long_diff) : I8(-e2);
		}
		if (en >= r) {                                                                   /* :152-155 */
			y[r] = I8(-qe), y2[r] = I8(-qe2);
			u[r] = r == 0 ? I8(-qe) : r < long_thres ? I8(-e) : r == long_thres ? I8(long_diff) : I8(-e2);
		}
		if (!generic) {                                                                  /* :158-173, 16-byte chunks from st0 */
			for (t = st0; t <= en0; t += 16) {
				int8_t tmp16[16];
				int i;
				for (i = 0; i < 16; ++i) {
					uint8_t a = sf[t + i], b = qrr[t + i];
					tmp16[i] = (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) ? sc_N : a == b ? sc_mch : sc_mis;
				}
				memcpy(s + t, tmp16, 16);
			}
		} else {
			for (t = st0; t <= en0; ++t) s[t] = mat[sf[t] * m + qrr[t]];
		}
		if (with_cigar) off[r] = st, off_end[r] = en;
		for (t = st; t <= en; ++t) {                                                     /* :183-314 lane by lane */
			int8_t z = s[t], xo = x[t], vo = v[t], x2o = x2[t], ut = u[t];
			int8_t a = I8(x1 + v1), b = I8(y[t] + ut), a2 = I8(x21 + v1), b2 = I8(y2[t] + ut), tmp;
			uint8_t d;
			if (!right) {
				d = a > z ? 1 : 0;  z = z > a ? z : a;
				d = b > z ? 2 : d;  z = z > b ? z : b;
				d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
				d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
			} else {
				d = z > a ? 0 : 1;  z = z > a ? z : a;
				d = z > b ? d : 2;  z = z > b ? z : b;
				d = z > a2 ? d : 3; z = z > a2 ? z : a2;
				d = z > b2 ? d : 4; z = z > b2 ? z : b2;
			}
			z = z < sc_mch ? z : sc_mch;
			u[t] = I8(z - v1); v[t] = I8(z - ut);
			tmp = I8(z - q);  a = I8(a - tmp);  b = I8(b - tmp);
			tmp = I8(z - q2); a2 = I8(a2 - tmp); b2 = I8(b2 - tmp);
			if (!right) {
				x[t]  = I8((a  > 0 ? a  : 0) - qe);  d |= a  > 0 ? 0x08 : 0;
				y[t]  = I8((b  > 0 ? b  : 0) - qe);  d |= b  > 0 ? 0x10 : 0;
				x2[t] = I8((a2 > 0 ? a2 : 0) - qe2); d |= a2 > 0 ? 0x20 : 0;
				y2[t] = I8((b2 > 0 ? b2 : 0) - qe2); d |= b2 > 0 ? 0x40 : 0;
			} else {
				x[t]  = I8((a  >= 0 ? a  : 0) - qe);  d |= a  >= 0 ? 0x08 : 0;
				y[t]  = I8((b  >= 0 ? b  : 0) - qe);  d |= b  >= 0 ? 0x10 : 0;
				x2[t] = I8((a2 >= 0 ? a2 : 0) - qe2); d |= a2 >= 0 ? 0x20 : 0;
				y2[t] = I8((b2 >= 0 ? b2 : 0) - qe2); d |= b2 >= 0 ? 0x40 : 0;
			}
			if (with_cigar) p[(size_t)r * n_col + (t - st)] = d;
			if (stats) {
				int mx = abs(u[t]), c;
				if ((c = abs(v[t])) > mx) mx = c;
				if ((c = abs(x[t])) > mx) mx = c;
				if ((c = abs(y[t])) > mx) mx = c;
				if ((c = abs(x2[t])) > mx) mx = c;
				if ((c = abs(y2[t])) > mx) mx = c;
				if ((c = abs(a)) > mx) mx = c;
				if ((c = abs(b)) > mx) mx = c;
				if ((c = abs(a2)) > mx) mx = c;
				if ((c = abs(b2)) > mx) mx = c;
				if (t < st0 || t > en0) { if (mx > st_out) st_out = mx; }
				else if (mx > st_in) st_in = mx;
			}
			x1 = xo, v1 = vo, x21 = x2o;
		}
		if (!approx) {                                                                   /* exact max :315-358 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t; /* NB: stores the chunk base t */
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v[t];
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {                                                                         /* approximate :359-375 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t];
				} else {
					++last_H0_t, H0 += u[last_H0_t];
				}
			} else H0 = v[0] - qe, last_H0_t = 0;
			if (approx_drop && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	if (with_cigar) {                                                                    /* :381-391 */
		if (!ez->zdropped && !extz_only)
			ez->n_cigar = backtrack(rev_cigar, p, off, off_end, n_col, tlen - 1, qlen - 1, cigar_out);
		else if (!ez->zdropped && extz_only && ez->mqe + end_bonus > ez->max) {
			ez->reach_end = 1;
			ez->n_cigar = backtrack(rev_cigar, p, off, off_end, n_col, ez->mqe_t, qlen - 1, cigar_out);
		} else if (ez->max_t >= 0 && ez->max_q >= 0)
			ez->n_cigar = backtrack(rev_cigar, p, off, off_end, n_col, ez->max_t, ez->max_q, cigar_out);
		free(p); free(off);
	}
	if (stats) stats[0] = st_out, stats[1] = st_in;
	free(blk); free(H);
}

/* ------------------------------------------------------------------------------------------------
 * ksw_exts2_sse (src/ksw2_exts2_sse.c:18-407): the splice-aware variant — one gap class with extension (q, e), one long
 * deletion class without extension (q2: an intron) whose opening is paid at the donor site and whose closing adds the acceptor
 * signal; no band, no second insertion class. Lane-exact scalar emulation like wmo_ksw_extd2; backtrack = src/ksw2.h:119-151
 * with min_intron_len = long_thres (state 3 -> N). `junc` (optional, tlen bytes) = annotated junction bits (src/index.c:690-803).
 * Flags: 0x01 score only, 0x02 right-align gaps, 0x04 generic scores, 0x08 approximate max, 0x10 approximate drop,
 * 0x40 extension only, 0x80 reverse CIGAR, 0x100 / 0x200 forward / reverse transcript strand, 0x400 flanking base of the signal.
 * ---------------------------------------------------------------------------------------------- */
static int backtrack_intron(int is_rev, int min_intron_len, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0, uint32_t *cig)
{ /* src/ksw2.h:119-151 with is_rot = 1 */
	int n = 0, i = i0, j = j0, state = 0, k;
	while (i >= 0 && j >= 0) {
		int r = i + j, force = -1;
		uint32_t d;
		if (i < off[r]) force = 2;
		if (i > off_end[r]) force = 1;
		d = force < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = d & 7;
		else if (!(d >> (state + 2) & 1)) state = 0;
		if (state == 0) state = d & 7;
		if (force >= 0) state = force;
		if (state == 0) n = push_op(cig, n, 0, 1), --i, --j;
		else if (state == 1 || (state == 3 && min_intron_len <= 0)) n = push_op(cig, n, 2, 1), --i;
		else if (state == 3 && min_intron_len > 0) n = push_op(cig, n, 3, 1), --i;
		else n = push_op(cig, n, 1, 1), --j;
	}
	if (i >= 0) n = push_op(cig, n, min_intron_len > 0 && i >= min_intron_len ? 3 : 2, i + 1);
	if (j >= 0) n = push_op(cig, n, 1, j + 1);
	if (!is_rev)
		for (k = 0; k < n >> 1; ++k) { uint32_t x = cig[k]; cig[k] = cig[n - 1 - k]; cig[n - 1 - k] = x; }
	return n;
}

/* donor[t] / acceptor[t] (:110-166): the cost of opening an intron right after target base t / closing one at base t */
void wmo_exts2_signals(int tlen, const uint8_t *target, int noncan, int junc_bonus, int flag, const uint8_t *junc, int8_t *donor, int8_t *acceptor)
{
	const int fwd = !!(flag & 0x100), rev = !!(flag & 0x200), semi = (flag & 0x400) ? -noncan / 2 : 0;
	int t;
	for (t = 0; t < tlen; ++t) donor[t] = acceptor[t] = I8(-noncan);
	if (!(flag & (0x100 | 0x200))) { for (t = 0; t < tlen; ++t) donor[t] = acceptor[t] = 0; return; }   /* (arrays stay zeroed: kcalloc) */
	if (!(flag & 0x80)) {                         /* the target is read left to right: GT[AG] ... [CT]AG (forward), CT[AG] ... [CT]AC (reverse) */
		for (t = 0; t < tlen - 4; ++t) {
			int can = 0;
			if (fwd && target[t + 1] == 2 && target[t + 2] == 3) can = 1;
			if (rev && target[t + 1] == 1 && target[t + 2] == 3) can = 1;
			if (can && (target[t + 3] == 0 || target[t + 3] == 2)) can = 2;
			if (can) donor[t] = can == 2 ? 0 : I8(semi);
		}
		if (junc) for (t = 0; t < tlen - 1; ++t) if ((fwd && (junc[t + 1] & 1)) || (rev && (junc[t + 1] & 8))) donor[t] = I8(donor[t] + junc_bonus);
		for (t = 2; t < tlen; ++t) {
			int can = 0;
			if (fwd && target[t - 1] == 0 && target[t] == 2) can = 1;
			if (rev && target[t - 1] == 0 && target[t] == 1) can = 1;
			if (can && (target[t - 2] == 1 || target[t - 2] == 3)) can = 2;
			if (can) acceptor[t] = can == 2 ? 0 : I8(semi);
		}
		if (junc) for (t = 0; t < tlen; ++t) if ((fwd && (junc[t] & 2)) || (rev && (junc[t] & 4))) acceptor[t] = I8(acceptor[t] + junc_bonus);
	} else {                                      /* left extension: the target arrives reversed, so do the signals */
		for (t = 0; t < tlen - 4; ++t) {
			int can = 0;
			if (fwd && target[t + 1] == 2 && target[t + 2] == 0) can = 1;
			if (rev && target[t + 1] == 1 && target[t + 2] == 0) can = 1;
			if (can && (target[t + 3] == 1 || target[t + 3] == 3)) can = 2;
			if (can) donor[t] = can == 2 ? 0 : I8(semi);
		}
		if (junc) for (t = 0; t < tlen - 1; ++t) if ((fwd && (junc[t + 1] & 2)) || (rev && (junc[t + 1] & 4))) donor[t] = I8(donor[t] + junc_bonus);
		for (t = 2; t < tlen; ++t) {
			int can = 0;
			if (fwd && target[t - 1] == 3 && target[t] == 2) can = 1;
			if (rev && target[t - 1] == 3 && target[t] == 1) can = 1;
			if (can && (target[t - 2] == 0 || target[t - 2] == 2)) can = 2;
			if (can) acceptor[t] = can == 2 ? 0 : I8(semi);
		}
		if (junc) for (t = 0; t < tlen; ++t) if ((fwd && (junc[t] & 1)) || (rev && (junc[t] & 8))) acceptor[t] = I8(acceptor[t] + junc_bonus);
	}
}

void wmo_ksw_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int q_, int e_, int q2_, int noncan_, int zdrop, int junc_bonus_, int flag, const uint8_t *junc,
                   wmo_ez_t *ez, uint32_t *cigar_out)
{
	const int8_t q = (int8_t)q_, e = (int8_t)e_, q2 = (int8_t)q2_, noncan = (int8_t)noncan_, junc_bonus = (int8_t)junc_bonus_;
	const int approx = !!(flag & 0x08), right = !!(flag & 0x02), extz_only = !!(flag & 0x40), rev_cigar = !!(flag & 0x80);
	const int generic = !!(flag & 0x04), with_cigar = !(flag & 0x01), approx_drop = !!(flag & 0x10);
	const int qe = q + e;
	int r, t, T, n_col, last_st = -1, last_en = -1, long_thres, long_diff, max_sc, min_sc;
	int8_t *blk, *u, *v, *x, *y, *x2, *donor, *acceptor, *s, sc_mch, sc_mis, sc_N;
	uint8_t *sf, *qr, *p = 0;
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int *off = 0, *off_end = 0;

	reset_ez(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0 || q2 <= q + e) return;                       /* :66 */
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? (int8_t)-e : mat[m * m - 1];                           /* :74 */
	T = (tlen + 15) / 16 * 16;
	n_col = (((qlen < tlen ? qlen : tlen) + 15) / 16 + 1) * 16;                         /* :78, in bytes */
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		if (mat[t] > max_sc) max_sc = mat[t];
		if (mat[t] < min_sc) min_sc = mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;                                                  /* :84 */
	long_thres = (q2 - q) / e - 1;                                                      /* :86-89 */
	if (q2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * e - (q2 - q);

	/* one block like the reference's (:91-96): u v x y x2 donor acceptor s | sf | qr — the score pass writes s in 16-byte chunks that start
	 * at st0 and may run past its end into sf, and reads sf / qr past their ends */
	blk = (int8_t*)calloc((size_t)9 * T + ((size_t)(qlen + 15) / 16 + 1) * 16 + 16, 1);
	u = blk, v = u + T, x = v + T, y = x + T, x2 = y + T, donor = x2 + T, acceptor = donor + T, s = acceptor + T;
	sf = (uint8_t*)(s + T), qr = sf + T;
	memset(u, -q - e, (size_t)T * 4);
	memset(x2, -q2, T);
	if (!approx) {
		H = (int32_t*)malloc((size_t)T * 4);
		for (t = 0; t < T; ++t) H[t] = WMO_NEG_INF;
	}
	if (with_cigar) {
		p = (uint8_t*)malloc((size_t)(qlen + tlen - 1) * n_col + 16);
		off = (int*)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);
	if (flag & (0x100 | 0x200)) {                                                        /* :109-166 (whole 16-lane groups get -noncan) */
		memset(donor, -noncan, T); memset(acceptor, -noncan, T);
		{
			int8_t *d2 = (int8_t*)malloc(tlen), *a2 = (int8_t*)malloc(tlen);
			wmo_exts2_signals(tlen, target, noncan, junc_bonus, flag, junc, d2, a2);
			memcpy(donor, d2, tlen); memcpy(acceptor, a2, tlen);
			free(d2); free(a2);
		}
	}

	for (r = 0; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1;
		const uint8_t *qrr = qr + (qlen - 1 - r);
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) {                                                                    /* :178-186 */
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			else x1 = I8(-q - e), x21 = I8(-q2), v1 = I8(-q - e);
		} else {
			x1 = I8(-q - e), x21 = I8(-q2);
			v1 = r == 0 ? I8(-q - e) : r < long_thres ? I8(-e) : r == long_thres ? I8(long_diff) : 0;
		}
		if (en >= r) {                                                                   /* :187-190 */
			y[r] = I8(-q - e);
			u[r] = r == 0 ? I8(-q - e) : r < long_thres ? I8(-e) : r == long_thres ? I8(long_diff) : 0;
		}
		if (!generic) {                                                                  /* :192-209, 16-byte chunks from st0 */
			for (t = st0; t <= en0; t += 16) {
				int8_t tmp16[16];
				int i;
				for (i = 0; i < 16; ++i) {
					uint8_t a = sf[t + i], b = qrr[t + i];
					tmp16[i] = (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) ? sc_N : a == b ? sc_mch : sc_mis;
				}
				memcpy(s + t, tmp16, 16);
			}
		} else {
			for (t = st0; t <= en0; ++t) s[t] = mat[sf[t] * m + qrr[t]];
		}
		if (with_cigar) off[r] = st, off_end[r] = en;
		for (t = st; t <= en; ++t) {                                                     /* :211-333 lane by lane */
			int8_t z = s[t], xo = x[t], vo = v[t], x2o = x2[t], ut = u[t];
			int8_t a = I8(x1 + v1), b = I8(y[t] + ut), a2 = I8(x21 + v1), a2a = I8(a2 + acceptor[t]), tmp, dn;
			uint8_t d;
			if (!right) {
				d = a > z ? 1 : 0;   z = z > a ? z : a;
				d = b > z ? 2 : d;   z = z > b ? z : b;
				d = a2a > z ? 3 : d; z = z > a2a ? z : a2a;
			} else {
				d = z > a ? 0 : 1;   z = z > a ? z : a;
				d = z > b ? d : 2;   z = z > b ? z : b;
				d = z > a2a ? d : 3; z = z > a2a ? z : a2a;
			}
			u[t] = I8(z - v1); v[t] = I8(z - ut);                                       /* (no clamp to the match score here) */
			tmp = I8(z - q); a = I8(a - tmp); b = I8(b - tmp);
			a2 = I8(a2 - I8(z - q2));
			dn = donor[t];
			if (!right) {
				x[t] = I8((a > 0 ? a : 0) - qe);   d |= a > 0 ? 0x08 : 0;
				y[t] = I8((b > 0 ? b : 0) - qe);   d |= b > 0 ? 0x10 : 0;
				x2[t] = I8((a2 > dn ? a2 : dn) - q2); d |= a2 > dn ? 0x20 : 0;
			} else {
				x[t] = I8((a >= 0 ? a : 0) - qe);  d |= a >= 0 ? 0x08 : 0;
				y[t] = I8((b >= 0 ? b : 0) - qe);  d |= b >= 0 ? 0x10 : 0;
				x2[t] = I8((dn > a2 ? dn : a2) - q2); d |= dn > a2 ? 0 : 0x20;
			}
			if (with_cigar) p[(size_t)r * n_col + (t - st)] = d;
			x1 = xo, v1 = vo, x21 = x2o;
		}
		if (!approx) {                                                                   /* exact max :334-381 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v[t];
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, 0)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {                                                                         /* approximate :382-398 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t];
				} else {
					++last_H0_t, H0 += u[last_H0_t];
				}
			} else H0 = v[0] - qe, last_H0_t = 0;
			if (approx_drop && apply_zdrop(ez, H0, r, last_H0_t, zdrop, 0)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	if (with_cigar) {                                                                    /* :400-406 */
		if (!ez->zdropped && !extz_only)
			ez->n_cigar = backtrack_intron(rev_cigar, long_thres, p, off, off_end, n_col, tlen - 1, qlen - 1, cigar_out);
		else if (ez->max_t >= 0 && ez->max_q >= 0)
			ez->n_cigar = backtrack_intron(rev_cigar, long_thres, p, off, off_end, n_col, ez->max_t, ez->max_q, cigar_out);
		free(p); free(off);
	}
	free(blk); free(H);
}

/* ------------------------------------------------------------------------------------------------
 * ksw_ll_qinit + ksw_ll_i16 (src/ksw2_ll_sse.c:32-147): striped (Farrar) local SW, int16 lanes,
 * signed saturating add / unsigned saturating subtract; emulated lane by lane (8 lanes per vector).
 * ---------------------------------------------------------------------------------------------- */
static int16_t adds16(int a, int b) { int s = a + b; return (int16_t)(s > 32767 ? 32767 : s < -32768 ? -32768 : s); }
static int16_t subsu16(int16_t a, int16_t b) { uint16_t ua = (uint16_t)a, ub = (uint16_t)b; return (int16_t)(ua > ub ? ua - ub : 0); }
static int16_t max16(int16_t a, int16_t b) { return a > b ? a : b; }

int wmo_ksw_ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int gapo, int gape, int *qe, int *te)
{
	const int slen = (qlen + 7) / 8, nl = slen * 8;
	int16_t *prof = (int16_t*)malloc((size_t)m * nl * 2);
	int16_t *H0 = (int16_t*)calloc(nl, 2), *H1 = (int16_t*)calloc(nl, 2), *E = (int16_t*)calloc(nl, 2), *Hm = (int16_t*)calloc(nl, 2);
	const int16_t goe = (int16_t)(gapo + gape), ge = (int16_t)gape;
	int a, i, j, k, l, gmax = 0;
	for (a = 0; a < m; ++a)                        /* profile: vector j lane l holds query position j + l*slen */
		for (j = 0; j < slen; ++j)
			for (l = 0; l < 8; ++l) {
				int pos = j + l * slen;
				prof[((size_t)a * slen + j) * 8 + l] = pos >= qlen ? 0 : mat[a * m + query[pos]];
			}
	*qe = *te = -1;
	for (i = 0; i < tlen; ++i) {
		int16_t h[8], f[8] = {0}, mx[8] = {0}, e[8];
		const int16_t *S = prof + (size_t)target[i] * slen * 8;
		int imax, done = 0;
		int16_t *tmp;
		h[0] = 0;
		for (l = 1; l < 8; ++l) h[l] = H0[(slen - 1) * 8 + l - 1];       /* shift left by one lane */
		for (j = 0; j < slen; ++j) {
			for (l = 0; l < 8; ++l) {
				int16_t hh = adds16(h[l], S[j * 8 + l]);
				e[l] = E[j * 8 + l];
				hh = max16(hh, e[l]); hh = max16(hh, f[l]);
				mx[l] = max16(mx[l], hh);
				H1[j * 8 + l] = hh;
				hh = subsu16(hh, goe);
				e[l] = subsu16(e[l], ge); e[l] = max16(e[l], hh);
				E[j * 8 + l] = e[l];
				f[l] = subsu16(f[l], ge); f[l] = max16(f[l], hh);
				h[l] = H0[j * 8 + l];
			}
		}
		for (k = 0; k < 8 && !done; ++k) {                                /* lazy-F loop :117-128 */
			for (l = 7; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (j = 0; j < slen; ++j) {
				int any = 0;
				for (l = 0; l < 8; ++l) {
					int16_t hh = max16(H1[j * 8 + l], f[l]);
					H1[j * 8 + l] = hh;
					hh = subsu16(hh, goe);
					f[l] = subsu16(f[l], ge);
					if (f[l] > hh) any = 1;
				}
				if (!any) { done = 1; break; }
			}
		}
		for (l = 0, imax = mx[0]; l < 8; ++l) if (mx[l] > imax) imax = mx[l];
		if (imax >= gmax) { gmax = imax; *te = i; memcpy(Hm, H1, (size_t)nl * 2); }
		tmp = H1; H1 = H0; H0 = tmp;
	}
	for (i = 0; i < nl; ++i)
		if ((int)(uint16_t)Hm[i] == gmax) *qe = i / 8 + i % 8 * slen;
	free(prof); free(H0); free(H1); free(E); free(Hm);
	return gmax;
}
