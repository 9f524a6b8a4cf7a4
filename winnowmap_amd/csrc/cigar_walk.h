// cigar_walk.h — the walks over a finished alignment's CIGAR that the reference runs on the host after every alignment, written ONCE for
// the host and the device: one function body, compiled by hipcc into the kernels that run them over the CIGAR pool a ksw call already holds
// in HBM (wm_gpu.hip: ksw_zdwalk_kernel runs wm_zdrop_walk; wm_extra_walk has no kernel — mm_update_extra needs the region's final CIGAR, which exists on the host only) and by g++ into the host mapper (host/wm_align.cpp: the same results for device
// operations that do not supply them, e.g. the oracle-backed ones of tests/host_harness). Sequences are 0..4 codes, one byte per base; CIGAR
// ops are BAM-encoded (len << 4 | op), in output order.
//   wm_zdrop_walk   = the scan of mm_test_zdrop + update_max_zdrop (src/align.c:32-66): the largest z-drop along the alignment and where it spans
//   wm_extra_walk   = the scan of mm_update_extra (src/align.c:240-286): dp_max, mlen, blen, n_ambi of a region's final CIGAR
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define WM_HD __host__ __device__
#else
#define WM_HD
#endif

#include "../../include/wm_gpu.h"                                           // wm_zd_t: max_zdrop and pos[0][0], pos[0][1] (target), pos[1][0], pos[1][1] (query) of src/align.c:51
typedef struct { int32_t dp_max, mlen, blen, n_ambi, qoff, toff; } wm_extra_t;

// length of the run of equal unambiguous bases at the start of t[0, n) / q[0, n) (eight bases per step)
WM_HD static inline uint32_t wm_match_run(const uint8_t *t, const uint8_t *q, uint32_t n)
{
	uint32_t l = 0;
	while (l + 8 <= n) {
		uint64_t a, b;
		memcpy(&a, t + l, 8); memcpy(&b, q + l, 8);
		const uint64_t bad = (a ^ b) | (a & 0xFCFCFCFCFCFCFCFCULL);
		if (bad) return l + (uint32_t)(__builtin_ctzll(bad) >> 3);
		l += 8;
	}
	while (l < n && t[l] == q[l] && t[l] < 4) ++l;
	return l;
}

// mat: the 5 x 5 scoring matrix of ksw_gen_simple_mat given by its three values (match = mat[0] > 0, mismatch = mat[1], ambi = mat[24]);
// q, e: the gap open / extension mm_test_zdrop is called with (opt->q, opt->e — not the cheaper piece ksw_extd2 swaps to the front)
WM_HD static inline void wm_zdrop_walk(const uint8_t *qseq, const uint8_t *tseq, const uint32_t *cigar, int n_cigar, int match, int mismatch, int ambi,
                                       int q, int e, wm_zd_t *out)
{
	int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
	int32_t p00 = -1, p01 = -1, p10 = -1, p11 = -1;
#define WM_ZD_UPD(sc_, ii_, jj_) do { \
		const int32_t sc__ = (sc_), ii__ = (ii_), jj__ = (jj_); \
		if (sc__ < max) { \
			const int li = ii__ - max_i, lj = jj__ - max_j, diff = li > lj ? li - lj : lj - li; \
			const int z = max - sc__ - diff * e; \
			if (z > max_zdrop) { max_zdrop = z; p00 = max_i; p01 = ii__; p10 = max_j; p11 = jj__; } \
		} else { max = sc__; max_i = ii__; max_j = jj__; } \
	} while (0)
	for (int k = 0; k < n_cigar; ++k) {
		const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			// (a run of matches only raises the score: inside a dip the drop z = max - score - diff * e shrinks, so nothing can be recorded
			// there, and above the old maximum the last base of the run is where max_i / max_j end up — one update per run is exact)
			for (uint32_t l = 0; l < len;) {
				const uint32_t run = match > 0 ? wm_match_run(tseq + i + l, qseq + j + l, len - l) : 0;
				if (run) {
					score += (int32_t)run * match; l += run;
					if (score >= max) { max = score; max_i = i + (int)l - 1; max_j = j + (int)l - 1; }
					continue;
				}
				const int ct = tseq[i + l], cq = qseq[j + l];
				score += (ct > 3 || cq > 3) ? ambi : ct == cq ? match : mismatch;
				WM_ZD_UPD(score, i + (int)l, j + (int)l);
				++l;
			}
			i += len; j += len;
		} else if (op == 1 || op == 2 || op == 3) {
			score -= q + e * (int32_t)len;
			if (op == 1) j += len; else i += len;
			WM_ZD_UPD(score, i, j);
		}
	}
#undef WM_ZD_UPD
	out->max_zdrop = max_zdrop; out->t0 = p00; out->t1 = p01; out->q0 = p10; out->q1 = p11;
}

// the scan of mm_update_extra over a region's final CIGAR (after mm_fix_cigar); n_ambi is what THIS walk adds (the caller accumulates)
WM_HD static inline void wm_extra_walk(const uint8_t *qseq, const uint8_t *tseq, const uint32_t *cigar, int n_cigar, int match, int mismatch, int ambi,
                                       int q, int e, wm_extra_t *out)
{
	int32_t s = 0, max = 0, toff = 0, qoff = 0, blen = 0, mlen = 0, n_ambi_all = 0;
	for (int k = 0; k < n_cigar; ++k) {
		const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			for (uint32_t l = 0; l < len;) {
				const uint32_t run = match > 0 ? wm_match_run(tseq + toff + l, qseq + qoff + l, len - l) : 0;
				if (run) { s += (int32_t)run * match; max = max > s ? max : s; l += run; continue; }   // (s only grows along the run)
				const int cq = qseq[qoff + l], ct = tseq[toff + l];
				if (ct > 3 || cq > 3) { ++n_ambi; s += ambi; }
				else if (ct != cq) { ++n_diff; s += mismatch; }
				else s += match;
				if (s < 0) s = 0; else max = max > s ? max : s;
				++l;
			}
			blen += (int32_t)len - n_ambi; mlen += (int32_t)len - (n_ambi + n_diff); n_ambi_all += n_ambi;
			toff += len; qoff += len;
		} else if (op == 1) {
			int n_ambi = 0;
			for (uint32_t l = 0; l < len; ++l) if (qseq[qoff + l] > 3) ++n_ambi;
			blen += (int32_t)len - n_ambi; n_ambi_all += n_ambi;
			s -= q + e * (int32_t)len;
			if (s < 0) s = 0;
			qoff += len;
		} else if (op == 2) {
			int n_ambi = 0;
			for (uint32_t l = 0; l < len; ++l) if (tseq[toff + l] > 3) ++n_ambi;
			blen += (int32_t)len - n_ambi; n_ambi_all += n_ambi;
			s -= q + e * (int32_t)len;
			if (s < 0) s = 0;
			toff += len;
		} else if (op == 3) toff += len;
	}
	out->dp_max = max; out->mlen = mlen; out->blen = blen; out->n_ambi = n_ambi_all; out->qoff = qoff; out->toff = toff;
}
