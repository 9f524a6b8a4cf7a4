// ksw_plan.h — host-side planning of one ksw job: traceback pitch, kernel class, cell counts.
// Pure C++ (no HIP); shared by the product shim (ksw_host.cpp) and the test emulator driver.
#pragma once
#include <stdint.h>
#include "wm_internal.h"

enum {            // klass = B-index*4 + CLIP*2 + HASN  for the register kernels; 12 = multi-wave LDS kernel; 13 = generic (global scratch)
	WM_KSW_B4 = 0, WM_KSW_B8 = 4, WM_KSW_B16 = 8, WM_KSW_BLOCK = 12, WM_KSW_GENERIC = 13, WM_KSW_NCLASS = 14
};
// geometry of the block kernel (ksw_dp_block<NWV, K>): NWV waves x K tiles x 64 lanes per row, LDS window of WN lanes
enum { WM_KSW_BLK_NWV = 8, WM_KSW_BLK_K = 6, WM_KSW_BLK_WN = 4096 };

// row pitch of the traceback in bytes: 16 * n_col_ of src/ksw2_extd2_sse.c:84-86
static inline int wm_ksw_ncol(int qlen, int tlen, int w)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int n = qlen < tlen ? qlen : tlen;
	n = n < w + 1 ? n : w + 1;
	return ((n + 15) / 16 + 1) * 16;
}

// DP cells inside the band (O(1) when the band never clips, else one pass over the target) and an estimate of the
// 16-aligned hull cells = traceback bytes written
static inline uint64_t wm_ksw_cells(int qlen, int tlen, int w, uint64_t *band_cells)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	uint64_t band;
	if (w >= qlen && w >= tlen) band = (uint64_t)qlen * tlen;
	else {
		band = 0;
		for (int i = 0; i < tlen; ++i) {             // row i of the target: query columns j with |i - j| within the band
			int lo = i - w, hi = i + w - 1;
			if (lo < 0) lo = 0;
			if (hi > qlen - 1) hi = qlen - 1;
			if (hi >= lo) band += (uint64_t)(hi - lo + 1);
		}
	}
	if (band_cells) *band_cells = band;
	return band + (uint64_t)16 * (uint64_t)(qlen + tlen - 1);
}

static inline int wm_ksw_has_n(const uint8_t *s, int n)
{
	for (int i = 0; i < n; ++i) if (s[i] >= 4) return 1;
	return 0;
}

// CLIP = 0 only when the band provably never limits a row (then out-of-band lanes never feed band cells)
static inline int wm_ksw_classify(int qlen, int tlen, int w, int has_n, int *n_col_out)
{
	const int n_col = wm_ksw_ncol(qlen, tlen, w);
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int clip = !(w >= qlen && w >= tlen);
	int k;
	if (n_col <= 64 * 4 - 16) k = WM_KSW_B4;
	else if (n_col <= 64 * 8 - 16) k = WM_KSW_B8;
	else if (n_col <= 64 * 16 - 16) k = WM_KSW_B16;
	else if (n_col + 16 <= 64 * WM_KSW_BLK_NWV * WM_KSW_BLK_K) { *n_col_out = n_col; return WM_KSW_BLOCK; }
	else { *n_col_out = n_col; return WM_KSW_GENERIC; }
	*n_col_out = n_col;
	return k + clip * 2 + (has_n ? 1 : 0);
}

// parameter ranges the kernels support (everything the reference's mm_check_opt admits, src/options.c:166-176)
static inline int wm_ksw_score_ok(const wm_ksw_score_t *sc)
{
	if (sc->match <= 0 || sc->mismatch >= 0 || sc->sc_ambi > 0) return 0;
	if (sc->q <= 0 || sc->e <= 0 || sc->q2 <= 0 || sc->e2 < 0) return 0;
	if ((sc->q + sc->e) + (sc->q2 + sc->e2) > 127) return 0;
	return 1;
}
