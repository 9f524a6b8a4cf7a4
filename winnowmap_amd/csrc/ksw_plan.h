// ksw_plan.h — host-side planning of one ksw job: traceback pitch, kernel class, cell counts.
// Pure C++ (no HIP); shared by the product shim (ksw_host.cpp) and the test emulator driver.
#pragma once
#include <stdint.h>
#include "wm_internal.h"

enum {            // register classes (ksw_dp_packed<BP,...>): klass = window*8 + EXACT*4 + CLIP*2 + HASN with window 0/1/2 = 4/8/16 chunk pairs (hull <= 496 /
	                  // 1008 / 2032 lanes); 24, 25 = multi-wave LDS kernels; 26 = multi-wave kernel with its state in global scratch; 27 = single-wave generic kernel
	WM_KSW_P4 = 0, WM_KSW_P8 = 8, WM_KSW_P16 = 16, WM_KSW_BLOCK = 24, WM_KSW_BLOCK2 = 25, WM_KSW_BLOCK3 = 26, WM_KSW_GENERIC = 27,
	// stripe-pipelined multi-wave kernels (ksw_stripe_kernel.h): klass = WM_KSW_STRIPE + geometry * 4 + CLIP * 2 + HASN (a job with an N runs on the
	// CLIP instantiation: 1 is unused); geometry 0..3 = <BP, NWV> of <2,4> <2,8> <4,8> <8,8>: traceback pitch n_col up to 768 / 1792 / 3584 / 7168;
	// geometry 4, 5 = <1,16> <2,16> (up to 1920 / 3840 lanes): half the pairs per wavefront for the same hull — the cells of a row are the largest
	// share of a stripe wavefront's row time (profiles/r04q_stripe_timing.txt). Opt-in (wide16 of wm_ksw_route) until a bench A/B has judged them.
	WM_KSW_STRIPE = 28,
	// chained-workgroup kernels (ksw_chain_kernel.h, round 6): klass = WM_KSW_CHAIN + geometry * 4 + CLIP * 2 + HASN, geometry 0 / 1 = 2 / 4 register pairs per
	// wavefront (256 / 512-lane stripes); ANY hull width and length — every wavefront of a job is a workgroup of its own
	WM_KSW_CHAIN = 52, WM_KSW_NCLASS = 60
};
static const int wm_ksw_chain_bp[2] = { 2, 4 };
// Which jobs run on the chained-workgroup kernels. mode bit 0: everything the stripe classes and the old wide-hull kernels (BLOCK2 = ksw_dp_pmulti<8,8>,
// BLOCK3 = ksw_dp_block, GENERIC) serve — hulls wider than one wavefront's window, and the long jobs of the 4- / 8-pair register classes that wm_ksw_route
// sends to a stripe class; bit 1: exact extensions (z-drop, exact maximum) of the 8-pair register classes from `min_rows_exact` rows on — one wavefront per
// alignment is one dependent chain of qlen + tlen rows of up to 8 pairs, and a launch lasts as long as its longest chain; bit 2: every job (tests).
// geom: wm_ksw_chain_bp index.
static inline int wm_ksw_route_chain(int klass, int qlen, int tlen, int w, int has_n, int flag, int mode, int min_rows_exact, int geom)
{
	if (klass < 0 || !mode) return klass;
	bool go = false;
	if (mode & 1) go = klass >= WM_KSW_P16;                       // (16-pair register classes, BLOCK.., GENERIC, every stripe class)
	if (!go && (mode & 2) && klass >= WM_KSW_P8 && klass < WM_KSW_P16 && (klass & 4) && qlen + tlen - 1 >= min_rows_exact) go = true;
	if (mode & 4) go = true;                                      // (tests: every job)
	if (!go || klass >= WM_KSW_CHAIN) return klass;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int clip = !(w >= qlen && w >= tlen) || has_n;
	(void)flag;
	return WM_KSW_CHAIN + geom * 4 + clip * 2 + (has_n ? 1 : 0);
}
static const int wm_ksw_stripe_max_ncol[6] = { 3 * 256, 7 * 256, 7 * 512, 7 * 1024, 15 * 128, 15 * 256 };      // ksw_stripe_lds<BP, NWV>::MAX_NCOL
// Which jobs leave their register / barrier class for a stripe class: every hull wider than the one-wave window of 8 pairs (the former 16-pair and
// BLOCK / BLOCK2 classes: one DP row cost 2.5 us there whatever the hull), and LONG alignments of the 4- and 8-pair classes — one alignment on
// one wavefront is one dependent chain of qlen + tlen rows, and the longest chain of a launch is the launch's duration (min_rows4 / min_rows8:
// rows from which the chain is split over four wavefronts; 0 = never).
static inline int wm_ksw_route(int klass, int n_col, int qlen, int tlen, int w, int has_n, int min_rows4, int min_rows8, int wide16 = 0)
{
	if (klass < 0 || klass >= WM_KSW_BLOCK3) return klass;
	const int n_rows = qlen + tlen - 1;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int geom;
	if (klass < WM_KSW_P8) { if (!min_rows4 || n_rows < min_rows4) return klass; geom = 0; }
	else if (klass < WM_KSW_P16) { if (!min_rows8 || n_rows < min_rows8) return klass; geom = n_col <= wm_ksw_stripe_max_ncol[0] ? 0 : 1; }
	else geom = n_col <= wm_ksw_stripe_max_ncol[1] ? 1 : n_col <= wm_ksw_stripe_max_ncol[2] ? 2 : n_col <= wm_ksw_stripe_max_ncol[3] ? 3 : -1;
	// sixteen wavefronts (opt-in, bits of wide16; isolated probe, profiles/r04q_stripe_timing.txt): bit 0 = <2,16> for the hulls <4,8> serves (1793..3584
	// lanes, and up to 3840): 2.03 vs 2.54 us per row exact, 1.56 vs 1.93 approximate; bit 1 = <1,16> for the long jobs of the one-wavefront classes
	// (hull <= 1008 lanes): 1.88 vs 2.04 / 1.29 vs 1.45. Hulls of 1009..1792 lanes stay on <2,8>: one pair per wavefront was slower there (3.47 vs 2.38).
	if ((wide16 & 1) && geom >= 0 && n_col > wm_ksw_stripe_max_ncol[1] && n_col <= wm_ksw_stripe_max_ncol[5]) geom = 5;
	if ((wide16 & 2) && geom >= 0 && klass < WM_KSW_P16) geom = 4;
	if (geom < 0) return klass;                              // (7169..8176 lanes: stays on ksw_dp_pmulti<8,8>)
	const int clip = !(w >= qlen && w >= tlen) || has_n;
	return WM_KSW_STRIPE + geom * 4 + clip * 2 + (has_n ? 1 : 0);
}
// geometry of the block kernels (ksw_dp_block<NWV, K>): NWV waves x K tiles x 64 lanes per row, LDS window of WN lanes
enum { WM_KSW_MULTI_B = 8, WM_KSW_MULTI_NWV = 8,                          // BLOCK: hulls up to 64 * 8 * 8 - 16 = 4080 lanes (ksw_dp_pmulti<4, 8>), BLOCK2: up to 8176 (<8, 8>)
       WM_KSW_BLK_NWV = 16, WM_KSW_BLK_K = 3, WM_KSW_BLK_WN = 4096,        // (LDS-state kernel at the same size: kept for tests)
       WM_KSW_BLK2_K = 7, WM_KSW_BLK2_WN = 8192 };                         // hulls up to 7168 lanes (unbanded fills across structural variants)
// bytes of LDS left for the staged sequences next to the state window (160 KB per CU, one block per CU for these classes)
enum { WM_KSW_BLK_SEQ_LDS = 96 * 1024, WM_KSW_BLK2_SEQ_LDS = 48 * 1024, WM_KSW_BLK3_SEQ_LDS = 128 * 1024 };
enum { WM_KSW_BLK_MAXC = 16, WM_KSW_BLK_PUB = 2 * 16 + 8 + 3 * 16 };      // chunks per row at most; ints of the publish area
// lanes of the state window of a BLOCK3 job (power of two covering every lane index the kernel can touch)
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t wm_ksw_blk3_wn(int tlen) { uint64_t n = 1024; while (n < (uint64_t)tlen + 80) n <<= 1; return n; }

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

// CLIP = 0 only when the band provably never limits a row (then out-of-band lanes never feed band cells);
// EXACT = the exact row maximum is wanted (no KSW_EZ_APPROX_MAX, src/ksw2.h:11)
static inline int wm_ksw_classify(int qlen, int tlen, int w, int has_n, int flag, int *n_col_out)
{
	const int n_col = wm_ksw_ncol(qlen, tlen, w);
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int clip = !(w >= qlen && w >= tlen), exact = !(flag & 0x08);
	*n_col_out = n_col;
	int k;
	if (n_col <= 128 * 4 - 16) k = WM_KSW_P4;
	else if (n_col <= 128 * 8 - 16) k = WM_KSW_P8;
	else if (n_col <= 128 * 16 - 16) k = WM_KSW_P16;
	else if (n_col + 16 <= 64 * WM_KSW_MULTI_NWV * WM_KSW_MULTI_B) return WM_KSW_BLOCK;
	else if (n_col + 16 <= 64 * 2 * WM_KSW_MULTI_NWV * WM_KSW_MULTI_B) return WM_KSW_BLOCK2;     // ksw_dp_pmulti<8,8>: 8192 lanes
	else if (n_col + 16 <= 64 * WM_KSW_BLK_NWV * WM_KSW_BLK2_K * WM_KSW_BLK_MAXC) return WM_KSW_BLOCK3;
	else return WM_KSW_GENERIC;
	return k + exact * 4 + clip * 2 + (has_n ? 1 : 0);
}

// parameter ranges the kernels support (everything the reference's mm_check_opt admits, src/options.c:166-176)
static inline int wm_ksw_score_ok(const wm_ksw_score_t *sc)
{
	if (sc->match <= 0 || sc->mismatch >= 0 || sc->sc_ambi > 0) return 0;
	if (sc->q <= 0 || sc->e <= 0 || sc->q2 <= 0 || sc->e2 < 0) return 0;
	if ((sc->q + sc->e) + (sc->q2 + sc->e2) > 127) return 0;
	return 1;
}
