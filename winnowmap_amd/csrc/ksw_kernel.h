// ksw_kernel.h — wave64 register-resident anti-diagonal DP for ksw_extd2_sse (src/ksw2_extd2_sse.c:26-393).
//
// One wavefront owns one alignment. The reference walks anti-diagonals r = i + j and keeps per-target-lane
// int8 arrays u,v,x,y,x2,y2 (and the score row s) that persist from row to row; only the 16-aligned hull
// [st,en] of the band is recomputed each row and everything else keeps its last value ("stale lanes", which
// DO feed back when the band is clipped by w — SURVEY.md §7). This kernel reproduces that machine exactly:
//
//   * a sliding window of 64*B consecutive lanes starting at `base` (= the hull start st, a multiple of 16)
//     lives in VGPRs: thread j holds lanes base+B*j .. base+B*j+B-1. No LDS, no per-row barrier.
//   * neighbour values (lane t-1 of the previous row) come from the register of the same thread or, for the
//     first lane of a thread, from one DPP/bpermute shift (simt::shr1).
//   * when the hull start advances by 16 the whole window is re-based with one cross-lane shift per register.
//   * int8 WRAPPING arithmetic is exact and free: every value is kept in the TOP BYTE of a 32-bit register
//     (v << 24), so 32-bit add/sub wrap exactly like _mm_add_epi8/_mm_sub_epi8 and signed compares agree. The
//     low bits carry a 3-bit tie-break tag so that the five-way "which state wins" selection of the reference
//     (left- vs right-aligned gaps, src/ksw2_extd2_sse.c:227-234 vs :274-281) is two v_max3_i32.
//   * traceback: 1 byte per cell, row r at tb + r*n_col, column t - st (the reference's layout, ksw2.h:128);
//     B=4 → one coalesced 256-byte store per row per wave. Byte = code<<4 | ext_a<<3 | ext_b<<2 | ext_a2<<1 |
//     ext_b2 with code = winning tag (decoded by ksw_backtrack_thread below).
//
// Template: B lanes per thread (4/8/16 → hull up to 64*B-16 lanes), CLIP = the band can be limited by w (then
// stale score lanes and out-of-band lanes are emulated exactly; otherwise they provably never feed a band cell
// and are computed loosely), HASN = either sequence contains code 4.
//
// Roofline note: ~34 VALU lane-ops per cell, 1 B of HBM written per cell.
#pragma once
// NB: the translation unit includes its simt.h first (device: csrc/simt.h; tests: tests/simt_emu/simt.h)
#ifndef WM_DEV
#error "include simt.h before ksw_kernel.h"
#endif
#include "wm_internal.h"

namespace wmk {
using namespace simt;

#define KSW_NEG_INF (-0x40000000)
#define KSW_BT_WATCHDOG (-3)          // wm_ksw_dres_t::bt_i of a job whose stripe kernel gave up waiting (ksw_stripe_kernel.h: WM_STRIPE_SPIN_BUDGET)
#define KSW_F_RIGHT 0x02
#define KSW_F_APPROX_MAX 0x08
#define KSW_F_EXTZ_ONLY 0x40
#define KSW_F_REV_CIGAR 0x80

// value (int8 semantics) -> top-byte representation
WM_DEV int tb8(int v) { return (int)((unsigned)v << 24); }

// read lane t (uniform) of a per-thread register array: thread (t-base)/B, register (t-base)%B
template <int B> WM_DEV int get_lane(const V<int> (&a)[B], int base, int t)
{
	const int o = t - base, jr = o / B, ir = o % B;
	WM_EMU_ASSERT(o >= 0 && o < 64 * B);
	int r = 0;
#pragma unroll
	for (int i = 0; i < B; ++i)
		if (ir == i) r = readlane(a[i], jr);
	return r;
}

// ------------------------------------------------------------------------------------------------------
// Generic kernel for band hulls wider than the register window (> 1008 lanes; long gaps of stage 2, LONG_JOIN
// segments of asm20): same machine, but the per-lane state lives in memory as int8 arrays indexed by the target
// lane — `mem` points to 7*T bytes (u v x y x2 y2 s), `Hm` to T ints (exact mode only) — either LDS (T*7 fits) or a
// global scratch slab. One wave sweeps the hull 64 lanes at a time; lane t only ever writes its own entries and gets
// lane t-1's previous-row values from the neighbouring thread's registers, so there is no intra-row hazard. COH =
// the arrays are in global memory and must bypass the per-CU L1 (another lane wrote them in the previous row).
// ------------------------------------------------------------------------------------------------------
template <bool COH> WM_DEV V<int> ld8(signed char *p, V<int> i) { return COH ? cld8(p, i) : cast<int>(gld(p, i)); }
template <bool COH> WM_DEV void st8(signed char *p, V<int> i, V<int> v) { if (COH) cst8(p, i, v); else gst(p, i, cast<signed char>(v)); }
template <bool COH> WM_DEV int ld8s(signed char *p, int i) { return COH ? cld8(p, (long long)i) : (int)gld(p, (long long)i); }
template <bool COH> WM_DEV V<int> ld32(int *p, V<int> i) { return COH ? cld(p, i) : gld(p, i); }
template <bool COH> WM_DEV void st32(int *p, V<int> i, V<int> v) { if (COH) cst(p, i, v); else gst(p, i, v); }
template <bool COH> WM_DEV int ld32s(int *p, int i) { return COH ? cld(p, (long long)i) : gld(p, (long long)i); }

template <bool COH>
WM_DEV void ksw_dp_generic(const wm_ksw_score_t sc, const wm_ksw_djob_t jb, const uint8_t *__restrict__ seqs,
                           uint8_t *__restrict__ tb_arena, signed char *mem, int *Hm, wm_ksw_dres_t *__restrict__ res)
{
	const int qlen = jb.qlen, tlen = jb.tlen, flag = jb.flag, zdrop = jb.zdrop;
	const int w = jb.w < 0 ? (tlen > qlen ? tlen : qlen) : jb.w;
	const bool approx = (flag & KSW_F_APPROX_MAX) != 0, right = (flag & KSW_F_RIGHT) != 0;
	const uint8_t *query = seqs + jb.q_off, *target = seqs + jb.t_off;
	uint8_t *tbp = tb_arena + jb.tb_off;
	const int T = (tlen + 15) / 16 * 16;
	signed char *u = mem, *v = u + T, *x = v + T, *y = x + T, *x2 = y + T, *y2 = x2 + T, *s = y2 + T;
	const int q = sc.q, e = sc.e, q2 = sc.q2, e2 = sc.e2, qe = q + e, qe2 = q2 + e2;
	const int Q = tb8(q), Q2 = tb8(q2), QE = tb8(qe), QE2 = tb8(qe2);
	const int tS = right ? 0 : 4, tA = right ? 1 : 3, tB = 2, tA2 = right ? 3 : 1, tB2 = right ? 4 : 0;
	const int hA = right ? tA - 1 : tA, hB = right ? tB - 1 : tB, hA2 = right ? tA2 - 1 : tA2, hB2 = right ? tB2 - 1 : tB2;
	const int MCH = tb8(sc.match);
	const int sc_n = sc.sc_ambi == 0 ? -e2 : sc.sc_ambi;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const V<int> ln = lane();
	// initial fill (src/ksw2_extd2_sse.c:99-112): the raw x/y bytes hold x+qe (resp. +qe2), so "-qe" is 0
	for (int t0 = 0; t0 < T; t0 += 64) {
		const V<int> t = ln + t0;
		WM_IF(t < T)
			st8<COH>(u, t, V<int>(-qe)); st8<COH>(v, t, V<int>(-qe)); st8<COH>(x, t, V<int>(0)); st8<COH>(y, t, V<int>(0));
			st8<COH>(x2, t, V<int>(0)); st8<COH>(y2, t, V<int>(0)); st8<COH>(s, t, V<int>(0));
			if (!approx) st32<COH>(Hm, t, V<int>(KSW_NEG_INF));
		WM_END
	}
	mem_sync();
	int ez_max = 0, ez_zdropped = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1;
	int ez_mte = KSW_NEG_INF, ez_mte_q = -1, ez_score = KSW_NEG_INF;
	int H0 = 0, last_H0_t = 0, last_st = -1, last_en = -1;
	const int n_rows = qlen + tlen - 1;
	for (int r = 0; r < n_rows; ++r) {
		int st0 = 0, en0 = tlen - 1;
		if (st0 < r - qlen + 1) st0 = r - qlen + 1;
		if (en0 > r) en0 = r;
		if (st0 < (r - w + 1) >> 1) st0 = (r - w + 1) >> 1;
		if (en0 > (r + w) >> 1) en0 = (r + w) >> 1;
		if (st0 > en0) { ez_zdropped = 1; break; }
		const int st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
		const int sched = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		int x1b = 0, x21b = 0, v1b;                                         // raw bytes of lane st-1 (:141-151)
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) { x1b = ld8s<COH>(x, st - 1); x21b = ld8s<COH>(x2, st - 1); v1b = ld8s<COH>(v, st - 1); }
			else v1b = -qe;
		} else v1b = sched;
		if (en >= r) {                                                       // :152-155
			WM_IF(ln == 0) st8<COH>(y, V<int>(r), V<int>(0)); st8<COH>(y2, V<int>(r), V<int>(0)); st8<COH>(u, V<int>(r), V<int>(sched)); WM_END
		}
		{   // score row: 16-byte chunks starting at st0 (:158-173); lanes >= T would spill into the next array and are never read
			const int cend = st0 + (en0 - st0) / 16 * 16 + 15;
			for (int t0 = st0; t0 <= cend; t0 += 64) {
				const V<int> t = ln + t0;
				WM_IF(t <= cend && t < T)
					V<int> tc = 0, qc = 0;
					WM_IF(t < tlen) tc = cast<int>(gld(target, t)); WM_END
					const V<int> qi = V<int>(r) - t;
					WM_IF(qi >= 0 && qi < qlen) qc = cast<int>(gld(query, qi)); WM_END
					V<int> sv = sel(tc == qc, (int)sc.match, (int)sc.mismatch);
					sv = sel((tc == 4) || (qc == 4), sc_n, sv);
					st8<COH>(s, t, sv);
				WM_END
			}
		}
		mem_sync();
		// sweep the hull; carry = previous-row values of the lane just below the current tile
		V<int> cx = tb8(x1b) | tA, cv = tb8(v1b), cx2 = tb8(x21b) | tA2;
		V<int> hcarry = KSW_NEG_INF;
		if (!approx && st > 0) hcarry = ld32s<COH>(Hm, st - 1);
		V<long long> key = (long long)(-0x7fffffffffffffffLL - 1);
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		for (int t0 = st; t0 <= en; t0 += 64) {
			const V<int> t = ln + t0;
			const vbool act = t <= en;
			V<int> ox = tA, ov = 0, ox2 = tA2, ou = 0, oy = tB, oy2 = tB2, os = tS, oh = KSW_NEG_INF;
			WM_IF(act)
				ou = ld8<COH>(u, t) << 24; ov = ld8<COH>(v, t) << 24;
				ox = (ld8<COH>(x, t) << 24) | tA; oy = (ld8<COH>(y, t) << 24) | tB;
				ox2 = (ld8<COH>(x2, t) << 24) | tA2; oy2 = (ld8<COH>(y2, t) << 24) | tB2;
				os = (ld8<COH>(s, t) << 24) | tS;
				if (!approx) oh = ld32<COH>(Hm, t);
			WM_END
			// lane-1 values of the previous row: neighbour thread, or the carry for lane 0
			const V<int> x1 = sel(ln == 0, cx, shr_n(ox, 1)), v1 = sel(ln == 0, cv, shr_n(ov, 1)), x21 = sel(ln == 0, cx2, shr_n(ox2, 1));
			const V<int> hl = sel(ln == 0, hcarry, shr_n(oh, 1));
			cx = V<int>(readlane(ox, 63)); cv = V<int>(readlane(ov, 63)); cx2 = V<int>(readlane(ox2, 63)); hcarry = V<int>(readlane(oh, 63));
			WM_IF(act)
				V<int> a = add3(x1, v1, -QE), b = add3(oy, ou, -QE), a2 = add3(x21, v1, -QE2), b2 = add3(oy2, ou, -QE2);
				V<int> zz = vmax3(vmax3(os, a, b), a2, b2);
				V<int> z = vmin(zz & (int)0xff000000, MCH);
				V<int> p = zz & 7;
				const V<int> nu = wsub(z, v1), nv = wsub(z, ou);
				V<int> tmp = wsub(z, Q), tmp2 = wsub(z, Q2);
				a = wsub(a, tmp); b = wsub(b, tmp); a2 = wsub(a2, tmp2); b2 = wsub(b2, tmp2);
				p = wadd(wadd(p, p), sel(a > hA, 1, 0));
				p = wadd(wadd(p, p), sel(b > hB, 1, 0));
				p = wadd(wadd(p, p), sel(a2 > hA2, 1, 0));
				p = wadd(wadd(p, p), sel(b2 > hB2, 1, 0));
				st8<COH>(u, t, nu >> 24); st8<COH>(v, t, nv >> 24);
				st8<COH>(x, t, vmax(a, tA) >> 24); st8<COH>(y, t, vmax(b, tB) >> 24);
				st8<COH>(x2, t, vmax(a2, tA2) >> 24); st8<COH>(y2, t, vmax(b2, tB2) >> 24);
				gst(tbp + (size_t)r * jb.n_col, t - st, cast<uint8_t>(p));
				if (!approx && r > 0) {
					const V<int> v8 = nv >> 24, u8 = nu >> 24;
					V<int> hn = oh + v8;
					hn = sel(t == en0, en0 > 0 ? V<int>(hl + u8) : hn, hn);
					const vbool inb = t >= st0 && t <= en0;
					WM_IF(inb) st32<COH>(Hm, t, hn); WM_END
					V<int> grp = sel(t == en0, 5, sel(t < en1, 4 - ((t - st0) & 3), 0));
					V<int> pri = (grp << 20) | (0xfffff - t);
					V<long long> k = cast<long long>(hn) * 4294967296LL + cast<long long>(pri);
					key = sel(inb && k > key, k, key);
				}
				if (!approx && r == 0) {
					WM_IF(t == 0)
						const V<int> h0 = (nv >> 24) - qe;
						st32<COH>(Hm, t, h0);
						key = cast<long long>(h0) * 4294967296LL + (long long)((5 << 20) | 0xfffff);
					WM_END
				}
			WM_END
		}
		mem_sync();
		if (!approx) {
			key = wave_max_i64(key);
			const long long kk = uniform(key);
			const int max_H = (int)(kk >> 32), pri = (int)(kk & 0xffffffffLL);
			const int max_t = 0xfffff - (pri & 0xfffff);
			if (en0 == tlen - 1) { const int h = ld32s<COH>(Hm, en0); if (h > ez_mte) ez_mte = h, ez_mte_q = r - en; }
			if (r - st0 == qlen - 1) { const int h = ld32s<COH>(Hm, st0); if (h > ez_mqe) ez_mqe = h, ez_mqe_t = st0; }
			if (max_H > ez_max) {
				ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
			} else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
				const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
				if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; break; }
			}
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = ld32s<COH>(Hm, tlen - 1);
		} else {
			if (r > 0) {
				const bool in0 = last_H0_t >= st0 && last_H0_t <= en0, in1 = last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0;
				if (in0 && in1) {
					const int d0 = ld8s<COH>(v, last_H0_t), d1 = ld8s<COH>(u, last_H0_t + 1);
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (in0) H0 += ld8s<COH>(v, last_H0_t);
				else { ++last_H0_t; H0 += ld8s<COH>(u, last_H0_t); }
			} else H0 = ld8s<COH>(v, 0) - qe, last_H0_t = 0;
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = H0;
		}
		last_st = st, last_en = en;
	}
	int bt_i = -1, bt_j = -1, reach_end = 0;
	if (!ez_zdropped && !(flag & KSW_F_EXTZ_ONLY)) bt_i = tlen - 1, bt_j = qlen - 1;
	else if (!ez_zdropped && (flag & KSW_F_EXTZ_ONLY) && ez_mqe + jb.end_bonus > ez_max) reach_end = 1, bt_i = ez_mqe_t, bt_j = qlen - 1;
	else if (ez_max_t >= 0 && ez_max_q >= 0) bt_i = ez_max_t, bt_j = ez_max_q;
	WM_IF(ln == 0)
		wm_ksw_dres_t o;
		o.max = ez_max; o.zdropped = ez_zdropped; o.max_q = ez_max_q; o.max_t = ez_max_t;
		o.mqe = ez_mqe; o.mqe_t = ez_mqe_t; o.mte = ez_mte; o.mte_q = ez_mte_q;
		o.score = ez_score; o.reach_end = reach_end; o.n_cigar = 0; o.bt_i = bt_i; o.bt_j = bt_j;
		*res = o;
	WM_END
}

// ------------------------------------------------------------------------------------------------------
// Block kernel for band hulls wider than the register window but narrow enough for LDS (the stage-2 gap fills of
// map-ont / map-pb: w = 3001, hull <= 3040 lanes): NWV wavefronts share ONE alignment. The per-lane state of the
// same machine lives in LDS as two packed words per lane (u|v|x|y and x2|y2|s, raw int8 bytes; x,y,x2,y2 biased by
// their gap-open+extend cost as in the generic kernel) plus an int32 H (exact-max mode), in a circular window of Wn
// lanes (Wn a power of two > hull + 64; a lane enters and leaves the hull exactly once, so slots are recycled and a
// lane that was never computed is substituted by its initial value instead of being pre-filled).
// The two code strings are staged in LDS as well when they fit (a global byte load per lane and row would put a
// full memory round trip on the critical path of every row).
// Row r: every wave LOADS its K tiles (own words + the words of lane t-1, previous-row values) | barrier | computes,
// stores, publishes the values the scalar bookkeeping needs (row maximum, H at en0 / st0, the approximate-max
// track) | barrier | all waves replay the same scalar bookkeeping from the published values, so they stay in step.
// ------------------------------------------------------------------------------------------------------
// one DP cell in the top-byte representation (src/ksw2_extd2_sse.c:205-311): inputs are the previous-row values, outputs the
// new u, v, x, y, x2, y2 (top byte) and the traceback byte
struct ksw_cell_cst_t { int Q, Q2, QE, QE2, MCH, tA, tB, tA2, tB2, hA, hB, hA2, hB2; };
WM_DEV void ksw_cell(const ksw_cell_cst_t &c, const V<int> os, const V<int> x1, const V<int> v1, const V<int> x21, const V<int> oy, const V<int> ou, const V<int> oy2,
                     V<int> &nu, V<int> &nv, V<int> &nx, V<int> &ny, V<int> &nx2, V<int> &ny2, V<int> &p)
{
	V<int> a = add3(x1, v1, -c.QE), b = add3(oy, ou, -c.QE), a2 = add3(x21, v1, -c.QE2), b2 = add3(oy2, ou, -c.QE2);
	const V<int> zz = vmax3(vmax3(os, a, b), a2, b2);
	const V<int> z = vmin(zz & (int)0xff000000, c.MCH);
	p = zz & 7;
	nu = wsub(z, v1); nv = wsub(z, ou);
	const V<int> tmp = wsub(z, c.Q), tmp2 = wsub(z, c.Q2);
	a = wsub(a, tmp); b = wsub(b, tmp); a2 = wsub(a2, tmp2); b2 = wsub(b2, tmp2);
	p = wadd(wadd(p, p), sel(a > c.hA, 1, 0));
	p = wadd(wadd(p, p), sel(b > c.hB, 1, 0));
	p = wadd(wadd(p, p), sel(a2 > c.hA2, 1, 0));
	p = wadd(wadd(p, p), sel(b2 > c.hB2, 1, 0));
	nx = vmax(a, c.tA); ny = vmax(b, c.tB); nx2 = vmax(a2, c.tA2); ny2 = vmax(b2, c.tB2);
}

// ------------------------------------------------------------------------------------------------------
// Helpers of the STRIPED lane layout used by ksw_dp_packed (ksw_packed_kernel.h): thread j holds lanes base + 64*i + j (i = 0..B-1,
// "chunk" i = 64 consecutive lanes). The window re-base (hull start +16 lanes) rotates every register by 16 threads and carries the low
// 16 threads of chunk i+1 into chunk i.
// ------------------------------------------------------------------------------------------------------
template <int B> WM_DEV int get_lane_striped(const V<int> (&a)[B], int base, int t)
{
	const int o = t - base, jr = o & 63, ir = o >> 6;
	WM_EMU_ASSERT(o >= 0 && o < 64 * B);
	int r = 0;
#pragma unroll
	for (int i = 0; i < B; ++i)
		if (ir == i) r = readlane(a[i], jr);
	return r;
}

// rotate an array of chunk registers down by 16 lanes; the last chunk's top 16 threads receive `fresh`
template <int B> WM_DEV void rebase_striped(V<int> (&a)[B], const V<int> fresh, const vbool low48)
{
	V<int> cur = rot_down(a[0], 16);
#pragma unroll
	for (int i = 0; i < B; ++i) {
		const V<int> nxt = i + 1 < B ? rot_down(a[i + 1 < B ? i + 1 : i], 16) : fresh;
		a[i] = sel(low48, cur, nxt);
		cur = nxt;
	}
}

// Hulls wider than one sweep of the block (CH = 64*NWV*K lanes) are processed in CHUNKS of CH lanes per row: the only
// cross-chunk dependency is the previous-row state of the lane just below a chunk, which one lane per chunk boundary
// saves (into `pub`) at the start of the row, before anybody has stored. GLOBAL = the state window lives in a global
// scratch slab instead of LDS (hulls beyond what LDS holds; all waves of a block share one CU's L1, so a workgroup
// barrier with a memory fence orders their accesses).
template <int NWV, int K, bool GLOBAL>
WM_DEV void ksw_dp_block(const wm_ksw_score_t sc, const wm_ksw_djob_t jb, const uint8_t *query, const uint8_t *target, uint8_t *__restrict__ tb_arena,
                         int *W0, int *W1, int *Hm, int Wn, int *pub, wm_ksw_dres_t *__restrict__ res)
{
	constexpr int CH = 64 * NWV * K, MAXC = 16;
	// query / target: the job's two code strings (the kernel stages them in LDS when they fit, else they are the global copies)
	const int qlen = jb.qlen, tlen = jb.tlen, flag = jb.flag, zdrop = jb.zdrop;
	const int w = jb.w < 0 ? (tlen > qlen ? tlen : qlen) : jb.w;
	const bool approx = (flag & KSW_F_APPROX_MAX) != 0, right = (flag & KSW_F_RIGHT) != 0;
	uint8_t *tbp = tb_arena + jb.tb_off;
	const int wmask = Wn - 1;
	const int q = sc.q, e = sc.e, q2 = sc.q2, e2 = sc.e2, qe = q + e, qe2 = q2 + e2;
	const int Q = tb8(q), Q2 = tb8(q2), QE = tb8(qe), QE2 = tb8(qe2);
	const int tS = right ? 0 : 4, tA = right ? 1 : 3, tB = 2, tA2 = right ? 3 : 1, tB2 = right ? 4 : 0;
	const int hA = right ? tA - 1 : tA, hB = right ? tB - 1 : tB, hA2 = right ? tA2 - 1 : tA2, hB2 = right ? tB2 - 1 : tB2;
	const int MCH = tb8(sc.match);
	const ksw_cell_cst_t cc = { Q, Q2, QE, QE2, MCH, tA, tB, tA2, tB2, hA, hB, hA2, hB2 };
	const int sc_n = sc.sc_ambi == 0 ? -e2 : sc.sc_ambi;
	const int MCHs = (int)sc.match, MISs = (int)sc.mismatch;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const V<int> ln = lane();
	const int wv = wave_in_block();
	const int INIT0 = (int)(((unsigned)(-qe) & 0xffu) | ((unsigned)(-qe) & 0xffu) << 8);      // u = v = -qe, x = y = 0 (biased)
	int *pub_key = pub, *pub_val = pub + 2 * NWV, *pub_bnd = pub + 2 * NWV + 8;                // wave maxima (lo,hi) | h_en0, h_st0, d0, d1 | chunk boundaries (3 ints each)

	int ez_max = 0, ez_zdropped = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1;
	int ez_mte = KSW_NEG_INF, ez_mte_q = -1, ez_score = KSW_NEG_INF;
	int H0 = 0, last_H0_t = 0, last_st = -1, last_en = -1, w1_hi = -1;
	const int n_rows = qlen + tlen - 1;
	for (int r = 0; r < n_rows; ++r) {
		int st0 = 0, en0 = tlen - 1;
		if (st0 < r - qlen + 1) st0 = r - qlen + 1;
		if (en0 > r) en0 = r;
		if (st0 < (r - w + 1) >> 1) st0 = (r - w + 1) >> 1;
		if (en0 > (r + w) >> 1) en0 = (r + w) >> 1;
		if (st0 > en0) { ez_zdropped = 1; break; }
		const int st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
		const int sched = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		const int cend = st0 + (en0 - st0) / 16 * 16 + 15;               // last lane of the rewritten score chunks (:158-173)
		const int top = en > cend ? en : cend;
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		const int n_chunks = (top - st + CH) / CH;
		WM_EMU_ASSERT(n_chunks <= MAXC);
		if (n_chunks > 1 && wv == 0) {                                     // previous-row state just below every later chunk
			const V<int> tm = st + (ln + 1) * CH - 1;
			WM_IF(ln < n_chunks - 1)
				V<int> b0 = INIT0, b1 = 0, bh = KSW_NEG_INF;
				WM_IF(tm >= last_st && tm <= last_en) b0 = gld(W0, tm & wmask); b1 = gld(W1, tm & wmask); WM_END
				if (!approx) { WM_IF(tm <= last_en) bh = gld(Hm, tm & wmask); WM_END }
				gst(pub_bnd, ln * 3, b0); gst(pub_bnd, ln * 3 + 1, b1); gst(pub_bnd, ln * 3 + 2, bh);
			WM_END
		}
		V<long long> key = (long long)(-0x7fffffffffffffffLL - 1);
		for (int ch = 0; ch < n_chunks; ++ch) {
		const int cst = st + ch * CH;                                      // first lane of this chunk

		// ---- loads: own state and the previous-row state of lane t-1 ------------------------------------
		V<int> o0[K], o1[K], oh[K], n0[K], n1[K], nh[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const V<int> t = ln + (cst + 64 * (wv + NWV * k));
			o0[k] = INIT0; o1[k] = 0; oh[k] = KSW_NEG_INF; n0[k] = INIT0; n1[k] = 0; nh[k] = KSW_NEG_INF;
			const int tl0 = cst + 64 * (wv + NWV * k);
			if (tl0 > top) continue;
			if (tl0 - 1 >= last_st && tl0 + 63 <= last_en && tl0 > 0) {  // interior tile: every slot it touches holds valid state
				o0[k] = gld(W0, t & wmask); o1[k] = gld(W1, t & wmask);
				n0[k] = gld(W0, (t - 1) & wmask); n1[k] = gld(W1, (t - 1) & wmask);
				if (!approx) { oh[k] = gld(Hm, t & wmask); nh[k] = gld(Hm, (t - 1) & wmask); }
				continue;
			}
			WM_IF(t <= top)
				WM_IF(t <= last_en) o0[k] = gld(W0, t & wmask); if (!approx) oh[k] = gld(Hm, t & wmask); WM_END
				WM_IF(t <= w1_hi) o1[k] = gld(W1, t & wmask); WM_END
				const V<int> tm = t - 1;
				WM_IF(tm >= last_st && tm <= last_en) n0[k] = gld(W0, tm & wmask); n1[k] = gld(W1, tm & wmask); WM_END
				if (!approx) { WM_IF(tm >= 0 && tm <= last_en) nh[k] = gld(Hm, tm & wmask); WM_END }
			WM_END
		}
		if (GLOBAL) block_sync(); else block_sync_lds();
		if (ch > 0 && wv == 0) {                                           // the chunk's first lane takes the saved boundary
			WM_IF(ln == 0) n0[0] = gld(pub_bnd, V<int>((ch - 1) * 3)); n1[0] = gld(pub_bnd, V<int>((ch - 1) * 3 + 1)); nh[0] = gld(pub_bnd, V<int>((ch - 1) * 3 + 2)); WM_END
		}

		// ---- compute, store, publish ------------------------------------------------------------------------
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int tile0 = cst + 64 * (wv + NWV * k);
			if (tile0 > top) continue;
			const V<int> t = ln + tile0;
			if (r > 0 && tile0 > st0 && tile0 + 63 < en1 && !(approx && last_H0_t + 1 >= tile0 && last_H0_t <= tile0 + 63)) {
				// interior tile: all 64 lanes are plain in-band cells (no boundary, no band edge, no published lane)
				const V<int> tc = cast<int>(gld(target, t)), qc = cast<int>(gld(query, V<int>(r) - t));
				V<int> s8i = sel(tc == qc, MCHs, MISs);
				s8i = sel((tc == 4) || (qc == 4), sc_n, s8i);
				const V<int> ou = o0[k] << 24, oy = (o0[k] & (int)0xff000000) | tB, oy2 = ((o1[k] << 16) & (int)0xff000000) | tB2;
				const V<int> x1 = ((n0[k] << 8) & (int)0xff000000) | tA, v1 = (n0[k] << 16) & (int)0xff000000, x21 = (n1[k] << 24) | tA2;
				V<int> nu, nv, nx, ny, nx2, ny2, p;
				ksw_cell(cc, (s8i << 24) | tS, x1, v1, x21, oy, ou, oy2, nu, nv, nx, ny, nx2, ny2, p);
				const V<int> w0n = cast<int>((cast<unsigned>(nu) >> 24) | ((cast<unsigned>(nv) >> 24) << 8) | ((cast<unsigned>(nx) >> 24) << 16) | (cast<unsigned>(ny) & 0xff000000u));
				const V<int> w1n = cast<int>((cast<unsigned>(nx2) >> 24) | ((cast<unsigned>(ny2) >> 24) << 8)) | ((s8i & 0xff) << 16);
				gst(W0, t & wmask, w0n); gst(W1, t & wmask, w1n);
				gst(tbp + (size_t)r * jb.n_col, t - st, cast<uint8_t>(p));
				if (!approx) {
					const V<int> hn = oh[k] + (nv >> 24);
					gst(Hm, t & wmask, hn);
					const V<int> pri = ((4 - ((t - st0) & 3)) << 20) | (0xfffff - t);
					const V<long long> kk = cast<long long>(hn) * 4294967296LL + cast<long long>(pri);
					key = sel(kk > key, kk, key);
				}
				continue;
			}
			// score of this row for the lanes inside the rewritten chunks; the others keep their old score byte
			V<int> s8 = (o1[k] << 8) >> 24;
			WM_IF(t >= st0 && t <= cend)
				V<int> tc = 0, qc = 0;
				WM_IF(t < tlen) tc = cast<int>(gld(target, t)); WM_END
				const V<int> qi = V<int>(r) - t;
				WM_IF(qi >= 0 && qi < qlen) qc = cast<int>(gld(query, qi)); WM_END
				s8 = sel(tc == qc, (int)sc.match, (int)sc.mismatch);
				s8 = sel((tc == 4) || (qc == 4), sc_n, s8);
			WM_END
			WM_IF(t > en && t <= cend)                                   // score-only lanes above the hull: x2 = y2 = initial
				gst(W1, t & wmask, (o1[k] & 0xffff) | ((s8 & 0xff) << 16));
			WM_END
			WM_IF(t <= en)
				V<int> ou = o0[k] << 24, ov = (o0[k] << 16) & (int)0xff000000;
				V<int> oy = (o0[k] & (int)0xff000000) | tB, oy2 = ((o1[k] << 16) & (int)0xff000000) | tB2;
				WM_IF(t == r)                                             // first column / first row boundary (:152-155)
					oy = tB; oy2 = tB2; ou = tb8(sched);
				WM_END
				const V<int> os = (s8 << 24) | tS;
				V<int> x1 = ((n0[k] << 8) & (int)0xff000000) | tA, v1 = (n0[k] << 16) & (int)0xff000000, x21 = (n1[k] << 24) | tA2;
				WM_IF(t == 0) x1 = tA; x21 = tA2; v1 = tb8(sched); WM_END    // :141-151 (st == 0)
				V<int> nu, nv, nx, ny, nx2, ny2, p;
				ksw_cell(cc, os, x1, v1, x21, oy, ou, oy2, nu, nv, nx, ny, nx2, ny2, p);
				const V<int> w0n = cast<int>((cast<unsigned>(nu) >> 24) | ((cast<unsigned>(nv) >> 24) << 8) | ((cast<unsigned>(nx) >> 24) << 16) | (cast<unsigned>(ny) & 0xff000000u));
				const V<int> w1n = cast<int>((cast<unsigned>(nx2) >> 24) | ((cast<unsigned>(ny2) >> 24) << 8)) | ((s8 & 0xff) << 16);
				gst(W0, t & wmask, w0n); gst(W1, t & wmask, w1n);
				gst(tbp + (size_t)r * jb.n_col, t - st, cast<uint8_t>(p));
				if (!approx) {
					V<int> hkeep = oh[k];                                  // every hull lane re-stores H so that "lane <= last_en" implies a valid slot
					if (r > 0) {
						const V<int> v8 = nv >> 24, u8 = nu >> 24;
						V<int> hn = oh[k] + v8;
						hn = sel(t == en0, en0 > 0 ? V<int>(nh[k] + u8) : hn, hn);
						const vbool inb = t >= st0 && t <= en0;
						hkeep = sel(inb, hn, hkeep);
						WM_IF(inb)
							WM_IF(t == en0) gst(pub_val, V<int>(0), hn); WM_END
							WM_IF(t == st0) gst(pub_val, V<int>(1), hn); WM_END
						WM_END
						V<int> grp = sel(t == en0, 5, sel(t < en1, 4 - ((t - st0) & 3), 0));
						V<int> pri = (grp << 20) | (0xfffff - t);
						V<long long> kk = cast<long long>(hn) * 4294967296LL + cast<long long>(pri);
						key = sel(inb && kk > key, kk, key);
					} else {
						WM_IF(t == 0)
							const V<int> h0 = (nv >> 24) - qe;
							hkeep = h0;
							gst(pub_val, V<int>(0), h0); gst(pub_val, V<int>(1), h0);
							key = cast<long long>(h0) * 4294967296LL + (long long)((5 << 20) | 0xfffff);
						WM_END
					}
					gst(Hm, t & wmask, hkeep);
				} else {
					WM_IF(t == last_H0_t) gst(pub_val, V<int>(2), nv >> 24); WM_END
					WM_IF(t == last_H0_t + 1) gst(pub_val, V<int>(3), nu >> 24); WM_END
				}
			WM_END
		}
		}   // chunks
		if (!approx) {
			key = wave_max_i64(key);
			const long long kk = uniform(key);
			WM_IF(ln == 0) gst(pub_key, V<int>(2 * wv), V<int>((int)(unsigned)(kk & 0xffffffffLL))); gst(pub_key, V<int>(2 * wv + 1), V<int>((int)(kk >> 32))); WM_END
		}
		if (GLOBAL) block_sync(); else block_sync_lds();

		// ---- scalar bookkeeping, identical in every wave ----------------------------------------------------
		if (!approx) {
			long long kk = -0x7fffffffffffffffLL - 1;
			for (int w2 = 0; w2 < NWV; ++w2) {
				const long long k2 = (long long)(((unsigned long long)(unsigned)gld(pub_key, (long long)(2 * w2 + 1)) << 32) | (unsigned)gld(pub_key, (long long)(2 * w2)));
				if (k2 > kk) kk = k2;
			}
			const int max_H = (int)(kk >> 32), pri = (int)(kk & 0xffffffffLL);
			const int max_t = 0xfffff - (pri & 0xfffff);
			if (en0 == tlen - 1) { const int h = gld(pub_val, 0LL); if (h > ez_mte) ez_mte = h, ez_mte_q = r - en; }
			if (r - st0 == qlen - 1) { const int h = gld(pub_val, 1LL); if (h > ez_mqe) ez_mqe = h, ez_mqe_t = st0; }
			if (max_H > ez_max) {
				ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
			} else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
				const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
				if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; break; }
			}
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = gld(pub_val, 0LL);
		} else {
			if (r > 0) {
				const bool in0 = last_H0_t >= st0 && last_H0_t <= en0, in1 = last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0;
				if (in0 && in1) {
					const int d0 = gld(pub_val, 2LL), d1 = gld(pub_val, 3LL);
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (in0) H0 += gld(pub_val, 2LL);
				else { ++last_H0_t; H0 += gld(pub_val, 3LL); }
			} else H0 = gld(pub_val, 2LL) - qe, last_H0_t = 0;
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = H0;
		}
		last_st = st, last_en = en;
		if (top > w1_hi) w1_hi = top;
	}
	int bt_i = -1, bt_j = -1, reach_end = 0;
	if (!ez_zdropped && !(flag & KSW_F_EXTZ_ONLY)) bt_i = tlen - 1, bt_j = qlen - 1;
	else if (!ez_zdropped && (flag & KSW_F_EXTZ_ONLY) && ez_mqe + jb.end_bonus > ez_max) reach_end = 1, bt_i = ez_mqe_t, bt_j = qlen - 1;
	else if (ez_max_t >= 0 && ez_max_q >= 0) bt_i = ez_max_t, bt_j = ez_max_q;
	if (wv == 0) {
		WM_IF(ln == 0)
			wm_ksw_dres_t o;
			o.max = ez_max; o.zdropped = ez_zdropped; o.max_q = ez_max_q; o.max_t = ez_max_t;
			o.mqe = ez_mqe; o.mqe_t = ez_mqe_t; o.mte = ez_mte; o.mte_q = ez_mte_q;
			o.score = ez_score; o.reach_end = reach_end; o.n_cigar = 0; o.bt_i = bt_i; o.bt_j = bt_j;
			*res = o;
		WM_END
	}
}

// ------------------------------------------------------------------------------------------------------
// ksw_backtrack (src/ksw2.h:119-151, is_rot = 1, min_intron_len = 0): ONE thread walks the traceback of one
// alignment and emits run-length CIGAR ops in backtrack order into cig[0..cap); the gather step reverses
// them unless KSW_EZ_REV_CIGAR. Row r covers lanes [st(r), en(r)] (recomputed here from the band formula).
// Returns the number of ops (or -needed if cap is too small).
// ------------------------------------------------------------------------------------------------------
WM_DEV int ksw_backtrack_thread(const wm_ksw_djob_t jb, const uint8_t *__restrict__ tb_arena, int i0, int j0,
                                uint32_t *__restrict__ cig, int cap)
{
	const uint8_t *p = tb_arena + jb.tb_off;
	const int qlen = jb.qlen, tlen = jb.tlen, n_col = jb.n_col;
	const int w = jb.w < 0 ? (tlen > qlen ? tlen : qlen) : jb.w;
	const bool right = (jb.flag & KSW_F_RIGHT) != 0;
	int n = 0, i = i0, j = j0, state = 0;
	uint32_t cur_op = 0xf, cur_len = 0;
#define WM_PUSH(op_, len_) do { if ((uint32_t)(op_) == cur_op) cur_len += (len_); else { if (cur_len) { if (n < cap) cig[n] = cur_len << 4 | cur_op; ++n; } cur_op = (op_); cur_len = (len_); } } while (0)
	while (i >= 0 && j >= 0) {
		const int r = i + j;
		int st0 = 0, en0 = tlen - 1, force = -1, d, ext;
		if (st0 < r - qlen + 1) st0 = r - qlen + 1;
		if (en0 > r) en0 = r;
		if (st0 < (r - w + 1) >> 1) st0 = (r - w + 1) >> 1;
		if (en0 > (r + w) >> 1) en0 = (r + w) >> 1;
		const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
		if (i < off) force = 2;
		if (i > off_end) force = 1;
		const int raw = force < 0 ? p[(size_t)r * n_col + (i - off)] : -1;
		// decode our byte into the reference's: low 3 bits = winning state, bits 3..6 = continuation flags
		if (raw >= 0) {
			const int code = raw >> 4;
			d = right ? code : 4 - code;
			ext = (raw >> 3 & 1) | (raw >> 2 & 1) << 1 | (raw >> 1 & 1) << 2 | (raw & 1) << 3;   // bit (state-1)
		} else d = 0, ext = 0;
		if (state == 0) state = d;
		else if (!(ext >> (state - 1) & 1)) state = 0;
		if (state == 0) state = d;
		if (force >= 0) state = force;
		if (state == 0) { WM_PUSH(0u, 1u); --i; --j; }
		else if (state == 1 || state == 3) { WM_PUSH(2u, 1u); --i; }
		else { WM_PUSH(1u, 1u); --j; }
	}
	if (i >= 0) WM_PUSH(2u, (uint32_t)(i + 1));
	if (j >= 0) WM_PUSH(1u, (uint32_t)(j + 1));
	if (cur_len) { if (n < cap) cig[n] = cur_len << 4 | cur_op; ++n; }
#undef WM_PUSH
	return n <= cap ? n : -n;
}

// ------------------------------------------------------------------------------------------------------
// The same backtrack by one WAVEFRONT per alignment. The walk itself is sequential, but a thread that follows it alone pays one
// dependent global load per step (~0.6 us: 4 ms per batch, tens of ms for the longest alignments). Here the 64 lanes first fetch a tile
// of the traceback matrix — the next KSW_BT_ROWS anti-diagonals, 64 target lanes ending at the current one — into LDS with independent
// loads, then every lane replays the same scalar walk on LDS bytes until it leaves the tile (at least KSW_BT_ROWS / 2 steps later).
// `tile` = KSW_BT_ROWS * 64 bytes of LDS per wavefront. Same op stream as ksw_backtrack_thread. Bit-exact on the emulator and on the GPU; timed in round 3
// (profiles/r03a_first_run.txt): no gain on BASELINE config 2 (0.184 vs 0.187 Gbp/s), so it stays behind WM_KSW_COOP_BT=1.
// ------------------------------------------------------------------------------------------------------
#define KSW_BT_ROWS 32
WM_DEV int ksw_backtrack_wave(const wm_ksw_djob_t jb, const uint8_t *__restrict__ tb_arena, int i0, int j0,
                              uint32_t *__restrict__ cig, int cap, uint8_t *tile)
{
	const uint8_t *p = tb_arena + jb.tb_off;
	const int qlen = jb.qlen, tlen = jb.tlen, n_col = jb.n_col;
	const int w = jb.w < 0 ? (tlen > qlen ? tlen : qlen) : jb.w;
	const bool right = (jb.flag & KSW_F_RIGHT) != 0;
	const V<int> ln = lane();
	int n = 0, i = i0, j = j0, state = 0;
	uint32_t cur_op = 0xf, cur_len = 0;
	auto hull = [&](int r, int &off, int &off_end) {
		int st0 = 0, en0 = tlen - 1;
		if (st0 < r - qlen + 1) st0 = r - qlen + 1;
		if (en0 > r) en0 = r;
		if (st0 < (r - w + 1) >> 1) st0 = (r - w + 1) >> 1;
		if (en0 > (r + w) >> 1) en0 = (r + w) >> 1;
		off = st0 / 16 * 16; off_end = (en0 + 16) / 16 * 16 - 1;
	};
#define WM_PUSH(op_, len_) do { if ((uint32_t)(op_) == cur_op) cur_len += (len_); else { if (cur_len) { if (n < cap) { WM_IF(ln == 0) gst(cig, (long long)n, (uint32_t)(cur_len << 4 | cur_op)); WM_END } ++n; } cur_op = (op_); cur_len = (len_); } } while (0)
	while (i >= 0 && j >= 0) {
		// ---- fetch the tile: rows r0, r0-1, .. r0-KSW_BT_ROWS+1; lane l holds target lane c = i - 63 + l ----
		const int r0 = i + j, c_lo = i - 63;
		const V<int> c = ln + c_lo;
#pragma unroll 4
		for (int k = 0; k < KSW_BT_ROWS; ++k) {
			const int rr = r0 - k;
			V<int> b = 0;
			if (rr >= 0) {
				int off, off_end;
				hull(rr, off, off_end);
				WM_IF(c >= off && c <= off_end && c >= 0) b = cast<int>(gld(p + (size_t)rr * n_col, c - off)); WM_END
			}
			gst(tile, ln + 64 * k, cast<uint8_t>(b));
		}
		lds_sync();
		// ---- walk inside the tile ----
		while (i >= 0 && j >= 0) {
			const int r = i + j, k = r0 - r, col = i - c_lo;
			if (k >= KSW_BT_ROWS || col < 0) break;
			int off, off_end, force = -1, d, ext;
			hull(r, off, off_end);
			if (i < off) force = 2;
			if (i > off_end) force = 1;
			if (force < 0) {
				const int raw = (int)gld(tile, (long long)(64 * k + col));
				const int code = raw >> 4;
				d = right ? code : 4 - code;
				ext = (raw >> 3 & 1) | (raw >> 2 & 1) << 1 | (raw >> 1 & 1) << 2 | (raw & 1) << 3;
			} else d = 0, ext = 0;
			if (state == 0) state = d;
			else if (!(ext >> (state - 1) & 1)) state = 0;
			if (state == 0) state = d;
			if (force >= 0) state = force;
			if (state == 0) { WM_PUSH(0u, 1u); --i; --j; }
			else if (state == 1 || state == 3) { WM_PUSH(2u, 1u); --i; }
			else { WM_PUSH(1u, 1u); --j; }
		}
		lds_sync();
	}
	if (i >= 0) WM_PUSH(2u, (uint32_t)(i + 1));
	if (j >= 0) WM_PUSH(1u, (uint32_t)(j + 1));
	if (cur_len) { if (n < cap) { WM_IF(ln == 0) gst(cig, (long long)n, (uint32_t)(cur_len << 4 | cur_op)); WM_END } ++n; }
#undef WM_PUSH
	return n <= cap ? n : -n;
}

} // namespace wmk
