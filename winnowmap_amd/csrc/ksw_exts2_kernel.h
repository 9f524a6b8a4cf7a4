// ksw_exts2_kernel.h — ksw_exts2_sse (src/ksw2_exts2_sse.c:18-407), the splice-aware extension: one gap class with extension (q, e),
// one long-deletion class without extension (q2: an intron, opened at the donor signal, closed at the acceptor signal), no band.
//
// Same machine as ksw_dp_generic (ksw_kernel.h): one wavefront owns one alignment, walks the anti-diagonals and sweeps the 16-aligned
// hull 64 lanes at a time; the per-lane int8 state (u v x y x2) and the per-lane constants (donor, acceptor) live in a global scratch slab
// of 8 * T bytes (+ 4 * T for the exact-maximum H), T = tlen rounded up to 16. Values sit in the top byte of a 32-bit register, the low
// bits carry the tie-break tag that turns the reference's chain of compare-and-blend (:272-278 left-aligned gaps, :309-315 right-aligned)
// into plain maxima. Differences from ksw_dp_generic that matter:
//   * z = max(s, a, b, a2 + acceptor[t]) — four states, and z is NOT capped at the match score;
//   * x2 = max(a2, donor[t]) - q2 (an intron can always be opened at the donor's price), no extension cost, no y2;
//   * no band, hence no stale-lane feedback: a lane outside [st0, en0] never feeds a lane inside it (the newest lane r gets fresh
//     boundary values, :187-190), so scores are computed on the fly for the whole hull;
//   * the z-drop test ignores the diagonal distance (ksw_apply_zdrop with e = 0, :375), there is no end bonus / reach_end;
//   * traceback bytes use the reference's own layout (state 0..3 | 0x08 | 0x10 | 0x20); ksw_exts2_backtrack_thread turns state 3 into N
//     when long_thres > 0 (src/ksw2.h:119-151 with min_intron_len = long_thres).
// STATUS: bit-exact against the oracle (which is pinned to the reference's function) on the wavefront emulator and on the GPU
// (tests/test_zz_exts2_gpu.py); serves every alignment of splice mode (DeviceOps::exts2_batch → wm_ksw_exts2_batch, tests/test_binding_gpu.py).
// One wavefront per alignment, not tuned.
#pragma once
#ifndef WM_DEV
#error "include simt.h before ksw_exts2_kernel.h"
#endif
#include "ksw_kernel.h"

namespace wmk {

#define KSW_F_SPLICE_FOR 0x100
#define KSW_F_SPLICE_REV 0x200
#define KSW_F_SPLICE_FLANK 0x400

// scratch bytes of one job
WM_DEV uint64_t ksw_exts2_scratch_bytes(int tlen) { const uint64_t T = ((uint64_t)tlen + 15) / 16 * 16; return 12 * T + 256; }

template <bool COH>
WM_DEV void ksw_dp_exts2(const wm_ksw_score_t sc, int noncan, int junc_bonus, const wm_ksw_djob_t jb, const uint8_t *__restrict__ seqs,
                         const uint8_t *__restrict__ junc_all, uint8_t *__restrict__ tb_arena, signed char *mem, int *Hm, wm_ksw_dres_t *__restrict__ res)
{
	const int qlen = jb.qlen, tlen = jb.tlen, flag = jb.flag, zdrop = jb.zdrop;
	const bool approx = (flag & KSW_F_APPROX_MAX) != 0, right = (flag & KSW_F_RIGHT) != 0;
	const uint8_t *query = seqs + jb.q_off, *target = seqs + jb.t_off;
	const uint8_t *junc = junc_all ? junc_all + jb.t_off : 0;
	uint8_t *tbp = tb_arena + jb.tb_off;
	const int T = (tlen + 15) / 16 * 16;
	signed char *u = mem, *v = u + T, *x = v + T, *y = x + T, *x2 = y + T, *dn = x2 + T, *ac = dn + T;
	const int q = (signed char)sc.q, e = (signed char)sc.e, q2 = (signed char)sc.q2, qe = q + e;
	const int Q = tb8(q), Q2 = tb8(q2), QE = tb8(qe);
	// tags: ties go to the earlier state with left-aligned gaps (s, a, b, a2), to the later one with right-aligned gaps
	const int tS = right ? 0 : 3, tA = right ? 1 : 2, tB = right ? 2 : 1, tA2 = right ? 3 : 0;
	const int hA = right ? tA - 1 : tA, hB = right ? tB - 1 : tB, hD = right ? tA2 - 1 : tA2;   // "a > 0" (left) / "a >= 0" (right) on tagged values
	const int sc_n = sc.sc_ambi == 0 ? -e : sc.sc_ambi;
	int long_thres = (q2 - q) / e - 1;
	if (q2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * e - (q2 - q);
	const V<int> ln = lane();
	// initial fill (:97-98: the raw x / y bytes hold x + qe, x2 holds x2 + q2, so the initial values are 0) and the splice signals (:109-166)
	{
		const bool fwd = (flag & KSW_F_SPLICE_FOR) != 0, rev = (flag & KSW_F_SPLICE_REV) != 0, any = fwd || rev, rc = (flag & KSW_F_REV_CIGAR) != 0;
		const int semi = (flag & KSW_F_SPLICE_FLANK) ? -noncan / 2 : 0;
		// donor: bases t+1, t+2 = GT / CT (reversed operands: GA / CA), flank t+3 = A or G (C or T); acceptor: bases t-1, t = AG / AC (TG / TC), flank t-2
		const int d2 = rc ? 0 : 3, dfa = rc ? 1 : 0, dfb = rc ? 3 : 2;
		const int a1 = rc ? 3 : 0, afa = rc ? 0 : 1, afb = rc ? 2 : 3;
		const int jd_f = rc ? 2 : 1, jd_r = rc ? 4 : 8, ja_f = rc ? 1 : 2, ja_r = rc ? 8 : 4;
		for (int t0 = 0; t0 < T; t0 += 64) {
			const V<int> t = ln + t0;
			WM_IF(t < T)
				st8<COH>(u, t, V<int>(-qe)); st8<COH>(v, t, V<int>(-qe)); st8<COH>(x, t, V<int>(0)); st8<COH>(y, t, V<int>(0)); st8<COH>(x2, t, V<int>(0));
				if (!approx) st32<COH>(Hm, t, V<int>(KSW_NEG_INF));
				V<int> dv = any ? -noncan : 0, av = dv;
				if (any) {
					V<int> c1 = 9, c2 = 9, c3 = 9, p0 = 9, p1 = 9, p2 = 9;
					WM_IF(t + 1 < tlen) c1 = cast<int>(gld(target, t + 1)); WM_END
					WM_IF(t + 2 < tlen) c2 = cast<int>(gld(target, t + 2)); WM_END
					WM_IF(t + 3 < tlen) c3 = cast<int>(gld(target, t + 3)); WM_END
					WM_IF(t < tlen) p0 = cast<int>(gld(target, t)); WM_END
					WM_IF(t >= 1 && t - 1 < tlen) p1 = cast<int>(gld(target, t - 1)); WM_END
					WM_IF(t >= 2 && t - 2 < tlen) p2 = cast<int>(gld(target, t - 2)); WM_END
					const vbool dcan = t < tlen - 4 && ((fwd && c1 == 2 && c2 == d2) || (rev && c1 == 1 && c2 == d2));
					dv = sel(dcan, sel(c3 == dfa || c3 == dfb, V<int>(0), V<int>(semi)), dv);
					const vbool acan = t >= 2 && t < tlen && ((fwd && p1 == a1 && p0 == 2) || (rev && p1 == a1 && p0 == 1));
					av = sel(acan, sel(p2 == afa || p2 == afb, V<int>(0), V<int>(semi)), av);
					if (junc) {
						V<int> j1 = 0, j0 = 0;
						WM_IF(t + 1 < tlen) j1 = cast<int>(gld(junc, t + 1)); WM_END
						WM_IF(t < tlen) j0 = cast<int>(gld(junc, t)); WM_END
						dv = sel(t < tlen - 1 && ((fwd && (j1 & jd_f) != 0) || (rev && (j1 & jd_r) != 0)), dv + junc_bonus, dv);
						av = sel(t < tlen && ((fwd && (j0 & ja_f) != 0) || (rev && (j0 & ja_r) != 0)), av + junc_bonus, av);
					}
				}
				st8<COH>(dn, t, dv); st8<COH>(ac, t, av);          // (stored as int8: the additions above wrap like the reference's)
			WM_END
		}
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
		const int st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
		const int sched = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : 0;
		int x1b = 0, x21b = 0, v1b;                                         // raw bytes of lane st-1 (:178-186)
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) { x1b = ld8s<COH>(x, st - 1); x21b = ld8s<COH>(x2, st - 1); v1b = ld8s<COH>(v, st - 1); }
			else v1b = -qe;
		} else v1b = sched;
		if (en >= r) {                                                       // :187-190
			WM_IF(ln == 0) st8<COH>(y, V<int>(r), V<int>(0)); st8<COH>(u, V<int>(r), V<int>(sched)); WM_END
		}
		mem_sync();
		V<int> cx = tb8(x1b) | tA, cv = tb8(v1b), cx2 = tb8(x21b) | tA2;
		V<int> hcarry = KSW_NEG_INF;
		if (!approx && st > 0) hcarry = ld32s<COH>(Hm, st - 1);
		V<long long> key = (long long)(-0x7fffffffffffffffLL - 1);
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		for (int t0 = st; t0 <= en; t0 += 64) {
			const V<int> t = ln + t0;
			const vbool act = t <= en;
			V<int> ox = tA, ov = 0, ox2 = tA2, ou = 0, oy = tB, os = tS, oh = KSW_NEG_INF, od = 0, oa = 0;
			WM_IF(act)
				ou = ld8<COH>(u, t) << 24; ov = ld8<COH>(v, t) << 24;
				ox = (ld8<COH>(x, t) << 24) | tA; oy = (ld8<COH>(y, t) << 24) | tB;
				ox2 = (ld8<COH>(x2, t) << 24) | tA2;
				od = (ld8<COH>(dn, t) << 24) | hD; oa = ld8<COH>(ac, t) << 24;
				if (!approx) oh = ld32<COH>(Hm, t);
				V<int> tc = 0, qc = 0;
				WM_IF(t < tlen) tc = cast<int>(gld(target, t)); WM_END
				const V<int> qi = V<int>(r) - t;
				WM_IF(qi >= 0 && qi < qlen) qc = cast<int>(gld(query, qi)); WM_END
				V<int> sv = sel(tc == qc, (int)sc.match, (int)sc.mismatch);
				sv = sel((tc == 4) || (qc == 4), sc_n, sv);
				os = (sv << 24) | tS;
			WM_END
			// lane-1 values of the previous row: neighbour thread, or the carry for lane 0
			const V<int> x1 = sel(ln == 0, cx, shr_n(ox, 1)), v1 = sel(ln == 0, cv, shr_n(ov, 1)), x21 = sel(ln == 0, cx2, shr_n(ox2, 1));
			const V<int> hl = sel(ln == 0, hcarry, shr_n(oh, 1));
			cx = V<int>(readlane(ox, 63)); cv = V<int>(readlane(ov, 63)); cx2 = V<int>(readlane(ox2, 63)); hcarry = V<int>(readlane(oh, 63));
			WM_IF(act)
				V<int> a = add3(x1, v1, -QE), b = add3(oy, ou, -QE), a2 = add3(x21, v1, -Q2);
				const V<int> a2a = wadd(a2, oa);
				const V<int> zz = vmax(vmax3(os, a, b), a2a);
				const V<int> z = zz & (int)0xff000000;
				const V<int> tag = zz & 7;
				V<int> p = right ? tag : V<int>(V<int>(3) - tag);                       // the winning state 0..3
				const V<int> nu = wsub(z, v1), nv = wsub(z, ou);
				const V<int> tmp = wsub(z, Q);
				a = wsub(a, tmp); b = wsub(b, tmp); a2 = wsub(a2, wsub(z, Q2));
				p = p | sel(a > hA, 0x08, 0) | sel(b > hB, 0x10, 0) | sel(a2 > od, 0x20, 0);
				st8<COH>(u, t, nu >> 24); st8<COH>(v, t, nv >> 24);
				st8<COH>(x, t, vmax(a, tA) >> 24); st8<COH>(y, t, vmax(b, tB) >> 24);
				st8<COH>(x2, t, vmax(a2, od) >> 24);
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
				if (zdrop >= 0 && ez_max - max_H > zdrop) { ez_zdropped = 1; break; }      // (:375: e = 0)
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
	int bt_i = -1, bt_j = -1;
	if (!ez_zdropped && !(flag & KSW_F_EXTZ_ONLY)) bt_i = tlen - 1, bt_j = qlen - 1;        // :400-405
	else if (ez_max_t >= 0 && ez_max_q >= 0) bt_i = ez_max_t, bt_j = ez_max_q;
	WM_IF(ln == 0)
		wm_ksw_dres_t o;
		o.max = ez_max; o.zdropped = ez_zdropped; o.max_q = ez_max_q; o.max_t = ez_max_t;
		o.mqe = ez_mqe; o.mqe_t = ez_mqe_t; o.mte = ez_mte; o.mte_q = ez_mte_q;
		o.score = ez_score; o.reach_end = 0; o.n_cigar = 0; o.bt_i = bt_i; o.bt_j = bt_j;
		*res = o;
	WM_END
}

// ksw_backtrack (src/ksw2.h:119-151, is_rot = 1, min_intron_len = long_thres): ops in backtrack order, like ksw_backtrack_thread
WM_DEV int ksw_exts2_backtrack_thread(const wm_ksw_score_t sc, const wm_ksw_djob_t jb, const uint8_t *__restrict__ tb_arena, int i0, int j0,
                                      uint32_t *__restrict__ cig, int cap)
{
	const uint8_t *p = tb_arena + jb.tb_off;
	const int qlen = jb.qlen, tlen = jb.tlen, n_col = jb.n_col;
	const int q = (signed char)sc.q, e = (signed char)sc.e, q2 = (signed char)sc.q2;
	int min_intron = (q2 - q) / e - 1;
	if (q2 > q + e + min_intron * e) ++min_intron;
	int n = 0, i = i0, j = j0, state = 0;
	uint32_t cur_op = 0xf, cur_len = 0;
#define WM_PUSH(op_, len_) do { if ((uint32_t)(op_) == cur_op) cur_len += (len_); else { if (cur_len) { if (n < cap) cig[n] = cur_len << 4 | cur_op; ++n; } cur_op = (op_); cur_len = (len_); } } while (0)
	while (i >= 0 && j >= 0) {
		const int r = i + j;
		int st0 = 0, en0 = tlen - 1, force = -1;
		if (st0 < r - qlen + 1) st0 = r - qlen + 1;
		if (en0 > r) en0 = r;
		const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
		if (i < off) force = 2;
		if (i > off_end) force = 1;
		const int d = force < 0 ? p[(size_t)r * n_col + (i - off)] : 0;
		if (state == 0) state = d & 7;
		else if (!(d >> (state + 2) & 1)) state = 0;
		if (state == 0) state = d & 7;
		if (force >= 0) state = force;
		if (state == 0) { WM_PUSH(0u, 1u); --i; --j; }
		else if (state == 1 || (state == 3 && min_intron <= 0)) { WM_PUSH(2u, 1u); --i; }
		else if (state == 3) { WM_PUSH(3u, 1u); --i; }
		else { WM_PUSH(1u, 1u); --j; }
	}
	if (i >= 0) WM_PUSH((min_intron > 0 && i >= min_intron) ? 3u : 2u, (uint32_t)(i + 1));
	if (j >= 0) WM_PUSH(1u, (uint32_t)(j + 1));
	if (cur_len) { if (n < cap) cig[n] = cur_len << 4 | cur_op; ++n; }
#undef WM_PUSH
	return n <= cap ? n : -n;
}

} // namespace wmk
