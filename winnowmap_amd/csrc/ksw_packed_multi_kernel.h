// ksw_packed_multi_kernel.h — ksw_extd2_sse (src/ksw2_extd2_sse.c:26-393) for band hulls wider than one wavefront's register window:
// the packed two-cells-per-lane machine of ksw_packed_kernel.h spread over the NWV wavefronts of one workgroup.
//
// STATUS: bit-exact against the oracle on the wavefront emulator (tests/test_kernels_emu.py, every geometry) and on the GPU
// (tests/test_ksw_gpu.py; first hardware run in round 3, profiles/r03a_first_run.txt). 104 / 178 VGPRs, no scratch. It serves the BLOCK / BLOCK2
// classes (hulls of 2033..8176 lanes) as <4,8> / <8,8> and, by default (WM_KSW_PMULTI=2), the 16-pair register classes as <4,4>: the long
// single-wave jobs whose 30..50 ms per alignment were the tail of every batch of heavy alignments.
//
// Layout: the window of 128 * BP * NWV lanes starting at the hull start `base` is striped over chunk PAIRS; pair g (lanes base + 128 g ..
// + 127: low halves = the first 64, high halves = the next 64) lives in register slot g / NWV of wavefront g % NWV, so the pairs that
// intersect the hull — a prefix of the window — are spread evenly over the wavefronts, and a row costs each of them only its share.
// What crosses a wavefront boundary goes through LDS, with ONE barrier per row:
//   * the previous-row values of the lane below a pair's first lane (x, v, x2, H of lane 63 / high half of pair g - 1): every wavefront
//     publishes them for its pairs at the end of a row (xch, double-buffered by row parity). On a row that starts with a re-base they
//     are not needed: the lane below pair g's first lane is then lane 15 of pair g's OWN pre-re-base registers;
//   * the window re-base (+16 lanes, every >= 16 rows): the high halves' top 16 threads of pair g take the low halves' first 16 threads of
//     pair g + 1 — published and picked up around one extra barrier on those rows (rb);
//   * what the scalar bookkeeping needs (exact row maximum with the reference's tie rule, H at en0 / st0, the approximate-max track):
//     published by the owners, replayed identically by every wavefront after the row's barrier.
// The two code strings are read through `query` / `target` (LDS copies when they fit, else global): the query code of lane t in row r is
// query[r - t], so nothing has to be shifted from lane to lane or across wavefronts; target codes are re-read after a re-base.
// Cell arithmetic, traceback bytes, stale-lane emulation (CLIP), exact / approximate maxima: exactly ksw_dp_packed.
#pragma once
#ifndef WM_DEV
#error "include simt.h before ksw_packed_multi_kernel.h"
#endif
#include "ksw_packed_kernel.h"

namespace wmk {

// LDS ints: xch 2 * NP * 4 | pub 2 * (2 * NWV + 4) | rb NP * (7 + 1) * 16   (NP = BP * NWV pairs)
template <int BP, int NWV> struct ksw_pmulti_lds { enum { NP = BP * NWV, XCH = 2 * NP * 4, PUB = 2 * (2 * NWV + 4), NARR = 8, RB = NP * NARR * 16, INTS = XCH + PUB + RB }; };

template <int BP, int NWV, bool CLIP, bool HASN, bool EXACT>
WM_DEV void ksw_dp_pmulti(const wm_ksw_score_t sc, const wm_ksw_djob_t jb, const uint8_t *query, const uint8_t *target,
                          uint8_t *__restrict__ tb_arena, int *lds, wm_ksw_dres_t *__restrict__ res)
{
	typedef ksw_pmulti_lds<BP, NWV> L;
	constexpr int NP = L::NP;
	int *xch = lds, *pub = lds + L::XCH, *rb = pub + L::PUB;
	const int qlen = jb.qlen, tlen = jb.tlen, flag = jb.flag, zdrop = jb.zdrop;
	const int w = jb.w < 0 ? (tlen > qlen ? tlen : qlen) : jb.w;
	const bool right = (flag & KSW_F_RIGHT) != 0;
	WM_EMU_ASSERT(EXACT == !(flag & KSW_F_APPROX_MAX));
	uint8_t *tbp = tb_arena + jb.tb_off;
	const int q = sc.q, e = sc.e, q2 = sc.q2, e2 = sc.e2, qe = q + e, qe2 = q2 + e2;
	const int tS = right ? 0 : 4, tA = right ? 1 : 3, tB = 2, tA2 = right ? 3 : 1, tB2 = right ? 4 : 0;
	const int hA = right ? tA - 1 : tA, hB = right ? tB - 1 : tB, hA2 = right ? tA2 - 1 : tA2, hB2 = right ? tB2 - 1 : tB2;
	const int MCHt = (((int)sc.match & 0xff) << 8) | tS, MISt = (((int)sc.mismatch & 0xff) << 8) | tS;
	const int NNt = (((sc.sc_ambi == 0 ? -e2 : (int)sc.sc_ambi) & 0xff) << 8) | tS;
	const int one2 = (int)sc.match > -128 ? 0x00010001 : 0x00020002;       // 1 | 1 << 16, opaque to the compiler (see ksw_dp_packed)
	const ksw_pcell_cst_t cc = { tb16(qe), tb16(qe2), tb16(q), tb16(q2), tb16(sc.match), rep16(tA), rep16(tB), rep16(tA2), rep16(tB2),
	                             rep16(hA), rep16(hB), rep16(hA2), rep16(hB2) };
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	const V<int> ln = lane();
	const int wv = wave_in_block();
	const vbool low48 = ln < 48;
	int base = 0;
	V<int> U[BP], Vv[BP], X[BP], Y[BP], X2[BP], Y2[BP];
	V<int> S[CLIP ? BP : 1];
	V<int> H[EXACT ? 2 * BP : 1];
	V<int> TC[BP];                                   // target codes of the pair's lanes: low chunk | high chunk << 16
#pragma unroll
	for (int s = 0; s < BP; ++s) {
		U[s] = tb16(-qe); Vv[s] = tb16(-qe); X[s] = rep16(tA); Y[s] = rep16(tB); X2[s] = rep16(tA2); Y2[s] = rep16(tB2);
		if constexpr (CLIP) S[s] = rep16(tS);
		if constexpr (EXACT) { H[2 * s] = KSW_NEG_INF; H[2 * s + 1] = KSW_NEG_INF; }
	}
	auto load_targets = [&](int b) {
#pragma unroll
		for (int s = 0; s < BP; ++s) {
			const V<int> t_lo = ln + (b + 128 * (s * NWV + wv)), t_hi = t_lo + 64;
			V<int> c0 = 0, c1 = 0;
			WM_IF(t_lo < tlen) c0 = cast<int>(gld(target, t_lo)); WM_END
			WM_IF(t_hi < tlen) c1 = cast<int>(gld(target, t_hi)); WM_END
			TC[s] = c0 | (c1 << 16);
		}
	};
	load_targets(0);
	// "previous row" of row 0 for every pair: the initial state (both parities, so that a pair that has never been computed reads it)
	WM_IF(ln == 63)
#pragma unroll
		for (int s = 0; s < BP; ++s)
			for (int par = 0; par < 2; ++par) {
				int *x = xch + (par * NP + (s * NWV + wv)) * 4;
				gst(x, V<int>(0), V<int>(tA)); gst(x, V<int>(1), V<int>((int)(((unsigned)(-qe) & 0xffu) << 8))); gst(x, V<int>(2), V<int>(tA2)); gst(x, V<int>(3), V<int>(KSW_NEG_INF));
			}
	WM_END
	block_sync_lds();

	int ez_max = 0, ez_zdropped = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1;
	int ez_mte = KSW_NEG_INF, ez_mte_q = -1, ez_score = KSW_NEG_INF;
	int H0 = 0, last_H0_t = 0;
	int Hbelow = KSW_NEG_INF;          // H of lane base - 1 as the last re-base left it (read when en0 == base, i.e. the band sits on the last target lane)
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
		const int par = r & 1, ppar = par ^ 1;
		int *pubr = pub + par * (2 * NWV + 4);           // wave maxima (lo, hi) | h_en0, h_st0, d0, d1

		// ---- previous-row values of the lane below each pair's first lane (16-bit patterns; H as a plain int) ----
		int px[BP], pv[BP], px2[BP], ph[BP];
		const bool rebase = st > base;
		if (rebase) {
			WM_EMU_ASSERT(st == base + 16);
			// that lane is lane 15 (low half) of the pair's own registers before they move
#pragma unroll
			for (int s = 0; s < BP; ++s) {
				px[s] = readlane(X[s], 15) & 0xffff; pv[s] = readlane(Vv[s], 15) & 0xffff; px2[s] = readlane(X2[s], 15) & 0xffff;
				ph[s] = EXACT ? readlane(H[EXACT ? 2 * s : 0], 15) : KSW_NEG_INF;
				if (s * NWV + wv == 0) Hbelow = ph[s];
			}
			// the first 16 threads (low halves) of every pair go to the pair below it
			WM_IF(ln < 16)
#pragma unroll
				for (int s = 0; s < BP; ++s) {
					int *o = rb + (s * NWV + wv) * (L::NARR * 16);
					gst(o, ln + 0 * 16, U[s]); gst(o, ln + 1 * 16, Vv[s]); gst(o, ln + 2 * 16, X[s]); gst(o, ln + 3 * 16, Y[s]);
					gst(o, ln + 4 * 16, X2[s]); gst(o, ln + 5 * 16, Y2[s]);
					if constexpr (CLIP) gst(o, ln + 6 * 16, S[s]);
					if constexpr (EXACT) gst(o, ln + 7 * 16, H[2 * s]);
				}
			WM_END
			block_sync_lds();
#pragma unroll
			for (int s = 0; s < BP; ++s) {
				const int g = s * NWV + wv;
				V<int> nU = tb16(-qe), nV = tb16(-qe), nX = rep16(tA), nY = rep16(tB), nX2 = rep16(tA2), nY2 = rep16(tB2), nS = rep16(tS), nH = KSW_NEG_INF;
				if (g + 1 < NP) {
					WM_IF(!low48)
						const int *o = rb + (g + 1) * (L::NARR * 16);
						const V<int> j = ln - 48;
						nU = gld(o, j + 0 * 16); nV = gld(o, j + 1 * 16); nX = gld(o, j + 2 * 16); nY = gld(o, j + 3 * 16);
						nX2 = gld(o, j + 4 * 16); nY2 = gld(o, j + 5 * 16);
						if constexpr (CLIP) nS = gld(o, j + 6 * 16);
						if constexpr (EXACT) nH = gld(o, j + 7 * 16);
					WM_END
				}
				// thread j < 48 takes thread j + 16 (both halves); thread j >= 48: low half <- own high half of thread j - 48, high half <- the next pair's low half
				auto mv = [&](V<int> &a, const V<int> nxt) { const V<int> cur = rot_down(a, 16); a = sel(low48, cur, alignbit(nxt, cur, 16)); };
				mv(U[s], nU); mv(Vv[s], nV); mv(X[s], nX); mv(Y[s], nY); mv(X2[s], nX2); mv(Y2[s], nY2);
				if constexpr (CLIP) mv(S[s], nS);
				if constexpr (EXACT) {
					const V<int> lo = rot_down(H[2 * s], 16), hi = rot_down(H[2 * s + 1], 16);
					H[2 * s] = sel(low48, lo, hi);               // (thread j >= 48 of the low chunk <- thread j - 48 of the old high chunk)
					H[2 * s + 1] = sel(low48, hi, nH);
				}
			}
			base = st;
			load_targets(base);
		} else {
#pragma unroll
			for (int s = 0; s < BP; ++s) {
				const int g = s * NWV + wv;
				if (g == 0) { px[s] = tA; pv[s] = ((st == 0 ? sched : -qe) & 0xff) << 8; px2[s] = tA2; ph[s] = Hbelow; }
				else {
					const int *x = xch + (ppar * NP + (g - 1)) * 4;
					px[s] = gld(x, (long long)0); pv[s] = gld(x, (long long)1); px2[s] = gld(x, (long long)2); ph[s] = gld(x, (long long)3);
				}
			}
		}

		WM_EMU_ASSERT(base == st);
		// ---- first-column / first-row boundary of lane r (:152-155): applied by the owner of the pair that holds the hull end (lane r shares
		// its 16-lane group, hence its pair, with `en`); bm selects the half of thread (r - base) & 63 that holds it
		V<int> bm = 0;
		if (en >= r) {
			const int o = r - base;
			WM_EMU_ASSERT(o >= 0 && o < 128 * NP && (o >> 7) == ((en - base) >> 7));
			bm = sel(ln == (o & 63), (o & 64) ? (int)0xffff0000 : 0x0000ffff, 0);
		}

		const int cend = st0 + (en0 - st0) / 16 * 16 + 15;           // last lane of the rewritten score chunks (:158-173)
		const int NI = ((en - base) >> 7) + 1;                         // pairs that intersect the hull
		const int NS = CLIP ? ((((cend > en ? cend : en) - base) >> 7) + 1) : NI;
		WM_EMU_ASSERT(NS <= NP);
		V<int> hmax = KSW_NEG_INF;
		uint8_t *trow = tbp + (size_t)r * jb.n_col;                    // column of lane t = t - st = t - base
		const int en0x = en0 > 0 ? en0 : -1;

#pragma unroll
		for (int s = 0; s < BP; ++s) {
			const int g = s * NWV + wv;
			if (g >= NS) continue;
			const bool top = g == NI - 1;          // the pair that holds the hull end: lane masks, the boundary lane, lane en0
			const int c0 = base + 128 * g;
			const V<int> t_lo = ln + c0, t_hi = ln + (c0 + 64);
			// query codes of the two chunks in this row: query[r - t]
			V<int> q0 = 0, q1 = 0;
			{
				const V<int> qi0 = r - t_lo, qi1 = r - t_hi;
				WM_IF(qi0 >= 0 && qi0 < qlen) q0 = cast<int>(gld(query, qi0)); WM_END
				WM_IF(qi1 >= 0 && qi1 < qlen) q1 = cast<int>(gld(query, qi1)); WM_END
			}
			const V<int> qc = q0 | (q1 << 16);
			const V<int> xq = TC[s] ^ qc;
			V<int> sv = pk_mad(pk_minu(xq, one2), rep16(MISt - MCHt), rep16(MCHt));
			if constexpr (HASN) {
				const V<int> isn = pk_lshr((TC[s] | qc) & 0x00040004, 2);
				sv = bfi(pk_sub(0, isn), rep16(NNt), sv);
			}
			if constexpr (CLIP) {   // the score row is persistent and only [st0, cend] is rewritten
				if (c0 >= st0 && c0 + 127 <= cend) S[s] = sv;
				else {
					const V<int> m = sel(t_lo >= st0 && t_lo <= cend, 0x0000ffff, 0) | sel(t_hi >= st0 && t_hi <= cend, (int)0xffff0000, 0);
					S[s] = bfi(m, sv, S[s]);
				}
				sv = S[s];
			}
			if (g >= NI) continue;
			int hprev = KSW_NEG_INF;                                   // H of lane en0 - 1 in the previous row
			if (top) {
				WM_KEEP_BRANCH();
				Y[s] = bfi(bm, rep16(tB), Y[s]); Y2[s] = bfi(bm, rep16(tB2), Y2[s]); U[s] = bfi(bm, tb16(sched), U[s]);
				if constexpr (EXACT) {
					const int le = en0 - 1 - c0;                           // lane en0 - 1 relative to the pair: -1 .. 126
					WM_EMU_ASSERT(le >= -1 && le < 127);
					if (le < 0) hprev = ph[s];
					else hprev = le < 64 ? readlane(H[2 * s], le & 63) : readlane(H[2 * s + 1], le & 63);
				}
			}
			const V<int> x1 = shr1(X[s], (int)((unsigned)readlane(X[s], 63) << 16 | (unsigned)px[s]));
			const V<int> v1 = shr1(Vv[s], (int)((unsigned)readlane(Vv[s], 63) << 16 | (unsigned)pv[s]));
			const V<int> x21 = shr1(X2[s], (int)((unsigned)readlane(X2[s], 63) << 16 | (unsigned)px2[s]));
			const V<int> ou = U[s];
			V<int> nu, nv, nx, ny, nx2, ny2, p;
			ksw_pcell(cc, sv, x1, v1, x21, Y[s], ou, Y2[s], nu, nv, nx, ny, nx2, ny2, p);
			if constexpr (CLIP) {              // lanes beyond the hull keep their stale values (they feed back when the band is clipped)
				V<int> m = -1;
				if (top) { WM_KEEP_BRANCH(); m = sel(t_lo <= en, 0x0000ffff, 0) | sel(t_hi <= en, (int)0xffff0000, 0); }
				U[s] = bfi(m, nu, U[s]); Vv[s] = bfi(m, nv, Vv[s]); X[s] = bfi(m, nx, X[s]); Y[s] = bfi(m, ny, Y[s]);
				X2[s] = bfi(m, nx2, X2[s]); Y2[s] = bfi(m, ny2, Y2[s]);
			} else {
				U[s] = nu; Vv[s] = nv; X[s] = nx; Y[s] = ny; X2[s] = nx2; Y2[s] = ny2;
			}
			if (top) {
				WM_IF(t_lo <= en) gst(trow, t_lo - base, cast<uint8_t>(p)); WM_END
				WM_IF(t_hi <= en) gst(trow, t_hi - base, cast<uint8_t>(lshr(p, 16))); WM_END
			} else {
				gst(trow, t_lo - base, cast<uint8_t>(p));
				gst(trow, t_hi - base, cast<uint8_t>(lshr(p, 16)));
			}
			if constexpr (EXACT) {
				// H += v (:320-345). Lanes outside the band keep their H; lane en0 takes H of its left neighbour + u. (Row 0 runs through here as
				// well: its only band lane is rewritten below, and hmax is not used.)
#pragma unroll
				for (int hf = 1; hf >= 0; --hf) {
					const int ci = 2 * s + hf, cb = c0 + 64 * hf;
					const V<int> v8 = hf ? vhi8(Vv[s]) : vlo8(Vv[s]);
					V<int> hn = H[ci] + v8;
					if (cb >= st0 && cb + 63 < en0) {
						H[ci] = hn;
						hmax = vmax(hmax, hn);
					} else {
						WM_KEEP_BRANCH();
						const V<int> t = hf ? t_hi : t_lo;
						const V<int> u8 = hf ? vhi8(U[s]) : vlo8(U[s]);
						hn = sel(t == en0x, V<int>(u8 + hprev), hn);
						const vbool inb = t >= st0 && t <= en0;
						H[ci] = sel(inb, hn, H[ci]);
						hmax = vmax(hmax, sel(inb, hn, V<int>(KSW_NEG_INF)));
					}
				}
			}
		}

		// ---- what the other wavefronts need: next row's neighbour values, and this row's bookkeeping inputs ----
		WM_IF(ln == 63)
#pragma unroll
			for (int s = 0; s < BP; ++s) {
				const int g = s * NWV + wv;
				if (g > NI) continue;              // (pairs up to the one just above the hull: it may join the hull next row)
				int *x = xch + (par * NP + g) * 4;
				gst(x, V<int>(0), lshr(X[s], 16)); gst(x, V<int>(1), lshr(Vv[s], 16)); gst(x, V<int>(2), lshr(X2[s], 16));
				if constexpr (EXACT) gst(x, V<int>(3), H[2 * s + 1]);
			}
		WM_END
		// the lane (uniform) `t` of the window: which wavefront / slot / half / thread holds it
		auto owner = [&](int t, int &slot, int &half, int &thr) { const int o = t - base, c = o >> 6, g = c >> 1; slot = g / NWV; half = c & 1; thr = o & 63; return g % NWV; };
		auto half_of = [&](const V<int> (&a)[BP], int slot, int half, int thr) {
			int rr = 0;
#pragma unroll
			for (int s = 0; s < BP; ++s) if (slot == s) rr = readlane(a[s], thr);
			return half ? rr >> 16 : (int)(short)(rr & 0xffff);
		};
		if constexpr (EXACT) {
			long long kk = -0x7fffffffffffffffLL - 1;
			if (r > 0) {
				const int hm = wave_max_i32(hmax);
				if (hm > KSW_NEG_INF) {
					// The lane that holds the maximum is consumed by a new maximum or by a z-drop test that can fire (ez_max - max_H > zdrop + l * e2 needs
					// ez_max - max_H > zdrop). Each wavefront decides with ITS maximum hm <= max_H: hm > ez_max (it may hold a new maximum), or
					// ez_max - hm > zdrop (a z-drop row needs this of every wavefront); a wavefront for which neither holds cannot be the holder of a
					// maximum anybody looks at, and publishes priority 0 (below every real one). The tie rule (src/ksw2_extd2_sse.c:315-358) as a lane
					// priority, evaluated on the vector unit only in chunks that hold the maximum (see ksw_dp_packed).
					int best_pri = 0;
					if (hm > ez_max || (zdrop >= 0 && ez_max - hm > zdrop)) {
						WM_KEEP_BRANCH();
						const int en1 = st0 + (en0 - st0) / 4 * 4;
						const V<int> g4 = (4 - ((ln + (base - st0)) & 3)) << 20;
						V<int> best = -1;
#pragma unroll
						for (int ci = 0; ci < 2 * BP; ++ci) {
							const int g = (ci >> 1) * NWV + wv, c0 = base + 128 * g + 64 * (ci & 1);
							if (g >= NI) continue;
							const V<int> t = ln + c0;
							const vbool hit = H[ci] == hm && cast<unsigned>(t - st0) <= (unsigned)(en0 - st0);
							if (any(hit)) {
								WM_KEEP_BRANCH();
								V<int> pri = sel(t < en1, g4, V<int>(0));
								pri = sel(t == en0, V<int>(5 << 20), pri) | (V<int>(0xfffff) - t);
								best = vmax(best, sel(hit, pri, V<int>(-1)));
							}
						}
						best_pri = wave_max_i32(best);
					}
					if (best_pri >= 0) kk = (long long)hm * 4294967296LL + (long long)best_pri;
				}
			} else if (wv == 0) {
				WM_IF(ln == 0) H[0] = vlo8(Vv[0]) - qe; WM_END
				kk = (long long)readlane(H[0], 0) * 4294967296LL + (long long)((5 << 20) | 0xfffff);
			}
			int slot, half, thr;
			if (owner(en0, slot, half, thr) == wv) {
				int h = 0;
#pragma unroll
				for (int ci = 0; ci < 2 * BP; ++ci) if (ci == 2 * slot + half) h = readlane(H[ci], thr);
				WM_IF(ln == 0) gst(pubr, V<int>(2 * NWV + 0), V<int>(h)); WM_END
			}
			if (owner(st0, slot, half, thr) == wv) {
				int h = 0;
#pragma unroll
				for (int ci = 0; ci < 2 * BP; ++ci) if (ci == 2 * slot + half) h = readlane(H[ci], thr);
				WM_IF(ln == 0) gst(pubr, V<int>(2 * NWV + 1), V<int>(h)); WM_END
			}
			WM_IF(ln == 0) gst(pubr, V<int>(2 * wv), V<int>((int)(unsigned)(kk & 0xffffffffLL))); gst(pubr, V<int>(2 * wv + 1), V<int>((int)(kk >> 32))); WM_END
		} else {
			int slot, half, thr;
			if (last_H0_t >= base && last_H0_t < base + 128 * NP && owner(last_H0_t, slot, half, thr) == wv) { const int d = half_of(Vv, slot, half, thr) >> 8; WM_IF(ln == 0) gst(pubr, V<int>(2 * NWV + 2), V<int>(d)); WM_END }
			if (last_H0_t + 1 >= base && last_H0_t + 1 < base + 128 * NP && owner(last_H0_t + 1, slot, half, thr) == wv) { const int d = half_of(U, slot, half, thr) >> 8; WM_IF(ln == 0) gst(pubr, V<int>(2 * NWV + 3), V<int>(d)); WM_END }
		}
		block_sync_lds();

		// ---- scalar bookkeeping, identical in every wavefront ----
		if constexpr (EXACT) {
			long long kk = -0x7fffffffffffffffLL - 1;
			for (int w2 = 0; w2 < NWV; ++w2) {
				const long long k2 = (long long)(((unsigned long long)(unsigned)gld(pubr, (long long)(2 * w2 + 1)) << 32) | (unsigned)gld(pubr, (long long)(2 * w2)));
				if (k2 > kk) kk = k2;
			}
			const int max_H = (int)(kk >> 32), pri = (int)(kk & 0xffffffffLL);
			const int max_t = 0xfffff - (pri & 0xfffff);
			if (en0 == tlen - 1) { const int h = gld(pubr, (long long)(2 * NWV)); if (h > ez_mte) ez_mte = h, ez_mte_q = r - en; }
			if (r - st0 == qlen - 1) { const int h = gld(pubr, (long long)(2 * NWV + 1)); if (h > ez_mqe) ez_mqe = h, ez_mqe_t = st0; }
			if (max_H > ez_max) {
				ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
			} else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
				const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
				if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; break; }
			}
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = gld(pubr, (long long)(2 * NWV));
		} else {
			if (r > 0) {
				const bool in0 = last_H0_t >= st0 && last_H0_t <= en0, in1 = last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0;
				if (in0 && in1) {
					const int d0 = gld(pubr, (long long)(2 * NWV + 2)), d1 = gld(pubr, (long long)(2 * NWV + 3));
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (in0) H0 += gld(pubr, (long long)(2 * NWV + 2));
				else { ++last_H0_t; H0 += gld(pubr, (long long)(2 * NWV + 3)); }
			} else H0 = gld(pubr, (long long)(2 * NWV + 2)) - qe, last_H0_t = 0;
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = H0;
		}
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

} // namespace wmk
