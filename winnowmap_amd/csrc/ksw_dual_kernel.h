// ksw_dual_kernel.h — ksw_extd2_sse (src/ksw2_extd2_sse.c:26-393) with TWO ALIGNMENTS PER WAVEFRONT (round 6), for the bulk of the DP
// cells: gap fills whose band never clips (w >= qlen, tlen), without ambiguous bases, with the approximate maximum (KSW_EZ_APPROX_MAX).
//
// ksw_dp_packed (ksw_packed_kernel.h) keeps chunk 2i of ONE alignment in the low 16-bit halves of register i and chunk 2i + 1 in the high halves: its
// granularity is 128 lanes, and the median gap fill (a hull of 100..300 lanes that grows from one lane and shrinks back to one) leaves a third of the lane
// slots outside the hull, while the row's scalar bookkeeping (hull arithmetic, query fetch, boundary lane, traceback row address: ~60 scalar instructions)
// is paid per wavefront and row. Here the low halves belong to alignment A and the high halves to alignment B — register i = chunk i (64 lanes) of BOTH —
// so that one packed instruction still advances 128 cells, but of two alignments whose hulls are 64-lane granular, and one row loop serves both:
//   * everything uniform across the halves is shared: the cell arithmetic (ksw_pcell: the reference's tie-break tags are per-half CONSTANTS, so the two
//     alignments may even differ in KSW_EZ_RIGHT), the neighbour shift (one DPP rotation per register and one v_cndmask: thread 0 takes thread 63 of the
//     register below, in both halves at once), the gap-open schedule of row r;
//   * per alignment: its hull [st0, en0], its window base (re-based by 16 lanes independently: a half-masked v_bfi after the rotation), its packed target /
//     query codes (one byte per chunk, as in ksw_dp_packed; the scores of register i take byte i of A's words into the low half and of B's into the high
//     half with ONE v_perm_b32), its boundary lane, its traceback rows (the byte layout is unchanged: row r at tb + r * n_col, column t - st, so
//     ksw_backtrack_thread and everything downstream are shared) and its H0 track along the hull's first lane (ksw_packed_kernel.h: WM_KSW_EDGE_TRACK);
//   * the row loop runs to the longer alignment's last row; the shorter one simply has no chunks after its own (no stores, no track): the launcher pairs
//     neighbours of the size-sorted job table.
// With an unclipped band lanes outside the hull never feed a band cell and are never read by the backtrack (ksw_packed_kernel.h), so neither the state
// update nor the other alignment's wider hull needs lane masks; only the chunk that holds a hull's end masks its stores.
#pragma once
#ifndef WM_DEV
#error "include simt.h before ksw_dual_kernel.h"
#endif
#include "ksw_packed_kernel.h"

namespace wmk {

// scalar state of one of the two alignments
struct ksw_dual_side_t {
	int qlen, tlen, n_rows, n_col, base, qb0, H0, score;
	int tA, tA2;                                     // (16-bit patterns of the lane-0 fills)
	const uint8_t *query, *target;
	uint8_t *tbp;
	// per row
	int st0, en, NI, bo, f_x, f_v, f_x2;
	uint8_t *trow;
};

template <int NC> WM_DEV void dual_rebase_half(V<int> (&a)[NC], const V<int> fresh, const vbool low48, int hm)
{
	V<int> cur = rot_down(a[0], 16);
#pragma unroll
	for (int i = 0; i < NC; ++i) {
		const V<int> nxt = i + 1 < NC ? rot_down(a[i + 1 < NC ? i + 1 : i], 16) : fresh;
		a[i] = bfi(V<int>(hm), sel(low48, cur, nxt), a[i]);
		cur = nxt;
	}
}

template <int NC>
WM_DEV void ksw_dp_dual(const wm_ksw_score_t sc, const wm_ksw_djob_t jbA, const wm_ksw_djob_t jbB, const bool hasB, const uint8_t *__restrict__ seqs,
                        uint8_t *__restrict__ tb_arena, wm_ksw_dres_t *__restrict__ resA, wm_ksw_dres_t *__restrict__ resB)
{
	constexpr int NW = NC / 4;                 // packed-code words per alignment (byte k of word w = chunk 4w + k)
	static_assert(NC == 4 || NC == 8 || NC == 16, "NC");
	const int q = sc.q, e = sc.e, q2 = sc.q2, e2 = sc.e2, qe = q + e, qe2 = q2 + e2;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const bool rightA = (jbA.flag & KSW_F_RIGHT) != 0, rightB = hasB && (jbB.flag & KSW_F_RIGHT) != 0;
	auto mk = [](int lo, int hi) { return (int)((((unsigned)hi & 0xffffu) << 16) | ((unsigned)lo & 0xffffu)); };      // low half: alignment A, high half: B
	auto tags = [](bool right, int &tS, int &tA, int &tA2, int &tB2) { tS = right ? 0 : 4; tA = right ? 1 : 3; tA2 = right ? 3 : 1; tB2 = right ? 4 : 0; };
	int tS0, tA0, tA20, tB20, tS1, tA1, tA21, tB21;
	tags(rightA, tS0, tA0, tA20, tB20); tags(rightB, tS1, tA1, tA21, tB21);
	const int tB = 2;
	const int hA0 = rightA ? tA0 - 1 : tA0, hB0 = rightA ? tB - 1 : tB, hA20 = rightA ? tA20 - 1 : tA20, hB20 = rightA ? tB20 - 1 : tB20;
	const int hA1 = rightB ? tA1 - 1 : tA1, hB1 = rightB ? tB - 1 : tB, hA21 = rightB ? tA21 - 1 : tA21, hB21 = rightB ? tB21 - 1 : tB21;
	const int MCHb = ((int)sc.match & 0xff) << 8, MISb = ((int)sc.mismatch & 0xff) << 8;
	const int MCH2 = mk(MCHb | tS0, MCHb | tS1);
	const int one2 = (int)sc.match > -128 ? 0x00010001 : 0x00020002;       // (opaque to the compiler, as in ksw_dp_packed)
	const ksw_pcell_cst_t cc = { tb16(qe), tb16(qe2), tb16(q), tb16(q2), tb16(sc.match), mk(tA0, tA1), mk(tB, tB), mk(tA20, tA21), mk(tB20, tB21),
	                             mk(hA0, hA1), mk(hB0, hB1), mk(hA20, hA21), mk(hB20, hB21) };

	const V<int> ln = lane();
	const vbool low48 = ln < 48, is0 = ln == 0;
	const V<int> qsel = sel(is0, 0x06050403, 0x07060504);          // code words: thread 0 takes {own bytes 2..0, prev byte 3}
	ksw_dual_side_t A, B;
	auto side_init = [&](ksw_dual_side_t &S, const wm_ksw_djob_t &jb, bool present, int tA_, int tA2_) {
		S.qlen = present ? jb.qlen : 0; S.tlen = present ? jb.tlen : 0; S.n_rows = present ? jb.qlen + jb.tlen - 1 : 0; S.n_col = jb.n_col;
		S.base = 0; S.qb0 = -(1 << 30); S.H0 = 0; S.score = KSW_NEG_INF; S.tA = tA_; S.tA2 = tA2_;
		S.query = seqs + jb.q_off; S.target = seqs + jb.t_off; S.tbp = tb_arena + jb.tb_off;
		S.st0 = 0; S.en = -1; S.NI = 0; S.bo = -1; S.f_x = S.f_v = S.f_x2 = 0; S.trow = S.tbp;
	};
	side_init(A, jbA, true, tA0, tA20); side_init(B, jbB, hasB, tA1, tA21);

	V<int> U[NC], Vv[NC], X[NC], Y[NC], X2[NC], Y2[NC];
	V<int> TP[2][NW], QP[2][NW], QB[2];
#pragma unroll
	for (int i = 0; i < NC; ++i) { U[i] = tb16(-qe); Vv[i] = tb16(-qe); X[i] = cc.tA; Y[i] = cc.tB; X2[i] = cc.tA2; Y2[i] = cc.tB2; }
	auto codes_init = [&](auto JX, const ksw_dual_side_t &S) {
		constexpr int jx = decltype(JX)::value;
#pragma unroll
		for (int wd = 0; wd < NW; ++wd) {
			V<int> pk = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				const V<int> t = ln + 64 * (wd * 4 + b);
				V<int> c = 0;
				WM_IF(t < S.tlen) c = cast<int>(gld(S.target, t)); WM_END
				pk = pk | (c << (8 * b));
			}
			TP[jx][wd] = pk; QP[jx][wd] = 0;
		}
		QB[jx] = 0;
	};
	codes_init(std::integral_constant<int, 0>{}, A); codes_init(std::integral_constant<int, 1>{}, B);

	const int max_rows = A.n_rows > B.n_rows ? A.n_rows : B.n_rows;
	for (int r = 0; r < max_rows; ++r) {
		const int sched = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		// ---- per alignment: hull, re-base, query codes of row r, boundary lane, traceback row ----
		auto side_row = [&](auto JX, ksw_dual_side_t &S) {
			constexpr int jx = decltype(JX)::value;
			constexpr int hm = jx ? (int)0xffff0000 : 0x0000ffff;
			if (r >= S.n_rows) { S.NI = 0; S.bo = -1; return; }
			int st0 = r - S.qlen + 1, en0 = S.tlen - 1;
			if (st0 < 0) st0 = 0;
			if (en0 > r) en0 = r;
			const int st = st0 & ~15, en = ((en0 + 16) & ~15) - 1;
			S.st0 = st0; S.en = en;
			S.f_x = S.tA; S.f_v = ((st == 0 ? sched : -qe) & 0xff) << 8; S.f_x2 = S.tA2;
			if (st > S.base) {
				WM_EMU_ASSERT(st == S.base + 16);
				const int rx = readlane(X[0], 15), rv = readlane(Vv[0], 15), rx2 = readlane(X2[0], 15);
				S.f_x = (jx ? lshr(rx, 16) : rx) & 0xffff; S.f_v = (jx ? lshr(rv, 16) : rv) & 0xffff; S.f_x2 = (jx ? lshr(rx2, 16) : rx2) & 0xffff;
				dual_rebase_half<NC>(U, V<int>(tb16(-qe)), low48, hm); dual_rebase_half<NC>(Vv, V<int>(tb16(-qe)), low48, hm);
				dual_rebase_half<NC>(X, V<int>(cc.tA), low48, hm); dual_rebase_half<NC>(Y, V<int>(cc.tB), low48, hm);
				dual_rebase_half<NC>(X2, V<int>(cc.tA2), low48, hm); dual_rebase_half<NC>(Y2, V<int>(cc.tB2), low48, hm);
				{   // packed characters (as in ksw_dp_packed): the fresh top 16 lanes take target codes from memory and the query codes of row r - 1
					const V<int> tnew = ln + (st + 64 * (NC - 1));
					V<int> c = 0, d = 0;
					WM_IF(!low48)
						WM_IF(tnew < S.tlen) c = cast<int>(gld(S.target, tnew)); WM_END
						const V<int> qi = (r - 1) - tnew;
						WM_IF(qi >= 0 && qi < S.qlen) d = cast<int>(gld(S.query, qi)); WM_END
					WM_END
					V<int> rt[NW], rq[NW];
#pragma unroll
					for (int wd = 0; wd < NW; ++wd) { rt[wd] = rot_down(TP[jx][wd], 16); rq[wd] = rot_down(QP[jx][wd], 16); }
#pragma unroll
					for (int wd = 0; wd < NW; ++wd) {
						const V<int> nt = wd + 1 < NW ? rt[wd + 1 < NW ? wd + 1 : wd] : c, nq = wd + 1 < NW ? rq[wd + 1 < NW ? wd + 1 : wd] : d;
						const V<int> ct = cast<int>((cast<unsigned>(rt[wd]) >> 8) | (cast<unsigned>(nt) << 24));
						const V<int> cq = cast<int>((cast<unsigned>(rq[wd]) >> 8) | (cast<unsigned>(nq) << 24));
						TP[jx][wd] = sel(low48, rt[wd], ct); QP[jx][wd] = sel(low48, rq[wd], cq);
					}
				}
				S.base = st;
			}
			{   // every lane takes the query code of lane t - 1; the first lane of the window takes query[r - base]
				const int qi0 = r - S.base;
				int newc = 0;
				if (qi0 < S.qlen) {
					WM_EMU_ASSERT(qi0 >= 0);
					if (qi0 < S.qb0 || qi0 >= S.qb0 + 64) {
						S.qb0 = qi0 < 16 ? 0 : qi0 - 16;
						const V<int> qidx = ln + S.qb0;
						QB[jx] = 0;
						WM_IF(qidx < S.qlen) QB[jx] = cast<int>(gld(S.query, qidx)); WM_END
						loads_land();
					}
					newc = readlane(QB[jx], qi0 - S.qb0);
				}
				V<int> rq[NW];
#pragma unroll
				for (int wd = 0; wd < NW; ++wd) rq[wd] = ror1(QP[jx][wd]);
#pragma unroll
				for (int wd = 0; wd < NW; ++wd) QP[jx][wd] = perm(rq[wd], wd ? rq[wd ? wd - 1 : 0] : V<int>(newc << 24), qsel);
			}
			S.bo = en >= r ? r - S.base : -1;                          // first-column / first-row boundary lane r (:152-155), when it is inside the window's hull chunk
			S.NI = ((en - S.base) >> 6) + 1;
			WM_EMU_ASSERT(S.NI <= NC && (S.bo < 0 || (S.bo >> 6) == S.NI - 1));
			S.trow = S.tbp + (size_t)r * S.n_col;
		};
		side_row(std::integral_constant<int, 0>{}, A); side_row(std::integral_constant<int, 1>{}, B);
		const int NImax = A.NI > B.NI ? A.NI : B.NI;
		V<int> bmA = 0, bmB = 0;
		if (A.bo >= 0) bmA = sel(ln == (A.bo & 63), 0x0000ffff, 0);
		if (B.bo >= 0) bmB = sel(ln == (B.bo & 63), (int)0xffff0000, 0);
		const V<int> Fx = V<int>(mk(A.f_x, B.f_x)), Fv = V<int>(mk(A.f_v, B.f_v)), Fx2 = V<int>(mk(A.f_x2, B.f_x2));
		V<int> xq[2][NW];
#pragma unroll
		for (int wd = 0; wd < NW; ++wd) { xq[0][wd] = TP[0][wd] ^ QP[0][wd]; xq[1][wd] = TP[1][wd] ^ QP[1][wd]; }
		V<int> crx = 0, crv = 0, crx2 = 0;                             // rotations handed from register i + 1 to register i

		auto chunk_body = [&](auto IC) {
			constexpr int i = decltype(IC)::value;
			constexpr int wd = i >> 2, kb = i & 3;
			constexpr int psel = (0x0c << 24) | ((4 + kb) << 16) | (0x0c << 8) | kb;      // {B's byte kb -> high half, A's byte kb -> low half}
			const bool topA = i == A.NI - 1, topB = i == B.NI - 1;
			const V<int> sv = pk_mad(pk_minu(perm(xq[1][wd], xq[0][wd], psel), one2), rep16(MISb - MCHb), MCH2);
			if (topA || topB) {
				WM_KEEP_BRANCH();
				V<int> bm = 0;
				if (topA) bm = bmA;
				if (topB) bm = bm | bmB;
				Y[i] = bfi(bm, V<int>(cc.tB), Y[i]); Y2[i] = bfi(bm, V<int>(cc.tB2), Y2[i]); U[i] = bfi(bm, V<int>(tb16(sched)), U[i]);
			}
			// previous-row values of lane t - 1: rotate by one thread; thread 0 takes thread 63 of the register below (or the window's fill), both halves at once
			if (i == NImax - 1) { WM_KEEP_BRANCH(); crx = ror1(X[i]); crv = ror1(Vv[i]); crx2 = ror1(X2[i]); }
			const V<int> rxo = crx, rvo = crv, rx2o = crx2;
			if constexpr (i > 0) { crx = ror1(X[i ? i - 1 : 0]); crv = ror1(Vv[i ? i - 1 : 0]); crx2 = ror1(X2[i ? i - 1 : 0]); }
			else { crx = Fx; crv = Fv; crx2 = Fx2; }
			const V<int> x1 = sel(is0, crx, rxo), v1 = sel(is0, crv, rvo), x21 = sel(is0, crx2, rx2o);
			V<int> nu, nv, nx, ny, nx2, ny2, p;
			ksw_pcell(cc, sv, x1, v1, x21, Y[i], U[i], Y2[i], nu, nv, nx, ny, nx2, ny2, p);
			U[i] = nu; Vv[i] = nv; X[i] = nx; Y[i] = ny; X2[i] = nx2; Y2[i] = ny2;
			if (i < A.NI) {
				if (topA) { WM_IF(ln + (A.base + 64 * i) <= A.en) gst(A.trow, ln + 64 * i, cast<uint8_t>(p)); WM_END }
				else gst(A.trow, ln + 64 * i, cast<uint8_t>(p));
			}
			if (i < B.NI) {
				if (topB) { WM_IF(ln + (B.base + 64 * i) <= B.en) gst(B.trow, ln + 64 * i, cast<uint8_t>(lshr(p, 16))); WM_END }
				else gst(B.trow, ln + 64 * i, cast<uint8_t>(lshr(p, 16)));
			}
		};
		static_for_desc<NC>([&](auto IC) {
			if (decltype(IC)::value < NImax) chunk_body(IC);
		});

		// ---- the approximate maximum along the hull's first lane (ksw_packed_kernel.h: WM_KSW_EDGE_TRACK): one v_readlane pair per alignment and row ----
		auto side_track = [&](auto JX, ksw_dual_side_t &S) {
			constexpr int jx = decltype(JX)::value;
			if (r >= S.n_rows) return;
			const int o = S.st0 - S.base;                                  // 0..15: chunk 0
			const int rv = readlane(Vv[0], o), ru = readlane(U[0], o);
			const int w16 = r < S.qlen ? rv : ru;
			const int d = ((int)(short)((jx ? lshr(w16, 16) : w16) & 0xffff)) >> 8;
			S.H0 = r ? S.H0 + d : d - qe;
			if (r == S.n_rows - 1) S.score = S.H0;
		};
		side_track(std::integral_constant<int, 0>{}, A); side_track(std::integral_constant<int, 1>{}, B);
	}

	auto side_result = [&](const ksw_dual_side_t &S, const wm_ksw_djob_t &jb, wm_ksw_dres_t *res) {
		int bt_i = -1, bt_j = -1;
		if (!(jb.flag & KSW_F_EXTZ_ONLY)) bt_i = S.tlen - 1, bt_j = S.qlen - 1;      // (no exact maximum here: an extension-only job without one has nothing to start from, as in ksw_dp_packed)
		WM_IF(ln == 0)
			wm_ksw_dres_t o;
			o.max = 0; o.zdropped = 0; o.max_q = -1; o.max_t = -1;
			o.mqe = KSW_NEG_INF; o.mqe_t = -1; o.mte = KSW_NEG_INF; o.mte_q = -1;
			o.score = S.score; o.reach_end = 0; o.n_cigar = 0; o.bt_i = bt_i; o.bt_j = bt_j;
			*res = o;
		WM_END
	};
	side_result(A, jbA, resA);
	if (hasB) side_result(B, jbB, resB);
}

} // namespace wmk
