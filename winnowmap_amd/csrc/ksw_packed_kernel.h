// ksw_packed_kernel.h — ksw_extd2_sse (src/ksw2_extd2_sse.c:26-393) as a wave64 register machine that evaluates TWO
// DP cells per 32-bit lane with packed 16-bit integer instructions (VOP3P: v_pk_add_u16 / v_pk_sub_u16 / v_pk_max_i16 /
// v_pk_min_i16 / v_pk_sub_i16 clamp / v_pk_mad_u16 — full rate on gfx950, profiles/r02_valu_bench.txt).
//
// One wavefront owns one alignment, walks the anti-diagonals r = i + j and
// keeps the reference's persistent per-target-lane int8 state (u, v, x, y, x2, y2, and the score row when the band clips)
// for a sliding window of lanes that starts at the 16-aligned hull start `base`. What changes is the representation:
//
//   * every int8 value lives in the HIGH BYTE of a 16-bit half (v << 8): 16-bit add / sub wrap exactly like
//     _mm_add_epi8 / _mm_sub_epi8 and signed 16-bit compares agree with _mm_max_epi8 / _mm_cmpgt_epi8; the low bits
//     carry the 3-bit tie-break tag that turns the reference's five-way state choice (:227-234 / :274-281) into plain
//     maxima, exactly as in the 32-bit kernels;
//   * the window is STRIPED over chunk PAIRS: register i of a state array holds chunk 2i (lanes base+128i+j) in its low
//     halves and chunk 2i+1 (lanes base+128i+64+j) in its high halves, thread j = 0..63. One packed instruction
//     therefore advances 128 cells; the left neighbour (lane t-1 of the previous row) is one DPP shift of the packed
//     register whose lane-0 fill is assembled on the scalar unit from lane 63 of this register (for the high half) and of
//     register i-1 (for the low half);
//   * a row only executes the pairs that intersect the hull; with an unclipped band (CLIP = false) lanes outside the band
//     never feed a band cell and are never read by the backtrack, so the state update needs no lane masking at all —
//     only the traceback stores are masked;
//   * the traceback byte layout is unchanged (code<<4 | ext bits, row r at tb + r*n_col, column t - st), so
//     ksw_backtrack_thread and everything downstream are shared with the other kernels; the high half is stored with
//     global_store_byte_d16_hi, no unpacking;
//   * EXACT selects the exact row maximum (H per lane in 32 bits, z-drop, mqe/mte) or the reference's approximate
//     one-track maximum (KSW_EZ_APPROX_MAX, :359-375): gap filling — most of the cells — never carries H or S.
//
// VALU work per pair and row, counted in the gfx950 code object (CLIP = HASN = EXACT = false, a pair inside the hull): 35 packed ops for the
// two cells and their traceback byte + 6 (new u, v and the four gap states) + 5 (scores) + 15 (neighbours: 6 v_readlane, 3 v_mov, 3
// v_alignbit, 3 DPP) = 61 per 128 cells (76 before the round-2 restructuring; 53 with WM_KSW_ROR=1, where the neighbours cost 3 DPP
// rotations + 3 byte permutes). EXACT adds 2 per chunk inside the band (H += v through SDWA, running maximum). Only the pair that holds the
// hull end (`top`) pays for lane masks, the boundary lane and predicated stores. Registers: 6 per pair (+1 with CLIP, +2 with EXACT).
//
// Code generation (profiles/r02z_codegen_flag.txt): the library is built with -mllvm -disable-promote-alloca-to-vector. The state arrays are
// indexed by fully unrolled loops; AMDGPUPromoteAllocaToVector runs before the unroller and would turn each array into one <N x i32>
// value that every uniform branch copies as a whole at its join (1536 v_mov_b64 in the EXACT 16-pair kernel). For the same reason the pair
// loop is a compile-time loop (static_for_desc) and its uniform special cases are real branches (WM_KEEP_BRANCH) around small blocks.
#pragma once
#ifndef WM_DEV
#error "include simt.h before ksw_packed_kernel.h"
#endif
#include "ksw_kernel.h"
#ifndef WM_KSW_ROR
#define WM_KSW_ROR 1          // 1 (default since round 3: +4 % in the isolated probe, +2 % in the bench, profiles/r03a_first_run.txt): neighbour values through wave_ror:1 + v_perm_b32; 0: v_readlane + scalar fill
#endif
#ifndef WM_KSW_EDGE_TRACK
#define WM_KSW_EDGE_TRACK 1   // 1: unclipped approximate-max jobs (the bulk of the gap fills) follow the hull's first lane instead of the reference's greedy
                              // track (same H at the end, see below) and skip the band terms of st0 / en0; 0: the round-3a code, for A/B runs
#endif
#include <type_traits>
#include <utility>

namespace wmk {

// f(integral_constant<int, N-1>) ... f(integral_constant<int, 0>): a loop whose index is a constant expression inside the body
template <class F, int... Is> WM_DEV void static_for_desc_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, (int)sizeof...(Is) - 1 - Is>{}), ...); }
template <int N, class F> WM_DEV void static_for_desc(F &&f) { static_for_desc_impl(f, std::make_integer_sequence<int, N>{}); }

// int8 value -> high byte of both 16-bit halves; 16-bit pattern -> both halves
WM_DEV int tb16(int v) { const unsigned h = ((unsigned)v & 0xffu) << 8; return (int)(h | h << 16); }
WM_DEV int rep16(int v) { const unsigned h = (unsigned)v & 0xffffu; return (int)(h | h << 16); }
// int8 value (sign-extended) held by the low / high half
template <class T> WM_DEV T vlo8(T x) { return (x << 16) >> 24; }
template <class T> WM_DEV T vhi8(T x) { return x >> 24; }

struct ksw_pcell_cst_t { int QE, QE2, Q, Q2, MCH, tA, tB, tA2, tB2, hA, hB, hA2, hB2; };

// two DP cells per lane (src/ksw2_extd2_sse.c:205-311): inputs are previous-row values, outputs the new u, v, x, y, x2, y2 and the
// traceback bytes (bits 0-7 of each half)
WM_DEV void ksw_pcell(const ksw_pcell_cst_t &c, const V<int> os, const V<int> x1, const V<int> v1, const V<int> x21, const V<int> oy, const V<int> ou, const V<int> oy2,
                      V<int> &nu, V<int> &nv, V<int> &nx, V<int> &ny, V<int> &nx2, V<int> &ny2, V<int> &p)
{
	const V<int> v1q = pk_sub(v1, c.QE), v1q2 = pk_sub(v1, c.QE2), ouq = pk_sub(ou, c.QE), ouq2 = pk_sub(ou, c.QE2);
	V<int> a = pk_add(x1, v1q), b = pk_add(oy, ouq), a2 = pk_add(x21, v1q2), b2 = pk_add(oy2, ouq2);
	const V<int> zz = pk_max(pk_max(pk_max(pk_max(os, a), b), a2), b2);
	const V<int> z = pk_min(zz & (int)0xff00ff00, c.MCH);
	p = zz & 0x00070007;
	nu = pk_sub(z, v1); nv = pk_sub(z, ou);
	const V<int> tmp = pk_sub(z, c.Q), tmp2 = pk_sub(z, c.Q2);
	a = pk_sub(a, tmp); b = pk_sub(b, tmp); a2 = pk_sub(a2, tmp2); b2 = pk_sub(b2, tmp2);
	// "gap continues" flags: the sign of sat(h - a) is the compare a > h; p = 2p + flag
	// (p holds at most 7 bits per half, so the shifts can run on the whole word: v_lshl_or_b32)
	p = (p << 1) | pk_lshr(pk_subsat(c.hA, a), 15);
	p = (p << 1) | pk_lshr(pk_subsat(c.hB, b), 15);
	p = (p << 1) | pk_lshr(pk_subsat(c.hA2, a2), 15);
	p = (p << 1) | pk_lshr(pk_subsat(c.hB2, b2), 15);
	nx = pk_max(a, c.tA); ny = pk_max(b, c.tB); nx2 = pk_max(a2, c.tA2); ny2 = pk_max(b2, c.tB2);
}

// the 16-bit half (sign-extended) that holds lane t (uniform) of a packed state array
template <int BP> WM_DEV int get_half(const V<int> (&a)[BP], int base, int t)
{
	const int o = t - base, jr = o & 63, c = o >> 6, ip = c >> 1;
	WM_EMU_ASSERT(o >= 0 && o < 128 * BP);
	int r = 0;
#pragma unroll
	for (int i = 0; i < BP; ++i)
		if (ip == i) r = readlane(a[i], jr);
	return (c & 1) ? r >> 16 : (int)(short)(r & 0xffff);
}

// window re-base (+16 lanes): every chunk rotates down by 16 threads; the top 16 threads of chunk c take the low 16 threads of
// chunk c+1 — for a packed register that is {hi: next register's low half, lo: this register's high half}
template <int BP> WM_DEV void rebase_packed(V<int> (&a)[BP], const V<int> fresh, const vbool low48)
{
	V<int> cur = rot_down(a[0], 16);
#pragma unroll
	for (int i = 0; i < BP; ++i) {
		const V<int> nxt = i + 1 < BP ? rot_down(a[i + 1 < BP ? i + 1 : i], 16) : fresh;
		a[i] = sel(low48, cur, alignbit(nxt, cur, 16));
		cur = nxt;
	}
}

template <int BP, bool CLIP, bool HASN, bool EXACT>
WM_DEV void ksw_dp_packed(const wm_ksw_score_t sc, const wm_ksw_djob_t jb, const uint8_t *__restrict__ seqs,
                          uint8_t *__restrict__ tb_arena, wm_ksw_dres_t *__restrict__ res)
{
	constexpr int B = 2 * BP;                  // 64-lane chunks in the window
	constexpr int NW = B / 4;                  // packed-code words (byte k of word w = chunk 4w+k)
	static_assert(BP == 2 || BP == 4 || BP == 8 || BP == 16, "BP");
	const int qlen = jb.qlen, tlen = jb.tlen, flag = jb.flag, zdrop = jb.zdrop;
	const int w = jb.w < 0 ? (tlen > qlen ? tlen : qlen) : jb.w;
	const bool right = (flag & KSW_F_RIGHT) != 0;
	WM_EMU_ASSERT(EXACT == !(flag & KSW_F_APPROX_MAX));
	const uint8_t *query = seqs + jb.q_off, *target = seqs + jb.t_off;
	uint8_t *tbp = tb_arena + jb.tb_off;
	const int q = sc.q, e = sc.e, q2 = sc.q2, e2 = sc.e2, qe = q + e, qe2 = q2 + e2;
	const int tS = right ? 0 : 4, tA = right ? 1 : 3, tB = 2, tA2 = right ? 3 : 1, tB2 = right ? 4 : 0;
	const int hA = right ? tA - 1 : tA, hB = right ? tB - 1 : tB, hA2 = right ? tA2 - 1 : tA2, hB2 = right ? tB2 - 1 : tB2;
	const int MCHt = (((int)sc.match & 0xff) << 8) | tS, MISt = (((int)sc.mismatch & 0xff) << 8) | tS;
	const int NNt = (((sc.sc_ambi == 0 ? -e2 : (int)sc.sc_ambi) & 0xff) << 8) | tS;
	const int one2 = (int)sc.match > -128 ? 0x00010001 : 0x00020002;       // 1 | 1 << 16, opaque to the compiler: min(x, 1) * k + c would otherwise become compares and selects
	const ksw_pcell_cst_t cc = { tb16(qe), tb16(qe2), tb16(q), tb16(q2), tb16(sc.match), rep16(tA), rep16(tB), rep16(tA2), rep16(tB2),
	                             rep16(hA), rep16(hB), rep16(hA2), rep16(hB2) };
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	const V<int> ln = lane();
	const vbool low48 = ln < 48;
#if WM_KSW_ROR
	const V<int> rsel = sel(ln == 0, 0x05040302, 0x07060504);      // v_perm_b32 selectors: {own.lo, prev.hi} for thread 0, own elsewhere
	const V<int> qsel = sel(ln == 0, 0x06050403, 0x07060504);      // code words: thread 0 takes {own bytes 2..0, prev byte 3}
#endif
	int base = 0;
	V<int> U[BP], Vv[BP], X[BP], Y[BP], X2[BP], Y2[BP];
	V<int> S[CLIP ? BP : 1];
	V<int> H[EXACT ? B : 1];
	V<int> TP[NW], QP[NW];
#pragma unroll
	for (int i = 0; i < BP; ++i) {
		U[i] = tb16(-qe); Vv[i] = tb16(-qe); X[i] = rep16(tA); Y[i] = rep16(tB); X2[i] = rep16(tA2); Y2[i] = rep16(tB2);
		if constexpr (CLIP) S[i] = rep16(tS);
	}
	if constexpr (EXACT) {
#pragma unroll
		for (int i = 0; i < B; ++i) H[i] = KSW_NEG_INF;
	}
#pragma unroll
	for (int wd = 0; wd < NW; ++wd) {
		V<int> pk = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const V<int> t = ln + 64 * (wd * 4 + b);
			V<int> c = 0;
			WM_IF(t < tlen) c = cast<int>(gld(target, t)); WM_END
			pk = pk | (c << (8 * b));
		}
		TP[wd] = pk; QP[wd] = 0;
	}
	// query codes are fetched 64 at a time: lane l of QB holds query[qb0 + l]
	V<int> QB = 0;
	int qb0 = -(1 << 30);

	int ez_max = 0, ez_zdropped = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1;
	int ez_mte = KSW_NEG_INF, ez_mte_q = -1, ez_score = KSW_NEG_INF;
	int H0 = 0, last_H0_t = 0, Hbelow = KSW_NEG_INF;
	const int n_rows = qlen + tlen - 1;

	for (int r = 0; r < n_rows; ++r) {
		int st0 = 0, en0 = tlen - 1;
		if (st0 < r - qlen + 1) st0 = r - qlen + 1;
		if (en0 > r) en0 = r;
		if (CLIP || !WM_KSW_EDGE_TRACK) {          // (CLIP = false: w >= qlen, tlen — the band terms never bind and the hull is never empty, ksw_plan.h)
			if (st0 < (r - w + 1) >> 1) st0 = (r - w + 1) >> 1;
			if (en0 > (r + w) >> 1) en0 = (r + w) >> 1;
			if (st0 > en0) { ez_zdropped = 1; break; }
		}
		const int st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
		const int sched = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;

		// previous-row values of lane st-1 for the first lane of the hull (:141-151), as 16-bit patterns: constants unless the hull
		// start just moved up, in which case lane st-1 is old chunk 0, thread 15
		int f_x = tA, f_v = ((st == 0 ? sched : -qe) & 0xff) << 8, f_x2 = tA2;
		if (st > base) {
			WM_EMU_ASSERT(st == base + 16);
			f_x = readlane(X[0], 15) & 0xffff; f_v = readlane(Vv[0], 15) & 0xffff; f_x2 = readlane(X2[0], 15) & 0xffff;
			if constexpr (EXACT) Hbelow = readlane(H[0], 15);
			rebase_packed<BP>(U, V<int>(tb16(-qe)), low48); rebase_packed<BP>(Vv, V<int>(tb16(-qe)), low48);
			rebase_packed<BP>(X, V<int>(rep16(tA)), low48); rebase_packed<BP>(Y, V<int>(rep16(tB)), low48);
			rebase_packed<BP>(X2, V<int>(rep16(tA2)), low48); rebase_packed<BP>(Y2, V<int>(rep16(tB2)), low48);
			if constexpr (CLIP) rebase_packed<BP>(S, V<int>(rep16(tS)), low48);
			if constexpr (EXACT) rebase_striped<B>(H, V<int>(KSW_NEG_INF), low48);
			{   // packed characters: the fresh top 16 lanes take target codes from memory and the query codes of row r-1
				const V<int> tnew = ln + (st + 64 * (B - 1));
				V<int> c = 0, d = 0;
				WM_IF(!low48)
					WM_IF(tnew < tlen) c = cast<int>(gld(target, tnew)); WM_END
					const V<int> qi = (r - 1) - tnew;
					WM_IF(qi >= 0 && qi < qlen) d = cast<int>(gld(query, qi)); WM_END
				WM_END
				V<int> rt[NW], rq[NW];
#pragma unroll
				for (int wd = 0; wd < NW; ++wd) { rt[wd] = rot_down(TP[wd], 16); rq[wd] = rot_down(QP[wd], 16); }
#pragma unroll
				for (int wd = 0; wd < NW; ++wd) {
					const V<int> nt = wd + 1 < NW ? rt[wd + 1 < NW ? wd + 1 : wd] : c, nq = wd + 1 < NW ? rq[wd + 1 < NW ? wd + 1 : wd] : d;
					const V<int> ct = cast<int>((cast<unsigned>(rt[wd]) >> 8) | (cast<unsigned>(nt) << 24));
					const V<int> cq = cast<int>((cast<unsigned>(rq[wd]) >> 8) | (cast<unsigned>(nq) << 24));
					TP[wd] = sel(low48, rt[wd], ct); QP[wd] = sel(low48, rq[wd], cq);
				}
			}
			base = st;
		}

		// ---- advance the query codes to row r: every lane takes the code of lane t-1; the first lane of the window takes query[r - base]
		{
			const int qi0 = r - base;
			int newc = 0;
			if (qi0 < qlen) {                              // (qi0 >= 0: base <= st0 <= r)
				WM_EMU_ASSERT(qi0 >= 0);
				if (qi0 < qb0 || qi0 >= qb0 + 64) {
					qb0 = qi0 < 16 ? 0 : qi0 - 16;          // a re-base steps the index back by 16: keep that much behind
					const V<int> qidx = ln + qb0;
					QB = 0;
					WM_IF(qidx < qlen) QB = cast<int>(gld(query, qidx)); WM_END
					loads_land();               // (here, once per 48+ rows — not at the join below, where the wait would also drain every row's traceback stores)
				}
				newc = readlane(QB, qi0 - qb0);
			}
#if WM_KSW_ROR
			// rotate every word by one thread; thread 0 (which received thread 63) shifts its four chunk bytes up by one and takes byte 3 of the
			// word below (or the new code) as byte 0
			V<int> rq[NW];
#pragma unroll
			for (int wd = 0; wd < NW; ++wd) rq[wd] = ror1(QP[wd]);
#pragma unroll
			for (int wd = 0; wd < NW; ++wd) QP[wd] = perm(rq[wd], wd ? rq[wd ? wd - 1 : 0] : V<int>(newc << 24), qsel);
#else
			int q63[NW];
#pragma unroll
			for (int wd = 0; wd < NW; ++wd) q63[wd] = readlane(QP[wd], 63);
#pragma unroll
			for (int wd = 0; wd < NW; ++wd) {
				const int fill = (int)(((unsigned)q63[wd] << 8) | (wd ? (unsigned)q63[wd ? wd - 1 : 0] >> 24 : (unsigned)newc));
				QP[wd] = shr1(QP[wd], fill);
			}
#endif
		}

		WM_EMU_ASSERT(base == st);
		// ---- first-column / first-row boundary of lane r (:152-155): applied inside the top pair's body (lane r shares its 16-lane group, hence
		// its pair, with the hull end `en`); bm selects the half of thread (r - base) & 63 that holds it
		V<int> bm = 0;
		if (en >= r) {
			const int o = r - base;
			WM_EMU_ASSERT(o >= 0 && o < 128 * BP && (o >> 7) == ((en - base) >> 7));
			bm = sel(ln == (o & 63), (o & 64) ? (int)0xffff0000 : 0x0000ffff, 0);
		}

		const int cend = st0 + (en0 - st0) / 16 * 16 + 15;           // last lane of the rewritten score chunks (:158-173)
		const int NI = ((en - base) >> 7) + 1;                         // pairs that intersect the hull
		const int NS = CLIP ? ((((cend > en ? cend : en) - base) >> 7) + 1) : NI;
		WM_EMU_ASSERT(NS <= BP);
		V<int> hmax = KSW_NEG_INF;
		uint8_t *trow = tbp + (size_t)r * jb.n_col;                    // column of lane t = t - st = t - base
#if WM_KSW_ROR
		V<int> crx = 0, crv = 0, crx2 = 0;                             // rotations handed from pair i + 1 to pair i
#endif
		int h_en0 = KSW_NEG_INF;                                       // exact mode: the new H of lane en0, when that is the last target lane
		int d0 = 0, d1 = 0;                                            // approximate-max track: new v of lane last_H0_t, new u of lane last_H0_t + 1 (int8)

		// match / mismatch (/ ambiguous) scores of pair i, tie-break tag in the low bits
		auto pair_scores = [&](auto IC) {
			constexpr int i = decltype(IC)::value;
			// bytes (2i & 3), (2i & 3) + 1 of the code words, spread to the halves
			constexpr int wd = i >> 1, psel = (i & 1) ? 0x0c030c02 : 0x0c010c00;
			const V<int> xq = TP[wd] ^ QP[wd];
			V<int> sv = pk_mad(pk_minu(perm(xq, xq, psel), one2), rep16(MISt - MCHt), rep16(MCHt));
			if constexpr (HASN) {
				const V<int> oq = TP[wd] | QP[wd];
				const V<int> isn = pk_lshr(perm(oq, oq, psel) & 0x00040004, 2);           // 1 where either code is 4
				sv = bfi(pk_sub(0, isn), rep16(NNt), sv);
			}
			if constexpr (CLIP) {   // the score row is persistent and only [st0, cend] is rewritten
				const int c0 = base + 128 * i;
				if (c0 >= st0 && c0 + 127 <= cend) S[i] = sv;
				else {
					const V<int> t_lo = ln + c0, t_hi = ln + (c0 + 64);
					const V<int> m = sel(t_lo >= st0 && t_lo <= cend, 0x0000ffff, 0) | sel(t_hi >= st0 && t_hi <= cend, (int)0xffff0000, 0);
					S[i] = bfi(m, sv, S[i]);
				}
				sv = S[i];
			}
			return sv;
		};
		// one pair of chunks: 128 cells. `top` (uniform) = the pair that holds the hull end: the only one with lanes beyond the hull (masked
		// stores, stale lanes when the band clips), with the boundary lane r and with lane en0 (whose H comes from its left neighbour)
		auto pair_body = [&](auto IC) {
			constexpr int i = decltype(IC)::value;
			const bool top = i == NI - 1;
			const int c0 = base + 128 * i;
			const V<int> t_lo = ln + c0, t_hi = ln + (c0 + 64);
			const V<int> sv = pair_scores(IC);
			int hprev = KSW_NEG_INF;                                   // H of lane en0 - 1 in the previous row
			if (top) {
				WM_KEEP_BRANCH();
				Y[i] = bfi(bm, rep16(tB), Y[i]); Y2[i] = bfi(bm, rep16(tB2), Y2[i]); U[i] = bfi(bm, tb16(sched), U[i]);
				if constexpr (EXACT) {
					const int le = en0 - 1 - c0;                           // lane en0 - 1 relative to the pair: -1 .. 126
					WM_EMU_ASSERT(le >= -1 && le < 127);
					if (le < 0) hprev = i ? readlane(H[i ? 2 * i - 1 : 0], 63) : Hbelow;
					else hprev = le < 64 ? readlane(H[2 * i], le & 63) : readlane(H[2 * i + 1], le & 63);
				}
			}
			// previous-row values of lane t-1
#if WM_KSW_ROR
			// rotate the packed register by one thread (thread 0 receives thread 63: {hi: lane c0+127, lo: lane c0+63}); thread 0 then takes its
			// low half from the high half of pair i-1's rotation (lane c0-1) and its high half from its own low half (lane c0+63): one byte
			// permute with a per-thread selector. The rotation of pair i-1 is kept for that pair's own turn.
			if (top) { WM_KEEP_BRANCH(); crx = ror1(X[i]); crv = ror1(Vv[i]); crx2 = ror1(X2[i]); }
			const V<int> rxo = crx, rvo = crv, rx2o = crx2;
			if constexpr (i > 0) { crx = ror1(X[i ? i - 1 : 0]); crv = ror1(Vv[i ? i - 1 : 0]); crx2 = ror1(X2[i ? i - 1 : 0]); }
			else { crx = f_x << 16; crv = f_v << 16; crx2 = f_x2 << 16; }
			const V<int> x1 = perm(rxo, crx, rsel), v1 = perm(rvo, crv, rsel), x21 = perm(rx2o, crx2, rsel);
#else
			const int px = i ? lshr(readlane(X[i ? i - 1 : 0], 63), 16) : f_x, pv = i ? lshr(readlane(Vv[i ? i - 1 : 0], 63), 16) : f_v;
			const int px2 = i ? lshr(readlane(X2[i ? i - 1 : 0], 63), 16) : f_x2;
			const V<int> x1 = shr1(X[i], (int)((unsigned)readlane(X[i], 63) << 16 | (unsigned)px));
			const V<int> v1 = shr1(Vv[i], (int)((unsigned)readlane(Vv[i], 63) << 16 | (unsigned)pv));
			const V<int> x21 = shr1(X2[i], (int)((unsigned)readlane(X2[i], 63) << 16 | (unsigned)px2));
#endif
			const V<int> ou = U[i];
			V<int> nu, nv, nx, ny, nx2, ny2, p;
			ksw_pcell(cc, sv, x1, v1, x21, Y[i], ou, Y2[i], nu, nv, nx, ny, nx2, ny2, p);
			if constexpr (CLIP) {              // lanes beyond the hull keep their stale values (they feed back when the band is clipped)
				V<int> m = -1;
				if (top) { WM_KEEP_BRANCH(); m = sel(t_lo <= en, 0x0000ffff, 0) | sel(t_hi <= en, (int)0xffff0000, 0); }
				U[i] = bfi(m, nu, U[i]); Vv[i] = bfi(m, nv, Vv[i]); X[i] = bfi(m, nx, X[i]); Y[i] = bfi(m, ny, Y[i]);
				X2[i] = bfi(m, nx2, X2[i]); Y2[i] = bfi(m, ny2, Y2[i]);
			} else {                           // (unclipped band: lanes beyond the hull never matter — see the header)
				U[i] = nu; Vv[i] = nv; X[i] = nx; Y[i] = ny; X2[i] = nx2; Y2[i] = ny2;
			}
#ifdef WM_KSW_NOSTORE           // timing experiment only (tools/ksw_probe.py): the traceback bytes are computed but never stored — results are WRONG
			if (!(flag & 0x40000000)) { WM_KEEP_BRANCH(); } else
#endif
			if (top) {
				WM_IF(t_lo <= en) gst(trow, ln + 128 * i, cast<uint8_t>(p)); WM_END
				WM_IF(t_hi <= en) gst(trow, ln + (128 * i + 64), cast<uint8_t>(lshr(p, 16))); WM_END
			} else {
				gst(trow, ln + 128 * i, cast<uint8_t>(p));
				gst(trow, ln + (128 * i + 64), cast<uint8_t>(lshr(p, 16)));
			}
			if constexpr (EXACT) {
				// H += v (:320-345). Lanes outside the band keep their H; lane en0 takes H of its left neighbour + u. (Row 0 runs through here
				// as well: its only band lane is rewritten below, and hmax is not used.)
				const int en0x = en0 > 0 ? en0 : -1;
#pragma unroll
				for (int hf = 1; hf >= 0; --hf) {
					const int ci = 2 * i + hf, cb = c0 + 64 * hf;
					const V<int> v8 = hf ? vhi8(Vv[i]) : vlo8(Vv[i]);
					V<int> hn = H[ci] + v8;
					if (cb >= st0 && cb + 63 < en0) {               // chunk strictly inside the band: every lane is a plain update
						H[ci] = hn;
						hmax = vmax(hmax, hn);
					} else {
						WM_KEEP_BRANCH();
						const V<int> t = hf ? t_hi : t_lo;
						const V<int> u8 = hf ? vhi8(U[i]) : vlo8(U[i]);
						hn = sel(t == en0x, V<int>(u8 + hprev), hn);
						const vbool inb = t >= st0 && t <= en0;
						H[ci] = sel(inb, hn, H[ci]);
						hmax = vmax(hmax, sel(inb, hn, V<int>(KSW_NEG_INF)));
					}
				}
				if (top && en0 == tlen - 1) {
					WM_KEEP_BRANCH();
					const int oe = en0 - c0;
					h_en0 = oe < 64 ? readlane(H[2 * i], oe & 63) : readlane(H[2 * i + 1], oe & 63);
				}
			} else if constexpr (CLIP || !WM_KSW_EDGE_TRACK) {
				// the approximate-max track reads two neighbouring lanes of this row (:359-375)
				const int o0 = last_H0_t - c0, o1 = o0 + 1;
				if ((unsigned)o0 < 128u) { const int rr = readlane(Vv[i], o0 & 63); d0 = ((o0 & 64) ? rr >> 16 : (int)(short)(rr & 0xffff)) >> 8; }
				if ((unsigned)o1 < 128u) { const int rr = readlane(U[i], o1 & 63); d1 = ((o1 & 64) ? rr >> 16 : (int)(short)(rr & 0xffff)) >> 8; }
			}
		};
		if constexpr (CLIP) {
			if (NS > NI) {                 // the rewritten score chunks reach one pair beyond the hull (cend - en < 16): its score row only
				WM_EMU_ASSERT(NS == NI + 1);
				static_for_desc<BP>([&](auto IC) { if (decltype(IC)::value == NI) pair_scores(IC); });
			}
		}
		static_for_desc<BP>([&](auto IC) {
			if (decltype(IC)::value < NI) pair_body(IC);
		});

		if constexpr (EXACT) {   // ---- exact max: 32-bit wave maximum, then — only when it matters — the lane that reaches it
			int max_H, max_t = en0;
			if (r > 0) {
				max_H = wave_max_i32(hmax);
				// max_t is consumed by a new maximum (:src/ksw2.h:160-163) or by a z-drop test that can fire: ez_max - max_H > zdrop + l * e2 needs
				// ez_max - max_H > zdrop (l >= 0). Every other row leaves it alone — most rows of a long extension.
				const bool need_t = max_H > ez_max || (zdrop >= 0 && ez_max - max_H > zdrop);
				if (need_t) {
					WM_KEEP_BRANCH();
					// The reference's SIMD tie rule (src/ksw2_extd2_sse.c:315-358) as a lane priority: en0 first, then residue groups 0..3 of [st0, en1),
					// then the tail [en1, en0); inside a group the lower lane. Chunk starts are multiples of 4, so the residue is the same in every chunk.
					// Chunks that do not hold the maximum cost one compare and a branch; no per-chunk scalar bookkeeping.
					const int en1 = st0 + (en0 - st0) / 4 * 4;
					const V<int> g4 = (4 - ((ln + (base - st0)) & 3)) << 20;
					V<int> best = -1;
					static_for_desc<B>([&](auto CC) {
						constexpr int ci = decltype(CC)::value;
						if (ci >= 2 * NI) return;
						const V<int> t = ln + (base + 64 * ci);
						const vbool hit = H[ci] == max_H && cast<unsigned>(t - st0) <= (unsigned)(en0 - st0);
						if (any(hit)) {
							WM_KEEP_BRANCH();
							V<int> pri = sel(t < en1, g4, V<int>(0));
							pri = sel(t == en0, V<int>(5 << 20), pri) | (V<int>(0xfffff) - t);
							best = vmax(best, sel(hit, pri, V<int>(-1)));
						}
					});
					max_t = 0xfffff - (wave_max_i32(best) & 0xfffff);
				}
			} else {
				WM_IF(ln == 0) H[0] = vlo8(Vv[0]) - qe; WM_END
				max_H = readlane(H[0], 0); max_t = 0;
				h_en0 = max_H;
			}
			if (en0 == tlen - 1) { if (h_en0 > ez_mte) ez_mte = h_en0, ez_mte_q = r - en; }
			if (r - st0 == qlen - 1) { const int h = readlane(H[0], st0 - base); if (h > ez_mqe) ez_mqe = h, ez_mqe_t = st0; }      // (base = st: lane st0 is in chunk 0)
			if (max_H > ez_max) {
				ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
			} else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
				const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
				if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; break; }
			}
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = h_en0;
		} else if constexpr (!CLIP && WM_KSW_EDGE_TRACK) {
			// ---- approximate max when the band never clips (w >= qlen, tlen: the bulk of the gap fills). H0 of :359-375 is H followed along a greedy
			// monotone track: staying in lane t adds v(r, t), stepping to lane t + 1 adds u(r, t + 1), so H0 == H(r, track lane) exactly as long as
			// the track visits band cells — and here every lane of [st0, en0] is one. Only ez.score = H0 of the LAST row is ever consumed, and
			// that row has a single cell: H there does not depend on the path. So follow the cheapest path instead — the hull's first lane st0
			// (lane 0 down the first column, then one lane to the right per row along the last query row): one v_readlane per row in thread
			// st0 - base of pair 0, no per-pair lookups, no track state.
			const int o = st0 - base;                                      // 0..15: chunk 0, low half
			const int rv = readlane(Vv[0], o), ru = readlane(U[0], o);
			const int d = ((int)(short)((r < qlen ? rv : ru) & 0xffff)) >> 8;
			H0 = r ? H0 + d : d - qe;
			if (r == n_rows - 1) ez_score = H0;                            // (en0 == tlen - 1 on the last row of an unclipped band)
		} else {        // ---- approximate max: follow one diagonal-ish track (:359-375); d0 / d1 were picked up by the pair bodies
			if (r > 0) {
				const bool in0 = last_H0_t >= st0 && last_H0_t <= en0, in1 = last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0;
				if (in0 && in1) {
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (in0) {
					H0 += d0;
				} else {
					++last_H0_t;
					H0 += d1;
				}
			} else H0 = d0 - qe, last_H0_t = 0;
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = H0;
		}
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

} // namespace wmk
