// ksw_packed_kernel.h — ksw_extd2_sse (src/ksw2_extd2_sse.c:26-393) as a wave64 register machine that evaluates TWO
// DP cells per 32-bit lane with packed 16-bit integer instructions (VOP3P: v_pk_add_u16 / v_pk_sub_u16 / v_pk_max_i16 /
// v_pk_min_i16 / v_pk_sub_i16 clamp / v_pk_mad_u16 — full rate on gfx950, profiles/r02_valu_bench.txt).
//
// Same machine as ksw_dp_striped (ksw_kernel.h): one wavefront owns one alignment, walks the anti-diagonals r = i + j and
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
// VALU work per pair and row (CLIP = HASN = EXACT = false): 39 packed ops for the two cells + 3 (scores) + 9 (neighbours)
// = 51 per 128 cells, against ~45 per 64 cells in the 32-bit kernel. Registers: 6 per pair (+1 with CLIP, +2 with EXACT).
#pragma once
#ifndef WM_DEV
#error "include simt.h before ksw_packed_kernel.h"
#endif
#include "ksw_kernel.h"

namespace wmk {

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
	p = pk_mad(p, 0x00020002, pk_lshr(pk_subsat(c.hA, a), 15));
	p = pk_mad(p, 0x00020002, pk_lshr(pk_subsat(c.hB, b), 15));
	p = pk_mad(p, 0x00020002, pk_lshr(pk_subsat(c.hA2, a2), 15));
	p = pk_mad(p, 0x00020002, pk_lshr(pk_subsat(c.hB2, b2), 15));
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
	const ksw_pcell_cst_t cc = { tb16(qe), tb16(qe2), tb16(q), tb16(q2), tb16(sc.match), rep16(tA), rep16(tB), rep16(tA2), rep16(tB2),
	                             rep16(hA), rep16(hB), rep16(hA2), rep16(hB2) };
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	const V<int> ln = lane();
	const vbool low48 = ln < 48;
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
		if (st0 < (r - w + 1) >> 1) st0 = (r - w + 1) >> 1;
		if (en0 > (r + w) >> 1) en0 = (r + w) >> 1;
		if (st0 > en0) { ez_zdropped = 1; break; }
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
			if (qi0 >= 0 && qi0 < qlen) {
				if (qi0 < qb0 || qi0 >= qb0 + 64) {
					qb0 = qi0 < 16 ? 0 : qi0 - 16;          // a re-base steps the index back by 16: keep that much behind
					const V<int> qidx = ln + qb0;
					QB = 0;
					WM_IF(qidx < qlen) QB = cast<int>(gld(query, qidx)); WM_END
				}
				newc = readlane(QB, qi0 - qb0);
			}
			int q63[NW];
#pragma unroll
			for (int wd = 0; wd < NW; ++wd) q63[wd] = readlane(QP[wd], 63);
#pragma unroll
			for (int wd = 0; wd < NW; ++wd) {
				const int fill = (int)(((unsigned)q63[wd] << 8) | (wd ? (unsigned)q63[wd ? wd - 1 : 0] >> 24 : (unsigned)newc));
				QP[wd] = shr1(QP[wd], fill);
			}
		}

		// ---- first-column / first-row boundary of lane r (:152-155)
		if (en >= r) {
			const int o = r - base, jr = o & 63, c = o >> 6, ip = c >> 1;
			const int hm = (c & 1) ? (int)0xffff0000 : 0x0000ffff;
			WM_EMU_ASSERT(o >= 0 && o < 128 * BP);
			WM_IF(ln == jr)
#pragma unroll
				for (int i = 0; i < BP; ++i)
					if (ip == i) { Y[i] = bfi(hm, rep16(tB), Y[i]); Y2[i] = bfi(hm, rep16(tB2), Y2[i]); U[i] = bfi(hm, tb16(sched), U[i]); }
			WM_END
		}

		const int cend = st0 + (en0 - st0) / 16 * 16 + 15;           // last lane of the rewritten score chunks (:158-173)
		const int NI = ((en - base) >> 7) + 1;                         // pairs that intersect the hull
		const int NS = CLIP ? ((((cend > en ? cend : en) - base) >> 7) + 1) : NI;
		WM_EMU_ASSERT(NS <= BP);
		V<int> hmax = KSW_NEG_INF;
		uint8_t *trow = tbp + (size_t)r * jb.n_col + (base - st);

#pragma unroll
		for (int i = BP - 1; i >= 0; --i) {
			if (i >= NS) continue;
			const int c0 = base + 128 * i;
			const V<int> t_lo = ln + c0, t_hi = ln + (c0 + 64);
			// match / mismatch scores of the two chunks: bytes (2i & 3), (2i & 3) + 1 of the code words, spread to the halves
			const int wd = i >> 1, psel = (i & 1) ? 0x0c030c02 : 0x0c010c00;
			const V<int> xq = TP[wd] ^ QP[wd];
			V<int> sv = pk_mad(pk_minu(perm(xq, xq, psel), 0x00010001), rep16(MISt - MCHt), rep16(MCHt));
			if constexpr (HASN) {
				const V<int> oq = TP[wd] | QP[wd];
				const V<int> isn = pk_lshr(perm(oq, oq, psel) & 0x00040004, 2);           // 1 where either code is 4
				sv = bfi(pk_sub(0, isn), rep16(NNt), sv);
			}
			if constexpr (CLIP) {   // the score row is persistent and only [st0, cend] is rewritten
				const V<int> m = sel(t_lo >= st0 && t_lo <= cend, 0x0000ffff, 0) | sel(t_hi >= st0 && t_hi <= cend, (int)0xffff0000, 0);
				S[i] = bfi(m, sv, S[i]); sv = S[i];
			}
			if (i >= NI) continue;
			// previous-row values of lane t-1
			const int px = i ? lshr(readlane(X[i ? i - 1 : 0], 63), 16) : f_x, pv = i ? lshr(readlane(Vv[i ? i - 1 : 0], 63), 16) : f_v;
			const int px2 = i ? lshr(readlane(X2[i ? i - 1 : 0], 63), 16) : f_x2;
			const V<int> x1 = shr1(X[i], (int)((unsigned)readlane(X[i], 63) << 16 | (unsigned)px));
			const V<int> v1 = shr1(Vv[i], (int)((unsigned)readlane(Vv[i], 63) << 16 | (unsigned)pv));
			const V<int> x21 = shr1(X2[i], (int)((unsigned)readlane(X2[i], 63) << 16 | (unsigned)px2));
			V<int> hl_lo = KSW_NEG_INF, hl_hi = KSW_NEG_INF;
			if constexpr (EXACT) {
				hl_lo = shr1(H[2 * i], i ? readlane(H[i ? 2 * i - 1 : 0], 63) : Hbelow);
				hl_hi = shr1(H[2 * i + 1], readlane(H[2 * i], 63));
			}
			const V<int> ou = U[i];
			V<int> nu, nv, nx, ny, nx2, ny2, p;
			ksw_pcell(cc, sv, x1, v1, x21, Y[i], ou, Y2[i], nu, nv, nx, ny, nx2, ny2, p);
			if (!CLIP || c0 + 127 <= en) {     // (unclipped band: lanes beyond the hull never matter — see the header)
				U[i] = nu; Vv[i] = nv; X[i] = nx; Y[i] = ny; X2[i] = nx2; Y2[i] = ny2;
			} else {                           // lanes beyond the hull keep their stale values (they feed back when the band is clipped)
				const V<int> m = sel(t_lo <= en, 0x0000ffff, 0) | sel(t_hi <= en, (int)0xffff0000, 0);
				U[i] = bfi(m, nu, U[i]); Vv[i] = bfi(m, nv, Vv[i]); X[i] = bfi(m, nx, X[i]); Y[i] = bfi(m, ny, Y[i]);
				X2[i] = bfi(m, nx2, X2[i]); Y2[i] = bfi(m, ny2, Y2[i]);
			}
			WM_IF(t_lo <= en) gst(trow, t_lo - base, cast<uint8_t>(p)); WM_END
			WM_IF(t_hi <= en) gst(trow, t_hi - base, cast<uint8_t>(lshr(p, 16))); WM_END
			if constexpr (EXACT) if (r > 0) {
#pragma unroll
				for (int hf = 1; hf >= 0; --hf) {
					const int ci = 2 * i + hf, cb = c0 + 64 * hf;
					if (cb > en) continue;
					const V<int> t = hf ? t_hi : t_lo;
					const V<int> v8 = hf ? vhi8(Vv[i]) : vlo8(Vv[i]);
					if (cb >= st0 && cb + 63 < en0) {               // chunk strictly inside the band: every lane is a plain update
						H[ci] = H[ci] + v8;
						hmax = vmax(hmax, H[ci]);
					} else {
						const V<int> u8 = hf ? vhi8(U[i]) : vlo8(U[i]);
						const V<int> hl = hf ? hl_hi : hl_lo;
						V<int> hn = H[ci] + v8;
						hn = sel(t == en0, en0 > 0 ? V<int>(hl + u8) : hn, hn);
						const vbool inb = t >= st0 && t <= en0;
						H[ci] = sel(inb, hn, H[ci]);
						hmax = vmax(hmax, sel(inb, H[ci], V<int>(KSW_NEG_INF)));
					}
				}
			}
		}

		if constexpr (EXACT) {   // ---- exact max: 32-bit wave maximum, then the lanes that reach it
			int max_H, max_t;
			const int NC = ((en - base) >> 6) + 1;
			if (r > 0) {
				max_H = wave_max_i32(hmax);
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				int best_pri = -1;
				max_t = en0;
#pragma unroll
				for (int i = 0; i < B; ++i) {
					if (i >= NC) continue;
					const int c0 = base + 64 * i;
					if (c0 > en0 || c0 + 63 < st0) continue;
					const int lo = st0 > c0 ? st0 - c0 : 0, hi = en0 - c0 < 63 ? en0 - c0 : 63;
					const uint64_t band = (hi == 63 ? ~(uint64_t)0 : (((uint64_t)1 << (hi + 1)) - 1)) & ~(((uint64_t)1 << lo) - 1);
					uint64_t m = ballot(H[i] == max_H) & band;
					while (m) {                                   // priority on ties: en0, then residue groups 0..3 of [st0,en1), then the tail
						const int tt = c0 + __builtin_ctzll(m);
						m &= m - 1;
						const int grp = tt == en0 ? 5 : tt < en1 ? 4 - ((tt - st0) & 3) : 0;
						const int pri = (grp << 20) | (0xfffff - tt);
						if (pri > best_pri) best_pri = pri, max_t = tt;
					}
				}
			} else {
				WM_IF(ln == 0) H[0] = vlo8(Vv[0]) - qe; WM_END
				max_H = readlane(H[0], 0); max_t = 0;
			}
			if (en0 == tlen - 1) { const int h = get_lane_striped<B>(H, base, en0); if (h > ez_mte) ez_mte = h, ez_mte_q = r - en; }
			if (r - st0 == qlen - 1) { const int h = get_lane_striped<B>(H, base, st0); if (h > ez_mqe) ez_mqe = h, ez_mqe_t = st0; }
			if (max_H > ez_max) {
				ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
			} else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
				const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
				if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; break; }
			}
			if (r == n_rows - 1 && en0 == tlen - 1) ez_score = get_lane_striped<B>(H, base, tlen - 1);
		} else {        // ---- approximate max: follow one diagonal-ish track (:359-375)
			if (r > 0) {
				const bool in0 = last_H0_t >= st0 && last_H0_t <= en0, in1 = last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0;
				if (in0 && in1) {
					const int d0 = get_half<BP>(Vv, base, last_H0_t) >> 8, d1 = get_half<BP>(U, base, last_H0_t + 1) >> 8;
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (in0) {
					H0 += get_half<BP>(Vv, base, last_H0_t) >> 8;
				} else {
					++last_H0_t;
					H0 += get_half<BP>(U, base, last_H0_t) >> 8;
				}
			} else H0 = (get_half<BP>(Vv, base, 0) >> 8) - qe, last_H0_t = 0;
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
