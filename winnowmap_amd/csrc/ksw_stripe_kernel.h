// ksw_stripe_kernel.h — ksw_extd2_sse (src/ksw2_extd2_sse.c:26-393) for long alignments and wide band hulls: the packed two-cells-per-lane
// machine of ksw_packed_kernel.h over the NWV wavefronts of one workgroup, WITHOUT a barrier in the row loop.
//
// Why it can be barrier-free: in the anti-diagonal recurrence cell (r, t) reads lane t and lane t - 1 of row r - 1, nothing else — data only
// flows from lower to higher target lanes. So the target is cut into STRIPES of SW = 128 * BP lanes in absolute coordinates; stripe s
// (lanes s * SW ..) lives in the registers of wavefront s % NWV for as long as the 16-aligned band hull [st, en] touches it (the hull is at
// most (NWV - 1) * SW lanes wide, so a wavefront is done with stripe s before stripe s + NWV is reached) and is never re-based. The only
// thing a wavefront needs from somebody else is, once per row, the previous-row values (x, v, x2, H) of the lane just below its stripe:
// one message per row from its left neighbour through an LDS ring, stamped with the row number. The wavefronts therefore run SKEWED — the
// left neighbour is (at least) one message ahead — instead of meeting at an s_barrier every row (ksw_packed_multi_kernel.h: 2.5 us per row
// whatever the hull, profiles/r03o_call_summary.txt).
//
// What the reference evaluates per row over the WHOLE band rides on the same message, left to right:
//   * exact row maximum (src/ksw2_extd2_sse.c:315-358): every stripe reduces its own lanes, compares with the prefix maximum it received and
//     passes (maximum, tie-rule priority of the lane that holds it) on. The stripe that holds the band's last lane en0 receives the row's
//     (max_H, max_t) and runs the reference's bookkeeping (ez.max / mqe / mte / score, ksw_apply_zdrop src/ksw2.h:160-175) for that row; the
//     running state moves to the next wavefront in the message when en0 crosses a stripe boundary. A z-drop stops everybody through a word
//     in LDS; rows that ran ahead only wrote traceback rows nobody reads.
//   * the lane priority is only evaluated where it can matter (a new maximum, or a row that can z-drop), judged against a slightly stale
//     copy of ez.max. That is a heuristic for SPEED only: the bookkeeping wavefront detects a priority it needs and did not get, and the
//     workgroup then repeats the alignment with the priority evaluated on every row (`safe`), so the result never depends on the guess.
//   * H of the band's first lane st0 (mqe) and the approximate-maximum track (src/ksw2_extd2_sse.c:359-375) travel the same way.
// Cell arithmetic, traceback bytes, stale-lane emulation (CLIP), the tie rule: exactly ksw_dp_packed (shared ksw_pcell).
#pragma once
#ifndef WM_DEV
#error "include simt.h before ksw_stripe_kernel.h"
#endif
#include "ksw_packed_kernel.h"
#ifndef WM_STRIPE_SPIN
#define WM_STRIPE_SPIN(where, r, a, wv, extra) ((void)0)      // test hook: a watchdog for the polling loops
#endif
// Every cross-wavefront polling loop has a budget: a wavefront that has polled WM_STRIPE_SPIN_BUDGET times in ONE wait (each poll
// is an LDS load + s_sleep: ~10^2 cycles, so the default is seconds — legitimate waits are a few rows of a neighbour, microseconds) declares the
// protocol broken: it raises C_RESTART = 2 and the stop word, everybody leaves, the job's result carries bt_i = KSW_BT_WATCHDOG, the traceback kernel
// turns that into the batch's error flag and wm_ksw_dev_run returns WM_EINTERNAL instead of hanging the mapping call.
#ifndef WM_STRIPE_SPIN_BUDGET
#define WM_STRIPE_SPIN_BUDGET (1 << 25)
#endif
// WM_STRIPE_EARLY_MSG=1 (build define, A/B): the left neighbour's message of a row is loaded BEFORE the row's cells and consumed after them, so that its LDS
// round trip hides behind the cells (in the steady state the neighbour is a row or more ahead). Measured in round 5 (profiles/r05_stripe_split_phase.txt):
// the isolated wide-hull probe gains 3-9 % per row (blk_3000x 2.00 -> 1.83 us), but the nine registers the message occupies across the cells take
// <2,16> from 104 to 128 VGPRs — four such wavefronts then fill a SIMD's register file and no bulk wavefront shares the CU — and the mapper lost 2-3 %
// (0.2996 / 0.3032 -> 0.2941 / 0.2898 Gbp/s at 32 768 reads per step). Off by default; the stop word, the stale ez.max and the back-pressure word
// are split-phase / cached in every build (two registers).
#ifndef WM_STRIPE_EARLY_MSG
#define WM_STRIPE_EARLY_MSG 0
#endif
#ifndef WM_STRIPE_EVENT
#define WM_STRIPE_EVENT(k) ((void)0)      // test hook (tests/simt_emu): counts how often the rare paths run
#endif
// Diagnostic build (WM_KERNEL_DEFINES="WM_STRIPE_TIMING=1", a variant library: winnowmap_amd/build.py): every wavefront accumulates the shader-clock
// cycles (s_memtime) it spends in each phase of the row loop and adds them to a device-global table at the end of the job; wm_debug_stripe_timing
// reads the table (tools/ksw_probe.py prints it per shape). Empty in every other build: the shipped kernels do not change.
#if defined(WM_STRIPE_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_wm_stripe_timing[16];
#endif
#if defined(WM_STRIPE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define WM_ST_DECL() unsigned long long st_acc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; unsigned long long st_t = __builtin_amdgcn_s_memtime(); const unsigned long long st_t0 = st_t
#define WM_ST_LAP(k) do { const unsigned long long st_n = __builtin_amdgcn_s_memtime(); st_acc[k] += st_n - st_t; st_t = st_n; } while (0)
#define WM_ST_COUNT(k) (++st_acc[k])
#define WM_ST_FLUSH() do { st_acc[9] = __builtin_amdgcn_s_memtime() - st_t0; \
		if (__lane_id() == 0) { for (int k_ = 0; k_ < 10; ++k_) atomicAdd(&g_wm_stripe_timing[k_], st_acc[k_]); atomicAdd(&g_wm_stripe_timing[10], 1ull); } } while (0)
#else
#define WM_ST_DECL() ((void)0)
#define WM_ST_LAP(k) ((void)0)
#define WM_ST_COUNT(k) ((void)0)
#define WM_ST_FLUSH() ((void)0)
#endif
enum { WM_ST_SCAN = 0, WM_ST_EPOCH = 1, WM_ST_CELLS = 2, WM_ST_WAIT_LEFT = 3, WM_ST_BOOK = 4, WM_ST_WAIT_RIGHT = 5, WM_ST_PUBLISH = 6, WM_ST_ROWS = 7, WM_ST_EPOCHS = 8, WM_ST_TOTAL = 9 };

namespace wmk {

// LDS ints: NWV rings of R slots (ring w: messages of wavefront w to its right neighbour) | progress per wavefront | control words
template <int BP, int NWV> struct ksw_stripe_lds {
	static constexpr int SW = 128 * BP, R = 8, SLOT = 20, RING = NWV * R * SLOT, PROG = RING, CTRL = PROG + NWV, INTS = CTRL + 4;
	static constexpr int C_STOP = 0, C_RESTART = 1, C_EZL = 2;        // control words: a z-drop ended the alignment in this row | repeat in safe mode | stale ez.max
	// message slot: stamp, 3 free | x v x2 h | pm ppri hst0 (exact maximum) or track H0, track lane (-1: no hand-over), free | ez state (8 ints; only on rows where it may move)
	static constexpr int M_STAMP = 0, M_X = 4, M_V = 5, M_X2 = 6, M_H = 7, M_PM = 8, M_PRI = 9, M_HST0 = 10, M_TH0 = 8, M_TL0 = 9, M_EZ = 12;      // (M_X and M_PM on 16-byte boundaries: lds_ld_msg)
	// widest traceback pitch n_col (>= the 16-aligned hull en - st + 1) this geometry can hold: hull plus the up to 15 lanes of score chunks beyond it
	// (cend, CLIP) touch at most NWV stripes
	static constexpr int MAX_NCOL = (NWV - 1) * SW;
};

struct ksw_geo_t { int st0, en0, st, en, cend; };
// band limits of row r (src/ksw2_extd2_sse.c:128-139); false: the band is empty
template <bool CLIP> WM_DEV bool ksw_geo(int r, int qlen, int tlen, int w, ksw_geo_t &g)
{
	int st0 = 0, en0 = tlen - 1;
	if (st0 < r - qlen + 1) st0 = r - qlen + 1;
	if (en0 > r) en0 = r;
	if (CLIP) {                                  // (CLIP = false: w >= qlen, tlen — the band terms never bind and the hull is never empty, ksw_plan.h)
		if (st0 < (r - w + 1) >> 1) st0 = (r - w + 1) >> 1;
		if (en0 > (r + w) >> 1) en0 = (r + w) >> 1;
		if (st0 > en0) return false;
	}
	g.st0 = st0; g.en0 = en0; g.st = st0 / 16 * 16; g.en = (en0 + 16) / 16 * 16 - 1;
	g.cend = st0 + (en0 - st0) / 16 * 16 + 15;   // last lane of the rewritten score chunks (:158-173)
	return true;
}

template <int BP, int NWV, bool CLIP, bool HASN, bool EXACT>
WM_DEV void ksw_dp_stripe(const wm_ksw_score_t sc, const wm_ksw_djob_t jb, const uint8_t *__restrict__ seqs,
                          uint8_t *__restrict__ tb_arena, int *lds, wm_ksw_dres_t *__restrict__ res)
{
	typedef ksw_stripe_lds<BP, NWV> L;
	constexpr int SW = L::SW, B = 2 * BP, NW = (B + 3) / 4, R = L::R;
	constexpr int BIG = 0x7fffffff;
	constexpr bool EARLY_MSG = WM_STRIPE_EARLY_MSG && BP <= 4;
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
	const int one2 = (int)sc.match > -128 ? 0x00010001 : 0x00020002;       // opaque to the compiler (see ksw_dp_packed)
	const ksw_pcell_cst_t cc = { tb16(qe), tb16(qe2), tb16(q), tb16(q2), tb16(sc.match), rep16(tA), rep16(tB), rep16(tA2), rep16(tB2),
	                             rep16(hA), rep16(hB), rep16(hA2), rep16(hB2) };
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const int n_rows = qlen + tlen - 1;
	const int negqe16 = ((-qe) & 0xff) << 8;
	// how far ez.max can have moved since the copy a wavefront looks at was taken: the leftmost wavefront runs at most R rows ahead of each
	// neighbour. Only the SPEED of the exact-maximum path depends on this margin (see the header), never the result.
#ifdef WM_STRIPE_TEST_SLACK
	const int ez_slack = WM_STRIPE_TEST_SLACK;              // test hook: a useless margin, so that the repeat-in-safe-mode path runs
#else
	const int ez_slack = (int)sc.match * (R * NWV / 2 + 2);
#endif

	const V<int> ln = lane();
	const int wv = wave_in_block();
	const V<int> rsel = sel(ln == 0, 0x05040302, 0x07060504);      // v_perm_b32 selectors: {own.lo, prev.hi} for thread 0, own elsewhere
	const V<int> qsel = sel(ln == 0, 0x06050403, 0x07060504);      // code words: thread 0 takes {own bytes 2..0, prev byte 3}
	int *ring_out = lds + wv * (R * L::SLOT), *ring_in = lds + ((wv + NWV - 1) % NWV) * (R * L::SLOT);
	int *prog = lds + L::PROG, *ctrl = lds + L::CTRL;
	const int right_wv = (wv + 1) % NWV;
	int spins = 0;                                                   // polls of the wait this wavefront is in (reset on entry to each polling loop: the budget bounds ONE wait, not the job — ADVICE r5)
	auto give_up = [&]() { lds_st_rel(ctrl, L::C_RESTART, 2); lds_st_rel(ctrl, L::C_STOP, -1); };      // (the loop that called it sees the stop word at its next poll)
	WM_ST_DECL();

	for (int safe = 0; safe < 2; ++safe) {
		// ---- the workgroup's LDS state: no message yet, nobody has consumed anything, nothing stops ----
		WM_IF(ln < R) lds_st(ring_out, ln * L::SLOT + L::M_STAMP, V<int>(-1)); WM_END
		lds_st_rel(prog, wv, -1);
		if (wv == 0) { lds_st_rel(ctrl, L::C_STOP, BIG); lds_st_rel(ctrl, L::C_RESTART, 0); lds_st_rel(ctrl, L::C_EZL, 0); }
		block_sync_lds();

		V<int> U[BP], Vv[BP], X[BP], Y[BP], X2[BP], Y2[BP];
		V<int> S[CLIP ? BP : 1];
		V<int> H[EXACT ? B : 1];
		V<int> TP[NW], QP[NW];
		V<int> QB = 0;
		int qb0 = -(1 << 30);
		int ez_max = 0, ez_zdropped = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1;
		int ez_mte = KSW_NEG_INF, ez_mte_q = -1, ez_score = KSW_NEG_INF;
		int H0 = 0, last_H0_t = 0;
		bool trk = !EXACT && wv == 0;           // this wavefront owns the approximate-maximum track
		bool my_stop = false;                   // this wavefront ended the alignment (z-drop)
		bool was_last = false;                  // it held the band's last lane in the last row it completed
		int row_done = -1;                      // last row this wavefront completed
		int end_row = n_rows;                   // first row that does not exist (n_rows, or the first row with an empty band)
		int r = 0, s = wv;
		bool all_done = false;
		int right_seen = -1;                    // lower bound of the right neighbour's progress word (it only grows during a pass): re-read when it no longer suffices

		while (!all_done) {
			// ================= find the first row that touches stripe s =================
			const int a = s * SW;
			if (a >= tlen) break;
			ksw_geo_t g;
			{   // no row before these can reach lane a - 15 (en0 <= r, en0 <= (r + w) >> 1): skip them without looking
				int rmin = a - 15;
				if (CLIP && 2 * (a - 15) - w > rmin) rmin = 2 * (a - 15) - w;
				if (r < rmin) r = rmin;
			}
			for (;; ++r) {
				if (r >= n_rows) { all_done = true; break; }
				if (!ksw_geo<CLIP>(r, qlen, tlen, w, g)) { end_row = r; all_done = true; break; }
				if (a <= (CLIP && g.cend > g.en ? g.cend : g.en)) break;
			}
			if (all_done) break;
			// ---- fresh registers: the initial values of src/ksw2_extd2_sse.c:99-112; codes of this stripe's lanes ----
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
				V<int> pk = 0, pq = 0;
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					if (wd * 4 + b >= B) continue;
					const V<int> t = ln + (a + 64 * (wd * 4 + b));
					V<int> c = 0, d = 0;
					WM_IF(t < tlen) c = cast<int>(gld(target, t)); WM_END
					const V<int> qi = (r - 1) - t;                  // the query codes of row r - 1: the row loop advances them to row r
					WM_IF(qi >= 0 && qi < qlen) d = cast<int>(gld(query, qi)); WM_END
					pk = pk | (c << (8 * b)); pq = pq | (d << (8 * b));
				}
				TP[wd] = pk; QP[wd] = pq;
			}
			qb0 = -(1 << 30);
			loads_land();
			int prev_st = -1;
			bool have_left = false;                 // the left neighbour's message of the previous row exists (its values are in m_*)
			int m_x = 0, m_v = 0, m_x2 = 0, m_h = KSW_NEG_INF;
			lds_st_rel(prog, wv, r - 2);            // nothing before row r - 1 is wanted from the left ring
			if (r > 0) {
				ksw_geo_t gp;
				if (ksw_geo<CLIP>(r - 1, qlen, tlen, w, gp)) {
					prev_st = gp.st;
					if (a > 0 && gp.st <= a - 1 && a - 1 <= gp.en) {      // lane a - 1 was computed in row r - 1: take its message
						const int *m = ring_in + ((r - 1) % R) * L::SLOT;
						bool gone = false;
						spins = 0;
						while (lds_ld_acq(m, L::M_STAMP) != r - 1) { if (lds_ld_acq(ctrl, L::C_STOP) < r) { gone = true; break; } WM_STRIPE_SPIN(0, r, a, wv, lds_ld_acq(m, L::M_STAMP)); if (++spins > WM_STRIPE_SPIN_BUDGET) give_up(); spin_pause(); }
						if (gone) { all_done = true; break; }
						m_x = lds_ld(m, (long long)L::M_X); m_v = lds_ld(m, (long long)L::M_V); m_x2 = lds_ld(m, (long long)L::M_X2); m_h = lds_ld(m, (long long)L::M_H);
						have_left = true;
						if (EXACT && gp.en == a - 1) { WM_STRIPE_EVENT(2);                      // the band's last lane sat right below this stripe: the bookkeeping state comes along
							ez_max = lds_ld(m, (long long)(L::M_EZ + 0)); ez_max_t = lds_ld(m, (long long)(L::M_EZ + 1)); ez_max_q = lds_ld(m, (long long)(L::M_EZ + 2));
							ez_mqe = lds_ld(m, (long long)(L::M_EZ + 3)); ez_mqe_t = lds_ld(m, (long long)(L::M_EZ + 4)); ez_mte = lds_ld(m, (long long)(L::M_EZ + 5));
							ez_mte_q = lds_ld(m, (long long)(L::M_EZ + 6)); ez_score = lds_ld(m, (long long)(L::M_EZ + 7));
						}
						if (!EXACT && lds_ld(m, (long long)L::M_TL0) >= 0) { trk = true; H0 = lds_ld(m, (long long)L::M_TH0); last_H0_t = lds_ld(m, (long long)L::M_TL0); }
					}
				}
			}

			// ================= the rows of stripe s, EPOCH by epoch =================
			// An epoch = a run of rows over which the 16-aligned hull [st, en] does not move (st0 and en0 advance by at most one lane per row, so an
			// epoch lasts ~8-16 rows). Everything that only depends on (st, en) — which pairs hold cells, which of them lie inside the hull with all
			// their lanes, the lane masks of the two that do not, which 64-lane chunks are inside the band whatever the row — is worked out once per
			// epoch; the row loop is left with the cells, one message in, one message out. (The first build evaluated all of it per row and pair:
			// ~900 instructions per row for two pairs, profiles/r04b: a lone wavefront issues one instruction per ~4.4 cycles.)
			bool leave = false;
			while (!all_done && !leave) {
				WM_ST_LAP(WM_ST_SCAN); WM_ST_COUNT(WM_ST_EPOCHS);               // (activation scan, first message, register set-up — or the tail of the last row)
				if (r >= n_rows) { all_done = true; break; }
				if (!ksw_geo<CLIP>(r, qlen, tlen, w, g)) { end_row = r; all_done = true; break; }
				const int st = g.st, en = g.en;
				if (st >= a + SW) { s += NWV; WM_STRIPE_EVENT(0); leave = true; break; }          // the hull has left this stripe for good
				int r_end = n_rows;                                            // first row of the next epoch
				{
					const int X = st + 16;                                     // st0 reaches X: r - qlen + 1 >= X, or (r - w + 1) >> 1 >= X
					int rs = X + qlen - 1;
					if (CLIP && 2 * X + w - 1 < rs) rs = 2 * X + w - 1;
					if (rs < r_end) r_end = rs;
					const int Y = en + 1;                                      // en0 reaches Y: Y <= tlen - 1, r >= Y and (r + w) >> 1 >= Y
					if (Y <= tlen - 1) {
						int re = Y;
						if (CLIP && 2 * Y - w > re) re = 2 * Y - w;
						if (re < r_end) r_end = re;
					}
				}
				WM_EMU_ASSERT(r_end > r);
				const bool have_cells = a <= en;
				const int i_lo = st > a ? (st - a) >> 7 : 0;
				const int i_hi = have_cells ? ((en - a) >> 7 < BP ? (en - a) >> 7 : BP - 1) : -1;
				const bool top_here = have_cells && en < a + SW;                // the pair i_hi holds the hull end
				const bool first_here = st >= a;                                // the pair i_lo holds the hull start
				const bool pub = a + SW - 1 <= en;                              // this stripe's last lane is computed: the right neighbour wants it
				const bool left_now = a > 0 && st <= a - 1 && a - 1 <= en;      // the left neighbour publishes a message for every row of the epoch
				// per pair: inside the hull with every lane (no lane mask anywhere)? its score row rewritten whole in every row? per chunk: strictly
				// inside the band in every row (st0 < st + 16, en0 >= en - 15)?
				int full_bits = 0, sfull_bits = 0, hin_bits = 0;
				V<int> vm[BP];                                                 // lanes of the pair inside [st, en], as a bit-field-insert mask (0: the pair is outside)
				V<int> smv[BP];                                                // the lane that holds the hull start, when that is strictly inside the stripe
#pragma unroll
				for (int i = 0; i < BP; ++i) {
					const int c0 = a + 128 * i;
					vm[i] = -1; smv[i] = 0;
					if (c0 >= st && c0 + 127 <= en) full_bits |= 1 << i;
					else {
						const V<int> t_lo = ln + c0, t_hi = ln + (c0 + 64);
						vm[i] = sel(t_lo >= st && t_lo <= en, 0x0000ffff, 0) | sel(t_hi >= st && t_hi <= en, (int)0xffff0000, 0);
					}
					if (first_here && st > a && ((st - a) >> 7) == i) { const int o = st - a; smv[i] = sel(ln == (o & 63), (o & 64) ? (int)0xffff0000 : 0x0000ffff, 0); }
					if (c0 >= st + 16 && c0 + 127 <= en - 15) sfull_bits |= 1 << i;
					if (c0 >= st + 16 && c0 + 63 < en - 15) hin_bits |= 1 << (2 * i);
					if (c0 + 64 >= st + 16 && c0 + 127 < en - 15) hin_bits |= 1 << (2 * i + 1);
				}
				// (hull start strictly inside the stripe: in every row but the one in which it moved there, lane st takes the constants of :147-150)
				bool moved = st > prev_st;                                     // lane st - 1 was computed in the last row (:141-146): first row of an epoch only
				prev_st = st;

				WM_ST_LAP(WM_ST_EPOCH);
				for (; r < r_end; ++r) {
					WM_ST_COUNT(WM_ST_ROWS);
					// split-phase loads (simt.h): issued here, waited for where the values are used — the stop word at the end of the row, the stale ez.max
					// in the bookkeeping, the left neighbour's message after the cells. In the steady state that neighbour is a row or more ahead, so the
					// message is already there and its LDS round trip hides behind the cells; if it is not, the polling loop below takes over.
					const int stop_raw = lds_ld_issue(ctrl, L::C_STOP);
					const int ezl_raw = EXACT ? lds_ld_issue(ctrl, L::C_EZL) : 0;
					lds_msg_raw early;
					if (EARLY_MSG && left_now) lds_ld_msg_issue(ring_in + (r % R) * L::SLOT, early);
					int st0 = 0, en0 = tlen - 1;
					if (st0 < r - qlen + 1) st0 = r - qlen + 1;
					if (en0 > r) en0 = r;
					if (CLIP) {
						if (st0 < (r - w + 1) >> 1) st0 = (r - w + 1) >> 1;
						if (en0 > (r + w) >> 1) en0 = (r + w) >> 1;
						if (st0 > en0) { end_row = r; all_done = true; break; }
					}
					WM_EMU_ASSERT(st0 / 16 * 16 == st && (en0 + 16) / 16 * 16 - 1 == en);
					const int cend = st0 + (en0 - st0) / 16 * 16 + 15;
					const int sched = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;

					// ---- advance the query codes to row r: every lane takes the code of lane t - 1; the stripe's first lane takes query[r - a] ----
					{
						const int qi0 = r - a;
						int newc = 0;
						if (qi0 >= 0 && qi0 < qlen) {
							if (qi0 < qb0 || qi0 >= qb0 + 64) {
								qb0 = qi0;
								const V<int> qidx = ln + qb0;
								QB = 0;
								WM_IF(qidx < qlen) QB = cast<int>(gld(query, qidx)); WM_END
								loads_land();
							}
							newc = readlane(QB, qi0 - qb0);
						}
						V<int> rq[NW];
#pragma unroll
						for (int wd = 0; wd < NW; ++wd) rq[wd] = ror1(QP[wd]);
#pragma unroll
						for (int wd = 0; wd < NW; ++wd) QP[wd] = perm(rq[wd], wd ? rq[wd ? wd - 1 : 0] : V<int>(newc << 24), qsel);
					}

					// previous-row values of the lane below the stripe: the left neighbour's message, or the constants of :141-151 when that lane was not
					// computed in the last row (which is also the case "hull start on the stripe's first lane and it did not move").
					// H of that lane is the one exception: lane en0 reads H[en0 - 1] whether or not it was updated in the last row (:322), i.e. the last
					// value the left neighbour ever sent (KSW_NEG_INF before the first)
					int px = tA, pv = (st == 0 ? sched & 0xff : (-qe) & 0xff) << 8, px2 = tA2;
					if (have_left) { px = m_x; pv = m_v; px2 = m_x2; }
					const bool is_last = have_cells && en0 < a + SW;                // the band's last lane is here: this wavefront closes the row
					// Every pair of the stripe runs in every row, in straight-line code: a pair (or a lane) outside the hull computes on whatever its
					// registers hold and its results are masked off (vm), instead of a web of per-pair branches whose joins copy the register arrays
					// (the first build: ~900 instructions per row for two pairs)
					V<int> hmax = KSW_NEG_INF;
					uint8_t *trow = tbp + (size_t)r * jb.n_col;                    // column of lane t = t - st
					// first-column / first-row boundary of lane r (:152-155): y, y2, u of that lane are reset before the cells
					if (en >= r && r >= a && r < a + SW) {
						WM_KEEP_BRANCH();
						const int o = r - a;
						const V<int> bm = sel(ln == (o & 63), (o & 64) ? (int)0xffff0000 : 0x0000ffff, 0);
						static_for_desc<BP>([&](auto IC) {
							constexpr int i = decltype(IC)::value;
							if ((o >> 7) == i) { Y[i] = bfi(bm, rep16(tB), Y[i]); Y2[i] = bfi(bm, rep16(tB2), Y2[i]); U[i] = bfi(bm, tb16(sched), U[i]); }
						});
					}
					// exact maximum: H of lane en0 - 1 in the previous row (lane en0 continues from its left neighbour, :322), read before H moves
					int hprev = KSW_NEG_INF, h_en0 = KSW_NEG_INF;
					auto h_of = [&](int t) {
						const int o = t - a, ci = o >> 6;
						int hh = 0;
#pragma unroll
						for (int k = 0; k < (EXACT ? B : 1); ++k) if (ci == k) hh = readlane(H[k], o & 63);
						return hh;
					};
					if constexpr (EXACT) { if (is_last) { WM_KEEP_BRANCH(); hprev = en0 - 1 < a ? m_h : h_of(en0 - 1); } }
					const bool inject = first_here && !moved && st > a;

					// rotations by one thread of x, v, x2 (lane t - 1 of the previous row, see ksw_dp_packed); thread 0 is patched from the pair below
					V<int> rX[BP], rV[BP], rX2[BP];
#pragma unroll
					for (int i = 0; i < BP; ++i) { rX[i] = ror1(X[i]); rV[i] = ror1(Vv[i]); rX2[i] = ror1(X2[i]); }
					static_for_desc<BP>([&](auto IC) {
						constexpr int i = decltype(IC)::value;
						constexpr int wd = i >> 1, psel = (i & 1) ? 0x0c030c02 : 0x0c010c00;
						const int c0 = a + 128 * i;
						// ---- match / mismatch (/ ambiguous) scores, tie-break tag in the low bits ----
						const V<int> xq = TP[wd] ^ QP[wd];
						V<int> sv = pk_mad(pk_minu(perm(xq, xq, psel), one2), rep16(MISt - MCHt), rep16(MCHt));
						if constexpr (HASN) {
							const V<int> oq = TP[wd] | QP[wd];
							const V<int> isn = pk_lshr(perm(oq, oq, psel) & 0x00040004, 2);           // 1 where either code is 4
							sv = bfi(pk_sub(0, isn), rep16(NNt), sv);
						}
						if constexpr (CLIP) {   // the score row is persistent and only [st0, cend] is rewritten
							if (sfull_bits >> i & 1) S[i] = sv;
							else {
								WM_KEEP_BRANCH();
								const V<int> t_lo = ln + c0, t_hi = ln + (c0 + 64);
								const V<int> m = sel(t_lo >= st0 && t_lo <= cend, 0x0000ffff, 0) | sel(t_hi >= st0 && t_hi <= cend, (int)0xffff0000, 0);
								S[i] = bfi(m, sv, S[i]);
							}
							sv = S[i];
						}
						// ---- the cells ----
						V<int> x1, v1, x21;
						if constexpr (i > 0) { x1 = perm(rX[i], rX[i ? i - 1 : 0], rsel); v1 = perm(rV[i], rV[i ? i - 1 : 0], rsel); x21 = perm(rX2[i], rX2[i ? i - 1 : 0], rsel); }
						else { x1 = perm(rX[0], V<int>(px << 16), rsel); v1 = perm(rV[0], V<int>(pv << 16), rsel); x21 = perm(rX2[0], V<int>(px2 << 16), rsel); }
						if (inject) {            // (sm_i is zero in every pair but the one that holds the hull start)
							WM_KEEP_BRANCH(); WM_STRIPE_EVENT(1);
							x1 = bfi(smv[i], rep16(tA), x1); v1 = bfi(smv[i], rep16(negqe16), v1); x21 = bfi(smv[i], rep16(tA2), x21);
						}
						const V<int> ou = U[i];
						V<int> nu, nv, nx, ny, nx2, ny2, p;
						ksw_pcell(cc, sv, x1, v1, x21, Y[i], ou, Y2[i], nu, nv, nx, ny, nx2, ny2, p);
						if (full_bits >> i & 1) {
							U[i] = nu; Vv[i] = nv; X[i] = nx; Y[i] = ny; X2[i] = nx2; Y2[i] = ny2;
							gst(trow, ln + (c0 - st), cast<uint8_t>(p));
							gst(trow, ln + (c0 + 64 - st), cast<uint8_t>(lshr(p, 16)));
						} else {
							WM_KEEP_BRANCH();
							// lanes beyond the hull keep their stale values when the band is clipped (they feed back); lanes below it are dead. An unclipped
							// band never reads either again. Traceback: column t - st, only lanes of the hull are part of the row
							if constexpr (CLIP) {
								const V<int> m = vm[i];
								U[i] = bfi(m, nu, U[i]); Vv[i] = bfi(m, nv, Vv[i]); X[i] = bfi(m, nx, X[i]); Y[i] = bfi(m, ny, Y[i]);
								X2[i] = bfi(m, nx2, X2[i]); Y2[i] = bfi(m, ny2, Y2[i]);
							} else { U[i] = nu; Vv[i] = nv; X[i] = nx; Y[i] = ny; X2[i] = nx2; Y2[i] = ny2; }
							WM_IF((vm[i] & 0xffff) != 0) gst(trow, ln + (c0 - st), cast<uint8_t>(p)); WM_END
							WM_IF(lshr(vm[i], 16) != 0) gst(trow, ln + (c0 + 64 - st), cast<uint8_t>(lshr(p, 16))); WM_END
						}
						if constexpr (EXACT) {
							// H += v (:320-345). Lanes outside the band keep their H; lane en0 takes H of its left neighbour + u
#pragma unroll
							for (int hf = 1; hf >= 0; --hf) {
								const int ci = 2 * i + hf;
								const V<int> v8 = hf ? vhi8(Vv[i]) : vlo8(Vv[i]);
								V<int> hn = H[ci] + v8;
								if (hin_bits >> ci & 1) {                       // chunk strictly inside the band: every lane is a plain update
									H[ci] = hn;
									hmax = vmax(hmax, hn);
								} else {
									WM_KEEP_BRANCH();
									const int en0x = en0 > 0 ? en0 : -1;
									const V<int> t = ln + (c0 + 64 * hf);
									const V<int> u8 = hf ? vhi8(U[i]) : vlo8(U[i]);
									hn = sel(t == en0x, V<int>(u8 + hprev), hn);
									const vbool inb = t >= st0 && t <= en0;
									H[ci] = sel(inb, hn, H[ci]);
									hmax = vmax(hmax, sel(inb, hn, V<int>(KSW_NEG_INF)));
								}
							}
						}
					});
					if constexpr (EXACT) { if (is_last && en0 == tlen - 1) { WM_KEEP_BRANCH(); h_en0 = h_of(en0); } }
					moved = false;
					WM_ST_LAP(WM_ST_CELLS);

					// ---- the lane (uniform) of this stripe as (register, half, thread) ----
					auto half_of = [&](const V<int> (&arr)[BP], int t) { return get_half<BP>(arr, a, t); };

					// ---- this stripe's share of the row's bookkeeping ----
					int hm = KSW_NEG_INF;
					// the left message of this row: values for the next row, and the row-wide quantities accumulated so far
					int pm = KSW_NEG_INF, ppri = -1, hst0 = KSW_NEG_INF;
					int in_th0 = 0, in_tl0 = -1;
					bool stopped = false;
					if (left_now) {
						const int *m = ring_in + (r % R) * L::SLOT;
						int o8[8];
						int stamp = EARLY_MSG ? lds_msg_take(early, o8) : lds_ld_msg(m, o8);
						if (stamp == r) WM_STRIPE_EVENT(9);
						spins = 0;
						while (stamp != r) {
							if (lds_ld_acq(ctrl, L::C_STOP) <= r) { stopped = true; break; }
							WM_STRIPE_SPIN(1, r, a, wv, lds_ld_acq(m, L::M_STAMP)); if (++spins > WM_STRIPE_SPIN_BUDGET) give_up(); spin_pause();
							stamp = lds_ld_msg(m, o8);
						}
						if (stopped) { all_done = true; break; }
						m_x = o8[0]; m_v = o8[1]; m_x2 = o8[2];
						if constexpr (EXACT) {
							m_h = o8[3]; pm = o8[4]; ppri = o8[5]; hst0 = o8[6];
							if (en == a - 1) {                                   // (not a cell of this stripe yet: keep the newest bookkeeping state)
								ez_max = lds_ld(m, (long long)(L::M_EZ + 0)); ez_max_t = lds_ld(m, (long long)(L::M_EZ + 1)); ez_max_q = lds_ld(m, (long long)(L::M_EZ + 2));
								ez_mqe = lds_ld(m, (long long)(L::M_EZ + 3)); ez_mqe_t = lds_ld(m, (long long)(L::M_EZ + 4)); ez_mte = lds_ld(m, (long long)(L::M_EZ + 5));
								ez_mte_q = lds_ld(m, (long long)(L::M_EZ + 6)); ez_score = lds_ld(m, (long long)(L::M_EZ + 7));
							}
						} else if (o8[5] >= 0) { in_th0 = o8[4]; in_tl0 = o8[5]; }
					}
					have_left = left_now;
					WM_ST_LAP(WM_ST_WAIT_LEFT);
					const int ezl = EXACT ? lds_uniform(ezl_raw) : 0;
					if constexpr (EXACT) { if (r > 0 && have_cells) hm = wave_max_i32(hmax); }

					int out_th0 = 0, out_tl0 = -1;
					if constexpr (EXACT) {
						if (r > 0) {
							if (have_cells && hm > KSW_NEG_INF && hm >= pm) {
								// this stripe may hold the row maximum. Its lane priority (the reference's SIMD tie rule, see ksw_dp_packed) is wanted by a new
								// maximum (hm > ez.max >= the stale copy) or by a z-drop test that can fire (ez.max - hm > zdrop; ez.max <= copy + slack)
								int my_pri = -1;
								if (safe || hm > ezl || (zdrop >= 0 && ezl + ez_slack - hm > zdrop)) {
									WM_KEEP_BRANCH();
									const int en1 = st0 + (en0 - st0) / 4 * 4;
									const V<int> g4 = (4 - ((ln + (a - st0)) & 3)) << 20;    // (a, chunk starts: multiples of 4 — the residue is the same in every chunk)
									V<int> best = -1;
									static_for_desc<B>([&](auto CC) {
										constexpr int ci = decltype(CC)::value;
										if ((ci >> 1) < i_lo || (ci >> 1) > i_hi) return;
										const V<int> t = ln + (a + 64 * ci);
										const vbool hit = H[ci] == hm && cast<unsigned>(t - st0) <= (unsigned)(en0 - st0);
										if (any(hit)) {
											WM_KEEP_BRANCH();
											V<int> pri = sel(t < en1, g4, V<int>(0));
											pri = sel(t == en0, V<int>(5 << 20), pri) | (V<int>(0xfffff) - t);
											best = vmax(best, sel(hit, pri, V<int>(-1)));
										}
									});
									my_pri = wave_max_i32(best);
								}
								if (my_pri >= 0) WM_STRIPE_EVENT(7); else WM_STRIPE_EVENT(8);
								if (hm > pm) { pm = hm; ppri = my_pri; }
								else if (my_pri > ppri) ppri = my_pri;
							}
						} else if (a == 0) {                                         // row 0: one cell (:346)
							WM_IF(ln == 0) H[0] = vlo8(Vv[0]) - qe; WM_END
							pm = readlane(H[0], 0); ppri = (5 << 20) | 0xfffff;
							h_en0 = pm;
						}
						if (r - st0 == qlen - 1 && st0 >= a && st0 < a + SW) hst0 = h_of(st0);
						if (is_last) {
							const int max_H = pm, max_t = 0xfffff - (ppri & 0xfffff);
							if (en0 == tlen - 1) { if (h_en0 > ez_mte) ez_mte = h_en0, ez_mte_q = r - en; }
							if (r - st0 == qlen - 1) { if (hst0 > ez_mqe) ez_mqe = hst0, ez_mqe_t = st0; }
							if (max_H > ez_max) {
								if (ppri < 0) { WM_STRIPE_EVENT(3); lds_st_rel(ctrl, L::C_RESTART, 1); lds_st_rel(ctrl, L::C_STOP, -1); all_done = true; break; }
								ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
								lds_st_rel(ctrl, L::C_EZL, ez_max);
							} else if (zdrop >= 0 && ez_max - max_H > zdrop) {       // (otherwise the test of src/ksw2.h:168 cannot fire whatever max_t is)
								if (ppri < 0) { WM_STRIPE_EVENT(3); lds_st_rel(ctrl, L::C_RESTART, 1); lds_st_rel(ctrl, L::C_STOP, -1); all_done = true; break; }
								if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
									const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
									if (ez_max - max_H > zdrop + l * e2) {
										ez_zdropped = 1; my_stop = true; WM_STRIPE_EVENT(6);
										lds_st_rel(ctrl, L::C_STOP, r);
										row_done = r; was_last = true; all_done = true;
										break;
									}
								}
							}
							if (r == n_rows - 1 && en0 == tlen - 1) ez_score = h_en0;
						}
					} else if constexpr (!CLIP && WM_KSW_EDGE_TRACK) {
						// approximate max, band never clips: follow the hull's first lane (see ksw_dp_packed). The owner is the stripe that holds st0
						if (trk) {
							WM_EMU_ASSERT(st0 >= a && st0 < a + SW);
							const int d = (r < qlen ? half_of(Vv, st0) : half_of(U, st0)) >> 8;
							H0 = r ? H0 + d : d - qe;
							if (r == n_rows - 1) ez_score = H0;
							const int st0n = r + 1 - qlen + 1 > 0 ? r + 1 - qlen + 1 : 0;       // st0 of the next row
							if (st0n >= a + SW && r + 1 < n_rows) { WM_STRIPE_EVENT(4); out_th0 = H0; out_tl0 = st0n; trk = false; WM_EMU_ASSERT(pub); }
						}
						if (in_tl0 >= 0) { trk = true; H0 = in_th0; }
					} else {
						// approximate max along one diagonal-ish track (:359-375): the step of row r reads v of lane L0 and u of lane L0 + 1 of THIS row. The
						// track never trails the band (after row r: L0 >= st0(r)), so both lanes are in the owner's stripe or L0 is the left neighbour's last
						// lane, whose v arrives in the left message. When the next step's lane L0 + 1 belongs to the right neighbour and that one computes cells
						// in the next row, the state goes there in this row's message (as long as it does not, L0 + 1 is beyond the band and the step is ours)
						if (trk) {
							if (r > 0) {
								const int L1 = last_H0_t + 1;
								const bool in0 = last_H0_t >= st0 && last_H0_t <= en0, in1 = L1 >= st0 && L1 <= en0;
								int d0 = 0, d1 = 0;
								if (in0) d0 = last_H0_t >= a ? half_of(Vv, last_H0_t) >> 8 : (int)(short)m_v >> 8;
								if (L1 >= a && L1 < a + SW) d1 = half_of(U, L1) >> 8;
								WM_EMU_ASSERT((in0 || (L1 >= a && L1 < a + SW)) && (!in1 || L1 < a + SW) && last_H0_t >= a - 1);
								if (in0 && in1) {
									if (d0 > d1) H0 += d0;
									else H0 += d1, ++last_H0_t;
								} else if (in0) H0 += d0;
								else { ++last_H0_t; H0 += d1; }
							} else { H0 = (half_of(Vv, 0) >> 8) - qe; last_H0_t = 0; }
							if (r == n_rows - 1 && en0 == tlen - 1) ez_score = H0;
							if (last_H0_t + 1 >= a + SW && r + 1 < n_rows) {
								ksw_geo_t gn;
								if (ksw_geo<CLIP>(r + 1, qlen, tlen, w, gn) && a + SW <= gn.en) { WM_STRIPE_EVENT(5); out_th0 = H0; out_tl0 = last_H0_t; trk = false; WM_EMU_ASSERT(pub); }
							}
						}
						if (in_tl0 >= 0) { trk = true; H0 = in_th0; last_H0_t = in_tl0; }
					}

					// ---- publish this row for the right neighbour ----
					WM_ST_LAP(WM_ST_BOOK);
					if (pub) {
						spins = 0;
						while (right_seen < r - R) {                               // (slot r % R still holds row r - R until the right neighbour has consumed it)
							right_seen = lds_ld_acq(prog, right_wv);
							if (right_seen >= r - R) break;
							if (lds_ld_acq(ctrl, L::C_STOP) <= r) { stopped = true; break; } WM_STRIPE_SPIN(2, r, a, wv, right_seen); if (++spins > WM_STRIPE_SPIN_BUDGET) give_up(); spin_pause();
						}
						if (stopped) { all_done = true; break; }
						WM_ST_LAP(WM_ST_WAIT_RIGHT);
						int *m = ring_out + (r % R) * L::SLOT;
						WM_IF(ln == 63)
							lds_st(m, V<int>(L::M_X), lshr(X[BP - 1], 16)); lds_st(m, V<int>(L::M_V), lshr(Vv[BP - 1], 16)); lds_st(m, V<int>(L::M_X2), lshr(X2[BP - 1], 16));
							if constexpr (EXACT) {
								lds_st(m, V<int>(L::M_H), H[EXACT ? B - 1 : 0]); lds_st(m, V<int>(L::M_PM), V<int>(pm)); lds_st(m, V<int>(L::M_PRI), V<int>(ppri)); lds_st(m, V<int>(L::M_HST0), V<int>(hst0));
								if (en == a + SW - 1) {
									lds_st(m, V<int>(L::M_EZ + 0), V<int>(ez_max)); lds_st(m, V<int>(L::M_EZ + 1), V<int>(ez_max_t)); lds_st(m, V<int>(L::M_EZ + 2), V<int>(ez_max_q));
									lds_st(m, V<int>(L::M_EZ + 3), V<int>(ez_mqe)); lds_st(m, V<int>(L::M_EZ + 4), V<int>(ez_mqe_t)); lds_st(m, V<int>(L::M_EZ + 5), V<int>(ez_mte));
									lds_st(m, V<int>(L::M_EZ + 6), V<int>(ez_mte_q)); lds_st(m, V<int>(L::M_EZ + 7), V<int>(ez_score));
								}
							} else { lds_st(m, V<int>(L::M_TH0), V<int>(out_th0)); lds_st(m, V<int>(L::M_TL0), V<int>(out_tl0)); }
						WM_END
						lds_st_rel(m, L::M_STAMP, r);
					}
					lds_st_rel(prog, wv, r - 1);                 // the left ring's messages up to row r - 1 may be overwritten
					row_done = r; was_last = is_last;
					WM_ST_LAP(WM_ST_PUBLISH);
					if (r >= lds_uniform(stop_raw)) { all_done = true; break; }   // (a z-drop in row stop_row: nothing after it exists)
				}
			}
		}

		// ---- the alignment is over: one wavefront holds the result ----
		// (a wavefront that has left the row loop reads no message any more: its left neighbour may still be publishing — e.g. the band runs empty a
		// few rows before the stripe this wavefront was waiting for is reached — and must not wait for it)
		lds_st_rel(prog, wv, BIG);
		WM_ST_LAP(WM_ST_SCAN);
		block_sync_lds();
		const int stop_row = lds_ld_acq(ctrl, L::C_STOP), restart = lds_ld_acq(ctrl, L::C_RESTART);
		if (restart == 2) {                                          // the watchdog: no result; the batch fails loudly (see WM_STRIPE_SPIN_BUDGET)
			if (wv == 0) {
				WM_IF(ln == 0)
					wm_ksw_dres_t o;
					o.max = 0; o.zdropped = 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1; o.mqe = o.mte = o.score = KSW_NEG_INF;
					o.reach_end = 0; o.n_cigar = 0; o.bt_i = KSW_BT_WATCHDOG; o.bt_j = -1;
					*res = o;
				WM_END
			}
			break;
		}
		if (restart) { block_sync_lds(); continue; }
		bool writer;
		if (EXACT) writer = stop_row != BIG ? my_stop : (was_last && row_done == end_row - 1);
		else writer = trk;
		if (writer) {
			if (end_row < n_rows) ez_zdropped = 1;          // the band ran empty (:136-139)
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
		break;
	}
	WM_ST_FLUSH();
}

} // namespace wmk
