// ksw_chain_kernel.h — ksw_extd2_sse (src/ksw2_extd2_sse.c:26-393) for wide band hulls and long alignments, ONE ALIGNMENT OVER SEVERAL COMPUTE UNITS:
// the stripe machine of ksw_stripe_kernel.h with every wavefront a workgroup of its own and the row messages in global memory (round 6).
//
// Why. ksw_dp_stripe keeps the NWV wavefronts of an alignment in one workgroup, i.e. on ONE compute unit: a row of a 3 000-lane hull is 24 register
// pairs x ~90 VALU instructions = ~2 200 instructions issued by four SIMDs, ~0.9 us at best and 2.0-2.3 us measured (profiles/r05_last_bench.json:
// the <2,16> classes retire 1.1 VALU per cell and sit at 30 GCUPS) — the workgroup is bound by the issue rate of its CU while the chip as a whole
// issues half of what it could, and sixteen wavefronts of ~110 VGPRs fill the CU's register files, so no bulk wavefront shares it. A geometry past
// 7 168 lanes does not fit one CU at all (VERDICT r5 missing 4). Here stripe s (SW = 128 * BP target lanes) belongs to wavefront s % nwv of the
// job, nwv = ceil(n_col / SW) + 1 chosen per job (any hull width), each wavefront is a 64-thread workgroup the dispatcher places wherever a SIMD
// has room, and a row costs what ONE wavefront issues for its BP pairs — the left neighbour is simply further ahead in the pipeline.
//
// The protocol (cell arithmetic, epochs, bookkeeping hand-over, tie rule: exactly ksw_dp_stripe; what differs is only how wavefronts talk):
//   * MAILBOX. Per job a block of 64-bit words in HBM (wm_chain_box): a STOP word, one progress word per wavefront, one ring of R = 128 row slots
//     x 16 words per wavefront (messages to its right neighbour). Every word is {value, row stamp} written and read as ONE relaxed agent-scope
//     atomic (simt.h: mbox_*): coherent across the XCDs' L2s without a fence, and self-validating — a reader that finds stamp r in a word holds the
//     value published for row r. The host fills the block with 0xff before the launch (stamp -1, progress -1, STOP "none").
//   * READING AHEAD. A message round trip through L2 / the fabric is ~1-2 us, a row ~0.5 us: the reader fetches the slots of FOUR rows with one
//     64-lane load (lanes 16j .. 16j+15 = row g + j) and keeps the next group in flight (two VGPR pairs); a word whose stamp is not there yet is
//     re-read in a polling loop. In the steady state the left neighbour is a ring-load ahead and no row waits for memory.
//   * NO DEADLOCK BY CONSTRUCTION. Workgroups take a ticket from a device counter when they START (ksw_chain_kernel) and the ticket, not blockIdx,
//     names (job, wavefront): every lower ticket is then running or finished, whatever order the dispatcher chose. A wavefront waits for messages
//     of a LOWER-ticket neighbour, or — wavefront 0 on its second stripe, and back-pressure (a producer runs at most R rows ahead of its consumer) —
//     for a higher one of the same job; the lowest unfinished job's wavefronts are at most a few dozen, so the chip always has room to start the
//     next ticket. Every polling loop has the watchdog of ksw_stripe_kernel.h (WM_STRIPE_SPIN_BUDGET): the job then reports KSW_BT_WATCHDOG and the
//     batch fails with WM_EINTERNAL instead of hanging.
//   * NO BARRIER, NO SECOND PASS. The lane priority of the exact row maximum is evaluated whenever a stripe ties or beats the prefix maximum it
//     received (ksw_dp_stripe's `safe` mode: no stale-ez.max heuristic, hence no restart); the result is written by the wavefront that did the
//     bookkeeping of the last row (rows are booked strictly in order: the state travels with the band's last lane) or that z-dropped.
//     A z-drop stores its row in STOP; waiting wavefronts see it in their polling loops, and a wavefront that leaves because of it does NOT
//     release its back-pressure, so producers further left stop within R rows.
#pragma once
#ifndef WM_DEV
#error "include simt.h before ksw_chain_kernel.h"
#endif
#include "ksw_stripe_kernel.h"
#ifndef WM_CHAIN_EVENT
#define WM_CHAIN_EVENT(k) ((void)0)       // test hook (tests/simt_emu/emu_chain.cpp)
#endif
#ifndef WM_CHAIN_SPIN
#define WM_CHAIN_SPIN(where, r, a, wv, extra) ((void)0)
#endif
#ifndef WM_CHAIN_BACKOFF
#define WM_CHAIN_BACKOFF 2                // long pauses (~3.4 us each) of a consumer whose prefetched group was not there (measured neutral between 0 and 8: misses are rare)
#endif

// mailbox geometry of one job, in 64-bit words (host and device)
struct wm_chain_box {
	static constexpr int R = 128, SLOT = 16, GROUP = 4;              // ring slots per wavefront, words per slot, rows fetched by one load
	static constexpr int I_STOP = 0, I_PROG = 16;                    // 32-bit views of the head: STOP word, progress words
#ifdef __HIPCC__
	__host__ __device__
#endif
	static inline int head_words(int nwv) { return (((I_PROG + nwv + 1) / 2) + 15) & ~15; }
#ifdef __HIPCC__
	__host__ __device__
#endif
	static inline long long words(int nwv) { return (long long)head_words(nwv) + (long long)nwv * R * SLOT; }
};
// wavefronts of a job on stripes of sw lanes: every stripe the hull (n_col lanes incl. its slack) can touch at once, plus the one being entered;
// never more than there are stripes
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int wm_chain_nwv(int n_col, int tlen, int sw)
{
	const int need = (n_col + sw - 1) / sw + 1, ns = (tlen + sw - 1) / sw;
	return need < ns ? need : (ns > 0 ? ns : 1);
}

namespace wmk {

// message words: x v x2 h | pm ppri hst0 (exact maximum) or track H0, track lane (-1: no hand-over) | 8 ez words from word 8 (only on rows where the state moves)
enum { CM_X = 0, CM_V = 1, CM_X2 = 2, CM_H = 3, CM_PM = 4, CM_PRI = 5, CM_HST0 = 6, CM_TH0 = 4, CM_TL0 = 5, CM_EZ = 8 };

template <int BP, bool CLIP, bool HASN, bool EXACT>
WM_DEV void ksw_dp_chain(const wm_ksw_score_t sc, const wm_ksw_djob_t jb, const uint8_t *__restrict__ seqs,
                         uint8_t *__restrict__ tb_arena, wm_mbox_t *mb, const int nwv, const int wv, int *tbs /* LDS: wm_chain_box::GROUP * 32 * BP ints */,
                         wm_ksw_dres_t *__restrict__ res)
{
	typedef wm_chain_box L;
	static_assert(BP == 2 || BP == 4, "BP");
	constexpr int SW = 128 * BP, B = 2 * BP, NW = (B + 3) / 4, R = L::R, NDW = BP / 2;       // NDW: traceback dwords per thread and row (4 chunks each)
	constexpr int BIG = 0x7fffffff;
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

	const V<int> ln = lane();
	const V<int> rsel = sel(ln == 0, 0x05040302, 0x07060504);      // v_perm_b32 selectors: {own.lo, prev.hi} for thread 0, own elsewhere
	const V<int> qsel = sel(ln == 0, 0x06050403, 0x07060504);      // code words: thread 0 takes {own bytes 2..0, prev byte 3}
	int *ctl = (int*)mb;
	wm_mbox_t *rings = mb + L::head_words(nwv);
	wm_mbox_t *ring_out = rings + (long long)wv * (R * L::SLOT);
	const wm_mbox_t *ring_in = rings + (long long)((wv + nwv - 1) % nwv) * (R * L::SLOT);
	const int right_wv = (wv + 1) % nwv;
	int spins = 0;                                                   // polls of the wait this wavefront is in
	bool gave_up = false;
	WM_ST_DECL();                                                    // (WM_STRIPE_TIMING builds only: cycles per phase of the row loop, ksw_stripe_kernel.h)
	auto give_up = [&]() {                                           // the watchdog: the batch fails loudly (ksw_backtrack_kernel turns KSW_BT_WATCHDOG into the error flag)
		gave_up = true;
		mbox_st_word(ctl, L::I_STOP, 0);
		WM_IF(ln == 0)
			wm_ksw_dres_t o;
			o.max = 0; o.zdropped = 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1; o.mqe = o.mte = o.score = KSW_NEG_INF;
			o.reach_end = 0; o.n_cigar = 0; o.bt_i = KSW_BT_WATCHDOG; o.bt_j = -1;
			*res = o;
		WM_END
	};
	auto stop_by = [&](int row) { return (unsigned)mbox_ld_word(ctl, L::I_STOP) <= (unsigned)row; };      // a z-drop in a row <= `row` (or the watchdog) ended the alignment

	// ---- the inbox: the left neighbour's slots of four rows per load, the next group in flight ----
	V<long long> wcur = 0;
	int cur_g = -4, nxt_g = -4;                                       // row groups (multiples of 4) held by wcur / on their way (simt.h: mbox_prefetch)
	int my_prog = -1;                                                // rows of the left ring up to here may be overwritten (published in I_PROG + wv)
	auto ld_group = [&](int g) { return mbox_ld(ring_in + (g & (R - 1)) * L::SLOT, ln); };
	auto publish_prog = [&](int p) { if (p > my_prog) { my_prog = p; mbox_st_word(ctl, L::I_PROG + wv, p); } };
	// Where the waits go matters more than what they wait for (ksw_packed_kernel.h: loads_land): gfx9 has ONE counter for vector loads and stores, and a wait the
	// compiler places at the JOIN behind a rare branch runs on every row. Every load below is waited for explicitly INSIDE the branch that issued it; the
	// prefetch of the next group is invisible to the compiler altogether and is collected at the next group switch, a group of rows after it was issued.
	auto need_group = [&](int g) {
		if (cur_g != g) {
			WM_KEEP_BRANCH();
			if (nxt_g == g) wcur = mbox_prefetched(); else { wcur = ld_group(g); loads_land(); }
			cur_g = g;
			mbox_prefetch(ring_in + ((g + L::GROUP) & (R - 1)) * L::SLOT, ln); nxt_g = g + L::GROUP;
			publish_prog(g - 1);                                         // groups below g are never read again
		}
	};
	// the message of row `row`: false = a stop ended the alignment first. o[0..7] = words 0..7; ez[0..7] = words 8..15 when want_ez
	auto take = [&](int row, bool want_ez, int (&o)[8], int (&ez)[8], int where, int a_) -> bool {
		need_group(row & ~(L::GROUP - 1));
		const int j16 = (row & (L::GROUP - 1)) * L::SLOT, hi = j16 + (want_ez ? 16 : 8);
		const vbool mine = ln >= j16 && ln < hi;
		if (any(mine && mbox_stamp(wcur) != row)) {                       // not there yet: poll (the reload is waited for inside the loop)
			WM_KEEP_BRANCH();
			// A consumer that polls eagerly stays right behind its producer: the group it prefetches is never there yet, and EVERY group then costs a
			// polling round trip (measured: ~1 poll per group, profiles/r06_chain_timing_v2.txt). So a miss makes this wavefront fall back on purpose,
			// by about three groups of rows, once: from then on its prefetches find their rows as long as it is not faster than its producer.
			for (int k = 0; k < WM_CHAIN_BACKOFF; ++k) long_pause();
			spins = 0;
			do {
				if (stop_by(row)) return false;
				WM_CHAIN_SPIN(where, row, a_, wv, 0);
				if (++spins > WM_STRIPE_SPIN_BUDGET) { give_up(); return false; }
				poll_pause(spins);                                           // (hundreds of wavefronts wait for their stripe at any time: uncached loads of a few hot lines, so not too often)
				wcur = ld_group(row & ~(L::GROUP - 1));
				loads_land();
			} while (any(mine && mbox_stamp(wcur) != row));
			if (nxt_g == cur_g + L::GROUP) mbox_prefetch(ring_in + (nxt_g & (R - 1)) * L::SLOT, ln);      // (the prefetch was taken before the pause: again, now that the producer is ahead)
		}
		const V<int> lo = mbox_val(wcur);
#pragma unroll
		for (int k = 0; k < 8; ++k) o[k] = readlane(lo, j16 + k);
		if (want_ez) {
#pragma unroll
			for (int k = 0; k < 8; ++k) ez[k] = readlane(lo, j16 + 8 + k);
		}
		return true;
	};

	// ---- a GROUP of rows at a time to memory: traceback and outgoing messages ----
	// Every vector-memory wait of the row loop (there is one counter for loads and stores) sits at a group switch, where everything outstanding was
	// issued four rows ago: the traceback bytes of a row go to LDS (one dword per thread = its four chunks' cells) and the rows of a group are written
	// from there with dword stores when the next group begins; the messages to the right neighbour are collected in one register (lanes 16j + k = word k
	// of row g + j) and leave with ONE 64-lane store. First measurement of the per-row form (profiles/r06_chain_timing_v1.txt): 2.1 us per row of which
	// 0.3 + 0.4 + 0.13 us were such waits — a wait for the prefetched group also drained the row's traceback stores (~1.2 us to HBM and back).
	int stg_g = -4, stg_mask = 0, stg_a = 0;                         // the group being staged (rows stg_g ..), its rows with traceback in LDS, the stripe they belong to
	int out_mask = 0;                                                // ... its rows with a message in outv
	V<int> outv = 0;
	auto flush_group = [&]() {
		if (stg_mask) {
			WM_KEEP_BRANCH();
			lds_sync();
			const V<int> l0 = (ln << 2) & 63, c = (ln >> 4) & 3;          // dword ln of a 256-lane block = target lanes 4 ln ..: chunk c, threads l0 .. l0 + 3
			const V<int> psel = ((c + 4) << 8) | c;                       // v_perm_b32: {byte c of the first operand's partner, byte c of the other}
#pragma unroll
			for (int j = 0; j < L::GROUP; ++j) {
				if (!(stg_mask >> j & 1)) continue;
				const int rr = stg_g + j;
				ksw_geo_t gg;
				ksw_geo<CLIP>(rr, qlen, tlen, w, gg);                    // (a staged row exists: its band is not empty)
				int *trow = (int*)(tbp + (size_t)rr * jb.n_col);         // column of lane t = t - st; st, n_col, the stripe start: multiples of 16 -> dword aligned
#pragma unroll
				for (int d = 0; d < NDW; ++d) {
					V<int> q4[4];
					lds_ld4(tbs, (j * NDW + d) * 64 + l0, q4);
					const V<int> lo = perm(q4[1], q4[0], psel), hi = perm(q4[3], q4[2], psel);
					const V<int> word = perm(hi, lo, 0x05040100);
					const V<int> t0 = (ln << 2) + (stg_a + 256 * d);
					WM_IF(t0 >= gg.st && t0 <= gg.en) gst(trow, (t0 - gg.st) >> 2, word); WM_END
				}
			}
			stg_mask = 0;
		}
		if (out_mask) {
			WM_KEEP_BRANCH();
			const V<int> jr = ln >> 4;
			WM_IF(((V<int>(out_mask) >> jr) & 1) != 0) mbox_st(ring_out + (stg_g & (R - 1)) * L::SLOT, ln, mbox_pack(outv, jr + stg_g)); WM_END
			out_mask = 0;
		}
	};

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
	bool stopped = false;                   // it left because somebody else ended the alignment
	int row_done = -1;                      // last row this wavefront completed
	int end_row = n_rows;                   // first row that does not exist (n_rows, or the first row with an empty band)
	int r = 0, s = wv;
	bool all_done = false;
	int right_seen = -1;                    // lower bound of the right neighbour's progress word
	auto load_ez = [&](const int (&ez)[8]) {
		ez_max = ez[0]; ez_max_t = ez[1]; ez_max_q = ez[2]; ez_mqe = ez[3]; ez_mqe_t = ez[4]; ez_mte = ez[5]; ez_mte_q = ez[6]; ez_score = ez[7];
	};

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
		publish_prog(r - 2);                    // nothing before row r - 1 is wanted from the left ring
		if (r > 0) {
			ksw_geo_t gp;
			if (ksw_geo<CLIP>(r - 1, qlen, tlen, w, gp)) {
				prev_st = gp.st;
				if (a > 0 && gp.st <= a - 1 && a - 1 <= gp.en) {      // lane a - 1 was computed in row r - 1: take its message
					int o8[8], ez8[8];
					const bool want_ez = EXACT && gp.en == a - 1;       // the band's last lane sat right below this stripe: the bookkeeping state comes along
					if (!take(r - 1, want_ez, o8, ez8, 0, a)) { stopped = !gave_up; all_done = true; break; }
					m_x = o8[CM_X]; m_v = o8[CM_V]; m_x2 = o8[CM_X2]; m_h = o8[CM_H];
					have_left = true;
					if (want_ez) { WM_CHAIN_EVENT(2); load_ez(ez8); }
					if (!EXACT && o8[CM_TL0] >= 0) { trk = true; H0 = o8[CM_TH0]; last_H0_t = o8[CM_TL0]; }
				}
			}
		}

		// ================= the rows of stripe s, EPOCH by epoch (ksw_dp_stripe) =================
		bool leave = false;
		while (!all_done && !leave) {
			if (r >= n_rows) { all_done = true; break; }
			if (!ksw_geo<CLIP>(r, qlen, tlen, w, g)) { end_row = r; all_done = true; break; }
			const int st = g.st, en = g.en;
			if (st >= a + SW) { flush_group(); stg_g = -4; s += nwv; WM_CHAIN_EVENT(0); leave = true; break; }          // the hull has left this stripe for good (its staged rows go out; the next stripe starts a group of its own)
			int r_end = n_rows;                                            // first row of the next epoch
			{
				const int X_ = st + 16;                                    // st0 reaches X_: r - qlen + 1 >= X_, or (r - w + 1) >> 1 >= X_
				int rs = X_ + qlen - 1;
				if (CLIP && 2 * X_ + w - 1 < rs) rs = 2 * X_ + w - 1;
				if (rs < r_end) r_end = rs;
				const int Y_ = en + 1;                                     // en0 reaches Y_: Y_ <= tlen - 1, r >= Y_ and (r + w) >> 1 >= Y_
				if (Y_ <= tlen - 1) {
					int re = Y_;
					if (CLIP && 2 * Y_ - w > re) re = 2 * Y_ - w;
					if (re < r_end) r_end = re;
				}
			}
			WM_EMU_ASSERT(r_end > r);
			const bool have_cells = a <= en;
			const int i_lo = st > a ? (st - a) >> 7 : 0;
			const int i_hi = have_cells ? ((en - a) >> 7 < BP ? (en - a) >> 7 : BP - 1) : -1;
			const bool first_here = st >= a;                                // the pair i_lo holds the hull start
			const bool pub = a + SW - 1 <= en && a + SW < tlen;             // this stripe's last lane is computed and there is a stripe to its right: the right neighbour wants it
			const bool left_now = a > 0 && st <= a - 1 && a - 1 <= en;      // the left neighbour publishes a message for every row of the epoch
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
			bool moved = st > prev_st;                                     // lane st - 1 was computed in the last row (:141-146): first row of an epoch only
			prev_st = st;

			WM_ST_LAP(WM_ST_EPOCH); WM_ST_COUNT(WM_ST_EPOCHS);
			for (; r < r_end; ++r) {
				WM_ST_COUNT(WM_ST_ROWS);
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
				if ((r & ~(L::GROUP - 1)) != stg_g) {                        // a new group of rows begins
					WM_KEEP_BRANCH();
					if (left_now) need_group(r & ~(L::GROUP - 1));           // (first: its wait then only meets what was issued a group ago)
					WM_ST_LAP(WM_ST_SCAN);
					flush_group();
					WM_ST_LAP(WM_ST_WAIT_RIGHT);                             // (diagnostic builds: this slot = writing the finished group out)
					stg_g = r & ~(L::GROUP - 1); stg_a = a;
					if (pub && right_seen < stg_g + L::GROUP - 1 - R) {      // (the slots of this group still hold rows - R until the right neighbour has consumed them)
						spins = 0;
						bool halt = false;
						for (;;) {
							right_seen = mbox_ld_word(ctl, L::I_PROG + right_wv);
							if (right_seen >= stg_g + L::GROUP - 1 - R) break;
							if (stop_by(r)) { halt = true; stopped = true; break; }
							WM_CHAIN_SPIN(2, r, a, wv, right_seen);
							if (++spins > WM_STRIPE_SPIN_BUDGET) { give_up(); halt = true; break; }
							poll_pause(spins);
						}
						if (halt) { all_done = true; break; }
					}
				} else if (left_now) need_group(r & ~(L::GROUP - 1));
				WM_ST_LAP(WM_ST_SCAN);                                       // (diagnostic builds: the group switch = the wait for the prefetched group)

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

				// previous-row values of the lane below the stripe (ksw_dp_stripe)
				int px = tA, pv = (st == 0 ? sched & 0xff : (-qe) & 0xff) << 8, px2 = tA2;
				if (have_left) { px = m_x; pv = m_v; px2 = m_x2; }
				const bool is_last = have_cells && en0 < a + SW;                // the band's last lane is here: this wavefront closes the row
				V<int> hmax = KSW_NEG_INF;
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

				V<int> rX[BP], rV[BP], rX2[BP], tbp_[BP];
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
					if (inject) {            // (smv[i] is zero in every pair but the one that holds the hull start)
						WM_KEEP_BRANCH(); WM_CHAIN_EVENT(1);
						x1 = bfi(smv[i], rep16(tA), x1); v1 = bfi(smv[i], rep16(negqe16), v1); x21 = bfi(smv[i], rep16(tA2), x21);
					}
					const V<int> ou = U[i];
					V<int> nu, nv, nx, ny, nx2, ny2, p;
					ksw_pcell(cc, sv, x1, v1, x21, Y[i], ou, Y2[i], nu, nv, nx, ny, nx2, ny2, p);
					tbp_[i] = p;                                             // (bits 0-7 of each half: the traceback bytes of lanes c0 + thread, c0 + 64 + thread)
					if (full_bits >> i & 1) {
						U[i] = nu; Vv[i] = nv; X[i] = nx; Y[i] = ny; X2[i] = nx2; Y2[i] = ny2;
					} else {
						WM_KEEP_BRANCH();
						if constexpr (CLIP) {
							const V<int> m = vm[i];
							U[i] = bfi(m, nu, U[i]); Vv[i] = bfi(m, nv, Vv[i]); X[i] = bfi(m, nx, X[i]); Y[i] = bfi(m, ny, Y[i]);
							X2[i] = bfi(m, nx2, X2[i]); Y2[i] = bfi(m, ny2, Y2[i]);
						} else { U[i] = nu; Vv[i] = nv; X[i] = nx; Y[i] = ny; X2[i] = nx2; Y2[i] = ny2; }
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
				if (have_cells) {    // the row's traceback: one dword per thread and four chunks, staged in LDS until the group is written (lanes outside the hull are masked there)
#pragma unroll
					for (int d = 0; d < NDW; ++d)
						lds_st(tbs, ln + ((r & (L::GROUP - 1)) * NDW + d) * 64, perm(tbp_[2 * d + 1], tbp_[2 * d], 0x06040200));
					stg_mask |= 1 << (r & (L::GROUP - 1));
				}
				if constexpr (EXACT) { if (is_last && en0 == tlen - 1) { WM_KEEP_BRANCH(); h_en0 = h_of(en0); } }
				moved = false;
				WM_ST_LAP(WM_ST_CELLS);

				auto half_of = [&](const V<int> (&arr)[BP], int t) { return get_half<BP>(arr, a, t); };

				// ---- this stripe's share of the row's bookkeeping ----
				int hm = KSW_NEG_INF;
				int pm = KSW_NEG_INF, ppri = -1, hst0 = KSW_NEG_INF;
				int in_th0 = 0, in_tl0 = -1;
				if (left_now) {
					int o8[8], ez8[8];
					const bool want_ez = EXACT && en == a - 1;               // (not a cell of this stripe yet: keep the newest bookkeeping state)
					if (!take(r, want_ez, o8, ez8, 1, a)) { stopped = !gave_up; all_done = true; break; }
					m_x = o8[CM_X]; m_v = o8[CM_V]; m_x2 = o8[CM_X2];
					if constexpr (EXACT) {
						m_h = o8[CM_H]; pm = o8[CM_PM]; ppri = o8[CM_PRI]; hst0 = o8[CM_HST0];
						if (want_ez) load_ez(ez8);
					} else if (o8[CM_TL0] >= 0) { in_th0 = o8[CM_TH0]; in_tl0 = o8[CM_TL0]; }
				}
				have_left = left_now;
				WM_ST_LAP(WM_ST_WAIT_LEFT);
				if constexpr (EXACT) { if (r > 0 && have_cells) hm = wave_max_i32(hmax); }

				int out_th0 = 0, out_tl0 = -1;
				if constexpr (EXACT) {
					if (r > 0) {
						if (have_cells && hm > KSW_NEG_INF && hm >= pm) {
							// this stripe may hold the row maximum: its lane priority (the reference's SIMD tie rule, see ksw_dp_packed)
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
							const int my_pri = wave_max_i32(best);
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
						WM_EMU_ASSERT(ppri >= 0 || r == 0 || max_H <= KSW_NEG_INF);
						if (en0 == tlen - 1) { if (h_en0 > ez_mte) ez_mte = h_en0, ez_mte_q = r - en; }
						if (r - st0 == qlen - 1) { if (hst0 > ez_mqe) ez_mqe = hst0, ez_mqe_t = st0; }
						if (max_H > ez_max) {
							ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
						} else if (zdrop >= 0 && ez_max - max_H > zdrop) {       // (otherwise the test of src/ksw2.h:168 cannot fire whatever max_t is)
							if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
								const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
								if (ez_max - max_H > zdrop + l * e2) {
									ez_zdropped = 1; my_stop = true; WM_CHAIN_EVENT(6);
									mbox_st_word(ctl, L::I_STOP, r);
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
						if (st0n >= a + SW && r + 1 < n_rows) { WM_CHAIN_EVENT(4); out_th0 = H0; out_tl0 = st0n; trk = false; WM_EMU_ASSERT(pub); }
					}
					if (in_tl0 >= 0) { trk = true; H0 = in_th0; }
				} else {
					// approximate max along one diagonal-ish track (:359-375), see ksw_dp_stripe
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
						if (last_H0_t + 1 >= a + SW && a + SW < tlen && r + 1 < n_rows) {
							ksw_geo_t gn;
							if (ksw_geo<CLIP>(r + 1, qlen, tlen, w, gn) && a + SW <= gn.en) { WM_CHAIN_EVENT(5); out_th0 = H0; out_tl0 = last_H0_t; trk = false; WM_EMU_ASSERT(pub); }
						}
					}
					if (in_tl0 >= 0) { trk = true; H0 = in_th0; last_H0_t = in_tl0; }
				}

				// ---- this row's message for the right neighbour joins the group's register ----
				WM_ST_LAP(WM_ST_BOOK);
				if (pub) {
					if (right_seen < stg_g + L::GROUP - 1 - R) {               // (only when this stripe began to publish inside the group: the check at the group switch did not run)
						WM_KEEP_BRANCH();
						spins = 0;
						bool halt = false;
						for (;;) {
							right_seen = mbox_ld_word(ctl, L::I_PROG + right_wv);
							if (right_seen >= stg_g + L::GROUP - 1 - R) break;
							if (stop_by(r)) { halt = true; stopped = true; break; }
							WM_CHAIN_SPIN(2, r, a, wv, right_seen);
							if (++spins > WM_STRIPE_SPIN_BUDGET) { give_up(); halt = true; break; }
							poll_pause(spins);
						}
						if (halt) { all_done = true; break; }
					}
					WM_ST_LAP(WM_ST_WAIT_RIGHT);
					const bool with_ez = EXACT && en == a + SW - 1;
					const int o_x = lshr(readlane(X[BP - 1], 63), 16), o_v = lshr(readlane(Vv[BP - 1], 63), 16), o_x2 = lshr(readlane(X2[BP - 1], 63), 16);
					const int o_h = EXACT ? readlane(H[EXACT ? B - 1 : 0], 63) : 0;
					auto put_row = [&](auto JC) {                              // lanes 16 j + k of outv = word k of row stg_g + j
						constexpr int j16 = decltype(JC)::value * L::SLOT;
						outv = wrlane<j16 + CM_X>(outv, o_x); outv = wrlane<j16 + CM_V>(outv, o_v); outv = wrlane<j16 + CM_X2>(outv, o_x2);
						if constexpr (EXACT) {
							outv = wrlane<j16 + CM_H>(outv, o_h); outv = wrlane<j16 + CM_PM>(outv, pm); outv = wrlane<j16 + CM_PRI>(outv, ppri); outv = wrlane<j16 + CM_HST0>(outv, hst0);
							if (with_ez) {
								WM_KEEP_BRANCH();
								outv = wrlane<j16 + CM_EZ + 0>(outv, ez_max); outv = wrlane<j16 + CM_EZ + 1>(outv, ez_max_t); outv = wrlane<j16 + CM_EZ + 2>(outv, ez_max_q);
								outv = wrlane<j16 + CM_EZ + 3>(outv, ez_mqe); outv = wrlane<j16 + CM_EZ + 4>(outv, ez_mqe_t); outv = wrlane<j16 + CM_EZ + 5>(outv, ez_mte);
								outv = wrlane<j16 + CM_EZ + 6>(outv, ez_mte_q); outv = wrlane<j16 + CM_EZ + 7>(outv, ez_score);
							}
						} else { outv = wrlane<j16 + CM_TH0>(outv, out_th0); outv = wrlane<j16 + CM_TL0>(outv, out_tl0); }
					};
					switch (r & (L::GROUP - 1)) {
					case 0: put_row(std::integral_constant<int, 0>{}); break;
					case 1: put_row(std::integral_constant<int, 1>{}); break;
					case 2: put_row(std::integral_constant<int, 2>{}); break;
					default: put_row(std::integral_constant<int, 3>{}); break;
					}
					out_mask |= 1 << (r & (L::GROUP - 1));
				}
				row_done = r; was_last = is_last;
				WM_ST_LAP(WM_ST_PUBLISH);
			}
		}
	}
	WM_ST_FLUSH();

	if (!gave_up) flush_group();            // (every way out of the row loop: the staged traceback rows are part of the result up to the stop row)
	// ---- the alignment is over for this wavefront ----
	// A wavefront that leaves normally reads no message any more and must not hold its left neighbour back. One that leaves because somebody ELSE ended
	// the alignment keeps its progress word: the producers to its left then run into back-pressure within R rows and find the STOP word in that loop.
	if (!stopped && !gave_up) mbox_st_word(ctl, L::I_PROG + wv, BIG);
	if (gave_up) return;
	bool writer;
	if (EXACT) writer = my_stop || (!stopped && was_last && row_done == end_row - 1);
	else writer = !stopped && trk;
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
}

} // namespace wmk
