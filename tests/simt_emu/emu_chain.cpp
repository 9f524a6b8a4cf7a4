// emu_chain.cpp — TEST INFRASTRUCTURE: the chained-workgroup ksw kernel (winnowmap_amd/csrc/ksw_chain_kernel.h) on the host wavefront emulator.
// One host thread per wavefront of the job (nwv = wm_chain_nwv: chosen per job, as the launcher does); the mailbox is plain memory filled with 0xff and
// the mailbox words are 64-bit atomics, exactly the device protocol. The host scheduler supplies the interleavings; `start_skew` > 0 additionally starts
// the wavefronts in ticket order with a delay, so that consumers really are not there yet when their producers run into back-pressure.
// WM_CHAIN_EVENT counts the rare paths, WM_CHAIN_SPIN is a watchdog on the polling loops (a protocol deadlock aborts with the place).
//   emu_chain_extd2(..., force): 400 + bp_index * 10 + (CLIP * 2 + HASN); bp_index 1, 2 = 2, 4 register pairs per wavefront (256 / 512-lane stripes)
#include <atomic>
static std::atomic<long> g_ev[16];
#define WM_CHAIN_EVENT(k) (++g_ev[k])
#include <stdio.h>
#include <stdlib.h>
static thread_local long g_spin = 0;
#define WM_CHAIN_SPIN(where, r, a, wv, extra) do { if (++g_spin > 20000000) { fprintf(stderr, "CHAIN SPIN where %d r %d a %d wv %d extra %d\n", where, r, a, wv, (int)(extra)); g_spin = 0; static int n = 0; if (++n > 6) abort(); } } while (0)
#include "simt.h"
#include "ksw_kernel.h"
#include "ksw_packed_kernel.h"
#include "ksw_chain_kernel.h"
#include "ksw_plan.h"
#include <vector>
#include <thread>
#include <chrono>
#include <algorithm>
static int g_start_skew_us = 0;
template <int BP> static int run_chain(int variant, const wm_ksw_score_t &sc, const wm_ksw_djob_t &jb, const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res)
{
	const bool exact = variant & 4, clip = variant & 2, hasn = variant & 1;
	const int nwv = wm_chain_nwv(jb.n_col, jb.tlen, 128 * BP);
	std::vector<wm_mbox_t> mb((size_t)wm_chain_box::words(nwv), ~0ull);
	std::vector<std::thread> th;
	for (int w = 0; w < nwv; ++w) {
		if (g_start_skew_us > 0 && w > 0) std::this_thread::sleep_for(std::chrono::microseconds(g_start_skew_us));
		th.emplace_back([&, w]() {
			simt::wave_slot() = 0; simt::exec_mask() = ~0ull;
			std::vector<int> tbsv(wm_chain_box::GROUP * 32 * BP, 0x5a5a5a5a);       // the wavefront's LDS: traceback rows of the group being staged
			int *tbs = tbsv.data();
			if (exact) {
				if (clip && hasn) wmk::ksw_dp_chain<BP, true, true, true>(sc, jb, seqs, tb, mb.data(), nwv, w, tbs, res);
				else if (clip) wmk::ksw_dp_chain<BP, true, false, true>(sc, jb, seqs, tb, mb.data(), nwv, w, tbs, res);
				else if (hasn) wmk::ksw_dp_chain<BP, false, true, true>(sc, jb, seqs, tb, mb.data(), nwv, w, tbs, res);
				else wmk::ksw_dp_chain<BP, false, false, true>(sc, jb, seqs, tb, mb.data(), nwv, w, tbs, res);
			} else {
				if (clip && hasn) wmk::ksw_dp_chain<BP, true, true, false>(sc, jb, seqs, tb, mb.data(), nwv, w, tbs, res);
				else if (clip) wmk::ksw_dp_chain<BP, true, false, false>(sc, jb, seqs, tb, mb.data(), nwv, w, tbs, res);
				else if (hasn) wmk::ksw_dp_chain<BP, false, true, false>(sc, jb, seqs, tb, mb.data(), nwv, w, tbs, res);
				else wmk::ksw_dp_chain<BP, false, false, false>(sc, jb, seqs, tb, mb.data(), nwv, w, tbs, res);
			}
		});
	}
	for (auto &t : th) t.join();
	return nwv;
}
extern "C" {
void emu_chain_events(long *out) { for (int i = 0; i < 16; ++i) out[i] = g_ev[i]; }
void emu_chain_start_skew(int us) { g_start_skew_us = us; }
int emu_chain_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                    int q, int e, int q2, int e2, int w, int zdrop, int end_bonus, int flag, int force_klass,
                    int32_t *ez_out, uint32_t *cigar_out, int cigar_cap, int *klass_out)
{
	wm_ksw_score_t sc;
	sc.match = mat[0]; sc.mismatch = mat[1]; sc.sc_ambi = mat[24];
	if (q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
	sc.q = q; sc.e = e; sc.q2 = q2; sc.e2 = e2;
	std::vector<uint8_t> seqs(qlen + tlen);
	memcpy(seqs.data(), query, qlen); memcpy(seqs.data() + qlen, target, tlen);
	wm_ksw_djob_t jb;
	memset(&jb, 0, sizeof(jb));
	jb.q_off = 0; jb.t_off = qlen; jb.qlen = qlen; jb.tlen = tlen; jb.w = w; jb.zdrop = zdrop; jb.end_bonus = end_bonus; jb.flag = flag;
	if (force_klass < 400 || force_klass >= 430) return -1;
	const int bpi = (force_klass - 400) / 10; force_klass = (force_klass - 400) % 10;
	const int n_col = wm_ksw_ncol(qlen, tlen, w);
	const int has_n = wm_ksw_has_n(query, qlen) | wm_ksw_has_n(target, tlen);
	int ww = w < 0 ? (tlen > qlen ? tlen : qlen) : w;
	const int need = (!(ww >= qlen && ww >= tlen) ? 2 : 0) | (has_n ? 1 : 0);
	if (need & ~(force_klass & 3)) return -1;
	int klass = (force_klass & 3) | ((flag & 0x08) ? 0 : 4);
	*klass_out = klass;
	jb.n_col = n_col; jb.tb_off = 0; jb.klass = klass;
	std::vector<uint8_t> tb((size_t)(qlen + tlen - 1) * n_col + 64, 0xEE);
	wm_ksw_dres_t res;
	memset(&res, 0x77, sizeof(res));
	const int variant = klass & 7;
	switch (bpi) {
	case 0: return -1;                              // (one pair per wavefront: not a geometry of this kernel any more)
	case 1: run_chain<2>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	default: run_chain<4>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	}
	int n = 0;
	if (res.bt_i == KSW_BT_WATCHDOG) return -4;
	if (res.bt_i >= 0) {
		n = wmk::ksw_backtrack_thread(jb, tb.data(), res.bt_i, res.bt_j, cigar_out, cigar_cap);
		if (n < 0) return -3;
		if (!(flag & KSW_F_REV_CIGAR)) std::reverse(cigar_out, cigar_out + n);
	}
	ez_out[0] = res.max; ez_out[1] = res.zdropped; ez_out[2] = res.max_q; ez_out[3] = res.max_t; ez_out[4] = res.mqe;
	ez_out[5] = res.mqe_t; ez_out[6] = res.mte; ez_out[7] = res.mte_q; ez_out[8] = res.score; ez_out[9] = res.reach_end;
	return n;
}
}
