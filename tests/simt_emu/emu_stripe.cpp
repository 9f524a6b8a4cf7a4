// emu_stripe.cpp — TEST INFRASTRUCTURE: the stripe-pipelined ksw kernel (winnowmap_amd/csrc/ksw_stripe_kernel.h) on the host wavefront emulator.
// NWV emulated wavefronts (one host thread each) talk through the LDS rings exactly as on the device: no barrier in the row loop, so the host
// scheduler supplies the interleavings. The kernel's test hooks are switched on here: WM_STRIPE_EVENT counts how often each rare path ran (the tests
// assert that they did), WM_STRIPE_SPIN is a watchdog on every polling loop (a protocol deadlock aborts with the place instead of hanging the suite).
// A library of its own (winnowmap_amd/build.py build_emu_stripe) so that the big emulator driver does not pay for its 64 instantiations.
//   emu_stripe_extd2(..., force): 300 + geometry * 10 + (CLIP * 2 + HASN); geometry 0 = <BP 1, 2 waves>, 1 = <1,3>, 2 = <2,3>, 3 = <2,4>, 4 = <4,4>,
//   5 = <4,8>, 6 = <8,8>, 7 = <1,4>, 8 = <1,16>, 9 = <2,16>; <2,4>, <2,8> (not here: same code as <2,4> with more waves), <4,8>, <8,8> and the
//   opt-in <1,16>, <2,16> are the product's geometries.
#include <atomic>
static std::atomic<long> g_ev[16];
#define WM_STRIPE_EVENT(k) (++g_ev[k])
#include <stdio.h>
#include <stdlib.h>
static thread_local long g_spin = 0;
#define WM_STRIPE_SPIN(where, r, a, wv, extra) do { if (++g_spin > 20000000) { fprintf(stderr, "SPIN where %d r %d a %d wv %d extra %d\n", where, r, a, wv, (int)(extra)); g_spin = 0; static int n = 0; if (++n > 6) abort(); } } while (0)
#include "simt.h"
#include "ksw_kernel.h"
#include "ksw_packed_kernel.h"
#include "ksw_stripe_kernel.h"
#include "ksw_plan.h"
#include <vector>
#include <thread>
#include <algorithm>
template <int BP, int NWV> static void run_stripe(int variant, const wm_ksw_score_t &sc, const wm_ksw_djob_t &jb, const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res)
{
	const bool exact = variant & 4, clip = variant & 2, hasn = variant & 1;
	std::vector<int> lds(wmk::ksw_stripe_lds<BP, NWV>::INTS, 0x5a5a5a5a);
	pthread_barrier_t bar;
	pthread_barrier_init(&bar, 0, NWV);
	simt::block_barrier() = &bar;
	std::vector<std::thread> th;
	for (int w = 0; w < NWV; ++w)
		th.emplace_back([&, w]() {
			simt::wave_slot() = w; simt::exec_mask() = ~0ull;
			if (exact) {
				if (clip && hasn) wmk::ksw_dp_stripe<BP, NWV, true, true, true>(sc, jb, seqs, tb, lds.data(), res);
				else if (clip) wmk::ksw_dp_stripe<BP, NWV, true, false, true>(sc, jb, seqs, tb, lds.data(), res);
				else if (hasn) wmk::ksw_dp_stripe<BP, NWV, false, true, true>(sc, jb, seqs, tb, lds.data(), res);
				else wmk::ksw_dp_stripe<BP, NWV, false, false, true>(sc, jb, seqs, tb, lds.data(), res);
			} else {
				if (clip && hasn) wmk::ksw_dp_stripe<BP, NWV, true, true, false>(sc, jb, seqs, tb, lds.data(), res);
				else if (clip) wmk::ksw_dp_stripe<BP, NWV, true, false, false>(sc, jb, seqs, tb, lds.data(), res);
				else if (hasn) wmk::ksw_dp_stripe<BP, NWV, false, true, false>(sc, jb, seqs, tb, lds.data(), res);
				else wmk::ksw_dp_stripe<BP, NWV, false, false, false>(sc, jb, seqs, tb, lds.data(), res);
			}
		});
	for (auto &t : th) t.join();
	simt::block_barrier() = 0;
	pthread_barrier_destroy(&bar);
}
extern "C" {
void emu_stripe_events(long *out) { for (int i = 0; i < 16; ++i) out[i] = g_ev[i]; }
int emu_stripe_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
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
	if (force_klass < 300 || force_klass >= 400) return -1;
	const int emu_stripe = 1 + (force_klass - 300) / 10; force_klass = (force_klass - 300) % 10;
	static const int max_ncol[10] = { 128, 2 * 128, 2 * 256, 3 * 256, 3 * 512, 7 * 512, 7 * 1024, 3 * 128, 15 * 128, 15 * 256 };
	int n_col = wm_ksw_ncol(qlen, tlen, w);
	if (n_col > max_ncol[emu_stripe - 1]) return -1;
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
	switch (emu_stripe) {
	case 1: run_stripe<1, 2>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	case 2: run_stripe<1, 3>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	case 3: run_stripe<2, 3>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	case 4: run_stripe<2, 4>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	case 5: run_stripe<4, 4>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	case 6: run_stripe<4, 8>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	case 7: run_stripe<8, 8>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	case 8: run_stripe<1, 4>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	case 9: run_stripe<1, 16>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	default: run_stripe<2, 16>(variant, sc, jb, seqs.data(), tb.data(), &res); break;
	}
	int n = 0;
	if (res.bt_i == KSW_BT_WATCHDOG) return -4;              // the kernel's own watchdog gave up (WM_STRIPE_SPIN_BUDGET): what ksw_backtrack_kernel turns into the batch's error
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
