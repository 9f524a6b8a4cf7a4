// tests/simt_emu/emu_driver.cpp — TEST INFRASTRUCTURE ONLY.
// Compiles the product kernel headers against the host wavefront emulator (tests/simt_emu/simt.h) and
// exposes them through a C ABI for ctypes, so kernel logic can be checked against the oracle on a CPU.
#include "simt.h"                    // the emulator (this directory is first on the include path)
#include "ksw_kernel.h"              // winnowmap_amd/csrc
#include "ksw_plan.h"
#include <vector>
#include <algorithm>

template <int B> static void run_dp(int clip, int hasn, const wm_ksw_score_t &sc, const wm_ksw_djob_t &jb, const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res)
{
	simt::exec_mask() = ~0ull;
	if (clip && hasn) wmk::ksw_dp_wave<B, true, true>(sc, jb, seqs, tb, res);
	else if (clip) wmk::ksw_dp_wave<B, true, false>(sc, jb, seqs, tb, res);
	else if (hasn) wmk::ksw_dp_wave<B, false, true>(sc, jb, seqs, tb, res);
	else wmk::ksw_dp_wave<B, false, false>(sc, jb, seqs, tb, res);
}

extern "C" {

// force_klass < 0: choose like the product host; otherwise use that class (to exercise CLIP/HASN variants on any input)
int emu_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
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
	int n_col, klass = wm_ksw_classify(qlen, tlen, w, wm_ksw_has_n(query, qlen) | wm_ksw_has_n(target, tlen), &n_col);
	if (force_klass >= 0) {
		if ((force_klass & ~3) < (klass & ~3)) return -1; // window too small for this job
		klass = force_klass;
	}
	*klass_out = klass;
	if (klass >= WM_KSW_GENERIC) return -2;
	jb.n_col = n_col; jb.tb_off = 0; jb.klass = klass;
	std::vector<uint8_t> tb((size_t)(qlen + tlen - 1) * n_col + 64, 0xEE);
	wm_ksw_dres_t res;
	memset(&res, 0, sizeof(res));
	const int clip = klass >> 1 & 1, hasn = klass & 1;
	switch (klass & ~3) {
	case WM_KSW_B4: run_dp<4>(clip, hasn, sc, jb, seqs.data(), tb.data(), &res); break;
	case WM_KSW_B8: run_dp<8>(clip, hasn, sc, jb, seqs.data(), tb.data(), &res); break;
	default: run_dp<16>(clip, hasn, sc, jb, seqs.data(), tb.data(), &res); break;
	}
	int n = 0;
	if (res.bt_i >= 0) {
		n = wmk::ksw_backtrack_thread(jb, tb.data(), res.bt_i, res.bt_j, cigar_out, cigar_cap);
		if (n < 0) return -3;
		if (!(flag & KSW_F_REV_CIGAR)) std::reverse(cigar_out, cigar_out + n);
	}
	ez_out[0] = res.max; ez_out[1] = res.zdropped; ez_out[2] = res.max_q; ez_out[3] = res.max_t; ez_out[4] = res.mqe;
	ez_out[5] = res.mqe_t; ez_out[6] = res.mte; ez_out[7] = res.mte_q; ez_out[8] = res.score; ez_out[9] = res.reach_end;
	return n;
}

} // extern "C"
