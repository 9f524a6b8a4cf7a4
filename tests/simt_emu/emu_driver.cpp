// tests/simt_emu/emu_driver.cpp — TEST INFRASTRUCTURE ONLY.
// Compiles the product kernel headers against the host wavefront emulator (tests/simt_emu/simt.h) and
// exposes them through a C ABI for ctypes, so kernel logic can be checked against the oracle on a CPU.
#include "simt.h"                    // the emulator (this directory is first on the include path)
#include "ksw_kernel.h"              // winnowmap_amd/csrc
#include "ksw_packed_kernel.h"
#include "ksw_packed_multi_kernel.h"
#include "ksw_dual_kernel.h"
#include "ksw_exts2_kernel.h"
#include "ksw_plan.h"
#include "sketch_kernel.h"
#include "seedchain_kernel.h"
#include "window_kernel.h"
#include <vector>
#include <algorithm>
#include <thread>

template <int BP> static void run_dpp(int variant, const wm_ksw_score_t &sc, const wm_ksw_djob_t &jb, const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res)
{
	simt::exec_mask() = ~0ull;
	const bool exact = variant & 4, clip = variant & 2, hasn = variant & 1;
	if (exact) {
		if (clip && hasn) wmk::ksw_dp_packed<BP, true, true, true>(sc, jb, seqs, tb, res);
		else if (clip) wmk::ksw_dp_packed<BP, true, false, true>(sc, jb, seqs, tb, res);
		else if (hasn) wmk::ksw_dp_packed<BP, false, true, true>(sc, jb, seqs, tb, res);
		else wmk::ksw_dp_packed<BP, false, false, true>(sc, jb, seqs, tb, res);
	} else {
		if (clip && hasn) wmk::ksw_dp_packed<BP, true, true, false>(sc, jb, seqs, tb, res);
		else if (clip) wmk::ksw_dp_packed<BP, true, false, false>(sc, jb, seqs, tb, res);
		else if (hasn) wmk::ksw_dp_packed<BP, false, true, false>(sc, jb, seqs, tb, res);
		else wmk::ksw_dp_packed<BP, false, false, false>(sc, jb, seqs, tb, res);
	}
}

// the packed multi-wave kernel: NWV emulated wavefronts (one host thread each) around a shared barrier
template <int BP, int NWV> static void run_pmulti(int variant, const wm_ksw_score_t &sc, const wm_ksw_djob_t &jb, const uint8_t *seqs, uint8_t *tb, wm_ksw_dres_t *res)
{
	const bool exact = variant & 4, clip = variant & 2, hasn = variant & 1;
	std::vector<int> lds(wmk::ksw_pmulti_lds<BP, NWV>::INTS, 0x5a5a5a5a);
	pthread_barrier_t bar;
	pthread_barrier_init(&bar, 0, NWV);
	simt::block_barrier() = &bar;
	std::vector<std::thread> th;
	for (int w = 0; w < NWV; ++w)
		th.emplace_back([&, w]() {
			simt::wave_slot() = w; simt::exec_mask() = ~0ull;
			const uint8_t *q_ = seqs + jb.q_off, *t_ = seqs + jb.t_off;
			if (exact) {
				if (clip && hasn) wmk::ksw_dp_pmulti<BP, NWV, true, true, true>(sc, jb, q_, t_, tb, lds.data(), res);
				else if (clip) wmk::ksw_dp_pmulti<BP, NWV, true, false, true>(sc, jb, q_, t_, tb, lds.data(), res);
				else if (hasn) wmk::ksw_dp_pmulti<BP, NWV, false, true, true>(sc, jb, q_, t_, tb, lds.data(), res);
				else wmk::ksw_dp_pmulti<BP, NWV, false, false, true>(sc, jb, q_, t_, tb, lds.data(), res);
			} else {
				if (clip && hasn) wmk::ksw_dp_pmulti<BP, NWV, true, true, false>(sc, jb, q_, t_, tb, lds.data(), res);
				else if (clip) wmk::ksw_dp_pmulti<BP, NWV, true, false, false>(sc, jb, q_, t_, tb, lds.data(), res);
				else if (hasn) wmk::ksw_dp_pmulti<BP, NWV, false, true, false>(sc, jb, q_, t_, tb, lds.data(), res);
				else wmk::ksw_dp_pmulti<BP, NWV, false, false, false>(sc, jb, q_, t_, tb, lds.data(), res);
			}
		});
	for (auto &t : th) t.join();
	simt::block_barrier() = 0;
	pthread_barrier_destroy(&bar);
}

static int g_coop_backtrack = 0;
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
	bool emu_blk3_small = false, emu_blk_lds = false, emu_bp2 = false;
	int emu_pmulti = 0;      // 200 + geometry * 10 + (CLIP * 2 + HASN): the packed multi-wave kernel; geometry 0 = <1,2> (256 lanes), 1 = <2,3> (768), 2 = <4,4> (2048), 3 = <8,4> (4096), 4 = <4,8> (4096) and 5 = <8,8> (8192): the product's WM_KSW_PMULTI geometries
	if (force_klass >= 200 && force_klass < 260) { emu_pmulti = 1 + (force_klass - 200) / 10; force_klass = WM_KSW_P4 + (force_klass - 200) % 10; }
	// force_klass: -1 = choose like the product host; 0..23 = that register class (window and CLIP / HASN bits as given, EXACT always follows
	// the job's flag); 100 + (CLIP*2 + HASN) = the 2-pair window (256 lanes: many re-bases and pair boundaries on small inputs; tests only);
	// 24..27 = that wide class; 124 / 125 = the LDS-state block kernel at the BLOCK / BLOCK2 size; 126 = BLOCK3 with a small geometry
	if (force_klass >= 100 && force_klass < 104) { emu_bp2 = true; force_klass = WM_KSW_P4 + (force_klass - 100); }
	if (force_klass == 124) { emu_blk_lds = true; force_klass = WM_KSW_BLOCK; }
	if (force_klass == 125) { emu_blk_lds = true; force_klass = WM_KSW_BLOCK2; }
	if (force_klass == 126) { emu_blk3_small = true; force_klass = WM_KSW_BLOCK3; }
	int n_col, klass = wm_ksw_classify(qlen, tlen, w, wm_ksw_has_n(query, qlen) | wm_ksw_has_n(target, tlen), flag, &n_col);
	if (force_klass >= 0 && !emu_pmulti) {
		if (force_klass < WM_KSW_BLOCK) {
			if (klass >= WM_KSW_BLOCK || (force_klass & ~7) < (klass & ~7)) return -1;      // window too small for this job
			if ((klass & 3) & ~(force_klass & 3)) return -1;                               // the job needs CLIP / HASN and the forced variant lacks it
			klass = (force_klass & ~4) | (klass & 4);
		} else {
			if (force_klass <= WM_KSW_BLOCK3 && klass > force_klass) return -1;
			klass = force_klass;
		}
	}
	if (emu_bp2 && n_col > 128 * 2 - 16) return -1;
	if (emu_pmulti) {
		static const int lanes[6] = { 256, 768, 2048, 4096, 4096, 8192 };
		n_col = wm_ksw_ncol(qlen, tlen, w);
		if (n_col > lanes[emu_pmulti - 1] - 16) return -1;
		const int has_n = wm_ksw_has_n(query, qlen) | wm_ksw_has_n(target, tlen);
		int ww = w < 0 ? (tlen > qlen ? tlen : qlen) : w;
		const int need = (!(ww >= qlen && ww >= tlen) ? 2 : 0) | (has_n ? 1 : 0);
		if (need & ~(force_klass & 3)) return -1;
		klass = (force_klass & 3) | ((flag & 0x08) ? 0 : 4);
	}
	*klass_out = klass;
	if (emu_blk3_small && n_col + 16 > 128 * WM_KSW_BLK_MAXC) return -1;
	jb.n_col = n_col; jb.tb_off = 0; jb.klass = klass;
	std::vector<uint8_t> tb((size_t)(qlen + tlen - 1) * n_col + 64, 0xEE);
	wm_ksw_dres_t res;
	memset(&res, 0, sizeof(res));
	if (emu_pmulti) {
		const int variant = (klass & 4) | ((klass & 2) || (klass & 1) ? 2 : 0) | (klass & 1);      // (jobs with an N run on the CLIP instantiation, as in the product)
		if (emu_pmulti == 1) run_pmulti<1, 2>(variant, sc, jb, seqs.data(), tb.data(), &res);
		else if (emu_pmulti == 2) run_pmulti<2, 3>(variant, sc, jb, seqs.data(), tb.data(), &res);
		else if (emu_pmulti == 3) run_pmulti<4, 4>(variant, sc, jb, seqs.data(), tb.data(), &res);
		else if (emu_pmulti == 4) run_pmulti<8, 4>(variant, sc, jb, seqs.data(), tb.data(), &res);
		else if (emu_pmulti == 5) run_pmulti<4, 8>(variant, sc, jb, seqs.data(), tb.data(), &res);
		else run_pmulti<8, 8>(variant, sc, jb, seqs.data(), tb.data(), &res);
	} else if ((klass == WM_KSW_BLOCK || klass == WM_KSW_BLOCK2) && !emu_blk_lds) {      // the product's kernels for these classes: ksw_dp_pmulti<4,8> / <8,8>, CLIP instantiation
		const int variant = ((flag & 0x08) ? 0 : 4) | 2 | ((wm_ksw_has_n(query, qlen) | wm_ksw_has_n(target, tlen)) ? 1 : 0);
		if (klass == WM_KSW_BLOCK) run_pmulti<4, 8>(variant, sc, jb, seqs.data(), tb.data(), &res);
		else run_pmulti<8, 8>(variant, sc, jb, seqs.data(), tb.data(), &res);
	} else if (klass == WM_KSW_BLOCK || klass == WM_KSW_BLOCK2 || klass == WM_KSW_BLOCK3) {
		constexpr int NWV = WM_KSW_BLK_NWV;
		const int WN = klass == WM_KSW_BLOCK ? WM_KSW_BLK_WN : klass == WM_KSW_BLOCK2 ? WM_KSW_BLK2_WN : (int)wm_ksw_blk3_wn(tlen);
		std::vector<int> W0(WN, 0x5a5a5a5a), W1(WN, 0x5a5a5a5a), Hm(WN, 0x5a5a5a5a), pub(WM_KSW_BLK_PUB);   // the state starts as garbage
		const int nw = klass == WM_KSW_BLOCK3 && emu_blk3_small ? 2 : NWV;
		pthread_barrier_t bar;
		pthread_barrier_init(&bar, 0, nw);
		simt::block_barrier() = &bar;
		std::vector<std::thread> th;
		for (int w = 0; w < nw; ++w)
			th.emplace_back([&, w]() {
				simt::wave_slot() = w; simt::exec_mask() = ~0ull;
				const uint8_t *q_ = seqs.data() + jb.q_off, *t_ = seqs.data() + jb.t_off;
				if (klass == WM_KSW_BLOCK) wmk::ksw_dp_block<NWV, WM_KSW_BLK_K, false>(sc, jb, q_, t_, tb.data(), W0.data(), W1.data(), Hm.data(), WN, pub.data(), &res);
				else if (klass == WM_KSW_BLOCK2) wmk::ksw_dp_block<NWV, WM_KSW_BLK2_K, false>(sc, jb, q_, t_, tb.data(), W0.data(), W1.data(), Hm.data(), WN, pub.data(), &res);
				else if (emu_blk3_small) wmk::ksw_dp_block<2, 1, true>(sc, jb, q_, t_, tb.data(), W0.data(), W1.data(), Hm.data(), WN, pub.data(), &res);
				else wmk::ksw_dp_block<NWV, WM_KSW_BLK2_K, true>(sc, jb, q_, t_, tb.data(), W0.data(), W1.data(), Hm.data(), WN, pub.data(), &res);
			});
		for (auto &t : th) t.join();
		simt::block_barrier() = 0;
		pthread_barrier_destroy(&bar);
	} else if (klass >= WM_KSW_GENERIC) {
		const int T = (tlen + 15) / 16 * 16;
		std::vector<signed char> mem((size_t)7 * T + 64);
		std::vector<int> Hm(T + 16);
		simt::exec_mask() = ~0ull;
		wmk::ksw_dp_generic<true>(sc, jb, seqs.data(), tb.data(), mem.data(), Hm.data(), &res);
	} else if (emu_bp2) run_dpp<2>(klass & 7, sc, jb, seqs.data(), tb.data(), &res);
	else switch (klass & ~7) {
	case WM_KSW_P4: run_dpp<4>(klass & 7, sc, jb, seqs.data(), tb.data(), &res); break;
	case WM_KSW_P8: run_dpp<8>(klass & 7, sc, jb, seqs.data(), tb.data(), &res); break;
	default: run_dpp<16>(klass & 7, sc, jb, seqs.data(), tb.data(), &res); break;
	}
	int n = 0;
	if (res.bt_i >= 0) {
		if (g_coop_backtrack) {
			std::vector<uint8_t> tile((size_t)KSW_BT_ROWS * 64, 0xAB);
			simt::exec_mask() = ~0ull;
			n = wmk::ksw_backtrack_wave(jb, tb.data(), res.bt_i, res.bt_j, cigar_out, cigar_cap, tile.data());
		} else n = wmk::ksw_backtrack_thread(jb, tb.data(), res.bt_i, res.bt_j, cigar_out, cigar_cap);
		if (n < 0) return -3;
		if (!(flag & KSW_F_REV_CIGAR)) std::reverse(cigar_out, cigar_out + n);
	}
	ez_out[0] = res.max; ez_out[1] = res.zdropped; ez_out[2] = res.max_q; ez_out[3] = res.max_t; ez_out[4] = res.mqe;
	ez_out[5] = res.mqe_t; ez_out[6] = res.mte; ez_out[7] = res.mte_q; ez_out[8] = res.score; ez_out[9] = res.reach_end;
	return n;
}
// two alignments on ONE emulated wavefront (ksw_dual_kernel.h): nc = chunks of 64 lanes per alignment (4 | 8 | 16); has_b = 0: the second alignment is absent.
// -1: a job does not qualify (band clips, an N, exact maximum wanted, hull wider than the window)
int emu_ksw_dual(int nc, int has_b, const int32_t *qlen, const int32_t *tlen, const uint8_t *const *query, const uint8_t *const *target, const int32_t *w, const int32_t *flag,
                 const int8_t *mat, int q, int e, int q2, int e2, int32_t *ez_out /* 2 x 10 */, uint32_t *cigar_out /* 2 x cigar_cap */, int cigar_cap, int32_t *n_cigar)
{
	wm_ksw_score_t sc;
	sc.match = mat[0]; sc.mismatch = mat[1]; sc.sc_ambi = mat[24];
	if (q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
	sc.q = q; sc.e = e; sc.q2 = q2; sc.e2 = e2;
	const int n = has_b ? 2 : 1;
	wm_ksw_djob_t jb[2];
	memset(jb, 0, sizeof(jb));
	std::vector<uint8_t> seqs;
	size_t tb_bytes = 0;
	for (int x = 0; x < n; ++x) {
		int n_col;
		const int has_n = wm_ksw_has_n(query[x], qlen[x]) | wm_ksw_has_n(target[x], tlen[x]);
		const int klass = wm_ksw_classify(qlen[x], tlen[x], w[x], has_n, flag[x], &n_col);
		if (klass < 0 || klass >= WM_KSW_BLOCK || (klass & 7) != 0 || n_col > 64 * nc - 16) return -1;      // (unclipped, no N, approximate maximum only)
		jb[x].q_off = (uint32_t)seqs.size(); seqs.insert(seqs.end(), query[x], query[x] + qlen[x]);
		jb[x].t_off = (uint32_t)seqs.size(); seqs.insert(seqs.end(), target[x], target[x] + tlen[x]);
		jb[x].qlen = qlen[x]; jb[x].tlen = tlen[x]; jb[x].w = w[x]; jb[x].zdrop = -1; jb[x].end_bonus = 0; jb[x].flag = flag[x]; jb[x].n_col = n_col; jb[x].klass = klass;
		jb[x].tb_off = tb_bytes; tb_bytes += (size_t)(qlen[x] + tlen[x] - 1) * n_col + 64;
	}
	if (!has_b) jb[1] = jb[0];
	std::vector<uint8_t> tb(tb_bytes + 64, 0xEE);
	wm_ksw_dres_t res[2];
	memset(res, 0, sizeof(res));
	simt::exec_mask() = ~0ull;
	if (nc == 4) wmk::ksw_dp_dual<4>(sc, jb[0], jb[1], has_b != 0, seqs.data(), tb.data(), &res[0], &res[1]);
	else if (nc == 8) wmk::ksw_dp_dual<8>(sc, jb[0], jb[1], has_b != 0, seqs.data(), tb.data(), &res[0], &res[1]);
	else wmk::ksw_dp_dual<16>(sc, jb[0], jb[1], has_b != 0, seqs.data(), tb.data(), &res[0], &res[1]);
	for (int x = 0; x < n; ++x) {
		int nn = 0;
		uint32_t *cg = cigar_out + (size_t)x * cigar_cap;
		if (res[x].bt_i >= 0) {
			nn = wmk::ksw_backtrack_thread(jb[x], tb.data(), res[x].bt_i, res[x].bt_j, cg, cigar_cap);
			if (nn < 0) return -3;
			if (!(flag[x] & KSW_F_REV_CIGAR)) std::reverse(cg, cg + nn);
		}
		n_cigar[x] = nn;
		int32_t *ez = ez_out + 10 * x;
		ez[0] = res[x].max; ez[1] = res[x].zdropped; ez[2] = res[x].max_q; ez[3] = res[x].max_t; ez[4] = res[x].mqe;
		ez[5] = res[x].mqe_t; ez[6] = res[x].mte; ez[7] = res[x].mte_q; ez[8] = res[x].score; ez[9] = res[x].reach_end;
	}
	return 0;
}
void emu_set_coop_backtrack(int on) { g_coop_backtrack = on; }       // 1: ksw_backtrack_wave instead of ksw_backtrack_thread


// ksw_exts2_sse through the emulated splice kernel + its backtrack; junc may be null
int emu_ksw_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int q, int e, int q2, int noncan, int zdrop,
                  int junc_bonus, int flag, const uint8_t *junc, int32_t *ez_out, uint32_t *cigar_out, int cigar_cap)
{
	wm_ksw_score_t sc;
	memset(&sc, 0, sizeof(sc));
	sc.match = mat[0]; sc.mismatch = mat[1]; sc.sc_ambi = mat[24]; sc.q = (int8_t)q; sc.e = (int8_t)e; sc.q2 = (int8_t)q2; sc.e2 = 0;
	std::vector<uint8_t> seqs((size_t)qlen + tlen + 64, 0), jn;
	memcpy(seqs.data(), query, qlen); memcpy(seqs.data() + qlen, target, tlen);
	if (junc) { jn.assign((size_t)qlen + tlen + 64, 0); memcpy(jn.data() + qlen, junc, tlen); }
	wm_ksw_djob_t jb;
	memset(&jb, 0, sizeof(jb));
	jb.q_off = 0; jb.t_off = qlen; jb.qlen = qlen; jb.tlen = tlen; jb.w = -1; jb.zdrop = zdrop; jb.end_bonus = 0; jb.flag = flag;
	jb.n_col = (((qlen < tlen ? qlen : tlen) + 15) / 16 + 1) * 16; jb.tb_off = 0;
	std::vector<uint8_t> tb((size_t)(qlen + tlen - 1) * jb.n_col + 64, 0xEE);
	const int T = (tlen + 15) / 16 * 16;
	std::vector<signed char> mem((size_t)8 * T + 64, 0x5a);
	std::vector<int> Hm(T + 16, 0x5a5a5a5a);
	wm_ksw_dres_t res;
	memset(&res, 0, sizeof(res));
	simt::exec_mask() = ~0ull;
	wmk::ksw_dp_exts2<true>(sc, noncan, junc_bonus, jb, seqs.data(), junc ? jn.data() : 0, tb.data(), mem.data(), Hm.data(), &res);
	int n = 0;
	if (res.bt_i >= 0) {
		n = wmk::ksw_exts2_backtrack_thread(sc, jb, tb.data(), res.bt_i, res.bt_j, cigar_out, cigar_cap);
		if (n < 0) return -3;
		if (!(flag & KSW_F_REV_CIGAR)) std::reverse(cigar_out, cigar_out + n);
	}
	ez_out[0] = res.max; ez_out[1] = res.zdropped; ez_out[2] = res.max_q; ez_out[3] = res.max_t; ez_out[4] = res.mqe;
	ez_out[5] = res.mqe_t; ez_out[6] = res.mte; ez_out[7] = res.mte_q; ez_out[8] = res.score; ez_out[9] = res.reach_end;
	return n;
}

// emu_set_packed(1): the sketch entry points below hand the sequences over as the mapper's resident reads do — 2 bits per base + an ambiguity bitmap
// (reads2bit.h: wm_pack_codes on the host, rd2_* in the kernels), the job offsets flagged WM_RD_PACKED_BIT — instead of bytes
static int g_packed = 0, g_hpc = 0;
void emu_set_packed(int on) { g_packed = on; }
void emu_set_hpc(int on) { g_hpc = on; }            // emu_sketch_coop with homopolymer compression (wm_sketch_params_t::hpc)
namespace {
struct PackedSeqs {
	std::vector<uint64_t> pk, nm;
	PackedSeqs(int n, const uint8_t *seqs, const uint64_t *offs, const int32_t *lens)
	{
		size_t tot = 0;
		for (int i = 0; i < n; ++i) if (lens[i] > 0) tot = std::max<size_t>(tot, (size_t)offs[i] + (size_t)lens[i]);
		pk.assign(wm_pk_words(tot), 0x5555555555555555ULL); nm.assign(wm_nm_words(tot), ~0ULL);      // (garbage first: the packer writes every word)
		if (g_packed) wm_pack_codes(seqs, tot, pk.data(), nm.data());
	}
	uint64_t off(uint64_t o) const { return g_packed ? (o | WM_RD_PACKED_BIT) : o; }
};
}

// mm_sketch through the emulated sketch kernel: n sequences (codes) packed in seqs; returns per-sequence counts and minimizers
int emu_sketch(int n, const uint8_t *seqs, const uint64_t *offs, const int32_t *lens, int w, int k, uint32_t table_bits, uint32_t salt0, uint32_t salt1,
               const uint8_t *bloom_bits, uint64_t *ox, uint64_t *oy, const uint64_t *out_offs, const int32_t *caps, int32_t *counts)
{
	std::vector<wm_sketch_job_t> jobs(n);
	uint64_t tot = 0;
	const PackedSeqs ps(n, seqs, offs, lens);
	for (int i = 0; i < n; ++i) { jobs[i].seq_off = ps.off(offs[i]); jobs[i].len = lens[i]; jobs[i].out_off = out_offs[i]; jobs[i].cap = caps[i]; jobs[i].scratch_off = 0; tot = std::max<uint64_t>(tot, out_offs[i] + caps[i]); }
	std::vector<wm128_t> out(tot + 1);
	wm_sketch_params_t P = { w, k, table_bits, salt0, salt1 };
	std::vector<double> ro((size_t)w * 64);
	std::vector<uint32_t> ry((size_t)w * 64);
	for (int wv = 0; wv * 64 < n; ++wv) {
		simt::exec_mask() = ~0ull;
		wmk::sketch_wave(P, jobs.data(), n, wv, g_packed ? 0 : seqs, ps.pk.data(), ps.nm.data(), bloom_bits, ro.data(), ry.data(), out.data(), counts);
	}
	for (uint64_t i = 0; i < tot; ++i) ox[i] = out[i].x, oy[i] = out[i].y;
	return 0;
}

// the same through the one-wavefront-per-sequence kernel (sketch_coop, odd k)
int emu_sketch_coop(int n, const uint8_t *seqs, const uint64_t *offs, const int32_t *lens, int w, int k, uint32_t table_bits, uint32_t salt0, uint32_t salt1,
                    const uint8_t *bloom_bits, uint64_t *ox, uint64_t *oy, const uint64_t *out_offs, const int32_t *caps, int32_t *counts)
{
	uint64_t tot = 0;
	wm_sketch_params_t P = { w, k, table_bits, salt0, salt1 };
	for (int i = 0; i < n; ++i) tot = std::max<uint64_t>(tot, out_offs[i] + caps[i]);
	std::vector<wm128_t> out(tot + 1);
	const PackedSeqs ps(n, seqs, offs, lens);
	for (int i = 0; i < n; ++i) {
		wm_sketch_job_t jb;
		jb.seq_off = ps.off(offs[i]); jb.len = lens[i]; jb.out_off = out_offs[i]; jb.cap = caps[i]; jb.scratch_off = 0;
		const size_t L = (size_t)(lens[i] > 0 ? lens[i] : 0) + 1;
		std::vector<double> so(L); std::vector<uint64_t> sx(L); std::vector<uint32_t> sy(L), sl(L), he(L, 0xdeadbeefu);
		std::vector<uint8_t> hc(L, 9);
		simt::exec_mask() = ~0ull;
		P.hpc = g_hpc;
		wmk::sketch_coop(P, jb, g_packed ? 0 : seqs, ps.pk.data(), ps.nm.data(), bloom_bits, so.data(), sx.data(), sy.data(), sl.data(), out.data(), counts + i, hc.data(), he.data());
	}
	for (uint64_t i = 0; i < tot; ++i) ox[i] = out[i].x, oy[i] = out[i].y;
	return 0;
}

// the same cut into chunks of `chunk` positions, one emulated wavefront per chunk (the sketch_long_* kernels of wm_gpu.hip: phase 1 per chunk, the first
// sync position of every chunk, phase 2 from sync to sync, the chunks' minimizers concatenated). poison != 0: every chunk's phase 1 runs on a scratch of
// its own that holds garbage outside the chunk, so that nothing depends on what a neighbour computed earlier than the protocol says.
int emu_sketch_chunked(int n, const uint8_t *seqs, const uint64_t *offs, const int32_t *lens, int w, int k, uint32_t table_bits, uint32_t salt0, uint32_t salt1,
                       const uint8_t *bloom_bits, uint64_t *ox, uint64_t *oy, const uint64_t *out_offs, const int32_t *caps, int32_t *counts, int chunk, int32_t *n_absorbed)
{
	uint64_t tot = 0;
	wm_sketch_params_t P = { w, k, table_bits, salt0, salt1 };
	for (int i = 0; i < n; ++i) tot = std::max<uint64_t>(tot, out_offs[i] + caps[i]);
	std::vector<wm128_t> out(tot + 1);
	int absorbed = 0;
	const PackedSeqs ps(n, seqs, offs, lens);
	for (int i = 0; i < n; ++i) {
		const int L = lens[i] > 0 ? lens[i] : 0;
		std::vector<double> so((size_t)L + 1, -7.0); std::vector<uint64_t> sx((size_t)L + 1, 0xdeadbeefULL); std::vector<uint32_t> sy((size_t)L + 1, 0xabcdu), sl((size_t)L + 1, 0u);
		const int n_ch = L > 0 ? (L + chunk - 1) / chunk : 1;
		simt::exec_mask() = ~0ull;
		for (int c = n_ch - 1; c >= 0; --c) {                 // (any order: phase 1 of a chunk depends on the sequence alone)
			const int b = c * chunk, e = std::min(L, b + chunk);
			wmk::sketch_p1_range(P, (long long)ps.off(offs[i]), L, g_packed ? 0 : seqs, ps.pk.data(), ps.nm.data(), bloom_bits, so.data(), sx.data(), sy.data(), sl.data(), b, e);
		}
		std::vector<int> sync(n_ch, -1);
		for (int c = 1; c < n_ch; ++c) { sync[c] = wmk::sketch_find_sync(w, so.data(), c * chunk, std::min(L, (c + 1) * chunk)); absorbed += sync[c] < 0; }
		int total = 0;
		for (int c = 0; c < n_ch; ++c) {
			if (c > 0 && sync[c] < 0) continue;
			int t_stop = -1;
			for (int d = c + 1; d < n_ch && t_stop < 0; ++d) t_stop = sync[d];
			std::vector<wm128_t> tmp((size_t)L + 2);                  // (a wavefront covers every chunk it absorbs)
			const int cnt = wmk::sketch_p2_range(P, L, so.data(), sx.data(), sy.data(), sl.data(), c == 0 ? 0 : sync[c], c != 0, t_stop, tmp.data(), (int)tmp.size());
			if (cnt > (int)tmp.size()) return -1;
			for (int j = 0; j < cnt; ++j) { if (total < caps[i]) out[out_offs[i] + total] = tmp[j]; ++total; }
		}
		counts[i] = total;
	}
	if (n_absorbed) *n_absorbed = absorbed;
	for (uint64_t i = 0; i < tot; ++i) ox[i] = out[i].x, oy[i] = out[i].y;
	return 0;
}

// seed lookup on a flat index given as arrays
int emu_seed(const uint64_t *hkey, const uint64_t *hval, const uint64_t *P, int hbits, const uint64_t *mx, const uint64_t *my, int n_mini, int qlen,
             int max_occ, int flag, uint64_t *ax, uint64_t *ay, int cap, int32_t *res_out)
{
	wm_index_view_t ix = { hkey, hval, P, hbits, 0 };
	std::vector<wm128_t> mini(n_mini + 1), anc(cap + 1);
	for (int i = 0; i < n_mini; ++i) mini[i].x = mx[i], mini[i].y = my[i];
	wm_seed_job_t jb = { 0, 0, n_mini, qlen, max_occ, cap, flag, 0 };
	std::vector<int> occ(n_mini + 1);
	wm_seed_res_t res = { 0, 0 };
	simt::exec_mask() = ~0ull;
	wmk::seed_wave(ix, jb, mini.data(), anc.data(), occ.data(), &res);
	for (int i = 0; i < res.n_anchors && i < cap; ++i) ax[i] = anc[i].x, ay[i] = anc[i].y;
	res_out[0] = res.n_anchors; res_out[1] = res.rep_len;
	return 0;
}

// chain DP fill: returns f, p, v
int emu_chain_fill(int64_t n, const uint64_t *ax, const uint64_t *ay, int max_dist_x, int min_dist_x, int max_dist_y, int bw, int max_skip, int max_iter,
                   float avg_qspan, float gap_scale, int32_t *f, int32_t *p, int32_t *v)
{
	std::vector<wm128_t> a(n + 1);
	for (int64_t i = 0; i < n; ++i) a[i].x = ax[i], a[i].y = ay[i];
	wm_chain_job_t jb = { 0, (int)n, max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter, avg_qspan, gap_scale, 0 };
	// test hook: bits 16.. of max_skip choose the LDS window (default 4096) so that wrap-around can be exercised on small inputs
	int W = 4096;
	if (max_skip >> 16) { W = max_skip >> 16; jb.max_skip &= 0xffff; }
	if (jb.max_skip & 0x8000) { jb.is_cdna = 1; jb.max_skip &= 0x7fff; }        // test hook: bit 15 of max_skip = splice mode (src/chain.c:69-74)
	const int kt_first_hook = (jb.max_skip >> 8) & 15;                          // test hook: bits 8..11 of max_skip = tiles per wavefront in the FIRST step of chain_block_wide (0: as many as in the others)
	jb.max_skip &= 0xff;
	std::vector<int> gt(n + 1), sf(W), sp(W), stt(W);
	std::vector<uint64_t> sx(W), sy(W);
	simt::exec_mask() = ~0ull;
	if (max_iter >> 24) {              // test hook: bits 24.. of max_iter = number of cooperating waves (chain_block)
		const int NWV = max_iter >> 24, KT = (max_iter >> 20) & 15;      // bits 20..23: tiles per wavefront and step (chain_block_wide); 0 = chain_block
		jb.max_iter &= 0xfffff;
		std::vector<int> pub(NWV * (KT ? KT : 1) * 69 + 16);
		pthread_barrier_t bar;
		pthread_barrier_init(&bar, 0, NWV);
		simt::block_barrier() = &bar;
		std::vector<std::thread> th;
		for (int w = 0; w < NWV; ++w)
			th.emplace_back([&, w]() {
				simt::wave_slot() = w; simt::exec_mask() = ~0ull;
				if (KT == 0) wmk::chain_block(jb, a.data(), NWV, W, sx.data(), sy.data(), sf.data(), sp.data(), stt.data(), pub.data(), f, p, gt.data());
				else if (KT == 1) wmk::chain_block_wide<1>(jb, a.data(), NWV, kt_first_hook ? kt_first_hook : 1, W, sx.data(), sy.data(), sf.data(), sp.data(), stt.data(), pub.data(), f, p, gt.data());
				else if (KT == 2) wmk::chain_block_wide<2>(jb, a.data(), NWV, kt_first_hook ? kt_first_hook : 2, W, sx.data(), sy.data(), sf.data(), sp.data(), stt.data(), pub.data(), f, p, gt.data());
				else if (KT == 3) wmk::chain_block_wide<3>(jb, a.data(), NWV, kt_first_hook ? kt_first_hook : 3, W, sx.data(), sy.data(), sf.data(), sp.data(), stt.data(), pub.data(), f, p, gt.data());
				else if (KT == 10) wmk::chain_block_wide<10>(jb, a.data(), NWV, kt_first_hook ? kt_first_hook : 10, W, sx.data(), sy.data(), sf.data(), sp.data(), stt.data(), pub.data(), f, p, gt.data());
				else wmk::chain_block_wide<5>(jb, a.data(), NWV, kt_first_hook ? kt_first_hook : 5, W, sx.data(), sy.data(), sf.data(), sp.data(), stt.data(), pub.data(), f, p, gt.data());
			});
		for (auto &t : th) t.join();
		simt::block_barrier() = 0;
		pthread_barrier_destroy(&bar);
	} else
		wmk::chain_wave(jb, a.data(), W, sx.data(), sy.data(), sf.data(), sp.data(), stt.data(), f, p, gt.data());
	for (int64_t i = 0; i < n; ++i) v[i] = p[i] >= 0 && v[p[i]] > f[i] ? v[p[i]] : f[i];   // the peak score is derived by the caller (as wm_chain_batch does)
	return 0;
}

// ---- the fused window path (window_kernel.h) ----
// collect_seed_hits through win_seed_wave: n_pre handed-in anchors first, then the seeded ones (unsorted); res_out = n_a, rep_len, err
int emu_win_seed(const uint64_t *hkey, const uint64_t *hval, const uint64_t *P, int hbits, const uint64_t *mx, const uint64_t *my, int n_mini, int qlen,
                 int max_occ, int flag, int n_pre, const uint64_t *px, const uint64_t *py, uint64_t *ax, uint64_t *ay, int cap, int32_t *res_out)
{
	wm_index_view_t ix = { hkey, hval, P, hbits, 0 };
	std::vector<wm128_t> mini(n_mini + 1), pre(n_pre + 1), pool(cap + 64);
	for (int i = 0; i < n_mini; ++i) mini[i].x = mx[i], mini[i].y = my[i];
	for (int i = 0; i < n_pre; ++i) pre[i].x = px[i], pre[i].y = py[i];
	wm_win_job_t jb;
	memset(&jb, 0, sizeof(jb));
	jb.seq_off = 0; jb.len = qlen; jb.n_pre = n_pre; jb.max_occ = max_occ; jb.seed_flag = flag;
	std::vector<int> occ(n_mini + 1), emit(n_mini + 1);
	std::vector<uint32_t> first(n_mini + 1);
	uint64_t used = 7;                          // (the pool is shared by the jobs of a call: this job does not start at 0)
	wm_win_res_t res;
	memset(&res, 0, sizeof(res));
	simt::exec_mask() = ~0ull;
	wmk::win_seed_wave(ix, jb, mini.data(), n_mini, pre.data(), occ.data(), first.data(), emit.data(), pool.data(), &used, (uint64_t)cap, &res);
	for (int i = 0; i < res.n_a; ++i) ax[i] = pool[res.a_off + i].x, ay[i] = pool[res.a_off + i].y;
	res_out[0] = res.n_a; res_out[1] = res.rep_len; res_out[2] = res.err;
	return 0;
}

// radix_sort_128x through win_sort_wave (in place); global != 0: the global-memory instantiation
int emu_win_sort(int n, uint64_t *x, uint64_t *y, int global)
{
	std::vector<wm128_t> a(n + 1);
	for (int i = 0; i < n; ++i) a[i].x = x[i], a[i].y = y[i];
	std::vector<int> ws(wmk::WIN_WS_INTS, 0x5a5a5a5a);
	simt::exec_mask() = ~0ull;
	if (global) wmk::win_sort_wave<true>(a.data(), n, ws.data()); else wmk::win_sort_wave<false>(a.data(), n, ws.data());
	for (int i = 0; i < n; ++i) x[i] = a[i].x, y[i] = a[i].y;
	return 0;
}

// avg_qspan + kernel class of the fill through win_plan_wave
int emu_win_plan(int n, const uint64_t *x, const uint64_t *y, int max_dist_x, float *avg_out, int *klass_out)
{
	std::vector<wm128_t> a(n + 1);
	for (int i = 0; i < n; ++i) a[i].x = x[i], a[i].y = y[i];
	wm_win_job_t jb;
	memset(&jb, 0, sizeof(jb));
	jb.max_dist_x = max_dist_x;
	wm_chain_job_t cj[1];
	int lists[4] = { -1, -1, -1, -1 }, counts[4] = { 0, 0, 0, 0 };
	simt::exec_mask() = ~0ull;
	wmk::win_plan_wave(jb, 0, 0, n, a.data(), cj, lists, counts, 1);
	*avg_out = cj[0].avg_qspan;
	*klass_out = -1;
	for (int k = 0; k < 4; ++k) if (counts[k] == 1 && lists[k] == 0) *klass_out = k;
	return 0;
}

// mm_chain_dp after the fill (src/chain.c:89-165) through win_extract_wave; a is overwritten with the chained anchors; returns n_v
int emu_win_extract(int n, uint64_t *ax, uint64_t *ay, const int32_t *f_in, const int32_t *p_in, int min_cnt, int min_sc, int global, int *n_u_out, uint64_t *u_out)
{
	std::vector<wm128_t> a(n + 1), b(n + 1), wb(n + 1);
	for (int i = 0; i < n; ++i) a[i].x = ax[i], a[i].y = ay[i];
	std::vector<int> f(f_in, f_in + n), p(p_in, p_in + n), v(n + 1, 0x5a5a5a5a), t(n + 1, 0x5a5a5a5a);
	f.push_back(0); p.push_back(0);
	std::vector<uint64_t> zu((size_t)n + 2, 0x5a5a5a5a5a5a5a5aull), up((size_t)n + 16);
	std::vector<wm128_t> vp((size_t)n + 16);
	std::vector<int> ws(wmk::WIN_WS_INTS, 0x5a5a5a5a);
	uint64_t ctr[2] = { 3, 5 };                 // (the pools are shared by the jobs of a call: this job does not start at 0)
	wm_win_res_t res;
	memset(&res, 0, sizeof(res));
	simt::exec_mask() = ~0ull;
	if (global) wmk::win_extract_wave<true>(n, min_cnt, min_sc, a.data(), f.data(), p.data(), v.data(), t.data(), zu.data(), b.data(), wb.data(), ws.data(), &res, up.data(), vp.data(), ctr);
	else wmk::win_extract_wave<false>(n, min_cnt, min_sc, a.data(), f.data(), p.data(), v.data(), t.data(), zu.data(), b.data(), wb.data(), ws.data(), &res, up.data(), vp.data(), ctr);
	*n_u_out = res.n_u;
	if (res.n_u && (res.u_out != 3 || res.v_out != 5 || ctr[0] != 3 + (uint64_t)res.n_u || ctr[1] != 5 + (uint64_t)res.n_v)) return -1;
	for (int i = 0; i < res.n_u; ++i) u_out[i] = up[res.u_out + i];
	for (int i = 0; i < res.n_v; ++i) ax[i] = vp[res.v_out + i].x, ay[i] = vp[res.v_out + i].y;
	return res.n_v;
}

// a small job from its unsorted anchors to its chains by one wavefront in LDS (win_small_wave): n_pre handed-in anchors first; seeded != 0: the job
// has a sequence (its seeded part is sorted, then the union). Returns n_v; chains in u_out, chained anchors in ax / ay
int emu_win_small(int n, int n_pre, int seeded, uint64_t *ax, uint64_t *ay, int max_dist_x, int min_dist_x, int max_dist_y, int bw, int max_skip, int max_iter,
                  int min_cnt, int min_sc, float gap_scale, int *n_u_out, uint64_t *u_out)
{
	if (n > wmk::WIN_SMALL) return -2;
	std::vector<wm128_t> a(n + 1), vp((size_t)n + 16);
	for (int i = 0; i < n; ++i) a[i].x = ax[i], a[i].y = ay[i];
	std::vector<uint64_t> up((size_t)n + 16);
	std::vector<unsigned char> lds(wmk::WIN_SMALL_LDS, 0x5a);
	wm_win_job_t jb;
	memset(&jb, 0, sizeof(jb));
	jb.seq_off = seeded ? 0 : -1; jb.n_pre = n_pre; jb.max_dist_x = max_dist_x; jb.min_dist_x = min_dist_x; jb.max_dist_y = max_dist_y; jb.bw = bw;
	jb.max_skip = max_skip; jb.max_iter = max_iter; jb.min_cnt = min_cnt; jb.min_sc = min_sc; jb.gap_scale = gap_scale;
	uint64_t ctr[2] = { 0, 0 };
	wm_win_res_t res;
	memset(&res, 0, sizeof(res));
	simt::exec_mask() = ~0ull;
	wmk::win_small_wave(jb, n, a.data(), lds.data(), &res, up.data(), vp.data(), ctr);
	*n_u_out = res.n_u;
	for (int i = 0; i < res.n_u; ++i) u_out[i] = up[res.u_out + i];
	for (int i = 0; i < res.n_v; ++i) ax[i] = vp[res.v_out + i].x, ay[i] = vp[res.v_out + i].y;
	return res.n_v;
}

// the workgroup-level stable LSD sort of large anchor sets (win_bigsort_block) on nwv emulated wavefronts; returns the tie flag, x / y sorted
int emu_win_bigsort(int n, uint64_t *x, uint64_t *y, int nwv)
{
	std::vector<wm128_t> a(n + 1), b0(n + 1), b1(n + 1);
	for (int i = 0; i < n; ++i) a[i].x = x[i], a[i].y = y[i];
	std::vector<int> lds(WIN_BIG_INTS(nwv), 0x5a5a5a5a);
	std::vector<int> cur(nwv, -7), tie(nwv, -7);
	pthread_barrier_t bar;
	pthread_barrier_init(&bar, 0, nwv);
	simt::block_barrier() = &bar;
	std::vector<std::thread> th;
	for (int w = 0; w < nwv; ++w)
		th.emplace_back([&, w]() {
			simt::wave_slot() = w; simt::exec_mask() = ~0ull;
			cur[w] = wmk::win_bigsort_block(nwv, a.data(), b0.data(), b1.data(), n, lds.data(), &tie[w]);
		});
	for (auto &t : th) t.join();
	simt::block_barrier() = 0;
	pthread_barrier_destroy(&bar);
	for (int w = 1; w < nwv; ++w) if (cur[w] != cur[0] || tie[w] != tie[0]) return -100;      // the result must be uniform over the workgroup
	const wm128_t *r = cur[0] < 0 ? a.data() : cur[0] == 0 ? b0.data() : b1.data();
	for (int i = 0; i < n; ++i) x[i] = r[i].x, y[i] = r[i].y;
	return tie[0];
}

} // extern "C"
