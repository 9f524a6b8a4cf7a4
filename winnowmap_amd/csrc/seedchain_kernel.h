// seedchain_kernel.h — seed lookup/expansion and the chaining DP fill on gfx950 (one wavefront per window).
//
//  seed_wave   : for every query minimizer, probe the flat index in HBM (open addressing, one 16-B slot + the
//                position run), drop over-represented minimizers (occ >= mid_occ, accumulating rep_len exactly as
//                collect_matches does), mark tandem seeds, and expand to anchors in minimizer order with a wave
//                prefix sum. The unstable radix_sort_128x that follows in the reference is applied by the caller
//                (its tie permutation is inherently sequential).
//  chain_wave  : mm_chain_dp's score fill. Anchor i is processed sequentially (f[i] depends on earlier f), but its
//                predecessor scan j = i-1 … st runs 64 at a time: every lane scores one j (integer + the reference's
//                fp64 gap cost), the "t[] marks" are scattered and re-read (marks only flow from larger to smaller
//                j, so scatter-then-read inside a tile is exact), and the sequential max_f / n_skip / break
//                automaton (src/chain.c:79-86) is replayed over two 64-bit ballots with a prefix max.
//                f, p, v are returned; chain extraction (:93-165) is O(n) bookkeeping done by the caller.
#pragma once
#ifndef WM_DEV
#error "include simt.h before seedchain_kernel.h"
#endif
#include "wm_internal.h"

namespace wmk {
using namespace simt;

WM_DEV void seed_wave(const wm_index_view_t ix, const wm_seed_job_t jb, const wm128_t *mini_pool, wm128_t *anchor_pool,
                      int *occ_scratch /* n_mini ints */, wm_seed_res_t *res)
{
	const V<int> ln = lane();
	const uint64_t *mini = (const uint64_t*)(mini_pool + jb.mini_off);
	uint64_t *outp = (uint64_t*)(anchor_pool + jb.out_off);
	const uint64_t hmask = ((uint64_t)1 << ix.hbits) - 1;
	const bool strand_filter = (jb.flag & (0x100000 | 0x200000)) != 0;
	int base = 0;                                        // anchors written so far
	for (int m0 = 0; m0 < jb.n_mini; m0 += 64) {
		const V<int> m = ln + m0;
		const vbool have = m < jb.n_mini;
		V<uint64_t> mx = (uint64_t)0, my = (uint64_t)0, first = (uint64_t)0;
		V<int> cnt = 0;
		WM_IF(have)
			mx = gld(mini, m * 2); my = gld(mini, m * 2 + 1);
			const V<uint64_t> key = mx >> 8;
			V<uint64_t> s = (key * (uint64_t)0x9E3779B97F4A7C15ULL) >> (64 - ix.hbits);
			vbool probing = s == s;                      // true
			for (int guard = 0; guard < (1 << 20) && any(probing); ++guard) {
				WM_IF(probing)
					V<uint64_t> hk = gld(ix.hkey, s);
					WM_IF(hk == key)
						V<uint64_t> hv = gld(ix.hval, s);
						cnt = cast<int>(hv & (uint64_t)0xffffffffu); first = hv >> 32;
					WM_END
					probing = (hk != key) && (hk != ~(uint64_t)0);
					s = (s + (uint64_t)1) & hmask;
				WM_END
			}
			cst(occ_scratch, m, cnt);
		WM_END
		// occurrence filter + (optional) strand filter decide how many anchors each minimizer contributes
		V<int> emit = sel(have && cnt < jb.max_occ, cnt, 0);
		WM_IF(strand_filter && emit > 0)
			V<int> kept = 0;
			const V<int> qstrand = cast<int>(my & (uint64_t)1);
			for (int h = 0; h < jb.max_occ && any(emit > h); ++h)
				WM_IF(emit > h)
					const V<int> rstrand = cast<int>(gld(ix.P, first + (uint64_t)h) & (uint64_t)1);
					const vbool fwd = rstrand == qstrand;
					kept = kept + sel((fwd && !(jb.flag & 0x200000)) || (!fwd && !(jb.flag & 0x100000)), 1, 0);
				WM_END
			emit = kept;
		WM_END
		// exclusive prefix sum across the tile
		V<int> incl = emit;
		for (int o = 1; o < 64; o <<= 1) incl = incl + sel(ln >= o, shr_n(incl, o), 0);
		const V<int> excl = incl - emit;
		const int tile_total = readlane(incl, 63);
		WM_IF(emit > 0)
			const V<uint32_t> q_pos = cast<uint32_t>(my), q_span = cast<uint32_t>(mx & (uint64_t)0xff);
			// tandem: the neighbouring minimizer (in the whole list) has the same key (src/map.c:121-122)
			vbool tandem = q_pos != q_pos;               // false
			WM_IF(m > 0) tandem = tandem || ((gld(mini, (m - 1) * 2) >> 8) == (mx >> 8)); WM_END
			WM_IF(m < jb.n_mini - 1) tandem = tandem || ((gld(mini, (m + 1) * 2) >> 8) == (mx >> 8)); WM_END
			V<int> w = base + excl;
			for (int h = 0; h < jb.max_occ && any(cnt > h); ++h)
				WM_IF(cnt > h)
					const V<uint64_t> r = gld(ix.P, first + (uint64_t)h);
					const V<uint64_t> rpos = (r & (uint64_t)0xffffffffu) >> 1;
					const vbool fwd = cast<uint32_t>(r & (uint64_t)1) == (q_pos & 1u);
					vbool keep = rpos == rpos;
					if (strand_filter) keep = (fwd && !(jb.flag & 0x200000)) || (!fwd && !(jb.flag & 0x100000));
					WM_IF(keep)
						V<uint64_t> ax = (r & (uint64_t)0xffffffff00000000ULL) | rpos;
						V<uint64_t> ay = cast<uint64_t>(q_span) << 32;
						WM_IF(fwd) ay = ay | cast<uint64_t>(q_pos >> 1); WM_ELSE
							ax = ax | ((uint64_t)1 << 63);
							ay = ay | cast<uint64_t>(cast<uint32_t>(V<int>(jb.qlen) - cast<int>((q_pos >> 1) + 1u - q_span) - 1));
						WM_END
						ay = sel(tandem, ay | ((uint64_t)1 << 42), ay);
						WM_IF(w < jb.cap) gst(outp, w * 2, ax); gst(outp, w * 2 + 1, ay); WM_END
						w = w + 1;
					WM_END
				WM_END
		WM_END
		base += tile_total;
	}
	// rep_len: sequential over the minimizers (src/map.c:111-116,126); uniform code, one result
	int rep_st = 0, rep_en = 0, rep_len = 0;
	for (int m = 0; m < jb.n_mini; ++m) {
		const int t = cld(occ_scratch, (long long)m);
		if (t >= jb.max_occ) {
			const uint64_t mx = gld(mini, (long long)m * 2), my = gld(mini, (long long)m * 2 + 1);
			const int en = (int)((uint32_t)my >> 1) + 1, st = en - (int)(mx & 0xff);
			if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st; rep_en = en; }
			else rep_en = en;
		}
	}
	rep_len += rep_en - rep_st;
	WM_IF(ln == 0)
		gst(&res->n_anchors, 0LL, base);
		gst(&res->rep_len, 0LL, rep_len);
	WM_END
}

// floor(log2(v)) for v > 0, 0 for v == 0 (ilog2_32, src/chain.c:15-20, and the dd > 0 guard at its call sites): one v_ffbh_u32
WM_DEV V<int> ilog2_pos(V<int> v) { return V<int>(31) - vclz(v | 1); }

// LDS holds a circular window of the W most recent anchors (W a power of two >= 128): their x, y and f, p, t — 28 B per
// anchor — so every dependent access of the sequential part is an LDS round trip. When the anchor set is larger than the
// window, finished f, p are copied to the global result slab gf/gp one 64-anchor tile at a time (never per anchor: an
// outstanding global store would stall the next anchor's s_waitcnt vmcnt), so that predecessors older than the window —
// which the reference may still visit when min_dist_x relaxes max_iter inside dense repeats (src/chain.c:51-55) — are
// served from global memory (L1-bypassing loads; marks for them go to gt). For n <= W nothing ever leaves the window.
// The peak score v[] (src/chain.c:89) is a function of f and p only and is not needed by the fill: the caller derives it.

// advance st over the sorted x: the two scalar loops of src/chain.c:50-55, 64 candidates per LDS round trip
WM_DEV long long chain_advance_st(long long st, long long i, uint64_t ri, uint64_t dist, long long keep_iter /* <0: no iteration clause */,
                                  long long lo, long long wm, const uint64_t *sx, const uint64_t *a)
{
	const V<int> ln = lane();
	for (;;) {
		if (st >= i || (keep_iter >= 0 && i - st <= keep_iter)) return st;
		const V<long long> k = cast<long long>(ln) + st;
		V<uint64_t> xk = ~(uint64_t)0;
		const vbool in = k < i;
		WM_IF(in && k >= lo) xk = gld(sx, k & wm); WM_END
		WM_IF(in && k < lo) xk = gld(a, k * 2LL); WM_END
		vbool go = in && V<uint64_t>(ri) > xk + dist;
		if (keep_iter >= 0) go = go && (V<long long>(i) - k > keep_iter);
		const uint64_t stay = ~ballot(go);                    // first lane that does not advance ends the run (conditions are monotone in k)
		if (stay) return st + __builtin_ctzll(stay);
		st += 64;
	}
}

// copy the finished anchors [from, to) of the window to the global slab (tile flush; all lanes)
WM_DEV void chain_flush_tile(long long from, long long to, long long wm, const int *sf, const int *sp, int *gf, int *gp)
{
	const V<long long> k = cast<long long>(lane()) + from;
	WM_IF(k < to) cst(gf, k, gld(sf, k & wm)); cst(gp, k, gld(sp, k & wm)); WM_END
	mem_sync();
}

// running state of one anchor's predecessor scan (uniform across the wave)
struct chain_scan_t { int max_f, n_skip; long long max_j; bool stop; };

// One tile's share of the sequential automaton of src/chain.c:79-86, without a scalar replay:
//   improvement lanes I : score > every earlier visited score and > the running max  (inclusive max scan, shifted by one)
//   marked lanes      M : t[j] == i and not I
//   n_skip is a counter that saturates at 0 on I and grows on M: n_k = S_k + max(n_0, -min_{m<=k} S_m) with S the
//   prefix sum of (-1 on I, +1 on M) (the Lindley recursion), so a prefix sum and a prefix min give it for all 64 lanes;
//   the scan stops at the first lane with n_k > max_skip.
WM_DEV void chain_tile_resolve(const vbool valid, const V<int> sc, const V<int> tj, int i, long long hi, int max_skip, chain_scan_t &cs)
{
	const V<int> NEG = -0x7fffffff - 1;
	const V<int> key = sel(valid, sc, NEG);
	vbool isI = valid && !valid;
	if (ballot(key > cs.max_f)) {
		const V<int> incl = wave_scan_max(key);
		const V<int> before = vmax(shr1(incl, NEG), V<int>(cs.max_f));
		isI = valid && sc > before;
	}
	const vbool isM = valid && tj == i && !isI;
	const uint64_t I = ballot(isI);
	if (!(I | ballot(isM))) return;
	const V<int> d = sel(isI, V<int>(-1), sel(isM, V<int>(1), V<int>(0)));
	const V<int> S = wave_scan_add(d);
	const V<int> mn = wave_scan_min(S);
	const V<int> nk = S + vmax(V<int>(cs.n_skip), V<int>(0) - mn);
	const uint64_t B = ballot(nk > max_skip);
	const int brk = B ? __builtin_ctzll(B) : 64;
	const uint64_t Ib = brk < 64 ? I & (((uint64_t)1 << brk) - 1) : I;
	if (Ib) {
		const int l = 63 - __builtin_clzll(Ib);
		cs.max_f = readlane(sc, l);
		cs.max_j = hi - l;
	}
	if (brk < 64) cs.stop = true; else cs.n_skip = readlane(nk, 63);
}

// score of predecessor j for anchor (ri, qi, span): src/chain.c:57-78. Every predecessor that is scored lies in [st, i), and st has been advanced
// past everything with x[j] + max_dist_x < x[i] (src/chain.c:50) before any relaxation of it, so 0 <= x[i] - x[j] <= max_dist_x < 2^31 there (x is
// sorted): the LOW words of x decide, and the score is 32-bit arithmetic throughout (the reference's int64 dr and its clamp at :64 never bind).
// xj, yj: low words of a[j].x (reference position) and a[j].y (query position).
WM_DEV void chain_score(const wm_chain_job_t &jb, uint32_t ri, int qi, int span, const V<uint32_t> xj, const V<uint32_t> yj, const V<int> fj, V<int> &sc, vbool &ok)
{
	const V<int> dr = cast<int>(V<uint32_t>(ri) - xj);
	const V<int> dq = V<int>(qi) - cast<int>(yj);
	ok = !(dr == 0 || dq <= 0) && !(dq > jb.max_dist_y || dq > jb.max_dist_x);                    // :60-61
	const V<int> dd = sel(dr > dq, dr - dq, dq - dr);
	ok = ok && !(dd > jb.bw);                                                                      // :63
	V<int> s0 = vmin(vmin(dq, dr), V<int>(span));                                                  // :65-66
	const V<int> lg = ilog2_pos(dd);
	const V<double> lin = cast<double>(dd) * .01 * (double)jb.avg_qspan;
	V<int> gc = cast<int>(lin) + (lg >> 1);                                                        // :76
	if (jb.is_cdna) gc = sel(dr > dq, vmin(cast<int>(lin), lg), gc);                               // :69-74: an intron (reference gap) costs min(linear, log)
	if (jb.gap_scale != 1.0f) { WM_KEEP_BRANCH(); gc = cast<int>(cast<double>(gc) * (double)jb.gap_scale + .499); }      // :77 (with the default scale (int)((double)gc + .499) == gc: gc >= 0)
	s0 = s0 - gc;
	sc = s0 + fj;
}

// the low words of x / y of anchor j: in the LDS window (index already wrapped) or in the job's anchor array
WM_DEV V<uint32_t> chain_lo_lds(const uint64_t *s, const V<int> k) { return gld((const uint32_t*)s, k * 2); }
WM_DEV V<uint32_t> chain_lo_far(const uint64_t *a, const V<long long> j, int which) { return gld((const uint32_t*)a, j * 4LL + (long long)(2 * which)); }

// U tiles (64 predecessors each) are scored together so that their memory latencies overlap; the automaton is then
// resolved tile by tile. Marks of a later tile never target an earlier one (p[j] < j), and marks written for tiles
// behind a break are harmless (they are only compared with this i).
template <int U>
WM_DEV void chain_group(const wm_chain_job_t &jb, long long hi0, long long st, long long lo, long long wm, int i, uint64_t ri, int qi, int span,
                        const uint64_t *a, const uint64_t *sx, const uint64_t *sy, const int *sf, const int *sp, int *st_, int *gf, int *gp, int *gt, chain_scan_t &cs)
{
	const V<int> ln = lane();
	V<int> sc[U], tj[U];
	vbool valid[U];
	V<long long> jj[U];
	bool any_far = false;
#pragma unroll
	for (int u = 0; u < U; ++u) {
		const V<long long> j = V<long long>(hi0 - 64 * u) - cast<long long>(ln);                  // lane 0 = first visited
		jj[u] = j;
		const vbool in = j >= st, res = j >= lo;
		const V<int> jw = cast<int>(j) & (int)wm;
		V<int> pj = -1, fj = 0;
		V<uint32_t> xj = 0u, yj = 0u;
		sc[u] = 0; tj[u] = 0;
		valid[u] = in && !in;                                                                      // false
		if (hi0 - 64 * u - 63 >= lo) {                                                             // whole tile resident (the common case)
			WM_IF(in) xj = chain_lo_lds(sx, jw); yj = chain_lo_lds(sy, jw); fj = gld(sf, jw); pj = gld(sp, jw); WM_END
		} else {
			WM_IF(in && res) xj = chain_lo_lds(sx, jw); yj = chain_lo_lds(sy, jw); fj = gld(sf, jw); pj = gld(sp, jw); WM_END
			WM_IF(in && !res) xj = chain_lo_far(a, j, 0); yj = chain_lo_far(a, j, 1); fj = cld(gf, j); pj = cld(gp, j); WM_END
		}
		WM_IF(in)
			vbool ok = in;
			chain_score(jb, (uint32_t)ri, qi, span, xj, yj, fj, sc[u], ok);
			valid[u] = ok;
		WM_END
		// marks (src/chain.c:86): a scored predecessor marks ITS predecessor, if that one can still be visited
		const V<long long> pjl = cast<long long>(pj);
		WM_IF(valid[u] && pj >= 0 && pjl >= lo) gst(st_, pjl & wm, V<int>(i)); WM_END
		if (st < lo) {
			const vbool far_mark = valid[u] && pj >= 0 && pjl < lo && pjl >= st;
			WM_IF(far_mark) cst(gt, pjl, V<int>(i)); WM_END
			any_far = any_far || any(far_mark);
		}
	}
	if (any_far) mem_sync(); else lds_sync();                                                      // global marks must reach L2 before they are re-read
#pragma unroll
	for (int u = 0; u < U; ++u) {
		if (hi0 - 64 * u - 63 >= lo) {
			WM_IF(valid[u]) tj[u] = gld(st_, jj[u] & wm); WM_END
		} else {
			const vbool res = jj[u] >= lo;
			WM_IF(valid[u] && res) tj[u] = gld(st_, jj[u] & wm); WM_END
			WM_IF(valid[u] && !res) tj[u] = cld(gt, jj[u]); WM_END
		}
	}
#pragma unroll
	for (int u = 0; u < U; ++u) {
		if (cs.stop || hi0 - 64 * u < st) break;
		chain_tile_resolve(valid[u], sc[u], tj[u], i, hi0 - 64 * u, jb.max_skip, cs);
	}
}

WM_DEV void chain_wave(const wm_chain_job_t jb, const wm128_t *anchor_pool, int W, uint64_t *sx, uint64_t *sy, int *sf, int *sp, int *st_,
                       int *gf, int *gp, int *gt)
{
	const V<int> ln = lane();
	const uint64_t *a = (const uint64_t*)(anchor_pool + jb.a_off);
	const int n = jb.n;
	const long long wm = (long long)W - 1;
	const bool wraps = n > W;
	if (wraps)
		for (int i0 = 0; i0 < n; i0 += 64) WM_IF(ln + i0 < n) cst(gt, cast<long long>(ln) + (long long)i0, V<int>(0)); WM_END
	long long st = 0;
	for (int i = 0; i < n; ++i) {
		if ((i & 63) == 0) {                                  // stage the next 64 anchors (evicts anchors i-W .. i-W+63)
			if (wraps && i > 0) chain_flush_tile((long long)i - 64, (long long)i, wm, sf, sp, gf, gp);
			const V<long long> k = cast<long long>(ln) + (long long)i;
			WM_IF(k < (long long)n)
				gst(sx, k & wm, gld(a, k * 2LL)); gst(sy, k & wm, gld(a, k * 2LL + 1LL)); gst(st_, k & wm, V<int>(0));
			WM_END
			lds_sync();
		}
		const long long lo = (long long)(i & ~63) + 64 - W;   // anchors >= lo are resident in LDS
		const uint64_t ri = gld(sx, (long long)i & wm), yi = gld(sy, (long long)i & wm);
		const int qi = (int)(uint32_t)yi, span = (int)(yi >> 32 & 0xff);
		chain_scan_t cs = { span, 0, -1, false };
		st = chain_advance_st(st, i, ri, (uint64_t)jb.max_dist_x, -1, lo, wm, sx, a);                                      // :50
		if (i - st > jb.max_iter) st = chain_advance_st(st, i, ri, (uint64_t)jb.min_dist_x, jb.max_iter, lo, wm, sx, a);    // :51-55
		long long hi0 = (long long)i - 1;
		if (hi0 >= st) {                                      // most scans end inside their first tile: score it alone, then four at a time
			chain_group<1>(jb, hi0, st, lo, wm, i, ri, qi, span, a, sx, sy, sf, sp, st_, gf, gp, gt, cs);
			for (hi0 -= 64; hi0 >= st && !cs.stop; hi0 -= 64 * 4)
				chain_group<4>(jb, hi0, st, lo, wm, i, ri, qi, span, a, sx, sy, sf, sp, st_, gf, gp, gt, cs);
		}
		WM_IF(ln == 0)
			const V<long long> ii = (long long)i;
			gst(sf, ii & wm, V<int>(cs.max_f)); gst(sp, ii & wm, V<int>((int)cs.max_j));
		WM_END
		lds_sync();
	}
	if (wraps) chain_flush_tile((long long)((n - 1) & ~63), (long long)n, wm, sf, sp, gf, gp);
	else                                                       // everything stayed in the window: one coalesced copy-out
		for (int i0 = 0; i0 < n; i0 += 64) {
			const V<long long> k = cast<long long>(ln) + (long long)i0;
			WM_IF(k < (long long)n) gst(gf, k, gld(sf, k)); gst(gp, k, gld(sp, k)); WM_END
		}
}


// ------------------------------------------------------------------------------------------------------
// chain_block: the same fill for LARGE anchor sets (satellite arrays: 10^4..10^6 anchors, up to max_iter = 5000
// predecessors per anchor). A workgroup of NWV wavefronts shares one LDS window; every step scores NWV*64 predecessors
// of anchor i (wave w takes tile w), so the serial latency per anchor drops by ~NWV. Cross-wave dependencies are the
// marks (barrier after the scatter), the running maximum (each wave publishes its tile maximum) and the n_skip/break
// automaton, which every wave replays redundantly from the published ballots so that all waves agree without a
// broadcast. pub: NWV * (2 + 64 + 2) ints of LDS.
// ------------------------------------------------------------------------------------------------------
WM_DEV void chain_block(const wm_chain_job_t jb, const wm128_t *anchor_pool, int NWV, int W, uint64_t *sx, uint64_t *sy, int *sf, int *sp, int *st_,
                        int *pub, int *gf, int *gp, int *gt)
{
	const V<int> ln = lane();
	const int wv = wave_in_block();
	const uint64_t *a = (const uint64_t*)(anchor_pool + jb.a_off);
	const int n = jb.n;
	const long long wm = (long long)W - 1;
	const bool wraps = n > W;
	int *pub_tmax = pub, *pub_mask = pub + NWV, *pub_sc = pub + NWV * 5;      // tile max | I lo,hi,M lo,hi per wave | 64 scores per wave
	if (wraps)
		for (int i0 = wv * 64; i0 < n; i0 += 64 * NWV) WM_IF(ln + i0 < n) cst(gt, cast<long long>(ln) + (long long)i0, V<int>(0)); WM_END
	long long st = 0;
	for (int i = 0; i < n; ++i) {
		if ((i & 63) == 0) {
			block_sync_lds();
			if (wv == 0) {
				if (wraps && i > 0) chain_flush_tile((long long)i - 64, (long long)i, wm, sf, sp, gf, gp);
				const V<long long> k = cast<long long>(ln) + (long long)i;
				WM_IF(k < (long long)n)
					gst(sx, k & wm, gld(a, k * 2LL)); gst(sy, k & wm, gld(a, k * 2LL + 1LL)); gst(st_, k & wm, V<int>(0));
				WM_END
			}
			block_sync_lds();
		}
		const long long lo = (long long)(i & ~63) + 64 - W;
		const uint64_t ri = gld(sx, (long long)i & wm), yi = gld(sy, (long long)i & wm);
		const int qi = (int)(uint32_t)yi, span = (int)(yi >> 32 & 0xff);
		int max_f = span, n_skip = 0;
		long long max_j = -1;
		st = chain_advance_st(st, i, ri, (uint64_t)jb.max_dist_x, -1, lo, wm, sx, a);
		if (i - st > jb.max_iter) st = chain_advance_st(st, i, ri, (uint64_t)jb.min_dist_x, jb.max_iter, lo, wm, sx, a);
		bool stop = false;
		for (long long hi0 = (long long)i - 1; hi0 >= st && !stop; hi0 -= 64LL * NWV) {
			const long long hi = hi0 - 64LL * wv;                                                   // this wave's tile
			const V<long long> j = V<long long>(hi) - cast<long long>(ln);
			const vbool in = j >= st, res = j >= lo;
			V<int> sc = 0, pj = -1, fj = 0, tj = 0;
			V<uint32_t> xj = 0u, yj = 0u;
			const V<int> jw = cast<int>(j) & (int)wm;
			vbool valid = in && !in;
			const bool all_res = hi - 63 >= lo;                                                     // whole tile resident (the common case)
			if (all_res) {
				WM_IF(in) xj = chain_lo_lds(sx, jw); yj = chain_lo_lds(sy, jw); fj = gld(sf, jw); pj = gld(sp, jw); WM_END
			} else {
				WM_IF(in && res) xj = chain_lo_lds(sx, jw); yj = chain_lo_lds(sy, jw); fj = gld(sf, jw); pj = gld(sp, jw); WM_END
				WM_IF(in && !res) xj = chain_lo_far(a, j, 0); yj = chain_lo_far(a, j, 1); fj = cld(gf, j); pj = cld(gp, j); WM_END
			}
			WM_IF(in)
				vbool ok = in;
				chain_score(jb, (uint32_t)ri, qi, span, xj, yj, fj, sc, ok);
				valid = ok;
			WM_END
			const V<long long> pjl = cast<long long>(pj);
			WM_IF(valid && pj >= 0 && pjl >= lo) gst(st_, pjl & wm, V<int>(i)); WM_END
			if (st < lo) {
				const vbool far_mark = valid && pj >= 0 && pjl < lo && pjl >= st;
				WM_IF(far_mark) cst(gt, pjl, V<int>(i)); WM_END
				if (any(far_mark)) mem_sync();
			}
			// publish this tile's maximum so that later tiles know the running maximum before them
			const V<int> key = sel(valid, sc, V<int>(-0x7fffffff - 1));
			V<int> pm = key;
			for (int o = 1; o < 64; o <<= 1) pm = vmax(pm, sel(ln >= o, shr_n(pm, o), V<int>(-0x7fffffff - 1)));
			WM_IF(ln == 63) gst(pub_tmax, V<int>(wv), pm); WM_END
			block_sync_lds();                                                                      // marks + tile maxima visible
			if (all_res) { WM_IF(valid) tj = gld(st_, j & wm); WM_END }
			else {
				WM_IF(valid && res) tj = gld(st_, j & wm); WM_END
				WM_IF(valid && !res) tj = cld(gt, j); WM_END
			}
			int run_max = max_f;
			for (int w2 = 0; w2 < wv; ++w2) { const int tm = gld(pub_tmax, (long long)w2); run_max = tm > run_max ? tm : run_max; }
			const V<int> before = vmax(sel(ln >= 1, shr_n(pm, 1), V<int>(-0x7fffffff - 1)), V<int>(run_max));
			const uint64_t I = ballot(valid && sc > before);
			const uint64_t M = ballot(valid && tj == i) & ~I;
			WM_IF(ln == 0)
				gst(pub_mask, V<int>(wv * 4), V<int>((int)(uint32_t)I)); gst(pub_mask, V<int>(wv * 4 + 1), V<int>((int)(uint32_t)(I >> 32)));
				gst(pub_mask, V<int>(wv * 4 + 2), V<int>((int)(uint32_t)M)); gst(pub_mask, V<int>(wv * 4 + 3), V<int>((int)(uint32_t)(M >> 32)));
			WM_END
			gst(pub_sc, ln + wv * 64, sc);
			block_sync_lds();                                                                      // ballots + scores visible
			for (int w2 = 0; w2 < NWV && !stop; ++w2) {                                            // every wave replays all tiles
				const long long hw = hi0 - 64LL * w2;
				if (hw < st) break;
				const uint64_t Iw = (uint64_t)(uint32_t)gld(pub_mask, (long long)w2 * 4) | (uint64_t)(uint32_t)gld(pub_mask, (long long)w2 * 4 + 1) << 32;
				const uint64_t Mw = (uint64_t)(uint32_t)gld(pub_mask, (long long)w2 * 4 + 2) | (uint64_t)(uint32_t)gld(pub_mask, (long long)w2 * 4 + 3) << 32;
				int brk = 64;
				uint64_t ev = Iw | Mw;
				while (ev) {
					const int l = __builtin_ctzll(ev);
					ev &= ev - 1;
					if (Iw >> l & 1) { if (n_skip > 0) --n_skip; }
					else if (++n_skip > jb.max_skip) { brk = l; break; }
				}
				const uint64_t Ib = brk < 64 ? Iw & (((uint64_t)1 << brk) - 1) : Iw;
				if (Ib) {
					const int l = 63 - __builtin_clzll(Ib);
					max_f = gld(pub_sc, (long long)w2 * 64 + l);
					max_j = hw - l;
				}
				if (brk < 64) stop = true;
			}
			block_sync_lds();                                                                      // pub area may be overwritten by the next step
		}
		if (wv == 0) {
			WM_IF(ln == 0)
				const V<long long> ii = (long long)i;
				gst(sf, ii & wm, V<int>(max_f)); gst(sp, ii & wm, V<int>((int)max_j));
			WM_END
		}
		block_sync_lds();
	}
	if (wraps) { if (wv == 0) chain_flush_tile((long long)((n - 1) & ~63), (long long)n, wm, sf, sp, gf, gp); }
	else
		for (int i0 = wv * 64; i0 < n; i0 += 64 * NWV) {
			const V<long long> k = cast<long long>(ln) + (long long)i0;
			WM_IF(k < (long long)n) gst(gf, k, gld(sf, k)); gst(gp, k, gld(sp, k)); WM_END
		}
}

// ------------------------------------------------------------------------------------------------------
// chain_block_wide (round 6): chain_block with the WHOLE predecessor window of an anchor scored in one step. chain_block's step is NWV tiles (512
// predecessors with eight wavefronts) behind three workgroup barriers; an anchor inside a satellite array scans up to max_iter = 5000 predecessors
// (src/chain.c:51-55), i.e. ten steps = thirty barriers per anchor, and a 5-Mb contig across such an array is ~10^6 anchors in ONE job: 14 s on one
// workgroup (profiles/r05_satellite_contig_kernels.txt). Nothing in a scan forces that order: the marks of every scored predecessor may be scattered
// before any t[j] is read (they only flow from larger to smaller j, and marks behind the break are never looked at), the running maximum before a tile
// is the maximum of the tiles in front of it — a prefix over per-tile maxima — and only the n_skip / break automaton (src/chain.c:79-86) is sequential,
// over the few improvement / marked lanes. So here every wavefront scores KT tiles (tile u = k * NWV + wv: a short scan still spreads over all
// wavefronts), one barrier, every wavefront reads its marks and turns the per-tile maxima (lanes = tiles) into the running maximum before its tiles with
// one wave scan, publishes the improvement / marked ballots, one barrier, and the automaton runs over the tiles that have events (found with one
// ballot over the published masks). NWV * KT tiles per step: 16 x 5 = 5 120 predecessors — a whole max_iter window — for the same three barriers.
// kt_first: tiles per wavefront in the FIRST step of an anchor (a step scores all of its tiles before the automaton runs, so a scan that ends early
// pays for tiles it never needed). On synthetic arrays whose scans break within a few hundred predecessors a narrow first step wins (8.4 vs 12.1 us per
// anchor, profiles/r06_chain_fill_probe.txt), but the 5-Mb contigs of BASELINE config 5 scan their whole window: 28.8 s with kt_first = 1 against 20.0 s
// with kt_first = KT (profiles/r06_closure.jsonl) — so the callers pass KT unless WM_CHAIN_WIDE_FIRST says otherwise.
// pub: NT * 69 ints (tile maxima | I lo,hi,M lo,hi per tile | 64 scores per tile, written only by tiles that have an improvement), NT = NWV * KT <= 128.
// ------------------------------------------------------------------------------------------------------
template <int KT>
WM_DEV void chain_block_wide(const wm_chain_job_t jb, const wm128_t *anchor_pool, int NWV, int kt_first, int W, uint64_t *sx, uint64_t *sy, int *sf, int *sp, int *st_,
                             int *pub, int *gf, int *gp, int *gt)
{
	const V<int> ln = lane();
	const int wv = wave_in_block();
	const uint64_t *a = (const uint64_t*)(anchor_pool + jb.a_off);
	const int n = jb.n;
	const long long wm = (long long)W - 1;
	const bool wraps = n > W;
	const int NT = NWV * KT;
	WM_EMU_ASSERT(NT <= 128);
	int *pub_tmax = pub, *pub_mask = pub + NT, *pub_sc = pub + NT * 5;      // tile max | I lo,hi,M lo,hi per tile | 64 scores per tile
	const V<int> NEG = -0x7fffffff - 1;
	if (wraps)
		for (int i0 = wv * 64; i0 < n; i0 += 64 * NWV) WM_IF(ln + i0 < n) cst(gt, cast<long long>(ln) + (long long)i0, V<int>(0)); WM_END
	long long st = 0;
	for (int i = 0; i < n; ++i) {
		if ((i & 63) == 0) {
			block_sync_lds();
			if (wv == 0) {
				if (wraps && i > 0) chain_flush_tile((long long)i - 64, (long long)i, wm, sf, sp, gf, gp);
				const V<long long> k = cast<long long>(ln) + (long long)i;
				WM_IF(k < (long long)n)
					gst(sx, k & wm, gld(a, k * 2LL)); gst(sy, k & wm, gld(a, k * 2LL + 1LL)); gst(st_, k & wm, V<int>(0));
				WM_END
			}
			block_sync_lds();
		}
		const long long lo = (long long)(i & ~63) + 64 - W;
		const uint64_t ri = gld(sx, (long long)i & wm), yi = gld(sy, (long long)i & wm);
		const int qi = (int)(uint32_t)yi, span = (int)(yi >> 32 & 0xff);
		int max_f = span, n_skip = 0;
		long long max_j = -1;
		st = chain_advance_st(st, i, ri, (uint64_t)jb.max_dist_x, -1, lo, wm, sx, a);
		if (i - st > jb.max_iter) st = chain_advance_st(st, i, ri, (uint64_t)jb.min_dist_x, jb.max_iter, lo, wm, sx, a);
		bool stop = false;
		int NTs = NWV * kt_first;                                                               // tiles of the first step (see the header), NT from the second on
		const int sti = (int)st, loi = (int)lo;                                                 // (anchor indices are ints, n < 2^31: the tile bounds are 32-bit scalar arithmetic)
		for (int hi0 = i - 1; hi0 >= sti && !stop; hi0 -= 64 * NTs, NTs = NT) {
			// tiles of this step that hold predecessors at all: 0 .. nt - 1
			const int span_j = hi0 - sti + 1;
			const int nt = span_j >= 64 * NTs ? NTs : (span_j + 63) >> 6;
			V<int> sc[KT], pmx[KT];
			vbool valid[KT];
			bool any_far = false;
			// ---- phase A: score, scatter the marks, publish the tile maxima ----
#pragma unroll
			for (int k = 0; k < KT; ++k) {
				const int u = k * NWV + wv;
				sc[k] = 0; pmx[k] = NEG;
				valid[k] = ln < 0;                                                              // false
				if (u >= nt) continue;
				const int hi = hi0 - 64 * u;
				const V<int> j = V<int>(hi) - ln;
				const V<int> jw = j & (int)wm;
				V<int> pj = -1, fj = 0;
				V<uint32_t> xj = 0u, yj = 0u;
				if (hi - 63 >= sti && hi - 63 >= loi) {                                         // every lane holds a resident predecessor (all but the last tile of a scan inside the window): no lane masks
					WM_KEEP_BRANCH();
					xj = chain_lo_lds(sx, jw); yj = chain_lo_lds(sy, jw); fj = gld(sf, jw); pj = gld(sp, jw);
					vbool ok = ln >= 0;
					chain_score(jb, (uint32_t)ri, qi, span, xj, yj, fj, sc[k], ok);
					valid[k] = ok;
				} else {
					const vbool in = j >= sti;
					if (hi - 63 >= loi) {                                                       // whole tile resident
						WM_IF(in) xj = chain_lo_lds(sx, jw); yj = chain_lo_lds(sy, jw); fj = gld(sf, jw); pj = gld(sp, jw); WM_END
					} else {
						const vbool res = j >= loi;
						const V<long long> jl = cast<long long>(j);
						WM_IF(in && res) xj = chain_lo_lds(sx, jw); yj = chain_lo_lds(sy, jw); fj = gld(sf, jw); pj = gld(sp, jw); WM_END
						WM_IF(in && !res) xj = chain_lo_far(a, jl, 0); yj = chain_lo_far(a, jl, 1); fj = cld(gf, jl); pj = cld(gp, jl); WM_END
					}
					WM_IF(in)
						vbool ok = in;
						chain_score(jb, (uint32_t)ri, qi, span, xj, yj, fj, sc[k], ok);
						valid[k] = ok;
					WM_END
				}
				if (sti >= loi) {                                                                 // every predecessor that can be marked is resident (p[j] < st is never visited: its mark would never be read)
					WM_IF(valid[k] && pj >= sti) gst(st_, pj & (int)wm, V<int>(i)); WM_END
				} else {
					WM_IF(valid[k] && pj >= 0 && pj >= loi) gst(st_, pj & (int)wm, V<int>(i)); WM_END
					const vbool far_mark = valid[k] && pj >= 0 && pj < loi && pj >= sti;
					WM_IF(far_mark) cst(gt, cast<long long>(pj), V<int>(i)); WM_END
					any_far = any_far || any(far_mark);
				}
				pmx[k] = wave_scan_max(sel(valid[k], sc[k], NEG));                              // inclusive prefix maximum inside the tile (lane 0 = first visited)
				WM_IF(ln == 63) gst(pub_tmax, V<int>(u), pmx[k]); WM_END
			}
			if (any_far) mem_sync();
			block_sync_lds();                                                                   // marks + tile maxima visible
			// ---- phase B: the running maximum before each tile (lanes = tiles), improvement / marked ballots ----
			V<int> t0 = NEG, t1 = NEG;
			WM_IF(ln < nt) t0 = gld(pub_tmax, ln); WM_END
			if (nt > 64) { WM_IF(ln + 64 < nt) t1 = gld(pub_tmax, ln + 64); WM_END }
			const V<int> s0 = wave_scan_max(t0);
			const int top0 = readlane(s0, 63);
			V<int> s1 = NEG;
			if (nt > 64) s1 = vmax(wave_scan_max(t1), V<int>(top0));
#pragma unroll
			for (int k = 0; k < KT; ++k) {
				const int u = k * NWV + wv;
				if (u >= nt) continue;
				const int hi = hi0 - 64 * u;
				const V<int> j = V<int>(hi) - ln;
				V<int> tj = 0;
				if (hi - 63 >= sti && hi - 63 >= loi) tj = gld(st_, j & (int)wm);              // (no mask: only valid lanes' marks are looked at)
				else if (hi - 63 >= loi) { WM_IF(valid[k]) tj = gld(st_, j & (int)wm); WM_END }
				else {
					const vbool res = j >= loi;
					WM_IF(valid[k] && res) tj = gld(st_, j & (int)wm); WM_END
					WM_IF(valid[k] && !res) tj = cld(gt, cast<long long>(j)); WM_END
				}
				int run_before = max_f;
				if (u > 0) { const int tb = u - 1 < 64 ? readlane(s0, u - 1) : readlane(s1, u - 1 - 64); run_before = tb > run_before ? tb : run_before; }
				const V<int> before = vmax(shr1(pmx[k], NEG), V<int>(run_before));
				const uint64_t I = ballot(valid[k] && sc[k] > before);
				const uint64_t M = ballot(valid[k] && tj == i) & ~I;
				WM_IF(ln == 0)
					gst(pub_mask, V<int>(u * 4), V<int>((int)(uint32_t)I)); gst(pub_mask, V<int>(u * 4 + 1), V<int>((int)(uint32_t)(I >> 32)));
					gst(pub_mask, V<int>(u * 4 + 2), V<int>((int)(uint32_t)M)); gst(pub_mask, V<int>(u * 4 + 3), V<int>((int)(uint32_t)(M >> 32)));
				WM_END
				if (I) gst(pub_sc, ln + u * 64, sc[k]);                                         // (only a tile with an improvement is ever asked for a score)
			}
			block_sync_lds();                                                                   // ballots + scores visible
			// ---- phase C: the automaton over the tiles that have events (every wavefront replays it: all agree without a broadcast) ----
			{
				vbool has0 = ln < 0, has1 = ln < 0;
				WM_IF(ln < nt) has0 = (gld(pub_mask, ln * 4) | gld(pub_mask, ln * 4 + 1) | gld(pub_mask, ln * 4 + 2) | gld(pub_mask, ln * 4 + 3)) != 0; WM_END
				if (nt > 64) { WM_IF(ln + 64 < nt) has1 = (gld(pub_mask, (ln + 64) * 4) | gld(pub_mask, (ln + 64) * 4 + 1) | gld(pub_mask, (ln + 64) * 4 + 2) | gld(pub_mask, (ln + 64) * 4 + 3)) != 0; WM_END }
				uint64_t ev_t[2] = { ballot(has0 && ln < nt), nt > 64 ? ballot(has1 && ln + 64 < nt) : 0 };
				for (int half = 0; half < 2 && !stop; ++half) {
					uint64_t evt = ev_t[half];
					while (evt && !stop) {
						const int u = __builtin_ctzll(evt) + 64 * half;
						evt &= evt - 1;
						const int hw = hi0 - 64 * u;
						const uint64_t Iw = (uint64_t)(uint32_t)gld(pub_mask, (long long)u * 4) | (uint64_t)(uint32_t)gld(pub_mask, (long long)u * 4 + 1) << 32;
						const uint64_t Mw = (uint64_t)(uint32_t)gld(pub_mask, (long long)u * 4 + 2) | (uint64_t)(uint32_t)gld(pub_mask, (long long)u * 4 + 3) << 32;
						int brk = 64;
						uint64_t ev = Iw | Mw;
						while (ev) {
							const int l = __builtin_ctzll(ev);
							ev &= ev - 1;
							if (Iw >> l & 1) { if (n_skip > 0) --n_skip; }
							else if (++n_skip > jb.max_skip) { brk = l; break; }
						}
						const uint64_t Ib = brk < 64 ? Iw & (((uint64_t)1 << brk) - 1) : Iw;
						if (Ib) {
							const int l = 63 - __builtin_clzll(Ib);
							max_f = gld(pub_sc, (long long)u * 64 + l);
							max_j = hw - l;
						}
						if (brk < 64) stop = true;
					}
				}
			}
			if (hi0 - 64 * NTs >= sti && !stop) block_sync_lds();                               // another step follows: it overwrites the pub area (after the last step the barrier below does)
		}
		if (wv == 0) {
			WM_IF(ln == 0)
				const V<long long> ii = (long long)i;
				gst(sf, ii & wm, V<int>(max_f)); gst(sp, ii & wm, V<int>((int)max_j));
			WM_END
		}
		block_sync_lds();
	}
	if (wraps) { if (wv == 0) chain_flush_tile((long long)((n - 1) & ~63), (long long)n, wm, sf, sp, gf, gp); }
	else
		for (int i0 = wv * 64; i0 < n; i0 += 64 * NWV) {
			const V<long long> k = cast<long long>(ln) + (long long)i0;
			WM_IF(k < (long long)n) gst(gf, k, gld(sf, k)); gst(gp, k, gld(sp, k)); WM_END
		}
}

} // namespace wmk
