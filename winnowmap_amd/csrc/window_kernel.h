// window_kernel.h — one MCAS window (or stage-2 pass) from minimizers to chains WITHOUT leaving HBM:
//
//   sketch_coop (sketch_kernel.h)  →  win_seed_wave  →  win_sort_wave  →  chain_wave / chain_block (seedchain_kernel.h)  →  win_extract_wave
//   mm_sketch                         collect_seed_hits  radix_sort_128x   mm_chain_dp fill                                   mm_chain_dp :93-165
//   src/sketch.c:128-219              src/map.c:222-251  src/ksort.h:101-151  src/chain.c:45-90                               src/chain.c:93-165
//
// Round 2 made three batched calls per window with the host in the middle (minimizers D2H → H2D, anchors D2H → host radix_sort_128x →
// H2D, f/p D2H → host chain extraction). Here the anchors of a job never leave the device: the seed kernel allocates the job's region
// in one anchor pool with a wave-uniform atomic bump, the sort kernel reproduces the reference's UNSTABLE in-place radix sort exactly —
// ties (two query minimizers hitting the same reference position) are common in repeats and their order decides the chains — and the
// extraction kernel replays src/chain.c:93-165 literally. Only (u, compacted anchors, rep_len) travel back.
//
// How the sort is made exact: the reference's rs_sort (src/ksort.h:116-146) is, per digit, (1) a histogram, (2) a prefix sum, (3) an
// in-place cycle-leader permutation whose swap order fixes where equal keys end up, (4) per bucket: recursion (> 64 elements) or
// insertion sort (stable). (1), (2) and (4)'s insertion sorts are order-free / stable and run wave-parallel (LDS atomics, a DPP scan, a
// 64-lane rank sort); (3) is sequential by nature and is replayed LITERALLY by lane 0 on the LDS-staged anchors. Digits that are the
// same in every key of a range leave it untouched and are skipped (as wm::FlagSort does on the host).
#pragma once
#ifndef WM_DEV
#error "include simt.h before window_kernel.h"
#endif
#include "wm_internal.h"

namespace wmk {
using namespace simt;

// lane 0 stores one int (uniform bookkeeping that lives in LDS)
WM_DEV void ust(int *p, int i, int v) { WM_IF(lane() == 0) gst(p, V<long long>((long long)i), V<int>(v)); WM_END }
// Cross-lane traffic of ONE wavefront through global memory: the lanes share a CU and its write-through L1, so what is needed is that the
// stores have left the wave (s_waitcnt) and that the compiler keeps the order — a WORKGROUP-scope fence. (An agent-scope release would write
// the XCD's L2 back and invalidate it — per job, a million times per mini-batch, under every other kernel's feet.)
WM_DEV void win_fence() { mem_sync(); }
template <bool G> WM_DEV void win_sync() { if (G) win_fence(); else lds_sync(); }

// ------------------------------------------------------------------------------------------------------------------------------
// collect_seed_hits (src/map.c:222-251) for one job, in two passes over the job's minimizers:
//   pass 1  probe the index (mm_idx_get, src/index.c:88-105), apply the occurrence filter (src/map.c:108-117), accumulate rep_len and
//           count the anchors; (cnt, first, emit) of every minimizer are cached so that pass 2 does not probe again;
//   then    the wave takes [n_pre + n_seeded) slots from the anchor pool (one atomic per job);
//   pass 2  copies the handed-in anchors and expands the minimizers in order (src/map.c:232-249).
// rep_len (src/map.c:111-116,126) is the length of the union of the intervals of the dropped minimizers; with their end positions
// increasing it is sum(en_i - max(st_i, en_prev)) over the dropped ones — a prefix maximum instead of the sequential merge.
// ------------------------------------------------------------------------------------------------------------------------------
WM_DEV void win_seed_wave(const wm_index_view_t ix, const wm_win_job_t jb, const wm128_t *mini_, int n_mini, const wm128_t *pre_,
                          int *occ, uint32_t *first_, int *emit_, wm128_t *anchor_pool, uint64_t *pool_used, uint64_t pool_cap, wm_win_res_t *res)
{
	const V<int> ln = lane();
	const uint64_t *mini = (const uint64_t*)mini_;
	const uint64_t hmask = ((uint64_t)1 << ix.hbits) - 1;
	const bool strand_filter = (jb.seed_flag & (0x100000 | 0x200000)) != 0;
	int total = 0, rep_len = 0, carry_en = 0;
	for (int m0 = 0; m0 < n_mini; m0 += 64) {
		const V<int> m = ln + m0;
		const vbool have = m < n_mini;
		V<uint64_t> mx = (uint64_t)0, my = (uint64_t)0, first = (uint64_t)0;
		V<int> cnt = 0;
		WM_IF(have)
			mx = gld(mini, m * 2); my = gld(mini, m * 2 + 1);
			const V<uint64_t> key = mx >> 8;
			V<uint64_t> s = (key * (uint64_t)0x9E3779B97F4A7C15ULL) >> (64 - ix.hbits);
			vbool probing = s == s;
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
		WM_END
		V<int> emit = sel(have && cnt < jb.max_occ, cnt, 0);
		WM_IF(strand_filter && emit > 0)
			V<int> kept = 0;
			const V<int> qstrand = cast<int>(my & (uint64_t)1);
			for (int h = 0; h < jb.max_occ && any(emit > h); ++h)
				WM_IF(emit > h)
					const V<int> rstrand = cast<int>(gld(ix.P, first + (uint64_t)h) & (uint64_t)1);
					const vbool fwd = rstrand == qstrand;
					kept = kept + sel((fwd && !(jb.seed_flag & 0x200000)) || (!fwd && !(jb.seed_flag & 0x100000)), 1, 0);
				WM_END
			emit = kept;
		WM_END
		WM_IF(have) gst(occ, m, cnt); gst(first_, m, cast<uint32_t>(first)); gst(emit_, m, emit); WM_END
		// rep_len: the dropped minimizers of this tile
		const vbool dropped = have && cnt >= jb.max_occ;
		const V<int> en = cast<int>(cast<uint32_t>(my) >> 1) + 1, st = en - cast<int>(mx & (uint64_t)0xff);
		const V<int> incl = wave_scan_max(sel(dropped, en, V<int>(0)));
		const V<int> prev = vmax(shr1(incl, 0), V<int>(carry_en));
		rep_len += readlane(wave_sum_i32(sel(dropped, en - vmax(st, prev), V<int>(0))), 0);
		carry_en = vmax(carry_en, readlane(incl, 63));
		total += readlane(wave_sum_i32(emit), 0);
	}
	const int n_a = jb.n_pre + total;
	const uint64_t a_off = wave_alloc(pool_used, (uint64_t)n_a);
	const bool fits = a_off + (uint64_t)n_a <= pool_cap;
	WM_IF(ln == 0)
		gst(&res->a_off, V<long long>(0LL), V<uint64_t>(a_off));
		gst(&res->n_a, V<long long>(0LL), V<int>(fits ? n_a : 0));
		gst(&res->rep_len, V<long long>(0LL), V<int>(rep_len));
		gst(&res->n_mini, V<long long>(0LL), V<int>(n_mini));
		gst(&res->n_u, V<long long>(0LL), V<int>(0)); gst(&res->n_v, V<long long>(0LL), V<int>(0));
	WM_END
	if (!fits) { WM_IF(ln == 0) gst(&res->err, V<long long>(0LL), V<int>(2)); WM_END return; }
	uint64_t *outp = (uint64_t*)(anchor_pool + a_off);
	const uint64_t *pre = (const uint64_t*)pre_;
	for (int i0 = 0; i0 < jb.n_pre; i0 += 64) {                   // the anchors handed in come first (src/map.c:818-826)
		const V<int> i = ln + i0;
		WM_IF(i < jb.n_pre) gst(outp, i * 2, gld(pre, i * 2)); gst(outp, i * 2 + 1, gld(pre, i * 2 + 1)); WM_END
	}
	int base = jb.n_pre;
	for (int m0 = 0; m0 < n_mini; m0 += 64) {
		const V<int> m = ln + m0;
		const vbool have = m < n_mini;
		V<int> cnt = 0, emit = 0;
		V<uint64_t> first = (uint64_t)0, mx = (uint64_t)0, my = (uint64_t)0;
		WM_IF(have) cnt = gld(occ, m); emit = gld(emit_, m); first = cast<uint64_t>(gld(first_, m)); mx = gld(mini, m * 2); my = gld(mini, m * 2 + 1); WM_END
		const V<int> incl = wave_scan_add(emit);
		const V<int> excl = incl - emit;
		const int tile_total = readlane(incl, 63);
		WM_IF(emit > 0)
			const V<uint32_t> q_pos = cast<uint32_t>(my), q_span = cast<uint32_t>(mx & (uint64_t)0xff);
			vbool tandem = q_pos != q_pos;                // the neighbouring minimizer has the same key (src/map.c:121-122)
			WM_IF(m > 0) tandem = tandem || ((gld(mini, (m - 1) * 2) >> 8) == (mx >> 8)); WM_END
			WM_IF(m < n_mini - 1) tandem = tandem || ((gld(mini, (m + 1) * 2) >> 8) == (mx >> 8)); WM_END
			V<int> w = excl + base;
			for (int h = 0; h < jb.max_occ && any(cnt > h); ++h)
				WM_IF(cnt > h)
					const V<uint64_t> r = gld(ix.P, first + (uint64_t)h);
					const V<uint64_t> rpos = (r & (uint64_t)0xffffffffu) >> 1;
					const vbool fwd = cast<uint32_t>(r & (uint64_t)1) == (q_pos & 1u);
					vbool keep = rpos == rpos;
					if (strand_filter) keep = (fwd && !(jb.seed_flag & 0x200000)) || (!fwd && !(jb.seed_flag & 0x100000));
					WM_IF(keep)
						V<uint64_t> ax = (r & (uint64_t)0xffffffff00000000ULL) | rpos;
						V<uint64_t> ay = cast<uint64_t>(q_span) << 32;
						WM_IF(fwd) ay = ay | cast<uint64_t>(q_pos >> 1); WM_ELSE
							ax = ax | ((uint64_t)1 << 63);
							ay = ay | cast<uint64_t>(cast<uint32_t>(V<int>(jb.len) - cast<int>((q_pos >> 1) + 1u - q_span) - 1));
						WM_END
						ay = sel(tandem, ay | ((uint64_t)1 << 42), ay);
						gst(outp, w * 2, ax); gst(outp, w * 2 + 1, ay);
						w = w + 1;
					WM_END
				WM_END
		WM_END
		base += tile_total;
	}
}

// ------------------------------------------------------------------------------------------------------------------------------
// radix_sort_128x (src/ksort.h:101-151) of one job by one wavefront; a = (x, y) pairs in LDS (G = false) or global memory (G = true).
// ws: WIN_WS_INTS ints of LDS.
// ------------------------------------------------------------------------------------------------------------------------------
enum { WIN_WS_HEAD = 0, WIN_WS_TAIL = 256, WIN_WS_NZ = 256 + 8 * 256, WIN_WS_FRAME = WIN_WS_NZ + 256, WIN_WS_INTS = WIN_WS_FRAME + 8 * 4 };

// rs_insertsort (src/ksort.h:105-115) of a[b0 .. b0 + m), m <= 64: a stable sort, so every element's final place is its rank
template <bool G> WM_DEV void rs_rank_sort(uint64_t *a, int b0, int m)
{
	const V<int> ln = lane();
	const vbool have = ln < m;
	V<uint64_t> kx = ~(uint64_t)0, ky = (uint64_t)0;
	WM_IF(have) kx = gld(a, cast<long long>(ln + b0) * 2LL); ky = gld(a, cast<long long>(ln + b0) * 2LL + 1LL); WM_END
	V<int> rank = 0;
	for (int j = 0; j < m; ++j) {
		const V<uint64_t> kj = readlane(kx, j);
		rank = rank + sel(kj < kx || (kj == kx && ln > j), 1, 0);
	}
	WM_IF(have) gst(a, cast<long long>(rank + b0) * 2LL, kx); gst(a, cast<long long>(rank + b0) * 2LL + 1LL, ky); WM_END
	win_sync<G>();
}

// the cycle-leader permutation of one digit (src/ksort.h:126-138), literally; scalar code, lane 0 only. nz[0..n_nz) = the non-empty
// buckets in ascending order (empty ones are skipped by the reference's `else ++k`)
WM_DEV void rs_permute(uint64_t *a, int *head, const int *tail, const int *nz, int n_nz, int shift)
{
	for (int q = 0; q < n_nz;) {
		const int d = nz[q], hd = head[d];
		if (hd == tail[d]) { ++q; continue; }
		uint64_t hx = a[2 * (long long)hd];
		int dst = (int)(hx >> shift & 0xff);
		if (dst == d) { head[d] = hd + 1; continue; }
		uint64_t hy = a[2 * (long long)hd + 1];
		while (dst != d) {
			const int p = head[dst];
			const uint64_t tx = a[2 * (long long)p], ty = a[2 * (long long)p + 1];
			a[2 * (long long)p] = hx; a[2 * (long long)p + 1] = hy;
			hx = tx; hy = ty;
			head[dst] = p + 1;
			dst = (int)(hx >> shift & 0xff);
		}
		a[2 * (long long)hd] = hx; a[2 * (long long)hd + 1] = hy;
		head[d] = hd + 1;
	}
}

template <bool G> WM_DEV void win_sort_wave(wm128_t *a_, int n, int *ws)
{
	uint64_t *a = (uint64_t*)a_;
	if (n <= 1) return;
	if (n <= 64) { rs_rank_sort<G>(a, 0, n); return; }                       // src/ksort.h:149
	const V<int> ln = lane();
	int *head = ws + WIN_WS_HEAD, *nz = ws + WIN_WS_NZ, *fr = ws + WIN_WS_FRAME;
	// frame d of the recursion = one range whose digit has been permuted and whose buckets are being visited: fr[4d] = first element,
	// fr[4d+1] = shift of the NEXT digit, fr[4d+2] = next bucket position to look at (q * 64 + lane: bucket 4 * lane + q), tails in ws
	int depth = 0, beg = 0, end = n, shift = 56;
	bool enter = true;                                                         // enter: process the digit of [beg, end) at `shift`
	for (;;) {
		if (enter) {
			// digits that are the same in every key of the range: one bucket, nothing moves, the range goes on to the next digit
			V<uint64_t> vo = (uint64_t)0, va = ~(uint64_t)0;
			for (int i0 = beg; i0 < end; i0 += 64) {
				const V<int> i = ln + i0;
				WM_IF(i < end) const V<uint64_t> k = gld(a, cast<long long>(i) * 2LL); vo = vo | k; va = va & k; WM_END
			}
			const uint64_t diff = wave_or_u64(vo) ^ wave_and_u64(va);
			bool nothing = false;
			while (!(diff >> shift & 0xff)) { if (shift == 0) { nothing = true; break; } shift = shift > 8 ? shift - 8 : 0; }
			if (nothing) { enter = false; --depth; if (depth < 0) return; continue; }
			int *tail = ws + WIN_WS_TAIL + depth * 256;
#pragma unroll
			for (int q = 0; q < 4; ++q) gst(head, ln * 4 + q, V<int>(0));
			lds_sync();
			for (int i0 = beg; i0 < end; i0 += 64) {
				const V<int> i = ln + i0;
				WM_IF(i < end) atomic_inc(head, cast<int>(gld(a, cast<long long>(i) * 2LL) >> shift & (uint64_t)0xff)); WM_END
			}
			lds_sync();
			V<int> c[4], h[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) c[q] = gld(head, ln * 4 + q);
			const V<int> s = c[0] + c[1] + c[2] + c[3];
			const V<int> incl = wave_scan_add(s);
			h[0] = incl - s + beg; h[1] = h[0] + c[0]; h[2] = h[1] + c[1]; h[3] = h[2] + c[2];
			// the non-empty buckets in ascending order (bucket = 4 * lane + q)
			const V<int> nzc = sel(c[0] > 0, 1, 0) + sel(c[1] > 0, 1, 0) + sel(c[2] > 0, 1, 0) + sel(c[3] > 0, 1, 0);
			const V<int> nzi = wave_scan_add(nzc);
			V<int> np = nzi - nzc;
			const int n_nz = readlane(nzi, 63);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				gst(head, ln * 4 + q, h[q]); gst(tail, ln * 4 + q, h[q] + c[q]);
				WM_IF(c[q] > 0) gst(nz, np, ln * 4 + q); np = np + 1; WM_END
			}
			win_sync<G>();
			WM_LANE0_BEGIN rs_permute(a, head, tail, nz, n_nz, shift); WM_LANE0_END
			win_sync<G>();
			if (shift == 0) { enter = false; --depth; if (depth < 0) return; continue; }      // the reference recurses only while s > 0 (:139)
			ust(fr, 4 * depth, beg); ust(fr, 4 * depth + 1, shift > 8 ? shift - 8 : 0); ust(fr, 4 * depth + 2, 0);
			lds_sync();
			// buckets of 2 .. 64 elements: insertion sort (:144) = stable rank sort; order among buckets is irrelevant
			{
				const int *tl = tail;
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					uint64_t small = ballot(c[q] > 1 && c[q] <= 64);
					while (small) {
						const int l = __builtin_ctzll(small);
						small &= small - 1;
						const int b0 = readlane(h[q], l), m = readlane(c[q], l);
						rs_rank_sort<G>(a, b0, m);
					}
				}
				(void)tl;
			}
			enter = false;
		}
		// ---- visit the next bucket of more than 64 elements of frame `depth` (:143) ----
		{
			const int *tail = ws + WIN_WS_TAIL + depth * 256;
			const int fbeg = uniform(gld(fr, (long long)(4 * depth))), fnext = uniform(gld(fr, (long long)(4 * depth + 1))), fpos = uniform(gld(fr, (long long)(4 * depth + 2)));
			V<int> t[4], sz[4], hd[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) t[q] = gld(tail, ln * 4 + q);
			const V<int> prev_last = shr1(t[3], fbeg);                       // tail of bucket 4 * lane - 1 (lane 0: the range start)
			hd[0] = prev_last; hd[1] = t[0]; hd[2] = t[1]; hd[3] = t[2];
#pragma unroll
			for (int q = 0; q < 4; ++q) sz[q] = t[q] - hd[q];
			int found_b = -1, found_e = -1, found_pos = -1;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				if (found_pos >= 0 || q < (fpos >> 6)) continue;
				uint64_t big = ballot(sz[q] > 64);
				if (q == (fpos >> 6)) big &= ~(uint64_t)0 << (fpos & 63);
				if (big) { const int l = __builtin_ctzll(big); found_b = readlane(hd[q], l); found_e = readlane(t[q], l); found_pos = q * 64 + l; }
			}
			if (found_pos < 0) { --depth; if (depth < 0) return; continue; }     // frame finished
			ust(fr, 4 * depth + 2, found_pos + 1);
			lds_sync();
			++depth;
			beg = found_b; end = found_e; shift = fnext; enter = true;
		}
	}
}

// ------------------------------------------------------------------------------------------------------------------------------
// Anchor sets beyond the LDS classes (reads inside a repeat family: 10^4 .. 10^6 anchors) sorted by a whole WORKGROUP: a stable LSD radix
// sort, 8 bits per pass, constant digits skipped. Wavefront w owns the w-th contiguous chunk of the array (stability = chunk order, then
// position): per pass every wavefront counts its chunk (LDS atomics), the counts are prefixed digit-major / wavefront-minor, and every
// wavefront scatters its chunk tile by tile — the rank of an element among the equal digits of its tile is a popcount over a match mask
// built from 8 ballots. Two workgroup barriers per pass, no global atomics. The result is the reference's order whenever the keys are
// distinct; if two keys tie (*tie != 0: found by comparing neighbours after the last pass) the order of equal keys is decided by the reference's
// unstable sort and the caller falls back to the literal replay (win_sort_wave) on the untouched input.
// src: n elements (not modified), b0 / b1: n elements each (ping-pong). Returns (uniform) 0 / 1 = the buffer that holds the sorted array, -1 = src
// itself (nothing to do). lds: WIN_BIG_INTS(NWV) ints.
// ------------------------------------------------------------------------------------------------------------------------------
#define WIN_BIG_INTS(NWV) ((NWV) * 256 + 256 + 4 * (NWV) + 8)
WM_DEV int win_bigsort_block(int NWV, const wm128_t *src_, wm128_t *b0_, wm128_t *b1_, int n, int *lds, int *tie)
{
	const V<int> ln = lane();
	const int wv = wave_in_block();
	int *hist = lds, *tot = lds + NWV * 256, *red = tot + 256, *flag = red + 4 * NWV;
	const uint64_t *src = (const uint64_t*)src_;
	uint64_t *buf[2] = { (uint64_t*)b0_, (uint64_t*)b1_ };
	const int per = (n + 64 * NWV - 1) / (64 * NWV) * 64;
	const int beg = wv * per < n ? wv * per : n, end = beg + per < n ? beg + per : n;
	// which digits vary
	V<uint64_t> vo = (uint64_t)0, va = ~(uint64_t)0;
	for (int i0 = beg; i0 < end; i0 += 64) {
		const V<int> i = ln + i0;
		WM_IF(i < end) const V<uint64_t> k = gld(src, cast<long long>(i) * 2LL); vo = vo | k; va = va & k; WM_END
	}
	{
		const uint64_t o = wave_or_u64(vo), a = wave_and_u64(va);
		ust(red, 4 * wv, (int)(uint32_t)o); ust(red, 4 * wv + 1, (int)(uint32_t)(o >> 32)); ust(red, 4 * wv + 2, (int)(uint32_t)a); ust(red, 4 * wv + 3, (int)(uint32_t)(a >> 32));
		if (wv == 0) ust(flag, 0, 0);
	}
	block_sync_lds();
	uint64_t all_or = 0, all_and = ~(uint64_t)0;
	for (int w = 0; w < NWV; ++w) {
		all_or |= (uint64_t)(uint32_t)uniform(gld(red, (long long)(4 * w))) | (uint64_t)(uint32_t)uniform(gld(red, (long long)(4 * w + 1))) << 32;
		all_and &= (uint64_t)(uint32_t)uniform(gld(red, (long long)(4 * w + 2))) | (uint64_t)(uint32_t)uniform(gld(red, (long long)(4 * w + 3))) << 32;
	}
	const uint64_t diff = all_or ^ all_and;
	int cur = -1;                                                             // where the array is now: -1 = src
	for (int shift = 0; shift < 64; shift += 8) {
		if (!(diff >> shift & 0xff)) continue;
		const uint64_t *in = cur < 0 ? src : buf[cur];
		uint64_t *out = buf[cur == 0 ? 1 : 0];
		int *myh = hist + wv * 256;
#pragma unroll
		for (int q = 0; q < 4; ++q) gst(myh, ln * 4 + q, V<int>(0));
		lds_sync();
		for (int i0 = beg; i0 < end; i0 += 64) {
			const V<int> i = ln + i0;
			WM_IF(i < end) atomic_inc(myh, cast<int>(gld(in, cast<long long>(i) * 2LL) >> shift & (uint64_t)0xff)); WM_END
		}
		block_sync_lds();
		// digit d: the chunks' counts become exclusive offsets inside the digit (wavefront-minor), tot[d] the digit's size
		for (int d0 = wv * 64; d0 < 256; d0 += 64 * NWV) {
			const V<int> d = ln + d0;
			V<int> run = 0;
			for (int w = 0; w < NWV; ++w) { const V<int> t = gld(hist + w * 256, d); gst(hist + w * 256, d, run); run = run + t; }
			gst(tot, d, run);
		}
		block_sync_lds();
		if (wv == 0) {                                                        // exclusive scan over the 256 digit sizes
			V<int> c[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) c[q] = gld(tot, ln * 4 + q);
			const V<int> s = c[0] + c[1] + c[2] + c[3];
			V<int> h = wave_scan_add(s) - s;
#pragma unroll
			for (int q = 0; q < 4; ++q) { gst(tot, ln * 4 + q, h); h = h + c[q]; }
		}
		block_sync_lds();
		for (int i0 = beg; i0 < end; i0 += 64) {                                // stable scatter of this wavefront's chunk, tile by tile
			const V<int> i = ln + i0;
			const vbool have = i < end;
			V<uint64_t> kx = (uint64_t)0, ky = (uint64_t)0;
			WM_IF(have) kx = gld(in, cast<long long>(i) * 2LL); ky = gld(in, cast<long long>(i) * 2LL + 1LL); WM_END
			const V<int> d = cast<int>(kx >> shift & (uint64_t)0xff);
			V<uint64_t> peers = ballot(have);
#pragma unroll
			for (int bit = 0; bit < 8; ++bit) {
				const vbool on = ((d >> bit) & 1) != 0;
				const uint64_t m = ballot(have && on);
				peers = peers & sel(on, V<uint64_t>(m), V<uint64_t>(~m));
			}
			const V<int> rank = vpopc64(peers & lanemask_lt());
			WM_IF(have)
				const V<int> pos = gld(tot, d) + gld(myh, d) + rank;
				gst(out, cast<long long>(pos) * 2LL, kx); gst(out, cast<long long>(pos) * 2LL + 1LL, ky);
			WM_END
			lds_sync();
			WM_IF(have && rank == 0) gst(myh, d, gld(myh, d) + vpopc64(peers)); WM_END
			lds_sync();
		}
		cur = cur == 0 ? 1 : 0;
		win_fence();
		block_sync_lds();
	}
	// equal neighbours?
	{
		const uint64_t *res = cur < 0 ? src : buf[cur];
		vbool t = ln != ln;
		for (int i0 = beg; i0 < end; i0 += 64) {
			const V<int> i = ln + i0;
			WM_IF(i < end && i + 1 < n) t = t || gld(res, cast<long long>(i) * 2LL) == gld(res, cast<long long>(i + 1) * 2LL); WM_END
		}
		if (any(t)) ust(flag, 0, 1);
	}
	block_sync_lds();
	*tie = uniform(gld(flag, 0LL));
	block_sync_lds();
	return cur;
}

// ------------------------------------------------------------------------------------------------------------------------------
// what mm_chain_dp needs before its fill (src/chain.c:36-40 avg_qspan) and the kernel class of the fill; appended to the class's list
// ------------------------------------------------------------------------------------------------------------------------------
// avg_qspan (src/chain.c:36-40): (float)sum / n with an exact integer sum
WM_DEV float win_avg_qspan(const wm128_t *a_, int n)
{
	const V<int> ln = lane();
	const uint64_t *a = (const uint64_t*)a_;
	V<int> part = 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const V<int> i = ln + i0;
		WM_IF(i < n) part = part + cast<int>(gld(a, cast<long long>(i) * 2LL + 1LL) >> 32 & (uint64_t)0xff); WM_END
	}
	const int sum = readlane(wave_sum_i32(part), 0);                          // (spans are < 256: the sum fits 32 bits below 8.4 M anchors, far beyond what a call's anchor pool holds per job)
	return (float)(uint64_t)(uint32_t)sum / (float)(long long)n;
}

WM_DEV void win_plan_wave(const wm_win_job_t jb, int j, uint64_t a_off, int n, const wm128_t *a_, wm_chain_job_t *cj, int *lists, int *counts, int n_jobs)
{
	const V<int> ln = lane();
	const uint64_t *a = (const uint64_t*)a_;
	if (n <= 0) return;
	const float avg = win_avg_qspan(a_, n);
	int klass = n > 256 ? (n > 1024 ? 1 : 2) : 3;
	if (n > 1024) {            // DENSE if a sample of anchors has more than ~900 predecessors within max_dist_x (satellite arrays), or (round 6) if a quarter of the sample has
		                       // more than two tiles of them: the workgroup kernel scores the whole window in one step, the one-wavefront kernel four tiles at a time
		V<int> worst = 0;
		WM_IF(ln >= 1 && ln <= 32)
			const V<long long> k = cast<long long>(ln) * (long long)n / 33LL;
			const V<uint64_t> xk = gld(a, k * 2LL);
			const V<uint64_t> lim = sel(xk > (uint64_t)jb.max_dist_x, xk - (uint64_t)jb.max_dist_x, V<uint64_t>((uint64_t)0));
			V<long long> lo = 0LL, hi = k;
			for (int it = 0; it < 40 && any(lo < hi); ++it)
				WM_IF(lo < hi)
					const V<long long> mid = (lo + hi) >> 1;
					const vbool less = gld(a, mid * 2LL) < lim;
					lo = sel(less, mid + 1LL, lo); hi = sel(less, hi, mid);
				WM_END
			worst = cast<int>(k - lo);
		WM_END
		if (readlane(wave_scan_max(worst), 63) > 900 || popc64(ballot(worst > 128)) >= 8) klass = 0;
	}
	WM_IF(ln == 0)
		wm_chain_job_t o;
		o.a_off = a_off; o.n = n; o.max_dist_x = jb.max_dist_x; o.min_dist_x = jb.min_dist_x; o.max_dist_y = jb.max_dist_y; o.bw = jb.bw;
		o.max_skip = jb.max_skip; o.max_iter = jb.max_iter; o.avg_qspan = avg; o.gap_scale = jb.gap_scale; o.is_cdna = jb.is_cdna;
		cj[j] = o;
	WM_END
	const int slot = wave_append(counts + klass);
	ust(lists + (long long)klass * n_jobs, slot, j);
}

// ------------------------------------------------------------------------------------------------------------------------------
// mm_chain_dp after the fill (src/chain.c:89-165) for one job: peak scores, chain ends, backtracking with the shared-anchor cut, compaction,
// chains ordered by their first anchor. f, p, v, t: the fill's arrays (LDS copies when G = false, the global slab when G = true; v and t are
// scratch as in the reference). zu: n uint64 of scratch (z / u), b: n anchors of scratch, wbuf: n_u <= n anchors of scratch, a: the job's
// sorted anchors. The chains (u) and their anchors go straight to the call's dense result pools; the job takes its slots with one atomic
// per pool (pool_ctr[0] chains, pool_ctr[1] anchors), so nothing has to be compacted afterwards.
// ------------------------------------------------------------------------------------------------------------------------------
// scalar pieces (lane 0)
WM_DEV void win_peaks(int n, const int *f, const int *p, int *v)             // src/chain.c:89
{
	for (int i = 0; i < n; ++i) { const int pi = p[i], fi = f[i]; v[i] = pi >= 0 && v[pi] > fi ? v[pi] : fi; }
}
WM_DEV void win_backtrack(int n_z, uint64_t *u, const int *f, const int *p, int *t, int *ord, int min_cnt, int min_sc, int *n_u_out, int *n_v_out)   // :119-135
{
	int n_v = 0, k = 0;
	for (int i = 0; i < n_z; ++i) {
		const int n_v0 = n_v, k0 = k;
		const uint64_t ui = u[i];
		int j = (int)(int32_t)(uint32_t)ui;
		do { ord[n_v++] = j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
		if (j < 0) { if (n_v - n_v0 >= min_cnt) u[k++] = ui >> 32 << 32 | (uint64_t)(uint32_t)(n_v - n_v0); }
		else if ((int32_t)(ui >> 32) - f[j] >= min_sc) { if (n_v - n_v0 >= min_cnt) u[k++] = ((ui >> 32) - (uint64_t)(int64_t)f[j]) << 32 | (uint64_t)(uint32_t)(n_v - n_v0); }
		if (k0 == k) n_v = n_v0;
	}
	*n_u_out = k; *n_v_out = n_v;
}

template <bool G> WM_DEV void win_extract_wave(int n, int min_cnt, int min_sc, const wm128_t *a_, int *f, int *p, int *v, int *t,
                                               uint64_t *zu, wm128_t *b_, wm128_t *wbuf_, int *ws, wm_win_res_t *res,
                                               uint64_t *u_pool, wm128_t *v_pool_, uint64_t *pool_ctr)
{
	const V<int> ln = lane();
	const uint64_t *a = (const uint64_t*)a_;
	uint64_t *b = (uint64_t*)b_, *wb = (uint64_t*)wbuf_, *vp = (uint64_t*)v_pool_;
	if (n <= 0) return;
	WM_LANE0_BEGIN win_peaks(n, f, p, v); WM_LANE0_END
	for (int i0 = 0; i0 < n; i0 += 64) { const V<int> i = ln + i0; WM_IF(i < n) gst(t, i, V<int>(0)); WM_END }
	win_sync<G>();
	for (int i0 = 0; i0 < n; i0 += 64) {                                      // :94-95
		const V<int> i = ln + i0;
		WM_IF(i < n) const V<int> pi = gld(p, i); WM_IF(pi >= 0) gst(t, pi, V<int>(1)); WM_END WM_END
	}
	win_sync<G>();
	int n_z = 0;                                                              // :96-110: chain ends and the peak that scores them
	for (int i0 = 0; i0 < n; i0 += 64) {
		const V<int> i = ln + i0;
		vbool is_end = i < n;
		WM_IF(i < n) is_end = gld(t, i) == 0 && gld(v, i) >= min_sc; WM_END
		V<int> j = i;
		vbool walking = is_end;
		while (any(walking)) {
			WM_IF(walking)
				const V<int> fj = gld(f, j), vj = gld(v, j);
				WM_IF(fj < vj) j = gld(p, j); WM_ELSE walking = fj != fj; WM_END
				walking = walking && j >= 0;
			WM_END
		}
		j = sel(j < 0, i, j);
		const uint64_t em = ballot(is_end);
		WM_IF(is_end)
			const V<int> slot = mbcnt(em);
			gst(zu, cast<long long>(slot + n_z), cast<uint64_t>(cast<uint32_t>(gld(f, j))) << 32 | cast<uint64_t>(cast<uint32_t>(j)));
		WM_END
		n_z += popc64(em);
	}
	win_fence();
	if (n_z == 0) { WM_IF(ln == 0) gst(&res->n_u, V<long long>(0LL), V<int>(0)); gst(&res->n_v, V<long long>(0LL), V<int>(0)); WM_END return; }
	// radix_sort_64 (:112) then reversed (:113-116): equal values are indistinguishable, so any exact sort gives the reference's array.
	// Through (z, 0) pairs in b: the 128x machinery sorts them; written back in descending order
	for (int i0 = 0; i0 < n_z; i0 += 64) {
		const V<int> i = ln + i0;
		WM_IF(i < n_z) gst(b, cast<long long>(i) * 2LL, gld(zu, cast<long long>(i))); gst(b, cast<long long>(i) * 2LL + 1LL, V<uint64_t>((uint64_t)0)); WM_END
	}
	win_fence();
	win_sort_wave<true>(b_, n_z, ws);
	win_fence();
	for (int i0 = 0; i0 < n_z; i0 += 64) {
		const V<int> i = ln + i0;
		WM_IF(i < n_z) gst(zu, cast<long long>(i), gld(b, cast<long long>(V<int>(n_z - 1) - i) * 2LL)); WM_END
	}
	for (int i0 = 0; i0 < n; i0 += 64) { const V<int> i = ln + i0; WM_IF(i < n) gst(t, i, V<int>(0)); WM_END }       // :119
	win_fence();
	win_sync<G>();
	int *cnts = ws + WIN_WS_FRAME;                                            // two ints handed from lane 0 to the wave
	WM_LANE0_BEGIN win_backtrack(n_z, zu, f, p, t, v, min_cnt, min_sc, cnts, cnts + 1); WM_LANE0_END
	win_fence();
	win_sync<false>();
	const int n_u = uniform(gld(cnts, 0LL)), n_v = uniform(gld(cnts, 1LL));
	// the job's slots in the dense result pools (what travels back to the host): one atomic per pool
	const uint64_t u_out = n_u ? wave_alloc(pool_ctr, (uint64_t)n_u) : 0, v_out = n_u ? wave_alloc(pool_ctr + 1, (uint64_t)n_v) : 0;
	WM_IF(ln == 0)
		gst(&res->n_u, V<long long>(0LL), V<int>(n_u)); gst(&res->n_v, V<long long>(0LL), V<int>(n_v));
		gst(&res->u_out, V<long long>(0LL), V<uint32_t>((uint32_t)u_out)); gst(&res->v_out, V<long long>(0LL), V<uint32_t>((uint32_t)v_out));
	WM_END
	if (n_u == 0) return;
	// :141-150: anchors of every chain in ascending order into b; w[i] = (x of the chain's first anchor, start << 32 | i)
	{
		int k0 = 0;
		for (int i = 0; i < n_u; ++i) {
			const int ni = uniform((int)(uint32_t)gld(zu, (long long)i));
			for (int j0 = 0; j0 < ni; j0 += 64) {
				const V<int> j = ln + j0;
				WM_IF(j < ni)
					const V<int> src = gld(v, V<int>(k0 + ni - 1) - j);
					gst(b, cast<long long>(j + k0) * 2LL, gld(a, cast<long long>(src) * 2LL)); gst(b, cast<long long>(j + k0) * 2LL + 1LL, gld(a, cast<long long>(src) * 2LL + 1LL));
				WM_END
			}
			WM_IF(ln == 0)
				const V<int> first = gld(v, V<int>(k0 + ni - 1));
				gst(wb, V<long long>((long long)i * 2), gld(a, cast<long long>(first) * 2LL));
				gst(wb, V<long long>((long long)i * 2 + 1), V<uint64_t>((uint64_t)k0 << 32 | (uint64_t)i));
			WM_END
			k0 += ni;
		}
	}
	win_fence();
	win_sort_wave<true>(wbuf_, n_u, ws);                                      // :155 radix_sort_128x(w): ties possible, exact permutation
	win_fence();
	{
		int k = 0;
		uint64_t *u2 = u_pool + u_out;
		vp += 2 * v_out;
		for (int i = 0; i < n_u; ++i) {                                       // :156-162 (written to the result pools instead of back over a)
			const uint64_t wy = gld(wb, (long long)i * 2 + 1);
			const int src = uniform((int)(uint32_t)wy), st = uniform((int)(wy >> 32));
			const uint64_t uj = gld(zu, (long long)src);
			const int cnt = uniform((int)(uint32_t)uj);
			WM_IF(ln == 0) gst(u2, V<long long>((long long)i), V<uint64_t>(uj)); WM_END
			for (int j0 = 0; j0 < cnt; j0 += 64) {
				const V<int> j = ln + j0;
				WM_IF(j < cnt)
					gst(vp, cast<long long>(j + k) * 2LL, gld(b, cast<long long>(j + st) * 2LL)); gst(vp, cast<long long>(j + k) * 2LL + 1LL, gld(b, cast<long long>(j + st) * 2LL + 1LL));
				WM_END
			}
			k += cnt;
		}
	}
}

// ------------------------------------------------------------------------------------------------------------------------------
// Jobs of at most W = WIN_SMALL anchors (the bulk: one MCAS window yields ~100): everything after the seed expansion — both sorts, avg_qspan, the
// chaining fill and the extraction — by ONE wavefront without leaving LDS. ga: the job's anchors as the seed kernel wrote them (global).
// lds: WIN_SMALL_LDS bytes. The results go to the call's result pools like those of win_extract_wave.
// ------------------------------------------------------------------------------------------------------------------------------
enum { WIN_SMALL = 256, WIN_WS_PAD = (WIN_WS_INTS + 3) & ~3,
       WIN_SMALL_LDS = WIN_WS_PAD * 4 + WIN_SMALL * (16 /* anchors */ + 28 /* fill window */ + 8 /* v, t */ + 8 /* z/u */ + 16 /* b */ + 16 /* w */) };
WM_DEV void win_small_wave(const wm_win_job_t jb, int n, const wm128_t *ga_, unsigned char *lds, wm_win_res_t *res, uint64_t *u_pool, wm128_t *v_pool, uint64_t *pool_ctr)
{
	const V<int> ln = lane();
	constexpr int W = WIN_SMALL;
	int *ws = (int*)lds;
	wm128_t *stage = (wm128_t*)(ws + WIN_WS_PAD);
	uint64_t *sx = (uint64_t*)(stage + W), *sy = sx + W;
	int *sf = (int*)(sy + W), *sp = sf + W, *st = sp + W, *lv = st + W, *lt = lv + W;
	uint64_t *zu = (uint64_t*)(lt + W);
	wm128_t *b = (wm128_t*)(zu + W), *wb = b + W;
	const uint64_t *ga = (const uint64_t*)ga_;
	uint64_t *la = (uint64_t*)stage;
	for (int i0 = 0; i0 < 2 * n; i0 += 64) { const V<int> i = ln + i0; WM_IF(i < 2 * n) gst(la, i, gld(ga, i)); WM_END }
	lds_sync();
	if (jb.seq_off >= 0) {                                                    // src/map.c:252, then :833 when anchors were handed in
		const int n_pre = jb.n_pre < n ? jb.n_pre : n;
		win_sort_wave<false>(stage + n_pre, n - n_pre, ws);
		if (n_pre > 0) win_sort_wave<false>(stage, n, ws);
	}
	wm_chain_job_t cj;
	cj.a_off = 0; cj.n = n; cj.max_dist_x = jb.max_dist_x; cj.min_dist_x = jb.min_dist_x; cj.max_dist_y = jb.max_dist_y; cj.bw = jb.bw;
	cj.max_skip = jb.max_skip; cj.max_iter = jb.max_iter; cj.avg_qspan = win_avg_qspan(stage, n); cj.gap_scale = jb.gap_scale; cj.is_cdna = jb.is_cdna;
	chain_wave(cj, stage, W, sx, sy, sf, sp, st, sf, sp, (int*)0);            // n <= W: nothing leaves the window; f and p stay where they are
	lds_sync();
	win_extract_wave<false>(n, jb.min_cnt, jb.min_sc, stage, sf, sp, lv, lt, zu, b, wb, ws, res, u_pool, v_pool, pool_ctr);
}

} // namespace wmk
