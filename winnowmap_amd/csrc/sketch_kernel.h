// sketch_kernel.h — weighted robust-winnowing minimizers (mm_sketch, src/sketch.c:128-219) on gfx950. Homopolymer compression (:152-163) in the
// one-wavefront-per-sequence kernel only (sketch_coop: sketch_hpc_steps compacts the sequence into its runs, the two phases then run over the runs).
//
// Mapping: ONE LANE per (sub)sequence — stage 1 of Winnowmap2 sketches ~9 windows per read, so a mini-batch holds
// 10^4..10^6 independent sequences and the machine fills with thread-level parallelism; the winnowing automaton
// itself is inherently sequential (the choice among equal-order k-mers depends on history) and is
// reproduced step by step. Per lane and window slot the ring keeps the fp64 order and (pos<<1|strand); the 2k-bit
// key of an emitted minimizer is rebuilt from the k bases ending at its position, so the ring needs 12 B/slot/lane
// (w=50: 38 KB of LDS per wave, slot-major so that lanes hit distinct banks).
//
// fp64 order: x = (double)fmix64(kmer) * 2^-64, -x or -(x^8 by three squarings) when the bloom filter holds the
// k-mer (src/sketch.c:70-89). Compiled with -ffp-contract=off; there is no add, so nothing can fuse anyway.
#pragma once
#ifndef WM_DEV
#error "include simt.h before sketch_kernel.h"
#endif
#include "wm_internal.h"
#include "reads2bit.h"

namespace wmk {
using namespace simt;

WM_DEV V<uint64_t> sk_hash64(V<uint64_t> key, uint64_t mask)
{   // src/sketch.c:53-63
	key = (~key + (key << 21)) & mask;
	key = key ^ (key >> 24);
	key = (key + (key << 3) + (key << 8)) & mask;
	key = key ^ (key >> 14);
	key = (key + (key << 2) + (key << 4)) & mask;
	key = key ^ (key >> 28);
	key = (key + (key << 31)) & mask;
	return key;
}
WM_DEV V<uint64_t> sk_fmix64(V<uint64_t> h)
{   // src/sketch.c:43-51
	h = h ^ (h >> 33); h = h * (uint64_t)0xff51afd7ed558ccdULL;
	h = h ^ (h >> 33); h = h * (uint64_t)0xc4ceb9fe1a85ec53ULL;
	h = h ^ (h >> 33);
	return h;
}
WM_DEV V<uint32_t> sk_bloom_hash(V<uint64_t> key, uint32_t salt)
{   // bloom_filter.hpp hash_ap, one 8-byte round
	V<uint32_t> lo = cast<uint32_t>(key), hi = cast<uint32_t>(key >> 32), h = salt;
	h = h ^ ((h << 7) ^ (lo * (h >> 3)) ^ (~((h << 11) + (hi ^ (h >> 5)))));
	return h;
}

// jobs[j]: sequence of 0..4 codes at seqs+seq_off (len bytes); minimizers go to out[out_off .. out_off+cap) and
// counts[j] receives how many the reference would emit (may exceed cap: the host then retries with a larger slot).
// ring_o / ring_y: w*64 entries each (LDS, or global scratch for very large w).
// a sequence's 0..4 code: byte i of the staged bytes at seqs + soff, or base i of the resident packed reads when soff carries WM_RD_PACKED_BIT (per lane)
WM_DEV V<int> sk_code(const uint8_t *seqs, const uint64_t *pk, const uint64_t *nm, V<long long> soff, V<long long> i)
{
	V<int> c = 0;
	const vbool packed = (soff & (long long)WM_RD_PACKED_BIT) != 0LL;
	WM_IF(packed) c = rd2_code(pk, nm, (soff & (long long)(WM_RD_PACKED_BIT - 1)) + i); WM_ELSE c = cast<int>(gld(seqs, soff + i)); WM_END
	return c;
}

WM_DEV void sketch_wave(const wm_sketch_params_t P, const wm_sketch_job_t *jobs, int n_jobs, int wave_id, const uint8_t *seqs, const uint64_t *pk, const uint64_t *nm,
                        const uint8_t *bloom_bits, double *ring_o, uint32_t *ring_y, wm128_t *out, int *counts)
{
	const int w = P.w, k = P.k;
	const uint64_t mask = (1ULL << 2 * k) - 1, top = 2ULL * (uint64_t)(k - 1);
	const V<int> ln = lane();
	const V<int> job = ln + wave_id * 64;
	const vbool have = job < n_jobs;
	V<int> len = 0, cap = 0;
	V<long long> soff = 0, ooff = 0;
	WM_IF(have)
		len = gld(&jobs[0].len, job * (int)(sizeof(wm_sketch_job_t) / 4));
		cap = gld(&jobs[0].cap, job * (int)(sizeof(wm_sketch_job_t) / 4));
		soff = cast<long long>(gld(&jobs[0].seq_off, job * (int)(sizeof(wm_sketch_job_t) / 8)));
		ooff = cast<long long>(gld(&jobs[0].out_off, job * (int)(sizeof(wm_sketch_job_t) / 8)));
	WM_END
	// uniform trip count = longest sequence in this wave
	int max_len = 0;
	for (int l = 0; l < 64; ++l) { const int v = readlane(len, l); max_len = v > max_len ? v : max_len; }

	V<uint64_t> fw = (uint64_t)0, rc = (uint64_t)0;
	V<int> run = 0, slot = 0, min_slot = 0, n_out = 0;
	V<double> min_o = 2.0;
	V<int> min_y = -1;                         // pos<<1|strand of the current minimum, -1 = none
	for (int j = 0; j < w; ++j) { gst(ring_o, ln + j * 64, V<double>(2.0)); gst(ring_y, ln + j * 64, V<uint32_t>(0xffffffffu)); }

	for (int i = 0; i < max_len; ++i) {
		WM_IF(have && len > i)
			V<int> c = sk_code(seqs, pk, nm, soff, V<long long>((long long)i));
			V<double> co = 2.0;
			V<int> cy = -1;
			vbool skip = c < 0;                // false
			WM_IF(c < 4)
				fw = ((fw << 2) | cast<uint64_t>(c)) & mask;
				rc = (rc >> 2) | ((cast<uint64_t>(c) ^ (uint64_t)3) << (int)top);
				skip = fw == rc;               // strand-ambiguous k-mer: the whole step is skipped (src/sketch.c:166)
				WM_IF(!skip)
					V<int> strand = sel(fw < rc, 0, 1);
					run = run + 1;
					WM_IF(run >= k)
						V<uint64_t> km = sel(strand == 1, rc, fw);
						// bloom probe: two hashes, bit = h % table_bits (32-bit: table_bits < 2^32 checked by the host)
						V<uint32_t> h0 = sk_bloom_hash(km, P.salt0) % P.table_bits, h1 = sk_bloom_hash(km, P.salt1) % P.table_bits;
						V<int> b0 = cast<int>(gld(bloom_bits, h0 >> 3)) >> cast<int>(h0 & 7u), b1 = cast<int>(gld(bloom_bits, h1 >> 3)) >> cast<int>(h1 & 7u);
						vbool down = ((b0 & b1) & 1) == 1;
						V<double> x = cast<double>(sk_fmix64(km)) * 1.0 / 18446744073709551616.0;
						V<double> x2 = x * x, x4 = x2 * x2;
						co = sel(down, -1.0 * (x4 * x4), -1.0 * x);
						cy = (V<int>(i) << 1) | strand;
					WM_END
				WM_END
			WM_ELSE
				run = 0;
			WM_END
			WM_IF(!skip)
				gst(ring_o, ln + slot * 64, co);
				gst(ring_y, ln + slot * 64, cast<uint32_t>(cy));
				vbool better = co < min_o, leaving = !better && (slot == min_slot);
				vbool emit = min_y >= 0 && ((better && run >= w + k) || (leaving && run >= w + k - 1));
				WM_IF(emit)                    // rebuild the key of the minimum from the k bases that end at its position
					V<int> pos = min_y >> 1, strand = min_y & 1;
					V<uint64_t> f2 = (uint64_t)0, r2 = (uint64_t)0;
					for (int t = 0; t < k; ++t) {
						V<uint64_t> cc = cast<uint64_t>(sk_code(seqs, pk, nm, soff, cast<long long>(pos - (k - 1) + t)));
						f2 = ((f2 << 2) | cc) & mask;
						r2 = (r2 >> 2) | ((cc ^ (uint64_t)3) << (int)top);
					}
					V<uint64_t> km = sel(strand == 1, r2, f2);
					WM_IF(n_out < cap)
						V<long long> o = (ooff + cast<long long>(n_out)) * 2LL;
						gst((uint64_t*)out, o, (sk_hash64(km, mask) << 8) | (uint64_t)k);
						gst((uint64_t*)out, o + 1LL, cast<uint64_t>(min_y));      // rid = 0; the host adds rid<<32 where needed
					WM_END
					n_out = n_out + 1;
				WM_END
				WM_IF(better)
					min_o = co; min_y = cy; min_slot = slot;
				WM_END
				WM_IF(leaving)                 // rescan the window oldest → newest; >= keeps the newest of equal orders
					min_o = 2.0; min_y = -1;
					for (int n = 1; n <= w; ++n) {
						V<int> jj = slot + n;
						jj = sel(jj >= w, jj - w, jj);
						V<double> o = gld(ring_o, ln + jj * 64);
						vbool take = min_o >= o;
						min_o = sel(take, o, min_o);
						min_y = sel(take, cast<int>(gld(ring_y, ln + jj * 64)), min_y);
						min_slot = sel(take, jj, min_slot);
					}
				WM_END
				slot = slot + 1;
				slot = sel(slot == w, 0, slot);
			WM_END
		WM_END
	}
	WM_IF(have && min_y >= 0)              // flush the last minimum (src/sketch.c:208-214)
		V<int> pos = min_y >> 1, strand = min_y & 1;
		V<uint64_t> f2 = (uint64_t)0, r2 = (uint64_t)0;
		for (int t = 0; t < k; ++t) {
			V<uint64_t> cc = cast<uint64_t>(sk_code(seqs, pk, nm, soff, cast<long long>(pos - (k - 1) + t)));
			f2 = ((f2 << 2) | cc) & mask;
			r2 = (r2 >> 2) | ((cc ^ (uint64_t)3) << (int)top);
		}
		V<uint64_t> km = sel(strand == 1, r2, f2);
		WM_IF(n_out < cap)
			V<long long> o = (ooff + cast<long long>(n_out)) * 2LL;
			gst((uint64_t*)out, o, (sk_hash64(km, mask) << 8) | (uint64_t)k);
			gst((uint64_t*)out, o + 1LL, cast<uint64_t>(min_y));
		WM_END
		n_out = n_out + 1;
	WM_END
	WM_IF(have) gst(counts, job, n_out); WM_END
}

// ------------------------------------------------------------------------------------------------------------------------------
// sketch_coop: ONE WAVEFRONT per sequence, for odd k (every preset: k = 15 / 19). With k odd a k-mer never equals its reverse
// complement, so the reference's "skip the palindrome" step (src/sketch.c:166) never fires, slot t of the winnowing ring is simply
// position t, and a k-mer is valid iff the last k bases are all unambiguous (l >= k; an N only resets l, src/sketch.c:175).
//
// Phase 1 (data parallel, 64 positions per step, coalesced byte loads): lane j builds the forward / reverse k-mers that end at its
// position from the k codes before it, picks the strand, probes the bloom filter, evaluates the fp64 order (applyWeight, :70-89) and
// stores (order, hash << 8 | k, pos << 1 | strand, l) of the slot — 24 B per position in a scratch slab in HBM (L2 resident).
// Phase 2 walks the CHAIN OF WINDOW MINIMA instead of every position. If slot m holds the current minimum after step m, the
// reference's automaton (:180-205) changes it next either at the first t in (m, m + w] whose order is STRICTLY smaller (a new minimum:
// emit m if l_t >= w + k) or, if there is none, at t = m + w when m is overwritten (emit m if l_t >= w + k - 1; the new minimum is the
// RIGHTMOST smallest slot of (m, m + w], the ">=" rescan of :199-204). Both questions are one ballot over the w orders that follow m,
// so a hop costs a few dozen instructions and advances ~(w + 1) / 2 positions; ties between identical k-mers (tandem repeats) and
// empty windows (N runs, masked reads) follow the same two rules, so the result is the reference's bit for bit. After the last
// position the current minimum is flushed (:208-214).
// Slot 0 is always empty for k >= 2 (l = 1 < k), which makes "m = 0, empty" the state after step 0.
// ------------------------------------------------------------------------------------------------------------------------------
// ---- the two phases over a RANGE of the sequence (round 5): a long sequence — a contig of the reference at index time, a 5-Mb query contig, the
// stage-2 pass of a long read — is cut into chunks and every chunk gets a wavefront of its own (sketch_long_* kernels, wm_gpu.hip) instead of one
// wavefront walking 10^6..10^8 positions (a 125-Mb contig: 34 s). Phase 1 is positional, so a chunk only has to know where the last ambiguous base
// before it lies (as far as the comparisons with w + k can see). Phase 2's state after step t is (m, order[m]); a chunk may start wherever that state
// is KNOWN without the history: at a position t whose order is strictly smaller than every order of the w - 1 positions before it (and is a real
// k-mer) the automaton holds m = t after step t whatever happened before — a smaller order replaces any minimum (:180-190), and if the minimum is
// being overwritten in that very step the rescan of (t - w, t] finds t as its only smallest slot (:199-204). sketch_find_sync finds the first such
// position of a chunk; the wavefront of the chunk before runs up to and including the event of that step (it emits the minimum that t replaces) and
// stops, the chunk's own wavefront starts from (m = t) without emitting. A chunk without such a position (a long low-complexity stretch) is simply
// absorbed by its predecessor. The chunks' minimizers concatenate to the sequence's.

// phase 1 for positions [begin, end) of a sequence of n codes: bytes at seqs + soff, or (PACKED) bases soff .. of the resident packed reads (reads2bit.h),
// where the k bases that end at a position are one shifted 64-bit window instead of k byte loads
// HPC: the "sequence" is the list of the automaton's steps (sketch_hpc_steps: seqs[s] = the code of step s, he[s] = the position of its last base):
// a k-mer's position is he[s], its span he[s] - he[s - k] — the k runs are contiguous — and k-mers that span 256 bases or more are not used (:168)
template <bool PACKED, bool HPC = false>
WM_DEV void sketch_p1_range_t(const wm_sketch_params_t P, long long soff, int n, const uint8_t *seqs, const uint64_t *pk, const uint64_t *nm, const uint8_t *bloom_bits,
                              double *so, uint64_t *sx, uint32_t *sy, uint32_t *sl, int begin, int end, const uint32_t *he = 0)
{
	const int w = P.w, k = P.k;
	const uint64_t mask = (1ULL << 2 * k) - 1;
	const V<int> ln = lane();
	// position of the last ambiguous base before `begin`, as far back as l = i - last_n is ever compared with (w + k): beyond that any value does
	int last_n = -1;
	if (begin > 0) {
		const int lo = begin - (w + k + 2) > 0 ? begin - (w + k + 2) : 0;
		last_n = lo - 1;                                       // (nothing ambiguous in [lo, begin): l >= w + k + 2 at `begin`, or exact when lo == 0)
		for (int t0 = lo; t0 < begin; t0 += 64) {
			const V<int> i = ln + t0;
			V<int> hit = -1;
			WM_IF(i < begin)
				vbool amb = false;
				if constexpr (PACKED) amb = rd2_is_n(nm, cast<long long>(i) + soff); else amb = cast<int>(gld(seqs, cast<long long>(i) + soff)) >= 4;
				WM_IF(amb) hit = i; WM_END
			WM_END
			const int mx = readlane(wave_scan_max(hit), 63);
			last_n = mx > last_n ? mx : last_n;
		}
	}
	for (int t0 = begin; t0 < end; t0 += 64) {
		const V<int> i = ln + t0;
		const vbool in = i < end;
		vbool amb = false;
		WM_IF(in)
			if constexpr (PACKED) amb = rd2_is_n(nm, cast<long long>(i) + soff); else amb = cast<int>(gld(seqs, cast<long long>(i) + soff)) >= 4;
		WM_END
		const V<int> lastN = vmax(wave_scan_max(sel(in && amb, i, V<int>(-1))), last_n);
		const V<int> l = i - lastN;                  // unambiguous bases ending here (0 on an N)
		last_n = readlane(lastN, 63);
		vbool valid = in && l >= k;
		V<int> span = k, epos = i;
		if constexpr (HPC) {
			WM_IF(valid)
				epos = cast<int>(gld(he, i));
				V<int> before = -1;
				WM_IF(i >= k) before = cast<int>(gld(he, i - k)); WM_END
				span = epos - before;
			WM_END
			valid = valid && span < 256;
		}
		V<double> co = 2.0;
		V<uint64_t> cx = ~(uint64_t)0;
		V<uint32_t> cy = 0xffffffffu;
		WM_IF(valid)
			V<uint64_t> f = (uint64_t)0, g = (uint64_t)0;
			if constexpr (PACKED) {                  // the window of the k bases i-k+1 .. i, earliest base in the low bits: its complement IS the reverse-complement k-mer
				const V<uint64_t> win = rd2_window(pk, cast<long long>(i - (k - 1)) + soff);
				g = ~win & mask;
				f = rd2_rev(win) >> (64 - 2 * k);
			} else {
				for (int j = 0; j < k; ++j) {        // base j steps back: digit j of the forward k-mer, digit k-1-j of the reverse complement
					const V<uint64_t> cj = cast<uint64_t>(gld(seqs, cast<long long>(i - j) + soff));
					f = f | (cj << (2 * j));
					g = g | ((cj ^ (uint64_t)3) << (2 * (k - 1 - j)));
				}
			}
			const V<int> strand = sel(f < g, 0, 1);
			const V<uint64_t> km = sel(strand == 1, g, f);
			const V<uint32_t> h0 = sk_bloom_hash(km, P.salt0) % P.table_bits, h1 = sk_bloom_hash(km, P.salt1) % P.table_bits;
			const V<int> b0 = cast<int>(gld(bloom_bits, h0 >> 3)) >> cast<int>(h0 & 7u), b1 = cast<int>(gld(bloom_bits, h1 >> 3)) >> cast<int>(h1 & 7u);
			const vbool down = ((b0 & b1) & 1) == 1;
			const V<double> x = cast<double>(sk_fmix64(km)) * 1.0 / 18446744073709551616.0;
			const V<double> x2 = x * x, x4 = x2 * x2;
			co = sel(down, -1.0 * (x4 * x4), -1.0 * x);
			cx = (sk_hash64(km, mask) << 8) | cast<uint64_t>(span);
			cy = cast<uint32_t>((epos << 1) | strand);
		WM_END
		WM_IF(in)
			gst(so, i, co); gst(sx, i, cx); gst(sy, i, cy); gst(sl, i, cast<uint32_t>(l));
		WM_END
	}
}

// soff: a byte offset into seqs, or WM_RD_PACKED_BIT | base index into the resident packed reads (uniform per call)
WM_DEV void sketch_p1_range(const wm_sketch_params_t P, long long soff, int n, const uint8_t *seqs, const uint64_t *pk, const uint64_t *nm, const uint8_t *bloom_bits,
                            double *so, uint64_t *sx, uint32_t *sy, uint32_t *sl, int begin, int end)
{
	if (soff & (long long)WM_RD_PACKED_BIT) sketch_p1_range_t<true>(P, soff & (long long)(WM_RD_PACKED_BIT - 1), n, seqs, pk, nm, bloom_bits, so, sx, sy, sl, begin, end);
	else sketch_p1_range_t<false>(P, soff, n, seqs, pk, nm, bloom_bits, so, sx, sy, sl, begin, end);
}

// the first position t of [from, to) that holds a real k-mer whose order is strictly smaller than the orders of the (up to) w - 1 positions before it
// (uniform; -1: none). so: phase-1 orders of the whole sequence (positions < from are read: they belong to the chunk before, written by phase 1)
WM_DEV int sketch_find_sync(int w, const double *so, int from, int to)
{
	const V<int> ln = lane();
	for (int t0 = from; t0 < to; t0 += 64) {
		const V<int> t = ln + t0;
		const vbool in = t < to;
		V<double> o = 2.0;
		WM_IF(in) o = gld(so, t); WM_END
		vbool ok = in && o < 2.0;
		for (int j = 1; j < w && any(ok); ++j) {
			WM_IF(ok)
				const V<int> q = t - j;
				WM_IF(q >= 0) ok = o < gld(so, q); WM_END
			WM_END
		}
		const uint64_t bm = ballot(ok);
		if (bm) return t0 + __builtin_ctzll(bm);
	}
	return -1;
}

// phase 2 from the state (m0, set0) — (0, false) at the start of a sequence, (t, true) at a sync position t — up to and including the event after
// which the minimum is t_stop (-1: to the end of the sequence, where the last minimum is flushed, :208-214). Minimizers go to out[0 .. cap); returns
// how many the reference emits over the stretch (may exceed cap).
WM_DEV int sketch_p2_range(const wm_sketch_params_t P, int n, const double *so, const uint64_t *sx, const uint32_t *sy, const uint32_t *sl,
                           int m0, bool set0, int t_stop, wm128_t *out, int cap)
{
	const int w = P.w, k = P.k;
	const V<int> ln = lane();
	int m = m0, n_out = 0;
	double om = set0 ? gld(so, (long long)m0) : 2.0;
	bool m_set = set0;                               // the current minimum is a real k-mer
	for (;;) {
		int found = -1;
		for (int b = m + 1; b <= m + w && b < n && found < 0; b += 64) {
			const V<int> t = ln + b;
			const vbool inr = t <= m + w && t < n;
			V<double> o = 2.0;
			WM_IF(inr) o = gld(so, t); WM_END
			const uint64_t bm = ballot(inr && o < om);
			if (bm) found = b + __builtin_ctzll(bm);
		}
		if (found >= 0) {                            // a strictly smaller order arrives at step `found` (:180-190)
			if (m_set && (int)gld(sl, (long long)found) >= w + k) {
				WM_IF(ln == 0 && n_out < cap)
					gst((uint64_t*)out, V<long long>((long long)n_out * 2), V<uint64_t>(gld(sx, (long long)m)));
					gst((uint64_t*)out, V<long long>((long long)n_out * 2 + 1), V<uint64_t>((uint64_t)gld(sy, (long long)m)));
				WM_END
				++n_out;
			}
			m = found; om = gld(so, (long long)m); m_set = true;
			if (m == t_stop) return n_out;           // the next chunk starts from here
			continue;
		}
		if (m + w > n - 1) break;                    // the minimum is never overwritten: flushed below
		{
			const int t = m + w;                     // slot m is overwritten at step m + w (:191-205)
			if (m_set && (int)gld(sl, (long long)t) >= w + k - 1) {
				WM_IF(ln == 0 && n_out < cap)
					gst((uint64_t*)out, V<long long>((long long)n_out * 2), V<uint64_t>(gld(sx, (long long)m)));
					gst((uint64_t*)out, V<long long>((long long)n_out * 2 + 1), V<uint64_t>((uint64_t)gld(sy, (long long)m)));
				WM_END
				++n_out;
			}
			// the rightmost smallest order of (m, m + w]: per 64 slots, descend from the rightmost slot to ever smaller orders
			double best = 3.0; int best_t = m + 1;
			for (int b = m + 1; b <= t; b += 64) {
				const V<int> tt = ln + b;
				const vbool inr = tt <= t;
				V<double> o = 2.0;
				WM_IF(inr) o = gld(so, tt); WM_END
				uint64_t cand = ballot(inr);
				int p = 63 - __builtin_clzll(cand);
				double v = readlane(o, p);
				for (;;) {
					const uint64_t lower = ballot(inr && o < v);
					if (!lower) break;
					p = 63 - __builtin_clzll(lower);
					v = readlane(o, p);
				}
				if (v <= best) { best = v; best_t = b + p; }   // (a later block wins ties)
			}
			m = best_t; om = best; m_set = om < 2.0;
			if (m == t_stop) return n_out;
		}
	}
	WM_EMU_ASSERT(t_stop < 0);                       // (a sync position ahead is always reached: see the header)
	if (m_set) {                                     // flush (:208-214)
		WM_IF(ln == 0 && n_out < cap)
			gst((uint64_t*)out, V<long long>((long long)n_out * 2), V<uint64_t>(gld(sx, (long long)m)));
			gst((uint64_t*)out, V<long long>((long long)n_out * 2 + 1), V<uint64_t>((uint64_t)gld(sy, (long long)m)));
		WM_END
		++n_out;
	}
	return n_out;
}

// homopolymer compression (src/sketch.c:152-163) as a compaction: the STEPS of the reference's automaton in order — every run of one unambiguous base, every
// ambiguous base on its own (:175) — hc[s] = the step's code, he[s] = the position of its last base (where :159 leaves i). Returns the number of steps.
WM_DEV int sketch_hpc_steps(long long soff, int n, const uint8_t *seqs, const uint64_t *pk, const uint64_t *nm, uint8_t *hc, uint32_t *he)
{
	const V<int> ln = lane();
	int S = 0;
	for (int t0 = 0; t0 < n; t0 += 64) {
		const V<int> i = ln + t0;
		const vbool in = i < n;
		V<int> c = 4, cn = 5;
		WM_IF(in)
			c = sk_code(seqs, pk, nm, V<long long>(soff), cast<long long>(i));
			WM_IF(i + 1 < n) cn = sk_code(seqs, pk, nm, V<long long>(soff), cast<long long>(i + 1)); WM_END
		WM_END
		const vbool last = in && (c >= 4 || cn != c);
		const uint64_t bm = ballot(last);
		const V<int> at = mbcnt(bm) + S;
		WM_IF(last) gst(hc, at, cast<uint8_t>(c)); gst(he, at, cast<uint32_t>(i)); WM_END
		S += popc64(bm);
	}
	return S;
}

// hc / he: scratch of jb.len entries each for the compacted sequence (P.hpc only)
WM_DEV void sketch_coop(const wm_sketch_params_t P, const wm_sketch_job_t jb, const uint8_t *seqs, const uint64_t *pk, const uint64_t *nm, const uint8_t *bloom_bits,
                        double *so, uint64_t *sx, uint32_t *sy, uint32_t *sl, wm128_t *out, int *count_out, uint8_t *hc = 0, uint32_t *he = 0)
{
	if (P.hpc) {
		const int S = sketch_hpc_steps((long long)jb.seq_off, jb.len, seqs, pk, nm, hc, he);
		mem_sync();
		sketch_p1_range_t<false, true>(P, 0, S, hc, pk, nm, bloom_bits, so, sx, sy, sl, 0, S, he);
		mem_sync();
		const int n_hpc = sketch_p2_range(P, S, so, sx, sy, sl, 0, false, -1, out + jb.out_off, jb.cap);
		WM_IF(lane() == 0) gst(count_out, V<long long>(0), V<int>(n_hpc)); WM_END
		return;
	}
	const int n = jb.len;
	sketch_p1_range(P, (long long)jb.seq_off, n, seqs, pk, nm, bloom_bits, so, sx, sy, sl, 0, n);
	mem_sync();                                      // phase 2 reads what other lanes of this wave wrote (same CU: a workgroup-scope fence; no L2 write-back per job)
	const int n_out = sketch_p2_range(P, n, so, sx, sy, sl, 0, false, -1, out + jb.out_off, jb.cap);
	WM_IF(lane() == 0) gst(count_out, V<long long>(0), V<int>(n_out)); WM_END
}

} // namespace wmk
