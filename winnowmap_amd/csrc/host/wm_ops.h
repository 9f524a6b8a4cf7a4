// wm_ops.h — the batched device operations the host mapper is written against. The product wires the HIP
// implementation (GpuOps in wm_gpu.hip → libwmgpu kernels). The interface exists so that the host glue can be
// exercised by the test-suite with a checker-backed implementation (tests/host_harness) on a machine with no GPU;
// nothing in the product constructs anything but GpuOps.
#pragma once
#include "wm_core.h"
#include "../cigar_walk.h"

namespace wm {

// Sequences the mapper hands to the device exist twice: as host views (plain pointers to 0..4 codes, used by the host-side judge /
// finish code and by checker-backed implementations) and as POSITIONS in data the device already holds — the codes of the whole
// mini-batch, uploaded once by DeviceOps::load_reads, and the packed reference uploaded with the index — so that a device
// implementation never has to copy, pack or ship a sequence per request. dev_off < 0: the sequence is not resident (use the view).
struct SketchReq {                 // mm_sketch of one (sub)sequence of 0..4 codes, rid = 0
	const uint8_t *seq = 0; int len = 0;
	int64_t dev_off = -1;          // offset of seq[0] in the resident read codes
	std::vector<m128> mini;        // out
};

struct SeedReq {                   // collect_seed_hits (src/map.c:222-254): lookup, occ filter, expand, sort by x
	const m128 *mini = 0; int n_mini = 0; int qlen = 0; int max_occ = 0; int64_t flag = 0;
	std::vector<m128> a;           // out: anchors sorted with radix_sort_128x
	int rep_len = 0;               // out (src/map.c:111-116,126)
};

struct ChainReq {                  // mm_chain_dp (src/chain.c:22); consumes `a`
	int max_dist_x = 0, min_dist_x = 0, max_dist_y = 0, bw = 0, max_skip = 0, max_iter = 0, min_cnt = 0, min_sc = 0;
	float gap_scale = 1.0f;
	bool is_cdna = false;          // splice mode: an intron-sized reference gap costs min(linear, log) (src/chain.c:69-74)
	std::vector<m128> a;           // in: sorted anchors; out: anchors grouped by chain
	std::vector<uint64_t> u;       // out: score<<32 | count per chain
};

// One MCAS window or stage-2 pass from the sequence to the chains (src/map.c:69-84, 222-254, 375-430; src/chain.c:22-167): mm_sketch of
// `seq` (if any), collect_seed_hits, the handed-in anchors `pre` (stage 2: what stage 1 collected, already sorted) in front of the seeded
// ones and the union sorted again when both are present (src/map.c:818-833), then mm_chain_dp. A device implementation keeps everything
// between the codes and the chains in HBM (wm_window_batch).
struct WindowReq {
	const uint8_t *seq = 0; int len = 0;   // 0..4 codes (host view); len == 0: no sequence, only `pre` is chained
	int64_t dev_off = -1;                  // offset of seq[0] in the resident read codes, -1 = not resident (the view is staged)
	std::vector<m128> pre;                 // in
	int max_occ = 0; int64_t flag = 0;     // collect_seed_hits
	int max_dist_x = 0, min_dist_x = 0, max_dist_y = 0, bw = 0, max_skip = 0, max_iter = 0, min_cnt = 0, min_sc = 0;   // mm_chain_dp
	float gap_scale = 1.0f;
	bool is_cdna = false;
	std::vector<m128> a;                   // out: anchors grouped by chain
	std::vector<uint64_t> u;               // out: score<<32 | count per chain
	int rep_len = 0, n_anchors = 0;        // out: src/map.c:126; anchors before chaining
};

struct KswReq {                    // ksw_extd2_sse (src/ksw2.h:60)
	// host views: element i of the query is qp[i * step], of the target tp[i * step]; step = -1 for the left extension, which aligns
	// both sequences reversed (src/align.c:690-705). Valid while the request is pending.
	const uint8_t *qp = 0, *tp = 0;
	int32_t ql = 0, tl = 0, step = 1;
	// the same operands as positions in resident data. Query: index q_pos of the TWO-STRAND SPACE of the (sub)read that starts at
	// qwin_off and is qwin_len long — [0, L) forward strand, [L, 2L) reverse complement, negative = N padding; this is the layout of the
	// reference's qseq0 buffer (src/align.c:871-877), including the few bases in front of a strand that mm_align1_inv may touch.
	// Target: base t_pos of contig rid. has_n: an operand may contain an ambiguous base (conservative).
	int64_t qwin_off = -1; int32_t qwin_len = 0, q_pos = 0, rid = -1, t_pos = 0;
	bool has_n = true;
	int w = 0, zdrop = 0, end_bonus = 0, flag = 0;
	// splice mode with a junction annotation (--junc-bed): the bits of mm_idx_bed_junc (src/index.c:768-803) for the target range, in the
	// order the target is presented (reversed for the left extension, src/align.c:693-696); empty = no annotation
	std::vector<uint8_t> junc;
	// want_zd: the mapper will judge this alignment's z-drop (mm_test_zdrop, src/align.c:32-89) — a device implementation may return the scan with the
	// alignment (has_zd + zd, see cigar_walk.h); without it the host walks the CIGAR itself
	bool want_zd = false, has_zd = false;
	wm_zd_t zd = { 0, -1, -1, -1, -1 };
	wm_ksw_result_t ez = {};       // out
	std::vector<uint32_t> cigar;   // out
	int qlen() const { return ql; }
	int tlen() const { return tl; }
	bool resident() const { return qwin_off >= 0 && rid >= 0; }
	void copy_query(uint8_t *dst) const { for (int i = 0; i < ql; ++i) dst[i] = qp[(ptrdiff_t)i * step]; }
	void copy_target(uint8_t *dst) const { for (int i = 0; i < tl; ++i) dst[i] = tp[(ptrdiff_t)i * step]; }
};

// The batch calls are synchronous (they return when the results are in the requests) and must be callable from several threads at
// once: the hub (wm_fiber.h) keeps up to max_inflight() of them running concurrently.
struct DeviceOps {
	virtual ~DeviceOps() {}
	virtual int max_inflight() const { return 1; }
	// true if a thread inside a batched call mostly sleeps (waiting for the device): the mapper then runs that many workers more than cores
	virtual bool waits_asleep() const { return false; }
	// the 0..4 codes of every read of the mini-batch, back to back (SketchReq::dev_off / KswReq::qwin_off index this buffer). Returns
	// true if the implementation keeps them resident; false = requests must be served from their host views.
	// `slot` (0 or 1): two mini-batches may be mapped concurrently by two calls that share one DeviceOps (the start-up and drain phases of one
	// hide behind the other); each keeps its codes in its own slab and adds *base to its offsets. release_reads: the call is over.
	virtual bool load_reads(const uint8_t *codes, size_t n, int slot, int64_t *base) { (void)codes; (void)n; (void)slot; *base = 0; return false; }
	virtual void release_reads(int slot) { (void)slot; }
	virtual void sketch_batch(int w, int k, std::vector<SketchReq*> &reqs) = 0;
	virtual void seed_batch(std::vector<SeedReq*> &reqs) = 0;
	virtual void chain_batch(std::vector<ChainReq*> &reqs) = 0;
	virtual void ksw_batch(const wm_ksw_score_t &sc, std::vector<KswReq*> &reqs) = 0;
	// splice mode: the same requests through ksw_exts2_sse (src/align.c:326-327) — no band, no end bonus, sc.q2 = the price of an intron;
	// KswReq::flag carries the KSW_EZ_SPLICE_* bits. No junction annotation (mm_idx_bed_junc: the BED reader is outside the path).
	virtual void exts2_batch(const wm_ksw_score_t &sc, int noncan, int junc_bonus, std::vector<KswReq*> &reqs) = 0;
	// the whole window in one call. The default composes it from the three operations above (checker-backed implementations in the
	// test-suite); the product's GpuOps overrides it with the HBM-resident wm_window_batch.
	virtual void window_batch(int w, int k, std::vector<WindowReq*> &reqs);
};

} // namespace wm
