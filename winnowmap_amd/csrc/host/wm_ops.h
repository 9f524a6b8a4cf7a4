// wm_ops.h — the batched device operations the host mapper is written against. The product wires the HIP
// implementation (GpuOps in wm_gpu_ops.cpp → libwmgpu kernels). The interface exists so that the host glue can be
// exercised by the test-suite with a checker-backed implementation (tests/host_harness) on a machine with no GPU;
// nothing in the product constructs anything but GpuOps.
#pragma once
#include "wm_core.h"

namespace wm {

struct SketchReq {                 // mm_sketch of one (sub)sequence of 0..4 codes, rid = 0
	const uint8_t *seq = 0; int len = 0;
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
	std::vector<m128> a;           // in: sorted anchors; out: anchors grouped by chain
	std::vector<uint64_t> u;       // out: score<<32 | count per chain
};

struct KswReq {                    // ksw_extd2_sse (src/ksw2.h:60)
	std::vector<uint8_t> q, t;
	int w = 0, zdrop = 0, end_bonus = 0, flag = 0;
	wm_ksw_result_t ez;            // out
	std::vector<uint32_t> cigar;   // out
};

// The batch calls are synchronous (they return when the results are in the requests) and must be callable from several threads at
// once: the hub (wm_fiber.h) keeps up to max_inflight() of them running concurrently.
struct DeviceOps {
	virtual ~DeviceOps() {}
	virtual int max_inflight() const { return 1; }
	virtual void sketch_batch(int w, int k, std::vector<SketchReq*> &reqs) = 0;
	virtual void seed_batch(std::vector<SeedReq*> &reqs) = 0;
	virtual void chain_batch(std::vector<ChainReq*> &reqs) = 0;
	virtual void ksw_batch(const wm_ksw_score_t &sc, std::vector<KswReq*> &reqs) = 0;
};

} // namespace wm
