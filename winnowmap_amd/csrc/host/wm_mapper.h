// wm_mapper.h — per-read orchestration: Winnowmap2's two-stage MCAS procedure (mm_map_frag, src/map.c:279-974)
// expressed as fibers over batched device operations (wm_fiber.h). This is the replacement of the
// kt_for(worker_for) seam at src/map.c:1164: a whole mini-batch of reads is mapped at once.
#pragma once
#include "wm_core.h"
#include "wm_index.h"
#include "wm_ops.h"
#include <functional>

namespace wm {

struct ReadIn { std::string name, seq, qual, comment; };        // mm_bseq1_t, src/bseq.h
struct ReadOut {
	std::vector<Reg> regs; int rep_len = 0, frag_gap = 0;
	// the reference ASSIGNS rep_len before mm_set_mapq reads it (src/map.c:933) only where it re-collects seeds for the whole read: the rescan of the
	// stretches stage 1 left unmapped (:808-813) and the fallback (:859-861). On the pure-MCAS path (collected anchors cover the read) its rep_len is an
	// uninitialised stack word (:281) and MAPQ / rl:i are not reproducible by the reference itself; there this flag is false and we use 0.
	bool rep_len_defined = true;
};

struct MapStats {
	uint64_t n_flush = 0, n_ksw = 0, n_chain = 0, n_seed = 0, n_sketch = 0;
	uint64_t n_batches[4] = {0, 0, 0, 0};                  // batched device calls per operation (window = sketch → seed → chain in one call, seed, chain, ksw)
	double cpu_fiber = 0, cpu_op[4] = {0, 0, 0, 0}, wall_op[4] = {0, 0, 0, 0}, wall_idle = 0, cpu_help = 0;
	double wall_fiber = 0, wall_lock = 0, wall_total = 0;   // host time accounting (wm_fiber.h), seconds over all workers
	std::string internal_error;                            // first violated invariant / exception of THIS call's workers (empty: none)
};

// Maps reads[i] → out[i] (out is resized). The caller chooses the batch (the reference uses ≤ 1 Gbase mini-batches).
// n_threads > 1: the host glue of the reads runs on that many threads (a SchedTeam); device batches span the whole team.
// on_read_done (optional): called by the worker that finishes read i (out[i] is final) — e.g. to format its records while the others are still mapped.
// slot (0 .. WM_MAX_SLOTS - 1): calls with different slots may run concurrently on one DeviceOps (DeviceOps::load_reads).
void map_batch(const Index &idx, const MapOpt &opt, DeviceOps *ops, const std::vector<ReadIn> &reads, std::vector<ReadOut> &out, MapStats *stats = 0, int n_threads = 1,
               const std::function<void(size_t)> *on_read_done = 0, int slot = 0);

} // namespace wm
