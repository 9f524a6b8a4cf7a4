// wm_align.h — base-level alignment driver on the host: decides WHAT to align (region arithmetic, bad-seed
// filtering, CIGAR stitching, z-drop / inversion tests) and hands every DP to the ksw kernels in batches.
// Restates mm_align_skeleton / mm_align1 / helpers (src/align.c) with all ksw2 calls of a round issued together.
#pragma once
#include "wm_core.h"
#include "wm_index.h"
#include "wm_hit.h"
#include "wm_fiber.h"

namespace wm {

// local SW score used by the inversion test: ksw_ll_qinit + ksw_ll_i16 (src/ksw2_ll_sse.c:32-147), exact incl. ties
int ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat5x5, int gapo, int gape, int *qe, int *te);

// mm_align_skeleton (src/align.c:864-920). qcodes: the query as 0..4 codes; q_dev_off: offset of qcodes[0] in the read codes the
// device holds (DeviceOps::load_reads), or -1. Runs inside a fiber of `sch`.
void align_skeleton(Scheduler &sch, const MapOpt &opt, const Index &idx, int qlen, const uint8_t *qcodes, int64_t q_dev_off, std::vector<Reg> &regs, m128 *a);

} // namespace wm
