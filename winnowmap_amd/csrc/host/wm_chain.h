// wm_chain.h — chain extraction after the DP fill kernel (mm_chain_dp, src/chain.c:93-165): O(n) bookkeeping.
#pragma once
#include "wm_core.h"
namespace wm {
// f, p, v: output of the chain_wave kernel for anchors a[0..n). Produces u (score<<32|count per chain, ordered by
// the reference position of the chain's first anchor) and b (anchors grouped by chain).
void chain_extract(int64_t n, const m128 *a, const int32_t *f, const int32_t *p, const int32_t *v, int min_cnt, int min_sc,
                   std::vector<uint64_t> &u, std::vector<m128> &b);
float chain_avg_qspan(int64_t n, const m128 *a);     // src/chain.c:42-43
}
