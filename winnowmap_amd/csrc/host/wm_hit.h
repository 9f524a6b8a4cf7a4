// wm_hit.h — chain → region bookkeeping on the host (the reference keeps this on the CPU too; semantics of
// src/hit.c must match exactly because MAPQ gates the MCAS control flow). Each function cites its source.
#pragma once
#include "wm_core.h"

namespace wm {

void reg_set_coor(Reg &r, int32_t qlen, const m128 *a);                                   // mm_reg_set_coor  src/hit.c:23
std::vector<Reg> gen_regs(uint32_t hash, int qlen, int n_u, const uint64_t *u, const m128 *a);   // mm_gen_regs      :52
// mm_gen_regs + mm_set_parent + mm_select_sub in one step (identical result; Regs are built for the chains select_sub keeps only)
std::vector<Reg> gen_regs_select(uint32_t hash, int qlen, int n_u, const uint64_t *u, const m128 *a, float mask_level, int mask_len, int sub_diff, int hard_mask_level,
                                 float pri_ratio, int min_diff, int best_n);
void split_reg(Reg &r, Reg &r2, int n, int qlen, const m128 *a);                          // mm_split_reg     :106
void set_parent(float mask_level, int mask_len, std::vector<Reg> &r, int sub_diff, int hard_mask_level);   // mm_set_parent :125
void hit_sort(std::vector<Reg> &r);                                                        // mm_hit_sort      :188
int set_sam_pri(std::vector<Reg> &r);                                                      // mm_set_sam_pri   :219
void sync_regs(std::vector<Reg> &r);                                                       // mm_sync_regs     :231
void select_sub(float pri_ratio, int min_diff, int best_n, std::vector<Reg> &r);           // mm_select_sub    :255
void filter_regs(const MapOpt &opt, int qlen, std::vector<Reg> &r);                        // mm_filter_regs   :274
int squeeze_a(std::vector<Reg> &r, m128 *a);                                               // mm_squeeze_a     :295
void join_long(const MapOpt &opt, int qlen, std::vector<Reg> &r, m128 *a);                 // mm_join_long     :315
void set_mapq(std::vector<Reg> &r, int min_chain_sc, int match_sc, int rep_len, int is_sr);   // mm_set_mapq   :463

} // namespace wm
