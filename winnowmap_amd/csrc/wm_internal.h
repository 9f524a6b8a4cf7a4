// wm_internal.h — structures shared by the host shim and the kernels (not part of the C-ABI).
#pragma once
#include <stdint.h>
#include "../../include/wm_gpu.h"

// device-side job: the public job plus where its traceback / cigar slots live
typedef struct {
	uint32_t q_off, t_off;
	int32_t qlen, tlen;
	int32_t w, zdrop, end_bonus, flag;
	uint64_t tb_off;      // byte offset of this job's traceback matrix in the arena (multiple of 16)
	int32_t n_col;        // traceback row pitch in bytes = 16 * n_col_ of the reference (ksw2_extd2_sse.c:85-86)
	int32_t klass;        // kernel class (see ksw_host.cpp)
	uint32_t cig_off;     // first op slot in the scratch cigar pool
	int32_t cig_cap;
} wm_ksw_djob_t;

// where the operands of a device job come from when they are expanded inside HBM (wm_ksw_batch_pos): parallel to the job table
typedef struct {
	int64_t qwin_off;     // first code of the (sub)read in the resident read codes
	int64_t t_base;       // global base index into the packed reference S of target element 0
	int32_t qwin_len, q_pos;
	int32_t step, pad;
} wm_ksw_dsrc_t;

typedef struct {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
	int32_t n_cigar;
	int32_t bt_i, bt_j;   // backtrack start cell (-1: no backtrack)
} wm_ksw_dres_t;

// ---- sketch ----
typedef struct {
	uint64_t seq_off;     // offset of the 0..4 codes inside the batch sequence arena
	uint64_t out_off;     // first minimizer slot of this job in the output pool
	int32_t len, cap;     // sequence length; capacity of the output slot
	uint64_t scratch_off; // first slot of this job in the per-position scratch arrays (sketch_coop)
} wm_sketch_job_t;

typedef struct {
	int32_t w, k;
	uint32_t table_bits;  // bloom geometry (0 bits never happens: the reference always allocates >= 14384)
	uint32_t salt0, salt1;
	int32_t hpc;          // homopolymer compression (MM_I_HPC, src/sketch.c:152-163): sketch_coop compacts a sequence into its runs first
} wm_sketch_params_t;

// ---- seed lookup (collect_matches + expansion of collect_seed_hits, src/map.c:97-130, 222-251) ----
typedef struct {
	uint64_t mini_off;    // first minimizer (wm128_t) of this job in the minimizer pool
	uint64_t out_off;     // first anchor slot in the anchor pool
	int32_t n_mini, qlen;
	int32_t max_occ, cap; // occurrence cut-off (mid_occ); capacity of the anchor slot
	int32_t flag, pad;    // MM_F_FOR_ONLY / MM_F_REV_ONLY bits
} wm_seed_job_t;

typedef struct { int32_t n_anchors, rep_len; } wm_seed_res_t;

typedef struct {          // flat index view in HBM (host/wm_index.h)
	const uint64_t *hkey, *hval, *P;
	int32_t hbits, pad;
} wm_index_view_t;

// ---- chain DP fill (mm_chain_dp, src/chain.c:45-90) ----
typedef struct {
	uint64_t a_off;       // first anchor of this job in the anchor pool (sorted by x)
	int32_t n, max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter;
	float avg_qspan, gap_scale;
	int32_t is_cdna;      // splice mode: the gap cost of src/chain.c:69-74
} wm_chain_job_t;

// ---- one MCAS window / stage-2 pass on the device: sketch → seed → sort → chain → extraction (window_kernel.h) ----
typedef struct {
	int64_t seq_off;      // first 0..4 code relative to the call's sequence base pointer (as wm_sketch_job_t::seq_off); < 0: no sequence, only handed-in anchors
	uint64_t pre_off;     // first handed-in anchor in the call's `pre` pool
	int32_t len, n_pre;
	int32_t max_occ, seed_flag;                                                        // collect_seed_hits: mid_occ, MM_F_FOR_ONLY / MM_F_REV_ONLY bits
	int32_t max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc;   // mm_chain_dp
	float gap_scale;
	int32_t is_cdna;
} wm_win_job_t;

typedef struct {          // per job on the device: where its anchors live and what came out
	uint64_t a_off;       // first slot of the job's region in the anchor pool (x 4 ints: in the f|p|v|t slab; x 2 uint64: in the u scratch)
	int32_t n_a;          // anchors before chaining (handed in + seeded)
	int32_t rep_len, n_mini;
	int32_t n_u, n_v;     // chains; anchors kept
	int32_t err;          // 1: the minimizer slot overflowed (retry with full-size slots), 2: the anchor pool overflowed
	uint32_t u_out, v_out;    // offsets in the dense output pools (after the scan)
} wm_win_res_t;
