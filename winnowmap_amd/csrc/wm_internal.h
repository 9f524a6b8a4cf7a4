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

typedef struct {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
	int32_t n_cigar;
	int32_t bt_i, bt_j;   // backtrack start cell (-1: no backtrack)
} wm_ksw_dres_t;
