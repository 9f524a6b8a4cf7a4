// oracle/ref_align_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// The reference's STATIC helpers of src/align.c that no object exports (mm_test_zdrop is inlined into mm_align1, mm_update_extra is a local
// symbol): the file is compiled once more FROM WHERE IT LIES (-I$(REFSRC); nothing is copied) inside a namespace, so that its definitions do not
// collide with align.o of the same library, and two thin C wrappers call the statics. The headers are included first, outside the namespace: what
// align.c calls in other objects (mm_idx_getseq, ksw_ll_i16, kalloc ...) binds to the reference's real objects.
//   refshim_test_zdrop   = mm_test_zdrop  (src/align.c:49-89): the z-drop / inversion verdict of a finished alignment
//   refshim_update_extra = mm_update_extra (src/align.c:240-286) incl. mm_fix_cigar (:91-167): the final CIGAR and its statistics
// Pins csrc/cigar_walk.h (host and device walk) and host/wm_align.cpp's vector scan to the reference itself.
#include <assert.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include "minimap.h"
#include "mmpriv.h"
#include "ksw2.h"
#include "kalloc.h"

extern unsigned char seq_nt4_table[256];                    // src/sketch.c
namespace refint {
#include "align.c"
// align.c:866 declares the table again INSIDE mm_align_skeleton, i.e. inside this namespace: a copy of the reference's own table, filled at load
unsigned char seq_nt4_table[256];
__attribute__((constructor)) static void copy_nt4_table() { memcpy(seq_nt4_table, ::seq_nt4_table, 256); }
}

extern "C" {

// opt fields mm_test_zdrop reads: flag, zdrop, zdrop_inv, q, e, max_gap, min_chain_score, a, min_dp_max
int refshim_test_zdrop(int64_t flag, int zdrop, int zdrop_inv, int q, int e, int max_gap, int min_chain_score, int a, int min_dp_max,
                       const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, const uint32_t *cigar, const int8_t *mat)
{
	mm_mapopt_t opt;
	mm_mapopt_init(&opt);
	opt.flag = flag; opt.zdrop = zdrop; opt.zdrop_inv = zdrop_inv; opt.q = q; opt.e = e; opt.max_gap = max_gap; opt.min_chain_score = min_chain_score; opt.a = a; opt.min_dp_max = min_dp_max;
	uint32_t *cg = (uint32_t*)malloc((n_cigar ? n_cigar : 1) * sizeof(uint32_t));
	memcpy(cg, cigar, n_cigar * sizeof(uint32_t));
	const int r = refint::mm_test_zdrop(0, &opt, qseq, tseq, n_cigar, cg, mat);
	free(cg);
	return r;
}

// in: a region's coordinates and stitched CIGAR; out6 = { blen, mlen, n_ambi, dp_max, qs, rs } after the call (mm_fix_cigar may move qs / qe / rs),
// cigar_out (cap entries) / *n_out = the CIGAR it leaves. qseq / tseq: the aligned stretches as mm_align1 passes them (src/align.c:790)
int refshim_update_extra(int rev, int qs, int qe, int rs, int re, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int q, int e,
                         uint32_t n_cigar, const uint32_t *cigar, int32_t *out6, uint32_t *cigar_out, int cap, int *n_out)
{
	mm_reg1_t r;
	memset(&r, 0, sizeof(r));
	r.rev = rev; r.qs = qs; r.qe = qe; r.rs = rs; r.re = re;
	r.p = (mm_extra_t*)calloc(1, sizeof(mm_extra_t) + (size_t)(n_cigar + 1) * sizeof(uint32_t));
	r.p->capacity = (uint32_t)((sizeof(mm_extra_t) + (size_t)(n_cigar + 1) * 4) / 4);
	r.p->n_cigar = n_cigar;
	memcpy(r.p->cigar, cigar, n_cigar * sizeof(uint32_t));
	refint::mm_update_extra(&r, qseq, tseq, mat, (int8_t)q, (int8_t)e, 0);
	out6[0] = r.blen; out6[1] = r.mlen; out6[2] = r.p->n_ambi; out6[3] = r.p->dp_max; out6[4] = rev ? r.qe : r.qs; out6[5] = r.rs;
	*n_out = (int)r.p->n_cigar;
	for (uint32_t i = 0; i < r.p->n_cigar && (int)i < cap; ++i) cigar_out[i] = r.p->cigar[i];
	free(r.p);
	return 0;
}

}
