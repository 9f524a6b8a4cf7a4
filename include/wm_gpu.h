/* include/wm_gpu.h — C-ABI of libwmgpu.so, the MI355X (gfx950) implementation of Winnowmap's
 * seed→chain→align hot path. Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference has no plugin layer: its seam is a set of link-level C functions.
 * Each entry point below names the reference function(s) it replaces (paths under /root/reference):
 *
 *   wm_ksw_batch          ← ksw_extd2_sse  src/ksw2.h:60-61  (src/ksw2_extd2_sse.c:26-393), with
 *                           ksw_backtrack / ksw_apply_zdrop  src/ksw2.h:119-176, called once per
 *                           (query,target) pair from mm_align_pair src/align.c:313-339.
 *                           Single-affine ksw_extz2_sse (src/ksw2.h:54-55) is served by the same kernel
 *                           with q2=q, e2=e (equivalence validated in tests).
 *   wm_sketch_batch       ← mm_sketch       src/mmpriv.h:61  (src/sketch.c:128-219) incl. the bloom
 *                           down-weighting of applyWeight src/sketch.c:70-89.
 *   wm_seed_chain_batch   ← collect_seed_hits src/map.c:222-254 (mm_idx_get src/index.c:88,
 *                           radix_sort_128x src/ksort.h:101-151) + mm_chain_dp src/mmpriv.h:73
 *                           (src/chain.c:22-167).
 *   wm_index_upload       ← the in-memory mm_idx_t (src/minimap.h:66-77) flattened for HBM.
 *
 * Batched forms do not exist in the reference; they are what the replacement of
 * kt_for(worker_for) at src/map.c:1164 calls. All functions return 0 on
 * success and a negative WM_E* code on failure; wm_last_error() gives a message. There is NO CPU
 * fallback: without a usable HIP device every compute entry point fails with WM_ENODEV.
 */
#ifndef WM_GPU_H
#define WM_GPU_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WM_OK        0
#define WM_ENODEV   (-1)   /* no HIP device / runtime error */
#define WM_EINVAL   (-2)   /* bad argument or unsupported parameter range */
#define WM_ENOMEM   (-3)   /* device arena too small for the request */
#define WM_EINTERNAL (-4)

typedef struct wm_ctx_s wm_ctx_t;

/* 2 x u64 record, same layout as mm128_t (src/minimap.h:55) */
typedef struct { uint64_t x, y; } wm128_t;

/* ---- context ------------------------------------------------------------------------------------ */
/* device: HIP ordinal. arena_bytes: scratch HBM reserved for traceback + batch buffers (0 = default). */
int wm_ctx_create(int device, size_t arena_bytes, wm_ctx_t **out);
void wm_ctx_destroy(wm_ctx_t *ctx);
const char *wm_last_error(void);
int wm_device_count(void);
/* time of the last batch call's kernels on the context's stream, measured with HIP events (ms) */
float wm_last_kernel_ms(const wm_ctx_t *ctx);

/* ---- ksw2 extension alignment ------------------------------------------------------------------- */
/* Scoring of one batch; mirrors the arguments of ksw_extd2_sse: match/mismatch/N scores are mat[0],
 * mat[1], mat[24] of the 5x5 matrix built by ksw_gen_simple_mat (src/align.c:9). */
typedef struct {
	int8_t match, mismatch, sc_ambi;   /* mat[0] (>0), mat[1] (<0), mat[24] (<=0; 0 means -e2) */
	int8_t q, e, q2, e2;               /* gap open / extend, two pieces (as passed to ksw_extd2_sse) */
} wm_ksw_score_t;

typedef struct {
	uint32_t q_off, t_off;             /* offsets of the 0..4-coded sequences inside `seqs` */
	int32_t qlen, tlen;
	int32_t w, zdrop, end_bonus, flag; /* exactly the ksw_extd2_sse arguments; flag = KSW_EZ_* bits */
} wm_ksw_job_t;

/* Result, same fields as ksw_extz_t (src/ksw2.h:23-32). cigar ops of job i are
 * cigar_pool[cig_off .. cig_off+n_cigar) in BAM encoding (len<<4|op), already in output order. */
typedef struct {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
	int32_t n_cigar;
	uint32_t cig_off;
} wm_ksw_result_t;

/* seqs: host buffer with all query/target codes of the batch. results[n_jobs] and cigar_pool
 * (capacity cigar_cap ops) are host buffers filled on return; *cigar_used receives the ops written.
 * Returns WM_ENOMEM if cigar_cap is too small (then *cigar_used holds the required size). */
int wm_ksw_batch(wm_ctx_t *ctx, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_job_t *jobs,
                 const uint8_t *seqs, size_t seqs_bytes,
                 wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used);

/* Same computation with the inputs already resident in HBM (device pointers from wm_dev_alloc /
 * wm_dev_upload); results stay on the device until wm_ksw_fetch. Used by bench.py so that the timed
 * region contains kernels only, and by the batched mapper which keeps reads and reference resident. */
typedef struct wm_ksw_dev_batch_s wm_ksw_dev_batch_t;
int wm_ksw_dev_prepare(wm_ctx_t *ctx, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_job_t *jobs,
                       const uint8_t *seqs, size_t seqs_bytes, wm_ksw_dev_batch_t **out);
int wm_ksw_dev_run(wm_ctx_t *ctx, wm_ksw_dev_batch_t *b);      /* launches kernels, synchronises */
int wm_ksw_dev_fetch(wm_ctx_t *ctx, wm_ksw_dev_batch_t *b, wm_ksw_result_t *results,
                     uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used);
/* algorithmic work of the batch: DP cells computed (sum over jobs of hull cells) and traceback bytes */
int wm_ksw_dev_stats(const wm_ksw_dev_batch_t *b, uint64_t *cells, uint64_t *tb_bytes, float *dp_ms, float *bt_ms);
void wm_ksw_dev_free(wm_ctx_t *ctx, wm_ksw_dev_batch_t *b);

/* Scalar drop-in with the reference's exact signature minus the kalloc handle (src/ksw2.h:60-61):
 * one alignment through the same kernels; cigar is malloc'd into *cigar_out (caller frees). */
int wm_ksw_extd2(wm_ctx_t *ctx, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m,
                 const int8_t *mat, int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus,
                 int flag, wm_ksw_result_t *ez, uint32_t **cigar_out);

#ifdef __cplusplus
}
#endif
#endif
