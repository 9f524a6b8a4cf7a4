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
 *   wm_seed_batch         ← collect_seed_hits src/map.c:222-254 (mm_idx_get src/index.c:88,
 *                           radix_sort_128x src/ksort.h:101-151)
 *   wm_chain_batch        ← mm_chain_dp src/mmpriv.h:73 (src/chain.c:22-167)
 *   wm_window_batch       ← the three above back to back for one MCAS window / stage-2 pass (src/map.c:69-84, 222-254, 375-430),
 *                           resident in HBM from the read codes to the chains: what wm_map_reads uses
 *   wm_map_reads          ← kt_for(worker_for) → mm_map_frag, src/map.c:1164, 1008, 279-974
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
/* the kernel-variant defines (WM_KERNEL_DEFINES of winnowmap_amd/build.py) this library was compiled with; "" for the default build */
const char *wm_build_defines(void);
int wm_device_count(void);
/* time of the last batch call's kernels on the context's stream, measured with HIP events (ms) */
float wm_last_kernel_ms(const wm_ctx_t *ctx);
/* the HIP device a context lives on (what wm_ctx_create was given); -1 for NULL */
int wm_ctx_device(const wm_ctx_t *ctx);

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

/* The same batch with the operands given as POSITIONS in data the device already holds — the packed reference (wm_index_upload:
 * the S array of mm_idx_t, 4 bits per base, src/index.c:329-335, read like mm_idx_getseq, src/index.c:161-171) and the 0..4 codes of the
 * reads of the current mini-batch (wm_reads_upload) — instead of bytes: nothing is unpacked, copied or shipped per alignment; a
 * kernel expands the operands inside HBM. Query element i = index q_pos + i*step of the TWO-STRAND SPACE of the (sub)read that starts
 * at code qwin_off and is qwin_len long: [0, L) forward strand, [L, 2L) reverse complement, outside = N — the layout of the reference's
 * qseq0 buffer (src/align.c:871-877). Target element i = base t_pos + i*step of contig rid. step = -1 aligns both operands back to
 * front (the left extension, src/align.c:690-705). has_n: 0 promises that neither operand holds an ambiguous base. */
typedef struct {
	int64_t qwin_off;
	int32_t qwin_len, q_pos;
	int32_t rid, t_pos;
	int32_t qlen, tlen;
	int32_t w, zdrop, end_bonus, flag;
	int8_t step, has_n, pad[6];
} wm_ksw_pos_t;
/* codes: the 0..4 codes (seq_nt4_table) of the mini-batch's reads back to back; they are packed on the host to 2 bits per base + 1 ambiguity bit per base
 * (0.375 B per base across PCIe and in HBM) — offsets into them stay BASE indices. */
int wm_reads_upload(wm_ctx_t *ctx, const uint8_t *codes, size_t n);
int wm_ksw_batch_pos(wm_ctx_t *ctx, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_pos_t *jobs,
                     wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used);
/* The same call, which also runs the scan of mm_test_zdrop + update_max_zdrop (src/align.c:32-66) on the device over the finished alignment of every
 * job whose flag carries WM_KSW_F_ZDWALK (the gap fills, src/align.c:736; forward jobs only): zd[i] = the largest z-drop along the CIGAR
 * (max - score - |di - dj| * e with the caller's q / e) and the target / query span between its maximum and its end — pos[0][0..1], pos[1][0..1] of
 * src/align.c:51; {0, -1, -1, -1, -1} for the other jobs. What is left for the host of mm_test_zdrop is the verdict (the inversion test by ksw_ll_i16 on
 * the few alignments whose drop exceeds zdrop_inv, src/align.c:68-89). The walk is one function for the host and the device: csrc/cigar_walk.h. */
#define WM_KSW_F_ZDWALK 0x10000
typedef struct { int32_t max_zdrop, t0, t1, q0, q1; } wm_zd_t;
int wm_ksw_batch_pos_zd(wm_ctx_t *ctx, const wm_ksw_score_t *sc, int n_jobs, const wm_ksw_pos_t *jobs,
                        wm_ksw_result_t *results, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used, wm_zd_t *zd);

/* diagnostics: per-phase cycle table of the stripe-pipelined wide-hull kernel; only in a library built with WM_KERNEL_DEFINES="WM_STRIPE_TIMING=1"
 * (WM_EINVAL otherwise). out16: cycles summed over wavefronts in scan / epoch set-up / cells / wait-left / bookkeeping / wait-right / publish, then rows,
 * epochs, total cycles, wavefronts. */
int wm_debug_stripe_timing(uint64_t *out16, int reset);

/* ksw_ll_qinit + ksw_ll_i16 (src/ksw2.h:82-83, src/ksw2_ll_sse.c:32-147): local alignment SCORE of query vs target with affine gaps
 * (16-bit striped lanes in the reference; ties and the striped layout's end coordinates are reproduced: *qe may be as low as -7 … see
 * host/wm_align.cpp). m = 5, mat = 5x5. Runs on the HOST (the mapper needs it twice per inversion candidate / boundary exon only);
 * exported so that the reference can be linked against it (oracle/wm_subst.cpp). Returns the score. */
int wm_ksw_ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat5x5, int gapo, int gape, int *qe, int *te);

/* ksw_exts2_sse as a batch (replaces src/ksw2.h:63-64, called at src/align.c:326-327 when MM_F_SPLICE is set): the splice-aware
 * extension. sc->q / e = gap open / extension, sc->q2 = the price of an intron (no extension), sc->e2 unused; noncan = the penalty of a
 * non-canonical splice site, junc_bonus = the bonus of an annotated junction; flag = KSW_EZ_* incl. SPLICE_FOR 0x100, SPLICE_REV 0x200,
 * SPLICE_FLANK 0x400 (job.w and job.end_bonus are ignored: the reference's function has no band and no end bonus). junc: NULL, or
 * seqs_bytes bytes parallel to seqs whose entries at a job's target hold the junction bits of mm_idx_bed_junc (src/index.c:690-803).
 * CIGAR op 3 = N. Results, pool and errors as wm_ksw_batch. wm_map_reads uses it for every alignment when MM_F_SPLICE is set. */
int wm_ksw_exts2_batch(wm_ctx_t *ctx, const wm_ksw_score_t *sc, int noncan, int junc_bonus, int n_jobs, const wm_ksw_job_t *jobs,
                       const uint8_t *seqs, size_t seqs_bytes, const uint8_t *junc,
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

/* Kernel routing knob (process-wide; results never depend on it): alignments of the 4-pair / 8-pair one-wavefront classes (traceback pitch <= 496 /
 * <= 1008 lanes) with at least rows4 / rows8 DP rows (qlen + tlen - 1) run on the stripe-pipelined multi-wave kernel (csrc/ksw_stripe_kernel.h), like
 * every wider hull does. 0 = never, < 0 = leave that threshold as it is; on = 0 switches the stripe classes off altogether (the barrier-per-row
 * kernels of round 3), 1 = on with the default geometries (since round 5: <2,16> — two lane pairs per wavefront, sixteen wavefronts — for hulls of
 * 1793..3840 lanes instead of <4,8>), 2 = both sixteen-wavefront geometries (also <1,16> for the long jobs of the one-wavefront classes), 3 = none of
 * them (the round-4 routing); environment: WM_KSW_STRIPE16 = 0 | 1 | 2 | 3, bits in that order; < 0 = leave. Defaults: WM_KSW_STRIPE_ROWS4 /
 * WM_KSW_STRIPE_ROWS8 / WM_KSW_STRIPE / WM_KSW_STRIPE16 from the environment. Not part of the
 * reference's interface: tests force every job through every kernel with it, tools/ tune the thresholds. */
void wm_ksw_set_routing(int on, int rows4, int rows8);
/* The same kind of knob for the chained-workgroup kernels (csrc/ksw_chain_kernel.h, round 6: one alignment over several compute units, every wavefront a
 * workgroup of its own, row messages through HBM — any hull width; ksw_extd2_sse's row loop src/ksw2_extd2_sse.c:123-376 is what every wavefront runs on its
 * stripe). mode bits: 1 = every job the stripe classes and the old wide-hull kernels would serve (default), 2 = also exact extensions of the 8-pair
 * one-wavefront classes with at least min_rows_exact rows, 4 = every job (tests); 0 = none (the round-5 routing). bp = 2 | 4 register pairs per wavefront
 * (256- / 512-lane stripes). < 0 / 0 = leave. Environment defaults: WM_KSW_CHAIN (1), WM_KSW_CHAIN_ROWS (2048), WM_KSW_CHAIN_BP (2). Results never depend on it. */
void wm_ksw_set_chain_routing(int mode, int min_rows_exact, int bp);
/* Two alignments per wavefront (round 6; csrc/ksw_dual_kernel.h) for the gap fills — band never clips (w >= qlen, tlen), no ambiguous base, KSW_EZ_APPROX_MAX: the classes
 * that hold 60 % of all DP cells. Workgroup g runs jobs 2g and 2g + 1 of the class's size-sorted list in the low / high 16-bit halves of the same registers
 * (ksw_extd2_sse's row loop src/ksw2_extd2_sse.c:123-313 once for both). on = 0 (the default: measured slower in the mapper): one alignment per
 * wavefront (ksw_dp_packed) for those classes too. Environment default: WM_KSW_DUAL (0). Results never depend on it. wm_ksw_dual_enabled(): the current setting. */
void wm_ksw_set_dual(int on);
int wm_ksw_dual_enabled(void);

/* Scalar drop-in with the reference's exact signature minus the kalloc handle (src/ksw2.h:60-61):
 * one alignment through the same kernels; cigar is malloc'd into *cigar_out (caller frees). */
int wm_ksw_extd2(wm_ctx_t *ctx, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m,
                 const int8_t *mat, int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus,
                 int flag, wm_ksw_result_t *ez, uint32_t **cigar_out);

/* ---- reference index ------------------------------------------------------------------------------ */
/* Host-side build (mm_idx_gen, src/index.c:378; reads the -W list like src/index.c:388-434) and upload of the
 * flat arrays (packed bases, key table, position runs, bloom bits) to HBM. kmer_file may be NULL/"". */
typedef struct wm_index_s wm_index_t;
int wm_index_build(const char *fasta, const char *kmer_file, int k, int w, int n_threads, wm_index_t **out);
/* The same index with the expensive part — mm_sketch over the whole reference (src/index.c:289-358; serial in the reference, ~31 Mb/s) — on the
 * device: one wavefront per contig runs the window-minimum chain kernel (sketch_coop) on the contig's codes, the host packs the bases and
 * builds the bloom filter before; the key table (src/index.c:200-252: per-bucket sort + hash fill) is built on the device as well — radix sort of
 * the minimizers, run-length encode, sort by home slot, linear probing as a prefix maximum (index_table_on_device) — with the host's builder as the
 * fall-back when the arena is too small for the sort. Bit-identical to wm_index_build (tests/test_aux_gpu.py).
 * Needs odd k (every preset); contigs are sketched in groups that fit the context's arena (~26 B per base). stats (optional, 4 doubles):
 * seconds reading + packing, sketching on the device (incl. transfers), building the table; minimizers. */
int wm_index_build_gpu(wm_ctx_t *ctx, const char *fasta, const char *kmer_file, int k, int w, int n_threads, wm_index_t **out, double *stats);
/* Both with the index flags of mm_idxopt_t::flag (src/minimap.h:41-43). Known here: MM_I_HPC = 1, the CLI's -H — minimizers over the homopolymer-
 * compressed sequence (src/sketch.c:152-163: a run of one base is one step, a minimizer sits on the last base of its last run and carries the summed length
 * of its k runs as span); a mapper on such an index sketches its reads the same way (the device sketch compacts every sequence into its runs first, needs
 * an odd k) and anchors are moved to the start of their runs before alignment (mm_adjust_minier, src/align.c:352-361). An index loaded from a file carries
 * its flag in the header. With -H the device build keeps every contig on ONE wavefront (the chunked sketch cuts in base space, the runs would have to be cut in
 * run space): for references with contigs of tens of Mb the host build (parallel per contig) is the faster of the two. */
int wm_index_build_flag(const char *fasta, const char *kmer_file, int k, int w, int idx_flag, int n_threads, wm_index_t **out);
int wm_index_build_gpu_flag(wm_ctx_t *ctx, const char *fasta, const char *kmer_file, int k, int w, int idx_flag, int n_threads, wm_index_t **out, double *stats);
void wm_index_destroy(wm_index_t *idx);
/* The reference's index file ("MMI\2": winnowmap -d, mm_idx_dump / mm_idx_load, src/index.c:515-608). Files written here load in
 * the reference and vice versa. The reference does not store its bloom filter; wm_index_save appends it as a trailer the reference
 * skips, and wm_index_load rebuilds it from kmer_file (the -W list, may be NULL) when the file has none. */
int wm_index_save(const wm_index_t *idx, const char *path);
int wm_index_load(const char *path, const char *kmer_file, wm_index_t **out);
int wm_index_upload(wm_ctx_t *ctx, const wm_index_t *idx);
/* mm_idx_bed_read(mi, fn, 1) (src/index.c:756-766, `--junc-bed`, src/main.c:416): annotated introns (the gaps between the blocks of BED12
 * records; plain BED intervals otherwise) for splice mode's junction bonus. wm_index_add_junc hands over intervals a front end has already
 * parsed (the bound CLI: the reference's mi->I). The annotation stays on the host: the mapper attaches the bits of mm_idx_bed_junc to each
 * alignment request (wm_ksw_exts2_batch's `junc`). */
int wm_index_read_junc_bed(wm_index_t *idx, const char *path);
int wm_index_add_junc(wm_index_t *idx, int ctg, int n, const int32_t *st, const int32_t *en, const int32_t *strand);
int wm_index_n_seq(const wm_index_t *idx);
const char *wm_index_seq_name(const wm_index_t *idx, int rid);
int wm_index_seq_len(const wm_index_t *idx, int rid);
uint64_t wm_index_n_minimizers(const wm_index_t *idx);
/* mm_idx_get (src/mmpriv.h:71): pointer into index memory (never freed by the caller), *n = 0 when absent */
const uint64_t *wm_index_get(const wm_index_t *idx, uint64_t minier, int *n);

/* The `-W` list of a FASTA (canonical k-mers above the 0.9998-distinct count threshold, README.md:29-30): what
 * `meryl count k=15` + `meryl print greater-than distinct=0.9998` produce; meryl cannot be built offline. */
int wm_write_repetitive_kmers(const char *fasta, int k, double distinct, const char *out_path, uint64_t *n_out);
/* The same list counted on the device: canonical k-mer of every reference position -> radix sort -> run lengths -> count histogram -> meryl's
 * threshold -> the k-mers above it (references below 4 Gbase; byte-identical output). stats (optional, 4 doubles): seconds reading +
 * encoding, on the device (incl. transfers), writing; distinct k-mers. */
int wm_write_repetitive_kmers_gpu(wm_ctx_t *ctx, const char *fasta, int k, double distinct, const char *out_path, uint64_t *n_out, double *stats);
/* Flat-array export / import (multi-GPU: one rank builds, RCCL broadcasts the arrays, the others import).
 * Call wm_index_export with S == NULL to obtain sizes9 = {|S| u32, |hkey|=|hval| u64, |P| u64, |bloom| bytes,
 * n_seq, name bytes, packed k/w/hbits/flag, bloom bits, bloom salts}; seq_meta holds (offset, len) per sequence. */
int wm_index_export(const wm_index_t *idx, uint64_t *sizes9, uint32_t *S, uint64_t *hkey, uint64_t *hval, uint64_t *P, uint8_t *bloom,
                    uint64_t *seq_meta, char *names);
int wm_index_import(const uint64_t *sizes9, const uint32_t *S, const uint64_t *hkey, const uint64_t *hval, const uint64_t *P, const uint8_t *bloom,
                    const uint64_t *seq_meta, const char *names, wm_index_t **out);

/* ---- sketch / seed / chain, batched (need wm_index_upload first) ------------------------------------ */
/* mm_sketch of n sequences of 0..4 codes (rid = 0). Minimizers of sequence i: out[out_off[i] .. +counts[i]). */
int wm_sketch_batch(wm_ctx_t *ctx, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len,
                    wm128_t *out, size_t out_cap, uint64_t *out_off, int32_t *counts);
/* wm_sketch_batch without an index on the context: the caller supplies the -W bloom filter (the bit table and hash salts of the reference's
 * bloom_filter object, ext/bloom/bloom_filter.hpp; table_bits = 0 / bits = NULL for none) and the sketch parameters. This is what a
 * link-level substitute of mm_sketch (src/mmpriv.h:61) needs: see oracle/wm_subst.cpp. */
int wm_sketch_set_filter(wm_ctx_t *ctx, const uint8_t *bits, size_t n_bytes, uint64_t table_bits, uint32_t salt0, uint32_t salt1, int k, int w);
/* collect_seed_hits: minimizers of job i are mini[mini_off[i] .. +n_mini[i]); anchors (sorted by x with the
 * reference's radix_sort_128x permutation) go to out[out_off[i] .. +n_anchors[i]); rep_len as src/map.c:126. */
int wm_seed_batch(wm_ctx_t *ctx, int n, const wm128_t *mini, const uint64_t *mini_off, const int32_t *n_mini, const int32_t *qlen,
                  int max_occ, int64_t flag, wm128_t *out, size_t out_cap, uint64_t *out_off, int32_t *n_anchors, int32_t *rep_len);
/* mm_chain_dp (src/mmpriv.h:73) for n anchor sets; chains of job i: u[u_off[i] .. +n_u[i]), anchors regrouped in
 * place: a[a_off[i] .. +n_v[i]). */
typedef struct { int32_t max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc; float gap_scale;
                 int32_t is_cdna; /* splice mode: the gap cost of src/chain.c:69-74 */ } wm_chain_par_t;
int wm_chain_batch(wm_ctx_t *ctx, int n, wm128_t *a, const uint64_t *a_off, const int32_t *n_a, const wm_chain_par_t *par,
                   uint64_t *u, uint64_t *u_off, int32_t *n_u, int32_t *n_v);
/* One device call per MCAS window or stage-2 pass (src/map.c:334-341, 700-900): mm_sketch → collect_seed_hits (incl. radix_sort_128x with the
 * reference's tie permutation, src/ksort.h:101-151) → mm_chain_dp (fill AND extraction, src/chain.c:22-167), nothing leaving HBM in between.
 * Job i: its sequence is len codes starting at code seq_off of the resident read codes (wm_reads_upload) when seq_off >= 0, at seqs + stage_off
 * when seq_off == -1 (e.g. stage 2's masked copy of a read), or absent (seq_off == -2). n_pre anchors pre[pre_off ..) are handed in (stage 2:
 * the anchors collected in stage 1, src/map.c:739-781); the seeded anchors are appended to them and, when both are present, the union is sorted
 * again (src/map.c:818-833). The anchors are then chained with `par`. Results: res[i] (rep_len as src/map.c:126), chains of job i =
 * u_pool[res[i].u_off .. + n_u), their anchors regrouped by chain = a_pool[res[i].a_off .. + n_v). *u_used / *a_used receive the pool sizes
 * needed; WM_ENOMEM if a capacity is too small (call again with larger pools). */
typedef struct {
	int64_t seq_off;
	uint64_t stage_off, pre_off;
	int32_t len, n_pre;
	wm_chain_par_t par;
} wm_window_job_t;
typedef struct { int32_t n_anchors, rep_len, n_mini, n_u, n_v; uint32_t u_off, a_off; } wm_window_res_t;
int wm_window_batch(wm_ctx_t *ctx, int n, const wm_window_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes, const wm128_t *pre, size_t n_pre_total,
                    int max_occ, int64_t flag, wm_window_res_t *res, uint64_t *u_pool, size_t u_cap, size_t *u_used, wm128_t *a_pool, size_t a_cap, size_t *a_used);
/* kernel time of the last sketch/seed/chain/window batch call (HIP events on the context stream), ms */
float wm_last_aux_ms(const wm_ctx_t *ctx);

/* ---- the mapper: replacement of kt_for(worker_for) (src/map.c:1164) ---------------------------------- */
/* PROCESS-WIDE SETTINGS the library makes for the mapper (both are defaults: a value the caller has set wins; WM_NO_PROCESS_DEFAULTS=1 switches both off):
 *  - by the first wm_ctx_create / wm_device_count of the process (NOT at load time any more: loading the library changes nothing): GPU_MAX_HW_QUEUES=20 in
 *    the environment, unless it is set — the HIP runtime reads it once, when it initialises, and a mapper runs 6 device contexts + 14 side streams that
 *    HIP's default of 4 hardware queues would serialise. A host program that has made HIP calls of its own before (torch, say) keeps whatever it had;
 *  - when the first mapper is created: glibc's allocator is told to grow its arenas in 64-MB steps, never to trim them and to serve up to 32 MB from
 *    them (mallopt; WM_MALLOPT=0 or the caller's own MALLOC_TOP_PAD_ / MALLOC_TRIM_THRESHOLD_ / MALLOC_MMAP_THRESHOLD_ win): the mapping calls allocate
 *    and free their per-call tables from dozens of threads, which with the defaults is one mprotect / brk per table (profiles/r04l, r04m). The
 *    process's resident memory after a mapping call therefore stays at its high-water mark.
 * A program that uses only the batched operations above is not touched by the second. */
typedef struct wm_mapper_s wm_mapper_t;
/* preset: NULL/"" or "map-ont" | "map-pb" | "map-pb-clr" | "asm5" | "asm10" | "asm20" (mm_set_opt, src/options.c:89);
 * flag: mm_mapopt_t::flag bits to OR in (MM_F_CIGAR 0x4, MM_F_OUT_SAM 0x8, MM_F_OUT_CG 0x20, ...). */
int wm_mapper_create(wm_ctx_t *ctx, const wm_index_t *idx, const char *preset, int64_t flag, wm_mapper_t **out);
void wm_mapper_destroy(wm_mapper_t *m);
/* host parallelism (the reference's -t): n_threads host threads run the per-read glue; they are organised in groups
 * (4 from 16 threads up, else 2 from 8 up; env WM_GROUPS) that each share one device batch per operation on their own HIP stream + arena slice
 * (arena_bytes_per_group; 0 = same size as ctx). */
int wm_mapper_set_threads(wm_mapper_t *m, int n_threads, size_t arena_bytes_per_group);
/* Map n reads (ASCII). Output records (PAF, or SAM when MM_F_OUT_SAM) of all reads in input order are appended to
 * an internal buffer returned through *text / *text_len (valid until the next call). hits (optional, 16 int32 per
 * hit: rid rs re qs qe rev mapq n_cigar score cnt mlen blen dp_score dp_max dp_max2 flags; flags = primary | inv<<1 | sam_pri<<2 | split<<3 |
 * trans_strand<<5) and their CIGARs are
 * returned for tests; hit_first[i] = index of read i's first hit, hit_first[n] = total. */
int wm_map_reads(wm_mapper_t *m, int n, const char *const *names, const char *const *seqs, const int32_t *lens,
                 const char **text, size_t *text_len, const int32_t **hits, const uint32_t **cigars, const int64_t **hit_first);
#define WM_MAX_SLOTS 4
/* wm_map_reads with its own result buffers (slot 0 .. WM_MAX_SLOTS - 1; wm_map_reads = slot 0): calls with different slots may run CONCURRENTLY from
 * different host threads on one mapper. They share the device contexts; each keeps the codes of its mini-batch in its own slab. A mapping call spends its first
 * and last few hundred milliseconds filling and draining its pipeline of dependent device calls: with several mini-batches in flight those phases of one
 * hide behind the steady state of the others (bench.py maps consecutive steps on WM_BENCH_SLOTS slots, wm_map_file on WM_MAP_LANES lanes; default 2). Results are valid until the next call on that slot. */
int wm_map_reads_slot(wm_mapper_t *m, int slot, int n, const char *const *names, const char *const *seqs, const int32_t *lens,
                      const char **text, size_t *text_len, const int32_t **hits, const uint32_t **cigars, const int64_t **hit_first);
/* The number of mini-batches the caller keeps in flight (slots 0 .. n - 1 of wm_map_reads_slot): the allocation of resident read codes is made with that
 * many slabs (default: the lanes of wm_map_file, WM_MAP_LANES / WM_READ_SLABS, at least 2). A mini-batch on a slot without a slab is still mapped, from host
 * views of its operands (slower; the library says so once on stderr). */
int wm_mapper_set_slots(wm_mapper_t *m, int n);
/* Per read of the slot's last wm_map_reads[_slot] call: flags[i] = 1 when the mapper assigned rep_len where the reference assigns it before mm_set_mapq
 * reads it (src/map.c:933) — the rescan of the stretches stage 1 left unmapped (src/map.c:808-813), the fallback to plain mapping (:859-861; every read below
 * SVawareMinReadLength and all of splice mode) — and 0 on the pure two-stage path, where the reference's rep_len is an uninitialised stack word (src/map.c:281)
 * and its own MAPQ / rl:i differ from run to run (we use rep_len = 0 there). Lets a parity check compare MAPQ on exactly the reads where it is defined. */
int wm_map_reads_rep_len_defined(const wm_mapper_t *m, int slot, const uint8_t **flags, size_t *n);
/* counters of the last wm_map_reads call: [0] super-steps, [1] ksw jobs, [2] chain jobs, [3] seed jobs,
 * [4] sketch jobs, [5] DP cells, [6] ksw kernel us, [7] aux kernel us, [8] read bases */
/* The file-level loop: replacement of mm_map_file / mm_map_file_frag (src/map.c:1226-1268, src/minimap.h:372-374) for
 * single-segment reads. Reads FASTA/FASTQ (optionally gzip) mini-batches of mini_batch_bases bases (0 = 1 Gbase, src/options.c:50),
 * orders each mini-batch like src/map.c:1124-1143, maps it and writes PAF/SAM records to out_path ("-" = stdout); reading,
 * mapping and writing overlap. stats (optional, 6 doubles): reads, bases, mini-batches, seconds reading / mapping / writing. */
int wm_map_file(wm_mapper_t *m, const char *reads_path, const char *out_path, int64_t mini_batch_bases, double *stats);
/* ---- several GPUs from one C process (reads shard, the index is broadcast, no data-path collective) ----
 * wm_index_upload_dev: like wm_index_upload, but the five flat arrays (S, hkey, hval, P, bloom bits — sizes and contig table from `idx`) come from DEVICE
 * memory on src_device: the receive buffers of an RCCL broadcast, or another context's copy; one device-to-device (xGMI peer) copy per array, no host
 * staging. wm_index_upload_peer: the index that context `src` holds, copied into `dst` (any GPU of the node) — the in-process index broadcast.
 * wm_map_file_multi: wm_map_file over n mappers (one per GPU, each with its own context / index copy / host threads): mini-batches go round-robin to
 * 2 lanes per mapper, the ordered writer restores input order, the output equals wm_map_file's. Replaces the role of mm_map_file_frag's pipeline
 * (src/map.c:1226-1268) for an N-GPU node; the reference has no counterpart (it has one address space). With n > 1 the 2n mapping calls in flight share the
 * host: each gets usable cores / 2n worker threads for the duration of the loop (never more than its mapper's own count, at least 2; usable = the affinity
 * mask cut by the cgroup CPU quota; WM_MULTI_THREADS=<per call> overrides, 0 = the mappers' own counts). */
int wm_index_upload_dev(wm_ctx_t *ctx, const wm_index_t *idx, const void *d_S, const void *d_hkey, const void *d_hval, const void *d_P, const void *d_bloom, int src_device);
int wm_index_upload_peer(wm_ctx_t *dst, const wm_index_t *idx, const wm_ctx_t *src);
int wm_map_file_multi(wm_mapper_t *const *mappers, int n, const char *reads_path, const char *out_path, int64_t mini_batch_bases, double *stats);
/* argv of the calling front end: with MM_F_OUT_SAM, wm_map_file starts the file with the @SQ lines and the @PG line carrying this
 * command line, as mm_write_sam_hdr does before mapping (src/format.c:118-139, src/main.c:393). Optional (no CL: field without it). */
int wm_mapper_set_cmdline(wm_mapper_t *m, int argc, const char *const *argv);
/* The same for wm_map_file_split, which owns its mappers (one per index part): process-wide. The reference prints the @PG line (with CL:) when it meets
 * the first of several index parts (mm_write_sam_hdr(0, ...), src/main.c:395) and the @SQ lines of every part in the merge pass (src/map.c:1304-1306):
 * @PG comes first there, and so it does in wm_map_file_split's SAM output. argc < 0: the front end prints @PG itself (the reference's main does),
 * wm_map_file_split then only lists the @SQ lines. */
int wm_set_cmdline(int argc, const char *const *argv);
/* All mapping options as plain data: the fields of mm_mapopt_t (src/minimap.h:112-175) this library honours, same names, same meaning, same
 * defaults (mm_mapopt_init / mm_set_opt, src/options.c:14-131). A front end that has parsed the reference's command line into an
 * mm_mapopt_t copies it field by field (oracle/wm_binding.cpp does exactly that inside the reference's own CLI).
 * wm_mapopt_preset = mm_set_opt (preset NULL or "" = defaults; also returns the preset's k and w); wm_mapper_create_opt = wm_mapper_create
 * with explicit options (checked like mm_check_opt, src/options.c:133-188). */
typedef struct {
	int64_t flag;
	int32_t seed, sdust_thres, max_qlen;
	int32_t bw, max_gap, max_gap_ref, min_gap_ref, max_frag_len;
	int32_t max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
	float chain_gap_scale;
	int32_t SVaware, SVawareMinReadLength, suffixSampleOffset, min_mapq;
	float min_qcov;
	int32_t minPrefixLength, maxPrefixLength;
	float prefixIncrementFactor;
	int32_t stage2_bw, stage2_zdrop_inv, stage2_max_gap;
	float mask_level;
	int32_t mask_len;
	float pri_ratio;
	int32_t best_n;
	int32_t max_join_long, max_join_short, min_join_flank_sc;
	float min_join_flank_ratio, alt_drop;
	int32_t a, b, q, e, q2, e2, sc_ambi;
	int32_t zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
	float max_clip_ratio;
	float mid_occ_frac;
	int32_t min_mid_occ, mid_occ, max_occ;
	int64_t mini_batch_size, max_sw_mat;
	int32_t noncan, junc_bonus;                 /* splice mode (src/minimap.h: mm_mapopt_t::noncan, junc_bonus) */
	int32_t anchor_ext_len, anchor_ext_shift;   /* mm_fix_bad_ends_splice (src/align.c:545-563) */
} wm_mapopt_t;
int wm_mapopt_preset(const char *preset, wm_mapopt_t *out, int *k, int *w);
int wm_mapper_create_opt(wm_ctx_t *ctx, const wm_index_t *idx, const wm_mapopt_t *opt, wm_mapper_t **out);
/* A reference indexed in PARTS (`-I bases`, src/main.c:193; the parts are formed like src/index.c:289-300, 378-384 forms them) and the reads
 * mapped against one part after the other with the hits merged at the end (`--split-prefix`: mm_split_merge / merge_hits, src/map.c:1050-1105,
 * 1278-1321): contig ids shifted by the parts before, mm_hit_sort, mm_set_parent, mm_select_sub, mm_set_sam_pri, mm_set_mapq with the largest
 * rep_len. The output file equals `winnowmap -I … --split-prefix …`'s (the reference keeps the per-part hits in <prefix>.NNNN.tmp files; here they
 * go to anonymous temporary files under $TMPDIR). Every part is an index of its own: destroy each with wm_index_destroy. */
int wm_index_build_parts(const char *fasta, const char *kmer_file, int k, int w, int n_threads, uint64_t batch_bases, wm_index_t **out, int cap, int *n_parts);
int wm_map_file_split(wm_ctx_t *ctx, int n_parts, wm_index_t *const *parts, const wm_mapopt_t *opt, int n_threads, const char *reads_path, const char *out_path,
                      int64_t mini_batch_bases, double *stats);
/* The same flow ONE PART AT A TIME, the way the reference's main runs it (src/main.c:398-429: mm_idx_reader_read builds a part, mm_map_file maps every
 * read against it, mm_idx_destroy, next part; mm_split_merge at the end) — host memory and the device hold one index part at a time:
 *   wm_split_begin      starts a run over `reads_path` (k, w: what every part must have been built with)
 *   wm_split_add_part   uploads the part to the run's context, maps the whole reads file against it and spills the hits to an anonymous temporary
 *                       file under $TMPDIR (the reference's <prefix>.NNNN.tmp, src/map.c:1174-1190); the part may be destroyed when it returns
 *   wm_split_finish     SAM header over the contigs of all parts (src/map.c:1304-1306), the merge pass, the output file; frees the run (also on error)
 *   wm_split_abort      frees a run that is not finished
 * wm_map_file_split above is these three over parts the caller already holds; wm_map_file_split_fasta reads the reference FASTA part by part
 * (mm_idx_gen's rule, src/index.c:289-300,383) and builds each part when its turn comes — on the host, or with on_device != 0 on the GPU
 * (wm_index_build_gpu's sketch + table kernels) — so a reference of any size needs the memory of ONE part. *n_parts receives the number of parts. */
typedef struct wm_split_s wm_split_t;
int wm_split_begin(wm_ctx_t *ctx, const wm_mapopt_t *opt, int k, int w, int n_threads, const char *reads_path, int64_t mini_batch_bases, wm_split_t **out);
int wm_split_add_part(wm_split_t *s, wm_index_t *part);
int wm_split_finish(wm_split_t *s, const char *out_path, double *stats);
void wm_split_abort(wm_split_t *s);
int wm_map_file_split_fasta(wm_ctx_t *ctx, const char *fasta, const char *kmer_file, int k, int w, int build_threads, uint64_t batch_bases, int on_device,
                            const wm_mapopt_t *opt, int n_threads, const char *reads_path, const char *out_path, int64_t mini_batch_bases, double *stats, int *n_parts);

/* wm_map_file prints the SAM header itself when MM_F_OUT_SAM is set; a front end that has already printed it (the reference's main does,
 * src/main.c:393) turns that off */
int wm_mapper_set_sam_header(wm_mapper_t *m, int on);
int wm_mapper_stats(const wm_mapper_t *m, uint64_t *out9);
/* per ksw kernel class (B4/B8/B16 x CLIP x HASN register kernels, the two multi-wave LDS kernels, the generic kernel) since
 * wm_mapper_create: out[3k] = summed launch durations (ms, HIP events on the launching stream), out[3k+1] = DP cells,
 * out[3k+2] = launches. cap = doubles available in out; *n_classes receives the class count. Feeds bench.py's roofline. */
int wm_mapper_kernel_stats(const wm_mapper_t *m, double *out, int cap, int *n_classes);
int wm_ksw_n_classes(void);      /* the class count (what out of wm_mapper_kernel_stats / _union must hold) */
/* Launches of one class overlap on different streams, so the summed durations above are residency. out[k] = ms during which AT LEAST ONE launch of
 * class k was running (union of the launch intervals on the device clock) after since_ms; *now_ms = the clock now (the next call's since_ms). */
int wm_mapper_kernel_union(const wm_mapper_t *m, double since_ms, double *out, int cap, double *now_ms);
/* where the host time of the mapping calls went since wm_mapper_create (seconds, summed over the worker threads; the host replaces
 * kt_for(worker_for), src/map.c:1164): out[0] = CPU time running the per-read glue, out[1] = wall time asleep waiting for device results,
 * out[2..5] = CPU time inside the batched window (sketch → seed → sort → chain, one call) / seed / chain / ksw calls, out[6..9] = their wall time, out[10..13] = number of
 * batched calls, out[14] = wall time of the mapping phase, out[15] = wall time formatting records, out[16] = worker threads, out[17] = CPU time idle workers spent helping the host-side loops of running batched calls; with cap >= 21 also out[18] = WALL time running
 * the glue (far above out[0]: the workers are being descheduled), out[19] = wall time waiting for the hub's mutex, out[20] = wall time of the workers in total. cap >= 18. */
int wm_mapper_host_stats(const wm_mapper_t *m, double *out, int cap);
/* SAM header lines (@SQ.., @PG) as mm_write_sam_hdr (src/format.c:118-139) */
int wm_sam_header(const wm_index_t *idx, int argc, const char *const *argv, const char **text, size_t *text_len);

#ifdef __cplusplus
}
#endif
#endif
