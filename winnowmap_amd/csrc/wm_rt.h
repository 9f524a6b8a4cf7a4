// wm_rt.h — what the translation units of libwmgpu.so share (NOT part of the C-ABI: include/wm_gpu.h is). Round 6: the library was one
// 3 900-line translation unit; it is five now —
//   wm_rt.hip      runtime: errors, device contexts, arena, pinned slab, process-wide defaults
//   wm_ksw.hip     the alignment kernels' entry points, routing, batch planning / launch / fetch (ksw2: src/ksw2_extd2_sse.c, src/ksw2_exts2_sse.c)
//   wm_index.hip   sketch / seed / chain batch operations, the index on the device, the -W counter (src/sketch.c, src/index.c, src/map.c:97-254, src/chain.c)
//   wm_window.hip  the fused window call (sketch -> seed -> sort -> chain -> extraction)
//   wm_mapper.hip  the mapper: device contexts as a pool, mapping calls, file loops, statistics; the host sources (host/*.cpp) are compiled here
// — compiled separately (winnowmap_amd/build.py) and linked into the one shared object. No device code is shared across units except through the kernel headers.
#pragma once
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <string.h>
#include <cstring>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <new>
#include <algorithm>
#include <numeric>
#include <thread>
#include <chrono>
#include <atomic>
#include <mutex>
#include <unistd.h>
#include "host/wm_core.h"
#include "host/wm_index.h"
#include "wm_internal.h"
#include "ksw_plan.h"
// large staging buffers: no value-initialisation (a std::vector would memset hundreds of MB per batch). When a context is
// given, the buffer comes from that context's PINNED host slab (LIFO bump allocation): device copies to/from pinned memory run
// at full PCIe rate and truly asynchronously, pageable memory is bounced through the runtime's staging buffers.
struct wm_ctx_s;
void *pin_take(wm_ctx_s *c, size_t bytes, size_t *mark);
void pin_release(wm_ctx_s *c, size_t mark);
template <class T> struct UBuf {
	T *p; size_t n; wm_ctx_s *c; size_t mark; bool pinned;
	explicit UBuf(size_t n_, wm_ctx_s *c_ = 0) : p(0), n(n_), c(c_), mark(0), pinned(false)
	{
		const size_t bytes = (n_ ? n_ : 1) * sizeof(T);
		if (c) { p = (T*)pin_take(c, bytes, &mark); pinned = p != 0; }
		if (!p) p = (T*)malloc(bytes);
		if (!p) throw std::bad_alloc();         // (caught where the batched calls are issued: reported as an error, never abort())
	}
	~UBuf() { if (pinned) pin_release(c, mark); else free(p); }
	UBuf(const UBuf&) = delete; UBuf &operator=(const UBuf&) = delete;
	T *data() { return p; } const T *data() const { return p; }
	size_t size() const { return n; }
	T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
	T *begin() { return p; } T *end() { return p + n; }
};
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

namespace wm { int read_fastx(const std::string &fn, std::vector<std::string> &names, std::vector<std::string> &seqs, std::vector<std::string> *quals, std::vector<std::string> *comments, std::string &err); }      // host/wm_seqio.cpp

// Wave priority (s_setprio: the SIMD's arbiter issues the ready wave with the highest priority first). The path mixes two kinds of kernels on
// one chip: the bulk DP classes — tens of thousands of independent waves, throughput work — and LATENCY-bound serial chains that a whole
// batched call waits for (one alignment over thousands of rows with a barrier per row, the window kernels' lane-0 sections, the traceback walk).
// Sharing a SIMD with seven bulk waves slows a serial chain several times while it costs the bulk nothing to yield: the chains run at
// raised priority, the bulk at the default 0 (profiles/r03c_window_profile.txt: a kernel's duration under load vs alone). WM_PRIO=0 (build define) turns it off for A/B.
#ifndef WM_PRIO
#define WM_PRIO 1
#endif
#ifndef WM_STRIPE_PRIO
#define WM_STRIPE_PRIO 2      // the stripe-pipelined classes: a few hundred wavefronts per launch, each a chain of dependent rows (3: above the exact / clipped register classes, for A/B)
#endif
#if WM_PRIO
#define WM_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define WM_SETPRIO(n) ((void)0)
#endif

int set_err(int code, const char *fmt, ...);
void wm_err_clear();                                                // forget this thread's message
const char *wm_err_text();                                          // this thread's message (wm_last_error)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err(WM_ENODEV, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

struct wm_ctx_s {
	int device;
	hipStream_t stream;
	hipStream_t kstream[4];                     // own side streams (created on first use): kernel classes of one batch run concurrently
	// the mapper's contexts draw their side streams from ONE pool instead, sized so that all streams of the mapper fit the hardware queues
	// (GPU_MAX_HW_QUEUES = 20 since round 4 — 6 main streams + 14 side streams; 16 / 24 / 32 measured slower, profiles/r04g_sched_sweep.txt;
	// more streams than queues share queues and serialise, profiles/r02f_stream_conc.txt)
	hipStream_t *side_pool; int n_side_pool; std::atomic<unsigned> *side_next;
	std::vector<hipStream_t> owned_pool; std::atomic<unsigned> owned_next[3];   // (the pool lives in the mapper's first context; the others point at it; one cursor per weight class)
	hipEvent_t kev[5];
	hipEvent_t cev[WM_KSW_NCLASS][2];           // per-class start/stop (on the stream the class was launched on)
	double k_ms[WM_KSW_NCLASS]; uint64_t k_cells[WM_KSW_NCLASS], k_launches[WM_KSW_NCLASS];   // accumulated per kernel class
	// every launch of a class as an interval on the device's clock (ms since the process-wide base event): launches of one class overlap on
	// different streams, so their SUMMED durations are residency, not time — the union of the intervals is (wm_mapper_kernel_union)
	std::vector<std::pair<float, float>> k_iv[WM_KSW_NCLASS];
	std::mutex iv_mu;                           // k_iv: appended by the batched call that holds the context, read by wm_mapper_kernel_union from any thread
	uint8_t *arena;
	size_t arena_bytes, arena_used;
	hipEvent_t ev[4];
	hipEvent_t sync_ev;                         // blocking-sync event: waiting threads sleep instead of spinning (the host cores are the scarce resource)
	float last_ms, aux_ms;
	uint64_t acc_cells; double t_prep, t_run, t_fetch;
	// flat index in HBM (wm_index_upload)
	uint64_t *d_hkey, *d_hval, *d_P;
	uint8_t *d_bloom;
	uint32_t *d_S;                              // packed reference (4 bits per base), for position jobs
	std::vector<uint64_t> seq_off; std::vector<uint32_t> seq_len;       // contig table of the uploaded index (bounds of position jobs)
	// the read codes of the current mini-batch(es), resident: 2 bits per base in d_reads, the ambiguity bitmap in d_reads_nm (reads2bit.h; one allocation);
	// reads_bytes = bases a job may address, reads_cap = bases the allocation holds (wm_reads_upload / GpuOps::load_reads)
	uint64_t *d_reads, *d_reads_nm; size_t reads_bytes, reads_cap; bool owns_reads;
	int hbits;
	wm_sketch_params_t skp;
	bool have_index, owns_index;
	bool owns_filter;                           // d_bloom came from wm_sketch_set_filter (no index on this context)
	int host_threads;                           // threads the batched entry points may use for their host-side packing / sorting
	uint8_t *pin; size_t pin_bytes, pin_used;   // pinned host slab for staging (allocated on first use)
	int *pin_small;                             // a few pinned words for scalar read-backs (an async copy into pageable memory makes the caller spin until the stream gets there)
};

hipError_t ctx_sync(wm_ctx_s *c);
void wm_default_malloc();
hipEvent_t device_base_event(int device);
void side_split(int P, int *light, int *heavy);
void *arena_take(wm_ctx_t *c, size_t bytes);
// releases what a batched call took from the arena when the call returns
struct ArenaMark { wm_ctx_t *c; size_t m; ArenaMark(wm_ctx_t *c_) : c(c_), m(c_->arena_used) {} ~ArenaMark() { c->arena_used = m; } };
int reads_alloc(wm_ctx_t *c, size_t cap);

struct wm_index_s { wm::Index ix; };

// wm_index.hip
int sketch_batch_impl(wm_ctx_t *c, int n, const uint8_t *seqs, size_t seqs_bytes, const uint64_t *seq_off, const int32_t *len, const uint8_t *resident,
                      wm128_t *out, size_t out_cap, uint64_t *out_off, int32_t *counts);
int wm_index_build_seqs_dev(wm_ctx_t *c, const wm::IdxOpt &io, std::vector<std::string> &names, std::vector<std::string> &seqs, const std::string &kmer_file, int n_threads,
                            wm_index_t **out, double *stats = 0, bool replace_ok = true, double t0 = -1);
size_t sketch_long_bytes(int n, const wm_sketch_job_t *h_jobs, bool allow_long, bool hpc = false);
int sketch_launch(wm_ctx_t *c, int n, const wm_sketch_job_t *h_jobs, const wm_sketch_job_t *d_jobs, const int *d_ord, const uint8_t *d_seqs,
                  double *d_so, uint64_t *d_sx, uint32_t *d_sy, uint32_t *d_sl, wm128_t *d_out, int *d_cnt, bool allow_long, uint8_t *mem = 0, size_t mem_bytes = 0);

// wm_window.hip
struct WinDev { wm_win_res_t *d_res; uint64_t *d_upool; wm128_t *d_vpool; uint64_t *d_ctr; uint64_t ctr[4]; uint32_t tot[3]; };
int window_launch(wm_ctx_t *c, int n, const wm_window_job_t *jobs, const uint8_t *seqs, size_t seqs_bytes, const wm128_t *pre, size_t n_pre_total,
                  int max_occ, int64_t flag, bool slot_full, WinDev &D);
int window_fetch(wm_ctx_t *c, const WinDev &D, int n, wm_window_res_t *res, uint64_t *u_pool, wm128_t *a_pool);
int window_verdict(const WinDev &D, int round);
