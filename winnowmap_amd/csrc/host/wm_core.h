// wm_core.h — host-side core types of the MI355X mapper (C++17). Names follow the reference's domain
// (minimizers, anchors, chains, regs); each structure cites the reference structure it mirrors.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include "../../../include/wm_gpu.h"

namespace wm {

typedef wm128_t m128;                      // mm128_t, src/minimap.h:55

// anchor flag bits, src/mmpriv.h:17-23
static const uint64_t SEED_LONG_JOIN = 1ULL << 40, SEED_IGNORE = 1ULL << 41, SEED_TANDEM = 1ULL << 42, SEED_SELF = 1ULL << 43;
static const int PARENT_UNSET = -1, PARENT_TMP_PRI = -2;   // src/mmpriv.h:8-9

// mm_mapopt_t::flag bits we honour, src/minimap.h:9-41
enum : int64_t {
	F_NO_DIAG = 0x001, F_NO_DUAL = 0x002, F_CIGAR = 0x004, F_OUT_SAM = 0x008, F_NO_QUAL = 0x010, F_OUT_CG = 0x020, F_OUT_CS = 0x040,
	F_SPLICE = 0x080, F_SPLICE_FOR = 0x100, F_SPLICE_REV = 0x200, F_NO_LJOIN = 0x400, F_OUT_CS_LONG = 0x800, F_SR = 0x1000, F_NO_PRINT_2ND = 0x4000, F_LONG_CIGAR = 0x10000,
	F_SPLICE_FLANK = 0x40000, F_SOFTCLIP = 0x80000, F_FOR_ONLY = 0x100000, F_REV_ONLY = 0x200000, F_HEAP_SORT = 0x400000, F_ALL_CHAINS = 0x800000,
	F_OUT_MD = 0x1000000, F_COPY_COMMENT = 0x2000000, F_EQX = 0x4000000, F_PAF_NO_HIT = 0x8000000, F_NO_END_FLT = 0x10000000,
	F_HARD_MLEVEL = 0x20000000, F_SAM_HIT_ONLY = 0x40000000
};

struct IdxOpt {                            // mm_idxopt_t, src/minimap.h:106-110; defaults src/options.c:5-12
	int k = 15, w = 50, flag = 0, bucket_bits = 14;
};

struct MapOpt {                            // mm_mapopt_t, src/minimap.h:112-175; defaults src/options.c:14-69
	int64_t flag = 0;
	int seed = 11, sdust_thres = 0, max_qlen = 0;
	int bw = 500, max_gap = 5000, max_gap_ref = -1, min_gap_ref = 1000, max_frag_len = 0;
	int max_chain_skip = 25, max_chain_iter = 5000, min_cnt = 3, min_chain_score = 40;
	float chain_gap_scale = 1.0f;
	bool SVaware = true;
	int SVawareMinReadLength = 10000, suffixSampleOffset = 2000, min_mapq = 5;
	float min_qcov = 0.5f;
	int minPrefixLength = 2000, maxPrefixLength = 16000;
	float prefixIncrementFactor = 0;
	int stage2_bw = 2000, stage2_zdrop_inv = 25, stage2_max_gap = 16000;
	float mask_level = 0.5f;
	int mask_len = 0x7fffffff;
	float pri_ratio = 0.8f;
	int best_n = 5;
	int max_join_long = 20000, max_join_short = 2000, min_join_flank_sc = 1000;
	float min_join_flank_ratio = 0.5f, alt_drop = 0.0f;
	int a = 2, b = 4, q = 4, e = 2, q2 = 24, e2 = 1, sc_ambi = 1;
	int zdrop = 400, zdrop_inv = 200, end_bonus = -1, min_dp_max = 80, min_ksw_len = 200;
	float max_clip_ratio = 1.0f;
	float mid_occ_frac = -1.0f;
	int min_mid_occ = 0, mid_occ = 5000, max_occ = 0;
	int64_t mini_batch_size = 1000000000;
	int64_t max_sw_mat = 0;
	int noncan = 0, junc_bonus = 0;        // splice mode: cost of a non-canonical splice site, bonus of an annotated junction (src/minimap.h:156-157)
	int anchor_ext_len = 20, anchor_ext_shift = 6;   // mm_fix_bad_ends_splice (src/options.c:48)
	std::string kmer_freq_filename;
};

void mapopt_init(MapOpt &o);                                          // mm_mapopt_init
int set_preset(const char *preset, IdxOpt &io, MapOpt &mo);           // mm_set_opt, src/options.c:89-131 (-1: unknown)
int check_opt(const IdxOpt &io, const MapOpt &mo, std::string &err);  // mm_check_opt, src/options.c:133-188

// alignment record: mm_reg1_t + mm_extra_t (src/minimap.h:80-103); the CIGAR lives in a vector
struct Reg {
	int32_t id = 0, cnt = 0, rid = 0, score = 0;
	int32_t qs = 0, qe = 0, rs = 0, re = 0;
	int32_t parent = 0, subsc = 0, as = 0, mlen = 0, blen = 0, n_sub = 0, score0 = 0;
	uint32_t mapq = 0, split = 0, rev = 0, inv = 0, sam_pri = 0, split_inv = 0;
	uint32_t hash = 0;
	float div = -1.0f;
	bool has_p = false;                    // "r->p != 0"
	int32_t dp_score = 0, dp_max = 0, dp_max2 = 0;
	uint32_t n_ambi = 0;
	uint32_t trans_strand = 0;             // mm_extra_t::trans_strand (splice mode): 1 +, 2 -, 3 undetermined
	std::vector<uint32_t> cigar;
	void drop_p() { has_p = false; dp_score = dp_max = dp_max2 = 0; n_ambi = 0; trans_strand = 0; cigar.clear(); cigar.shrink_to_fit(); }
};

// ---- sorts with the reference's exact (unstable) permutation, src/ksort.h:101-151 ----
void radix_sort_128x(m128 *beg, m128 *end);
// identical result (incl. the order of equal keys); the independent buckets of large inputs are sorted on up to n_threads threads
void radix_sort_128x_parallel(m128 *beg, m128 *end, int n_threads);
void radix_sort_64(uint64_t *beg, uint64_t *end);

// ---- hashing, src/sketch.c:43-63 and khash.h Wang / X31 used for the per-read tie-break hash (src/map.c:355-357) ----
uint64_t hash64_masked(uint64_t key, uint64_t mask);
uint64_t hash64_full(uint64_t key);                                   // src/hit.c:40-50
uint32_t wang_hash32(uint32_t key);                                   // __ac_Wang_hash, src/khash.h
uint32_t x31_hash_string(const char *s);                              // __ac_X31_hash_string, src/khash.h

extern const uint8_t *const nt4_table;                                  // seq_nt4_table, src/sketch.c:19-36

// Library code never aborts (include/wm_gpu.h: "integer return codes"): a violated internal invariant — the places where the
// reference has assert() (src/align.c:166, :282, :645, :782) — is recorded here (first one wins) and surfaced by the C-ABI entry
// point as WM_EINTERNAL after the batch.
// Two mapping calls may run at once (wm_map_reads_slot): a worker thread of a call records into the call's own sink (installed by map_batch for
// its workers); threads without one (a plain parallel_for helper) record into the process-wide slot that take_internal_error empties.
struct ErrorSink { std::mutex mu; std::string msg; void put(const std::string &m) { std::lock_guard<std::mutex> lk(mu); if (msg.empty()) msg = m; } };
inline ErrorSink *&tl_error_sink() { static thread_local ErrorSink *p = 0; return p; }
void note_internal_error(const char *expr, const char *file, int line);
void put_internal_error(const std::string &msg);     // an already formatted message into the process-wide slot
bool take_internal_error(std::string &msg);          // true + message if one was recorded since the last call; clears it
#define WM_INVARIANT(x) do { if (!(x)) ::wm::note_internal_error(#x, __FILE__, __LINE__); } while (0)

// A thread that issues a batched device call may borrow the mapper's idle workers for the call's host-side loops (packing, sorting,
// unpacking): the hub (wm_fiber.h) installs a hook for the duration of the call; parallel_for uses it when present.
struct ParHook { virtual ~ParHook() {} virtual void run(size_t n, const std::function<void(size_t)> &fn) = 0; };
inline ParHook *&tl_par_hook() { static thread_local ParHook *p = 0; return p; }
// a label for the NEXT parallel_for of this thread (diagnostics: the hub accounts the CPU time of shared loops per label)
inline const char *&tl_par_site() { static thread_local const char *s = 0; return s; }
#define WM_SITE(name) (::wm::tl_par_site() = (name))

// chunk size for the NEXT hooked loop of this thread (0 = the hook's default): 1 for loops over a few coarse tasks
inline size_t &tl_par_chunk() { static thread_local size_t c = 0; return c; }

// host-side helper: fn(i) for i in [0, n) on up to n_threads threads (dynamic chunks); used for packing / unpacking batches
template <class F> inline void parallel_for(int n_threads, size_t n, F fn)
{
	if (n >= 256) if (ParHook *h = tl_par_hook()) { const std::function<void(size_t)> f = fn; h->run(n, f); return; }
	if (n_threads <= 1 || n < 2) { for (size_t i = 0; i < n; ++i) fn(i); return; }
	const size_t T = (size_t)n_threads < n ? (size_t)n_threads : n;
	const size_t chunk = n / (T * 8) > 0 ? n / (T * 8) : 1;
	std::atomic<size_t> next(0);
	auto body = [&]() { for (;;) { const size_t b = next.fetch_add(chunk); if (b >= n) break; const size_t e = b + chunk < n ? b + chunk : n; for (size_t i = b; i < e; ++i) fn(i); } };
	std::vector<std::thread> th;
	for (size_t t = 1; t < T; ++t) th.emplace_back(body);
	body();
	for (auto &x : th) x.join();
}

// fn(i) for a few coarse tasks, one at a time per thread (largest first is the caller's business)
template <class F> inline void parallel_tasks(int n_threads, size_t n, F fn)
{
	if (n >= 2) if (ParHook *h = tl_par_hook()) { const std::function<void(size_t)> f = fn; tl_par_chunk() = 1; h->run(n, f); return; }
	if (n_threads <= 1 || n < 2) { for (size_t i = 0; i < n; ++i) fn(i); return; }
	const size_t T = (size_t)n_threads < n ? (size_t)n_threads : n;
	std::atomic<size_t> next(0);
	auto body = [&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) break; fn(i); } };
	std::vector<std::thread> th;
	for (size_t t = 1; t < T; ++t) th.emplace_back(body);
	body();
	for (auto &x : th) x.join();
}

// ---- optional host-side profile (env WM_PROF=1): per-thread time accumulated per named region, printed by prof_report().
//      Regions must not contain a fiber yield.
struct ProfSlot { const char *name; double ms; uint64_t n; };
inline bool prof_on() { static const bool on = getenv("WM_PROF") != 0; return on; }
std::vector<ProfSlot> &prof_slots();
std::mutex &prof_mutex();
struct ProfScope {
	int id; std::chrono::steady_clock::time_point t0;
	explicit ProfScope(int id_) : id(id_) { if (id >= 0) t0 = std::chrono::steady_clock::now(); }
	~ProfScope() { if (id >= 0) { const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); std::lock_guard<std::mutex> g(prof_mutex()); prof_slots()[id].ms += ms; prof_slots()[id].n += 1; } }
};
// host cores this process may use: the affinity mask, cut by the cgroup CPU quota (cpu.max, or cfs_quota_us / cfs_period_us) when there is one
int usable_cores();
int prof_region(const char *name);
void prof_report(FILE *f);
#define WM_PROF(name) static const int wm_prof_id = wm::prof_on() ? wm::prof_region(name) : -1; wm::ProfScope wm_prof_scope(wm_prof_id)   /* one per block */

} // namespace wm
