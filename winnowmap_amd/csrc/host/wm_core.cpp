// wm_core.cpp — options/presets, exact-permutation sorts, small hashes (see wm_core.h for citations).
#include "wm_core.h"
#include <sched.h>
#include <unistd.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <string.h>
#include <string>
#include <math.h>
#include <algorithm>

namespace wm {

struct Nt4Table {
	uint8_t t[256];
	constexpr Nt4Table() : t()
	{
		for (int i = 0; i < 256; ++i) t[i] = i < 4 ? (uint8_t)i : 4;
		t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = t['U'] = t['u'] = 3;
	}
};
static constexpr Nt4Table g_nt4;
const uint8_t *const nt4_table = g_nt4.t;

void mapopt_init(MapOpt &o)
{
	o = MapOpt();
	o.min_dp_max = o.min_chain_score * o.a;                                              // src/options.c:47
	o.prefixIncrementFactor = (float)pow((o.maxPrefixLength - 1) * 1.0 / o.minPrefixLength, 0.5);   // :57
	o.stage2_max_gap = o.maxPrefixLength;
}

int set_preset(const char *p, IdxOpt &io, MapOpt &mo)
{
	if (p == 0) { io = IdxOpt(); mapopt_init(mo); return 0; }
	std::string s(p);
	if (s == "map-ont") { io.flag = 0; io.k = 15; }
	else if (s == "map-pb") {
		io.flag = 0; io.k = 15;
		mo.maxPrefixLength = mo.stage2_max_gap = 8000;
		mo.suffixSampleOffset = mo.minPrefixLength = 1000;
		mo.stage2_bw = 1000;
		mo.prefixIncrementFactor = (float)pow((mo.maxPrefixLength - 1) * 1.0 / mo.minPrefixLength, 0.33);
	} else if (s == "map-pb-clr") { mo.SVaware = false; }
	else if (s == "asm5") { io.flag = 0; io.k = 19; mo.a = 1; mo.b = 19; mo.q = 39; mo.q2 = 81; mo.e = 3; mo.e2 = 1; mo.zdrop = mo.zdrop_inv = 200; mo.min_dp_max = 200; }
	else if (s == "asm10") { io.flag = 0; io.k = 19; mo.a = 1; mo.b = 9; mo.q = 16; mo.q2 = 41; mo.e = 2; mo.e2 = 1; mo.zdrop = mo.zdrop_inv = 200; mo.min_dp_max = 200; }
	else if (s == "asm20") { io.flag = 0; io.k = 19; mo.a = 1; mo.b = 4; mo.q = 6; mo.q2 = 26; mo.e = 2; mo.e2 = 1; mo.zdrop = mo.zdrop_inv = 200; mo.min_dp_max = 200; }
	else if (s.compare(0, 6, "splice") == 0 || s == "cdna") {                              // src/options.c:116-129
		mo.SVaware = false;
		io.w = 25; io.flag = 0; io.k = 15;
		mo.flag |= F_SPLICE | F_SPLICE_FOR | F_SPLICE_REV | F_SPLICE_FLANK;
		mo.max_gap = 2000; mo.max_gap_ref = mo.bw = 200000;
		mo.a = 1; mo.b = 2; mo.q = 2; mo.e = 1; mo.q2 = 32; mo.e2 = 0;
		mo.noncan = 9; mo.junc_bonus = 9;
		mo.zdrop = 200; mo.zdrop_inv = 100;
		if (s == "splice:hq") { mo.junc_bonus = 5; mo.b = 4; mo.q = 6; mo.q2 = 24; }
	} else return -1;
	return 0;
}

int check_opt(const IdxOpt &io, const MapOpt &mo, std::string &err)
{
	if (io.k <= 0 || io.w <= 0) { err = "-k and -w must be positive"; return -5; }
	if (io.k > 28 || io.w >= 256) { err = "need k <= 28 and w < 256 (src/sketch.c:140)"; return -5; }
	if (mo.best_n < 0) { err = "-N must be no less than 0"; return -4; }
	if (mo.pri_ratio < 0.0f || mo.pri_ratio > 1.0f) { err = "-p must be within 0 and 1 (including 0 and 1)"; return -4; }
	if ((mo.flag & F_FOR_ONLY) && (mo.flag & F_REV_ONLY)) { err = "--for-only and --rev-only can't be applied at the same time"; return -3; }
	if (mo.e <= 0 || mo.q <= 0) { err = "-O and -E must be positive"; return -1; }
	if ((mo.q != mo.q2 || mo.e != mo.e2) && !(mo.e > mo.e2 && mo.q + mo.e < mo.q2 + mo.e2)) { err = "dual gap penalties violating E1>E2 and O1+E1<O2+E2"; return -2; }
	if ((mo.q + mo.e) + (mo.q2 + mo.e2) > 127) { err = "scoring system violating ({-O}+{-E})+({-O2}+{-E2}) <= 127"; return -1; }
	if (mo.zdrop < mo.zdrop_inv) { err = "Z-drop should not be less than inversion-Z-drop"; return -5; }
	if ((mo.flag & F_NO_PRINT_2ND) && (mo.flag & F_ALL_CHAINS)) { err = "-X/-P and --secondary=no can't be applied at the same time"; return -5; }
	return 0;
}

// ------------------------------------------------------------------------------------------------
// In-place MSD byte radix sort ("American flag"), 8-bit digits from the top byte, ranges of <= 64
// elements finished by insertion sort. Ties end up in an order fixed by this exact swap sequence, which
// downstream code (chain DP, region order) depends on — SURVEY.md Appendix G. Only the key is compared.
// ------------------------------------------------------------------------------------------------
template <class T, class KeyFn> struct FlagSort {
	KeyFn key;
	void insertion(T *b, T *e) const
	{
		for (T *i = b + 1; i < e; ++i) {
			if (!(key(*i) < key(*(i - 1)))) continue;
			T held = *i;
			T *j = i;
			while (j > b && key(held) < key(*(j - 1))) { *j = *(j - 1); --j; }
			*j = held;
		}
	}
	// one digit of one range: skips constant digits, counts, permutes in place; tail[d] = end of bucket d. Returns the digit's shift, or -1
	// if every remaining digit is constant (nothing to do).
	int digit(T *beg, T *end, int shift, T **tail) const
	{
		// A digit that is the same in every key leaves the range as it is (one bucket, every element already in place) and hands the whole
		// range to the next digit: skip such digits without counting them. (Anchor keys are strand | contig | position: four of the eight
		// digits are constant within a job, more within a bucket.)
		{
			uint64_t all_or = 0, all_and = ~(uint64_t)0;
			for (const T *i = beg; i != end; ++i) { const uint64_t k = key(*i); all_or |= k; all_and &= k; }
			const uint64_t diff = all_or ^ all_and;
			while (!(diff >> shift & 0xff)) {
				if (shift == 0) return -1;
				shift = shift > 8 ? shift - 8 : 0;
			}
		}
		size_t hist[256] = {0};
		T *head[256];
		for (T *i = beg; i != end; ++i) ++hist[key(*i) >> shift & 0xff];
		T *cur = beg;
		for (int d = 0; d < 256; ++d) { head[d] = cur; cur += hist[d]; tail[d] = cur; }
		for (int d = 0; d < 256; ) {
			if (head[d] == tail[d]) { ++d; continue; }
			int dst = (int)(key(*head[d]) >> shift & 0xff);
			if (dst == d) { ++head[d]; continue; }
			T held = *head[d];
			while (dst != d) {               // follow the displacement cycle until something for bucket d turns up
				std::swap(held, *head[dst]);
				++head[dst];
				dst = (int)(key(held) >> shift & 0xff);
			}
			*head[d]++ = held;
		}
		return shift;
	}
	void pass(T *beg, T *end, int shift) const
	{
		T *tail[256];
		shift = digit(beg, end, shift, tail);
		if (shift <= 0) return;
		const int next = shift > 8 ? shift - 8 : 0;
		T *cur = beg;
		for (int d = 0; d < 256; ++d) {
			T *stop = tail[d];
			if (stop - cur > 64) pass(cur, stop, next);
			else if (stop - cur > 1) insertion(cur, stop);
			cur = stop;
		}
	}
	void run(T *beg, T *end) const
	{
		if (end - beg <= 64) insertion(beg, end);
		else pass(beg, end, 56);
	}
	// The same sort — the same swaps in the same order within every range — with the ranges of one level spread over threads: a range's digit
	// is sequential by nature (its swap sequence fixes the order of equal keys), but its buckets are independent of each other. Levels are
	// expanded until there are enough ranges; then every range is finished by the sequential recursion.
	void run_parallel(T *beg, T *end, int n_threads) const
	{
		if (end - beg <= 64) { insertion(beg, end); return; }
		struct Range { T *b, *e; int shift; };
		std::vector<Range> level(1, Range{beg, end, 56});
		for (int depth = 0; depth < 3 && level.size() < 4 * (size_t)n_threads; ++depth) {
			std::vector<std::vector<Range>> kids(level.size());
			parallel_tasks(n_threads, level.size(), [&](size_t i) {
				const Range r = level[i];
				T *tail[256];
				const int shift = digit(r.b, r.e, r.shift, tail);
				if (shift <= 0) return;
				const int next = shift > 8 ? shift - 8 : 0;
				T *cur = r.b;
				for (int d = 0; d < 256; ++d) {
					T *stop = tail[d];
					if (stop - cur > 64) kids[i].push_back(Range{cur, stop, next});
					else if (stop - cur > 1) insertion(cur, stop);
					cur = stop;
				}
			});
			std::vector<Range> nxt;
			for (auto &k : kids) nxt.insert(nxt.end(), k.begin(), k.end());
			level.swap(nxt);
			if (level.empty()) return;
		}
		std::sort(level.begin(), level.end(), [](const Range &x, const Range &y) { return x.e - x.b > y.e - y.b; });       // largest first
		parallel_tasks(n_threads, level.size(), [&](size_t i) { pass(level[i].b, level[i].e, level[i].shift); });
	}
};
struct KeyX { uint64_t operator()(const m128 &a) const { return a.x; } };
struct KeyId { uint64_t operator()(uint64_t a) const { return a; } };
void radix_sort_128x(m128 *beg, m128 *end) { FlagSort<m128, KeyX>().run(beg, end); }
void radix_sort_128x_parallel(m128 *beg, m128 *end, int n_threads) { FlagSort<m128, KeyX>().run_parallel(beg, end, n_threads); }
void radix_sort_64(uint64_t *beg, uint64_t *end) { FlagSort<uint64_t, KeyId>().run(beg, end); }

uint64_t hash64_masked(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = (key * 265) & mask;               // key + (key<<3) + (key<<8)
	key = key ^ key >> 14;
	key = (key * 21) & mask;                // key + (key<<2) + (key<<4)
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}
uint64_t hash64_full(uint64_t key) { return hash64_masked(key, ~0ULL); }
uint32_t wang_hash32(uint32_t key)
{
	key += ~(key << 15); key ^= key >> 10; key += key << 3; key ^= key >> 6; key += ~(key << 11); key ^= key >> 16;
	return key;
}
uint32_t x31_hash_string(const char *s)
{
	uint32_t h = (uint32_t)*s;
	if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
	return h;
}

std::vector<ProfSlot> &prof_slots() { static std::vector<ProfSlot> v; return v; }
std::mutex &prof_mutex() { static std::mutex m; return m; }
int usable_cores()
{
	int n = 0;
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
	if (n < 1) { const long c = sysconf(_SC_NPROCESSORS_ONLN); n = c > 0 ? (int)c : 1; }
	double quota = 0;
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
		char q[64]; double per = 0;
		if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per;
		fclose(f);
	} else {
		double qv = -1, per = 0;
		if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &qv) != 1) qv = -1; fclose(g); }
		if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &per) != 1) per = 0; fclose(g); }
		if (qv > 0 && per > 0) quota = qv / per;
	}
	if (quota > 0 && quota < n) n = (int)quota > 0 ? (int)quota : 1;      // (whole cores, like winnowmap_amd/dist.py: available_cores)
	return n;
}

int prof_region(const char *name)
{
	std::lock_guard<std::mutex> g(prof_mutex());
	prof_slots().push_back(ProfSlot{ name, 0.0, 0 });
	return (int)prof_slots().size() - 1;
}
void prof_report(FILE *f)
{
	std::lock_guard<std::mutex> g(prof_mutex());
	for (const ProfSlot &s : prof_slots()) fprintf(f, "[prof] %-28s %10.2f ms  %10llu calls\n", s.name, s.ms, (unsigned long long)s.n);
}


static std::mutex g_ie_mu;
static std::string g_ie_msg;
void note_internal_error(const char *expr, const char *file, int line)
{
	const char *b = strrchr(file, '/');
	const std::string m = std::string("internal invariant violated: ") + expr + " (" + (b ? b + 1 : file) + ":" + std::to_string(line) + ")";
	if (ErrorSink *sink = tl_error_sink()) { sink->put(m); return; }
	std::lock_guard<std::mutex> lk(g_ie_mu);
	if (g_ie_msg.empty()) g_ie_msg = m;
}
void put_internal_error(const std::string &m)
{
	std::lock_guard<std::mutex> lk(g_ie_mu);
	if (g_ie_msg.empty()) g_ie_msg = m;
}
bool take_internal_error(std::string &msg)
{
	std::lock_guard<std::mutex> lk(g_ie_mu);
	if (g_ie_msg.empty()) return false;
	msg.swap(g_ie_msg); g_ie_msg.clear();
	return true;
}

} // namespace wm
