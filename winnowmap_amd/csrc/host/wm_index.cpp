// wm_index.cpp — host-side sketching (index build; the per-read sketch runs on the GPU) and the flat index.
#include "wm_index.h"
#include <math.h>
#include <zlib.h>
#include <ctype.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <functional>
#include <fstream>
#include <thread>
#include <atomic>
#include <limits>

namespace wm {

// ---- bloom filter --------------------------------------------------------------------------------------
void Bloom::init(uint64_t n_kmers)
{   // compute_optimal_parameters (bloom_filter.hpp:108-160) with projected = max(n,1000), fpp = 0.001; the
	// hash count is capped at 2 (src/index.c:414) but the table is sized for the unconstrained optimum
	const double n = (double)(n_kmers > 1000 ? n_kmers : 1000);
	double min_m = std::numeric_limits<double>::infinity();
	for (double kk = 1.0; kk < 1000.0; kk += 1.0) {
		const double m = (-kk * n) / std::log(1.0 - std::pow(0.001, 1.0 / kk));
		if (m < min_m) min_m = m;
	}
	table_bits = (uint64_t)min_m;
	if (table_bits % 8) table_bits += 8 - table_bits % 8;
	// generate_unique_salt (:513-528) with random_seed_ = seed*0xA5A5A5A5+1 (:186), two salts, updated in place
	const uint64_t seed = 0xA5A5A5A55A5A5A5AULL * 0xA5A5A5A5ULL + 1;
	salt[0] = 0xAAAAAAAAu; salt[1] = 0x55555555u;
	for (int i = 0; i < 2; ++i) salt[i] = salt[i] * salt[(i + 3) % 2] + (uint32_t)seed;
	bits.assign(table_bits / 8, 0);
	n_inserted = 0;
}
uint32_t Bloom::hash_ap8(uint64_t key, uint32_t h)
{   // hash_ap (:551-608) for an 8-byte little-endian key: exactly one round of the 8-byte loop
	const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
	return h ^ ((h << 7) ^ lo * (h >> 3) ^ (~((h << 11) + (hi ^ (h >> 5)))));
}
void Bloom::insert(uint64_t key)
{
	for (int i = 0; i < 2; ++i) {
		const uint64_t bit = hash_ap8(key, salt[i]) % table_bits;
		bits[bit >> 3] |= (uint8_t)(1u << (bit & 7));
	}
	++n_inserted;
}
bool Bloom::contains(uint64_t key) const
{
	if (table_bits == 0) return false;
	for (int i = 0; i < 2; ++i) {
		const uint64_t bit = hash_ap8(key, salt[i]) % table_bits;
		if (!((bits[bit >> 3] >> (bit & 7)) & 1)) return false;
	}
	return true;
}

uint64_t encode_kmer(const char *s, int k)
{
	uint64_t fwd = 0, rev = 0;
	for (int i = 0; i < k; ++i) {
		const uint64_t c = nt4_table[(uint8_t)s[i]];
		fwd = fwd << 2 | c;
		rev = rev >> 2 | (3ULL ^ c) << (2 * (k - 1));
	}
	return fwd < rev ? fwd : rev;
}

static inline uint64_t fmix64(uint64_t h)
{
	h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
	return h;
}
double minimizer_order(uint64_t kmer, bool down)
{   // larger hash = smaller order = preferred; down-weighted k-mers use x^8 (three squarings) — no FMA here
	const double x = (double)fmix64(kmer) * 1.0 / 18446744073709551616.0;
	if (!down) return -1.0 * x;
	const double x2 = x * x, x4 = x2 * x2;
	return -1.0 * (x4 * x4);
}

// ---- robust weighted winnowing as a small streaming machine -----------------------------------------------
namespace {
struct Winnower {
	static constexpr uint64_t NONE = ~0ULL;
	int w, k;
	uint64_t mask, top_shift, fwd = 0, rev = 0;
	int run = 0, slot = 0, min_slot = 0;
	uint64_t ring_x[256], ring_y[256], min_x = NONE, min_y = NONE;
	double ring_o[256], min_o = 2.0;
	const Bloom *bloom;
	std::vector<m128> &out;
	Winnower(int w_, int k_, const Bloom *b, std::vector<m128> &o) : w(w_), k(k_), mask((1ULL << 2 * k_) - 1), top_shift(2ULL * (k_ - 1)), bloom(b), out(o)
	{
		for (int j = 0; j < w; ++j) ring_x[j] = ring_y[j] = NONE, ring_o[j] = 2.0;
	}
	void emit() { m128 m = { min_x, min_y }; out.push_back(m); }
	// span: what the low byte of x carries — k, or with homopolymer compression the length of the k runs that make the k-mer (then only < 256 counts)
	void push(uint32_t rid, uint32_t pos, int c, int span)
	{
		uint64_t x = NONE, y = NONE;
		double o = 2.0;
		if (c < 4) {
			fwd = (fwd << 2 | (uint64_t)c) & mask;
			rev = rev >> 2 | (3ULL ^ (uint64_t)c) << top_shift;
			if (fwd == rev) return;                            // strand-ambiguous k-mer: nothing happens at all
			const int strand = fwd < rev ? 0 : 1;
			if (++run >= k && span < 256) {
				const uint64_t km = strand ? rev : fwd;
				x = hash64_masked(km, mask) << 8 | (uint64_t)span;
				y = (uint64_t)rid << 32 | (uint64_t)pos << 1 | (uint64_t)strand;
				o = minimizer_order(km, bloom && bloom->contains(km));
			}
		} else run = 0;
		ring_x[slot] = x; ring_y[slot] = y; ring_o[slot] = o;
		if (o < min_o) {                                          // strictly better: report the minimum it replaces
			if (run >= w + k && min_x != NONE) emit();
			min_x = x; min_y = y; min_o = o; min_slot = slot;
		} else if (slot == min_slot) {                            // the minimum just fell out of the window
			if (run >= w + k - 1 && min_x != NONE) emit();
			min_x = min_y = NONE; min_o = 2.0;
			for (int n = 1; n <= w; ++n) {                        // oldest to newest; the newest of equal orders wins
				const int j = (slot + n) % w;
				if (min_o >= ring_o[j]) { min_x = ring_x[j]; min_y = ring_y[j]; min_o = ring_o[j]; min_slot = j; }
			}
		}
		if (++slot == w) slot = 0;
	}
	void finish() { if (min_x != NONE) emit(); }
};
}

void sketch(const char *seq, int len, int w, int k, uint32_t rid, const Bloom *bloom, std::vector<m128> &out, bool hpc)
{
	Winnower wn(w, k, bloom, out);
	if (!hpc) {
		for (int i = 0; i < len; ++i) wn.push(rid, (uint32_t)i, nt4_table[(uint8_t)seq[i]], k);
	} else {
		// homopolymer compression (src/sketch.c:152-163): a run of one base is one step, at the position of its last base; the span is the summed length
		// of the last k runs since the last ambiguous base
		int rl[32], n_rl = 0, head = 0, span = 0;
		for (int i = 0; i < len; ++i) {
			const int c = nt4_table[(uint8_t)seq[i]];
			if (c < 4) {
				int n = 1;
				while (i + n < len && nt4_table[(uint8_t)seq[i + n]] == c) ++n;
				i += n - 1;
				rl[(head + n_rl++) & 31] = n; span += n;
				if (n_rl > k) { span -= rl[head & 31]; ++head; --n_rl; }
			} else n_rl = head = 0, span = 0;
			wn.push(rid, (uint32_t)i, c, span);
		}
	}
	wn.finish();
}

// ---- flat index ------------------------------------------------------------------------------------------
const uint64_t *Index::get(uint64_t minier, int *n) const
{
	*n = 0;
	if (hkey.empty()) return 0;
	const uint64_t msk = ((uint64_t)1 << hbits) - 1;
	for (uint64_t s = slot_of(minier, hbits);; s = (s + 1) & msk) {
		if (hkey[s] == minier) { *n = (int)(uint32_t)hval[s]; return &P[hval[s] >> 32]; }
		if (hkey[s] == ~0ULL) return 0;
	}
}

int Index::getseq(uint32_t rid, uint32_t st, uint32_t en, uint8_t *out) const
{
	if (rid >= seq.size() || st >= seq[rid].len) return -1;
	if (en > seq[rid].len) en = seq[rid].len;
	const uint64_t o = seq[rid].offset;
	uint64_t i = o + st;
	const uint64_t e = o + en;
	uint8_t *d = out;
	for (; i < e && (i & 7); ++i) *d++ = (uint8_t)(S[i >> 3] >> ((i & 7) << 2) & 0xf);         // up to a word boundary
	for (; i + 8 <= e; i += 8, d += 8) {                                                     // one packed word = 8 bases
		const uint32_t wv = S[i >> 3];
		const uint64_t lo = wv & 0xffffu, hi = wv >> 16;
		// spread the 4 nibbles of a half word into 4 bytes
		uint64_t a = (lo | lo << 8) & 0x00ff00ffULL; a = (a | a << 4) & 0x0f0f0f0fULL;
		uint64_t b = (hi | hi << 8) & 0x00ff00ffULL; b = (b | b << 4) & 0x0f0f0f0fULL;
		const uint64_t v = a | b << 32;
		memcpy(d, &v, 8);
	}
	for (; i < e; ++i) *d++ = (uint8_t)(S[i >> 3] >> ((i & 7) << 2) & 0xf);
	return (int)(en - st);
}

void Index::scan_n_runs()
{
	n_runs.clear();
	uint64_t run_b = 0; bool in_run = false;
	const uint64_t n_words = (total_len + 7) / 8;
	for (uint64_t wi = 0; wi < n_words && wi < S.size(); ++wi) {
		const uint32_t wv = S[wi];
		if (!(wv & 0xccccccccu)) {                         // eight codes < 4
			if (in_run) { n_runs.emplace_back(run_b, wi * 8); in_run = false; }
			continue;
		}
		for (int b = 0; b < 8; ++b) {
			const uint64_t pos = wi * 8 + b;
			if (pos >= total_len) break;
			const bool isn = (wv >> (4 * b) & 0xf) >= 4;
			if (isn && !in_run) { run_b = pos; in_run = true; }
			else if (!isn && in_run) { n_runs.emplace_back(run_b, pos); in_run = false; }
		}
	}
	if (in_run) n_runs.emplace_back(run_b, total_len);
}

bool Index::has_n(uint32_t rid, uint32_t st, uint32_t en) const
{
	if (n_runs.empty() || rid >= seq.size() || en <= st) return false;
	const uint64_t b = seq[rid].offset + st, e = seq[rid].offset + en;
	// first run that ends after b
	size_t lo = 0, hi = n_runs.size();
	while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (n_runs[mid].second <= b) lo = mid + 1; else hi = mid; }
	return lo < n_runs.size() && n_runs[lo].first < e;
}

int index_begin(const IdxOpt &io, const std::vector<std::string> &names, const std::vector<std::string> &seqs, const std::string &kmer_file, int n_threads, Index &ix, std::string &err)
{
	ix = Index();
	ix.k = io.k; ix.w = io.w; ix.flag = io.flag;
	// -W list: count the lines, size the filter, insert the canonical k-mers (src/index.c:388-434)
	{
		std::vector<uint64_t> kms;
		std::string last;
		if (!kmer_file.empty()) {
			std::ifstream in(kmer_file);
			std::string km;
			uint64_t freq;
			while (in >> km >> freq) { kms.push_back(encode_kmer(km.c_str(), (int)km.size())); last = km; }
		}
		if (!kms.empty() && (int)last.size() != io.k) { err = "input list of k-mers and winnowmap parameter k are inconsistent"; return -1; }
		ix.bloom.init(kms.size());
		for (uint64_t km : kms) ix.bloom.insert(km);
	}
	// sequences: names, offsets, 4-bit packing (src/index.c:316-339)
	uint64_t sum = 0;
	for (size_t i = 0; i < seqs.size(); ++i) {
		RefSeq r; r.name = names[i]; r.offset = sum; r.len = (uint32_t)seqs[i].size();
		ix.seq.push_back(r); sum += r.len;
	}
	ix.total_len = sum;
	ix.S.assign((sum + 7) / 8 + 1, 0);
	// (contigs share a packed word only at their first / last word: pack word-aligned interiors in parallel, the edges serially)
	parallel_for(n_threads, seqs.size(), [&](size_t i) {
		const uint64_t o = ix.seq[i].offset, n = seqs[i].size();
		uint64_t j = 0;
		const uint64_t head = (8 - (o & 7)) & 7;                // bases before the first word this contig owns alone
		for (j = head < n ? head : n; j + 8 <= n && ((o + j) >> 3) < ((o + n) >> 3); j += 8) {
			uint32_t wv = 0;
			for (int b = 0; b < 8; ++b) wv |= (uint32_t)nt4_table[(uint8_t)seqs[i][j + b]] << (4 * b);
			ix.S[(o + j) >> 3] = wv;
		}
	});
	for (size_t i = 0; i < seqs.size(); ++i) {
		const uint64_t o = ix.seq[i].offset, n = seqs[i].size();
		const uint64_t head = (8 - (o & 7)) & 7;
		for (uint64_t j = 0; j < (head < n ? head : n); ++j) { const uint64_t p = o + j; ix.S[p >> 3] |= (uint32_t)nt4_table[(uint8_t)seqs[i][j]] << ((p & 7) << 2); }
		uint64_t j = head < n ? head : n;
		while (j + 8 <= n && ((o + j) >> 3) < ((o + n) >> 3)) j += 8;      // (what the parallel pass covered)
		for (; j < n; ++j) { const uint64_t p = o + j; ix.S[p >> 3] |= (uint32_t)nt4_table[(uint8_t)seqs[i][j]] << ((p & 7) << 2); }
	}
	return 0;
}

int index_build(const IdxOpt &io, const std::vector<std::string> &names, const std::vector<std::string> &seqs,
                const std::string &kmer_file, int n_threads, Index &ix, std::string &err)
{
	if (index_begin(io, names, seqs, kmer_file, n_threads, ix, err) < 0) return -1;
	// sketch every contig (independent → one task per contig)
	std::vector<std::vector<m128>> per(seqs.size());
	std::atomic<size_t> next(0);
	auto work = [&]() {
		for (size_t i; (i = next.fetch_add(1)) < seqs.size();) {
			const std::string &s = seqs[i];
			if (!s.empty()) sketch(s.data(), (int)s.size(), io.w, io.k, (uint32_t)i, &ix.bloom, per[i], (io.flag & 1) != 0);
		}
	};
	{
		const int nt = std::max(1, std::min<int>(n_threads, (int)seqs.size()));
		std::vector<std::thread> th;
		for (int t = 1; t < nt; ++t) th.emplace_back(work);
		work();
		for (auto &t : th) t.join();
	}
	std::vector<m128> all;
	size_t tot = 0;
	for (auto &v : per) tot += v.size();
	all.reserve(tot);
	for (auto &v : per) { all.insert(all.end(), v.begin(), v.end()); std::vector<m128>().swap(v); }
	index_table_from_minimizers(ix, all);
	return 0;
}

// (key, position) records -> the flat table: grouped by minimizer key, positions ascending (src/index.c:200-252)
void index_table_from_minimizers(Index &ix, std::vector<m128> &all)
{
	ix.scan_n_runs();                                      // (S is complete on every path that gets here)
	ix.n_minimizers = all.size();
	std::sort(all.begin(), all.end(), [](const m128 &a, const m128 &b) { return (a.x >> 8) != (b.x >> 8) ? (a.x >> 8) < (b.x >> 8) : a.y < b.y; });
	size_t nk = 0;
	for (size_t i = 0; i < all.size(); ++i) if (i == 0 || (all[i].x >> 8) != (all[i - 1].x >> 8)) ++nk;
	ix.n_keys = nk;
	ix.hbits = 4;
	while (((uint64_t)1 << ix.hbits) < 2 * nk + 2) ++ix.hbits;
	ix.hkey.assign((size_t)1 << ix.hbits, ~0ULL);
	ix.hval.assign((size_t)1 << ix.hbits, 0);
	ix.P.resize(all.size());
	// The keys enter the table in the order of (home slot, key). Linear probing then has a closed form — the j-th key lands on
	// max(home_j, slot of the key before it + 1), a prefix maximum — which is what lets the device build the same table with a sort and a scan
	// (wm_gpu.hip: index_table_on_device); the few keys that run past the last slot wrap around to the first free slots in the same order.
	struct Grp { uint64_t home, key, val; };
	std::vector<Grp> g;
	g.reserve(nk);
	for (size_t i = 0; i < all.size();) {
		size_t j = i;
		const uint64_t key = all[i].x >> 8;
		while (j < all.size() && (all[j].x >> 8) == key) { ix.P[j] = all[j].y; ++j; }
		g.push_back(Grp{ Index::slot_of(key, ix.hbits), key, (uint64_t)i << 32 | (uint64_t)(j - i) });
		i = j;
	}
	std::stable_sort(g.begin(), g.end(), [](const Grp &a, const Grp &b) { return a.home < b.home; });      // (g is in key order: stable = ties by key)
	index_table_insert(ix, g.size(), [&](size_t t, uint64_t *key, uint64_t *val) { *key = g[t].key; *val = g[t].val; return g[t].home; });
}

// sequential linear probing of n keys given in (home slot, key) order
void index_table_insert(Index &ix, size_t n, const std::function<uint64_t(size_t, uint64_t*, uint64_t*)> &item)
{
	const uint64_t msk = ((uint64_t)1 << ix.hbits) - 1;
	for (size_t t = 0; t < n; ++t) {
		uint64_t key, val;
		uint64_t s = item(t, &key, &val);
		while (ix.hkey[s] != ~0ULL) s = (s + 1) & msk;
		ix.hkey[s] = key; ix.hval[s] = val;
	}
}

// ---- the reference's index file ("MMI\2", mm_idx_dump / mm_idx_load, src/index.c:515-608) --------------------------------
// header: magic, u32 w k b n_seq flag | per sequence: u8 name length, name, u32 length | per bucket i < 2^b: u32 n, n x u64
// positions, u32 n_keys, n_keys x (u64 key = minier >> b << 1 | singleton, u64 val = position | start << 32 | count) | packed
// 4-bit bases. The reference does not store its bloom filter (it must be given -W again); we append an optional trailer
// "WMB1" with ours, which the reference's reader skips (it stops at the first block that is not an index part).
static const int MMI_B = 14;

int index_save_mmi(const Index &ix, const std::string &path, std::string &err)
{
	FILE *fp = fopen(path.c_str(), "wb");
	if (!fp) { err = "cannot write '" + path + "'"; return -1; }
	const uint32_t hdr[5] = { (uint32_t)ix.w, (uint32_t)ix.k, (uint32_t)MMI_B, (uint32_t)ix.seq.size(), (uint32_t)ix.flag };
	fwrite("MMI\2", 1, 4, fp);
	fwrite(hdr, 4, 5, fp);
	for (const RefSeq &r : ix.seq) {
		const uint8_t l = (uint8_t)std::min<size_t>(r.name.size(), 255);
		fwrite(&l, 1, 1, fp);
		fwrite(r.name.data(), 1, l, fp);
		fwrite(&r.len, 4, 1, fp);
	}
	// distribute the keys over the 2^b buckets
	const uint64_t mask = ((uint64_t)1 << MMI_B) - 1;
	std::vector<std::vector<uint64_t>> slots((size_t)1 << MMI_B);        // table slots per bucket
	for (size_t s = 0; s < ix.hkey.size(); ++s) if (ix.hkey[s] != ~0ULL) slots[ix.hkey[s] & mask].push_back(s);
	std::vector<uint64_t> p, kv;
	for (size_t b = 0; b < slots.size(); ++b) {
		p.clear(); kv.clear();
		std::sort(slots[b].begin(), slots[b].end(), [&](uint64_t x, uint64_t y) { return ix.hkey[x] < ix.hkey[y]; });
		for (uint64_t s : slots[b]) {
			const uint64_t first = ix.hval[s] >> 32, cnt = ix.hval[s] & 0xffffffffu, key = ix.hkey[s] >> MMI_B << 1;
			if (cnt == 1) { kv.push_back(key | 1); kv.push_back(ix.P[first]); }
			else { kv.push_back(key); kv.push_back((uint64_t)p.size() << 32 | cnt); for (uint64_t k = 0; k < cnt; ++k) p.push_back(ix.P[first + k]); }
		}
		const uint32_t n = (uint32_t)p.size(), size = (uint32_t)(kv.size() / 2);
		fwrite(&n, 4, 1, fp);
		fwrite(p.data(), 8, n, fp);
		fwrite(&size, 4, 1, fp);
		fwrite(kv.data(), 8, kv.size(), fp);
	}
	if (!(ix.flag & 2)) fwrite(ix.S.data(), 4, (ix.total_len + 7) / 8, fp);      // MM_I_NO_SEQ: the reference's loader expects no S then (src/index.c:601)
	{   // trailer: the bloom filter
		const uint64_t t[4] = { ix.bloom.table_bits, (uint64_t)ix.bloom.salt[0] | (uint64_t)ix.bloom.salt[1] << 32, ix.bloom.n_inserted, (uint64_t)ix.bloom.bits.size() };
		fwrite("WMB1", 1, 4, fp);
		fwrite(t, 8, 4, fp);
		fwrite(ix.bloom.bits.data(), 1, ix.bloom.bits.size(), fp);
	}
	const bool ok = fflush(fp) == 0 && !ferror(fp);
	fclose(fp);
	if (!ok) { err = "write error on '" + path + "'"; return -1; }
	return 0;
}

// kmer_file: the -W list to rebuild the bloom filter from when the file has no trailer (an index written by the reference);
// empty = keep an empty filter (no down-weighting of the READS' minimizers)
int index_load_mmi(const std::string &path, const std::string &kmer_file, Index &ix, std::string &err)
{
	FILE *fp = fopen(path.c_str(), "rb");
	if (!fp) { err = "cannot read '" + path + "'"; return -1; }
	auto bad = [&](const char *what) { err = std::string("'") + path + "': " + what; fclose(fp); return -1; };
	char magic[4];
	uint32_t hdr[5];
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "MMI\2", 4) != 0) return bad("not an MMI\\2 index");
	if (fread(hdr, 4, 5, fp) != 5) return bad("truncated header");
	ix = Index();
	ix.w = (int)hdr[0]; ix.k = (int)hdr[1]; ix.flag = (int)hdr[4];
	const int b = (int)hdr[2];
	if (b < 1 || b > 28) return bad("bad bucket bits");
	uint64_t sum = 0;
	for (uint32_t i = 0; i < hdr[3]; ++i) {
		uint8_t l;
		RefSeq r;
		if (fread(&l, 1, 1, fp) != 1) return bad("truncated sequence table");
		r.name.resize(l);
		if (l && fread(&r.name[0], 1, l, fp) != l) return bad("truncated sequence table");
		if (fread(&r.len, 4, 1, fp) != 1) return bad("truncated sequence table");
		r.offset = sum; sum += r.len;
		ix.seq.push_back(r);
	}
	ix.total_len = sum;
	std::vector<m128> all;
	std::vector<uint64_t> p, kv;
	for (uint64_t bi = 0; bi < ((uint64_t)1 << b); ++bi) {
		uint32_t n, size;
		if (fread(&n, 4, 1, fp) != 1) return bad("truncated bucket");
		p.resize(n);
		if (n && fread(p.data(), 8, n, fp) != n) return bad("truncated bucket");
		if (fread(&size, 4, 1, fp) != 1) return bad("truncated bucket");
		kv.resize((size_t)size * 2);
		if (size && fread(kv.data(), 8, kv.size(), fp) != kv.size()) return bad("truncated bucket");
		for (uint32_t j = 0; j < size; ++j) {
			const uint64_t key = kv[2 * j], val = kv[2 * j + 1];
			const uint64_t minier = (key >> 1) << b | bi;
			if (key & 1) { m128 e; e.x = minier << 8; e.y = val; all.push_back(e); }
			else {
				const uint64_t st = val >> 32, cnt = val & 0xffffffffu;
				if (st + cnt > n) return bad("corrupt bucket");
				for (uint64_t k = 0; k < cnt; ++k) { m128 e; e.x = minier << 8; e.y = p[st + k]; all.push_back(e); }
			}
		}
	}
	ix.S.assign((sum + 7) / 8 + 1, 0);
	if (!(ix.flag & 2) && fread(ix.S.data(), 4, (sum + 7) / 8, fp) != (sum + 7) / 8) return bad("truncated sequence");   // MM_I_NO_SEQ = 2
	index_table_from_minimizers(ix, all);
	// bloom filter: our trailer, else rebuild from the -W list
	char tg[4] = {0, 0, 0, 0};
	bool have_bloom = false;
	if (fread(tg, 1, 4, fp) == 4 && memcmp(tg, "WMB1", 4) == 0) {
		uint64_t t[4];
		if (fread(t, 8, 4, fp) != 4) return bad("truncated bloom trailer");
		ix.bloom.table_bits = t[0]; ix.bloom.salt[0] = (uint32_t)t[1]; ix.bloom.salt[1] = (uint32_t)(t[1] >> 32); ix.bloom.n_inserted = t[2];
		ix.bloom.bits.resize(t[3]);
		if (t[3] && fread(ix.bloom.bits.data(), 1, t[3], fp) != t[3]) return bad("truncated bloom trailer");
		have_bloom = true;
	} else if (memcmp(tg, "MMI\2", 4) == 0 && !feof(fp))
		// `winnowmap -d` concatenates one "MMI\2" part per -I batch of the reference (src/index.c:515-608 called per part, src/main.c)
		return bad("multi-part index files are not supported (the reference was split by -I; rebuild with a larger -I)");
	fclose(fp);
	if (!have_bloom) {
		std::vector<uint64_t> kms;
		std::string last;
		if (!kmer_file.empty()) {
			std::ifstream in(kmer_file);
			std::string km;
			uint64_t freq;
			while (in >> km >> freq) { kms.push_back(encode_kmer(km.c_str(), (int)km.size())); last = km; }
		}
		if (!kms.empty() && (int)last.size() != ix.k) { err = "input list of k-mers and the index's k are inconsistent"; return -1; }
		ix.bloom.init(kms.size());
		for (uint64_t km : kms) ix.bloom.insert(km);
	}
	return 0;
}

int index_build_from_fasta(const IdxOpt &io, const std::string &fasta, const std::string &kmer_file, int n_threads, Index &out, std::string &err)
{
	extern int read_fastx(const std::string &fn, std::vector<std::string> &names, std::vector<std::string> &seqs, std::vector<std::string> *quals, std::vector<std::string> *comments, std::string &err);
	std::vector<std::string> names, seqs;
	if (read_fastx(fasta, names, seqs, 0, 0, err) < 0) return -1;
	if (seqs.empty()) { err = "no sequences in " + fasta; return -1; }
	return index_build(io, names, seqs, kmer_file, n_threads, out, err);
}

int index_build_parts_from_fasta(const IdxOpt &io, const std::string &fasta, const std::string &kmer_file, int n_threads, uint64_t batch_bases,
                                 std::vector<Index> &parts, std::string &err)
{
	extern int read_fastx(const std::string &fn, std::vector<std::string> &names, std::vector<std::string> &seqs, std::vector<std::string> *quals, std::vector<std::string> *comments, std::string &err);
	std::vector<std::string> names, seqs;
	if (read_fastx(fasta, names, seqs, 0, 0, err) < 0) return -1;
	if (seqs.empty()) { err = "no sequences in " + fasta; return -1; }
	const uint64_t mini = batch_bases < 50000000ULL ? batch_bases : 50000000ULL;      // mm_idx_gen: min(mini_batch_size, batch_size), src/index.c:383
	parts.clear();
	size_t i = 0;
	while (i < seqs.size()) {
		std::vector<std::string> pn, ps;
		uint64_t sum = 0;
		while (i < seqs.size() && sum <= batch_bases) {                // step 0 of the reference's pipeline: another mini-batch unless sum_len > batch_size (:295)
			uint64_t mb = 0;
			while (i < seqs.size()) {                                  // mm_bseq_read: sequences until the chunk reaches `mini` bases
				mb += seqs[i].size();
				pn.push_back(std::move(names[i])); ps.push_back(std::move(seqs[i])); ++i;
				if (mb >= mini) break;
			}
			sum += mb;
		}
		parts.emplace_back();
		if (index_build(io, pn, ps, kmer_file, n_threads, parts.back(), err) < 0) return -1;
	}
	return (int)parts.size();
}

int32_t Index::cal_max_occ(float f) const
{
	if (f <= 0.) return INT32_MAX;
	std::vector<uint32_t> occ;
	occ.reserve(n_keys);
	for (size_t i = 0; i < hkey.size(); ++i) if (hkey[i] != UINT64_MAX) occ.push_back((uint32_t)hval[i]);   // (count of the key; 1 for singletons)
	if (occ.empty()) return 1;                                         // (the reference reads a[0] of an empty array here)
	size_t kth = (size_t)(uint32_t)((1. - f) * occ.size());
	if (kth >= occ.size()) kth = occ.size() - 1;
	std::nth_element(occ.begin(), occ.begin() + kth, occ.end());
	return (int32_t)(occ[kth] + 1);
}

void mapopt_update(MapOpt &opt, const Index &ix)
{
	if ((opt.flag & F_SPLICE_FOR) || (opt.flag & F_SPLICE_REV)) opt.flag |= F_SPLICE;
	if (opt.mid_occ_frac >= 0 && opt.mid_occ_frac < 1) opt.mid_occ = ix.cal_max_occ(opt.mid_occ_frac);
	if (opt.mid_occ < opt.min_mid_occ) opt.mid_occ = opt.min_mid_occ;
}

// ---- junction annotation (src/index.c:690-803) ----
int index_read_bed(Index &ix, const std::string &path, bool read_junc, std::string &err)
{
	gzFile fp = path == "-" ? gzdopen(0, "r") : gzopen(path.c_str(), "r");
	if (!fp) { err = "failed to open file '" + path + "'"; return -1; }
	std::vector<std::vector<JuncIntv>> I(ix.seq.size());
	std::vector<char> buf(1 << 16);
	std::string line, carry;
	auto name2id = [&](const char *nm) { for (size_t i = 0; i < ix.seq.size(); ++i) if (ix.seq[i].name == nm) return (int)i; return -1; };
	auto do_line = [&](std::string &ln) {
		if (!ln.empty() && ln.back() == '\r') ln.pop_back();
		JuncIntv t = { -1, -1, 0 };
		int32_t id = -1, n_blk = 0, i = 0;
		char *bl = 0, *bs = 0;
		char *p, *q;
		ln.push_back(0);
		for (p = q = &ln[0];; ++p) {
			if (*p == 0 || *p == '\t') {
				const int c = *p;
				*p = 0;
				if (i == 0) { id = name2id(q); if (id < 0) break; }
				else if (i == 1) { t.st = (int32_t)atol(q); if (t.st < 0) break; }
				else if (i == 2) { t.en = (int32_t)atol(q); if (t.en < 0) break; }
				else if (i == 5) t.strand = *q == '+' ? 1 : *q == '-' ? -1 : 0;
				else if (i == 9) { if (!isdigit((unsigned char)*q)) break; n_blk = (int32_t)atol(q); }
				else if (i == 10) bl = q;
				else if (i == 11) { bs = q; break; }
				if (c == 0) break;
				++i, q = p + 1;
			}
		}
		if (id < 0 || t.st < 0 || t.st >= t.en) return;
		std::vector<JuncIntv> &r = I[id];
		if (i >= 11 && read_junc) {                                    // BED12: the gaps between consecutive blocks
			int32_t st = (int32_t)strtol(bs, &bs, 10); ++bs;
			int32_t sz = (int32_t)strtol(bl, &bl, 10); ++bl;
			int32_t en = t.st + st + sz;
			for (int32_t b = 1; b < n_blk; ++b) {
				JuncIntv s = t;
				st = (int32_t)strtol(bs, &bs, 10); ++bs;
				sz = (int32_t)strtol(bl, &bl, 10); ++bl;
				s.st = en, s.en = t.st + st;
				en = t.st + st + sz;
				if (s.en > s.st) r.push_back(s);
			}
		} else r.push_back(t);
	};
	for (;;) {
		const int n = gzread(fp, buf.data(), (unsigned)buf.size());
		if (n <= 0) break;
		size_t st = 0;
		for (int k = 0; k < n; ++k)
			if (buf[k] == '\n') { carry.append(buf.data() + st, k - st); do_line(carry); carry.clear(); st = (size_t)k + 1; }
		carry.append(buf.data() + st, (size_t)n - st);
	}
	if (!carry.empty()) do_line(carry);
	gzclose(fp);
	for (auto &r : I) std::stable_sort(r.begin(), r.end(), [](const JuncIntv &a, const JuncIntv &b) { return a.st < b.st; });   // (order among equal starts does not matter: the bits are ORed)
	ix.I.swap(I);
	return 0;
}

int Index::bed_junc(int32_t ctg, int32_t st, int32_t en, uint8_t *s) const
{   // mm_idx_bed_junc, src/index.c:768-803: bit 1 / 2 = first / last base of a + strand intron inside [st, en), 8 / 4 the same for the - strand
	memset(s, 0, (size_t)(en > st ? en - st : 0));
	if (I.empty() || ctg < 0 || ctg >= (int32_t)I.size()) return -1;
	const std::vector<JuncIntv> &r = I[ctg];
	int32_t left = 0, right = (int32_t)r.size();
	while (right > left) {
		const int32_t mid = left + ((right - left) >> 1);
		if (r[mid].st >= st) right = mid; else left = mid + 1;
	}
	for (int32_t i = left; i < (int32_t)r.size(); ++i) {
		if (r[i].st >= en) break;                                      // (sorted by start: nothing further right can lie inside; the reference scans on)
		if (st <= r[i].st && en >= r[i].en && r[i].strand != 0) {
			if (r[i].strand > 0) s[r[i].st - st] |= 1, s[r[i].en - 1 - st] |= 2;
			else s[r[i].st - st] |= 8, s[r[i].en - 1 - st] |= 4;
		}
	}
	return left;
}

} // namespace wm
