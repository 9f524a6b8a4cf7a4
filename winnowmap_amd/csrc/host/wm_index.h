// wm_index.h — reference index of the mapper, laid out flat so the same arrays are uploaded to HBM unchanged.
// Mirrors mm_idx_t (src/minimap.h:66-77) / mm_idx_bucket_t (src/index.c:33-38) by behaviour: a minimizer key
// maps to its reference positions in ascending order (src/index.c:88-105,239).
#pragma once
#include "wm_core.h"

namespace wm {

struct Bloom {                              // ext/bloom/bloom_filter.hpp as configured by src/index.c:411-423
	uint64_t table_bits = 0;
	uint32_t salt[2] = {0, 0};
	std::vector<uint8_t> bits;
	uint64_t n_inserted = 0;
	void init(uint64_t n_kmers);
	static uint32_t hash_ap8(uint64_t key, uint32_t h);
	void insert(uint64_t key);
	bool contains(uint64_t key) const;
};

uint64_t encode_kmer(const char *s, int k);                    // encodeKmer, src/index.c:362-376
double minimizer_order(uint64_t kmer, bool down_weighted);     // applyWeight, src/sketch.c:70-89

// mm_sketch (src/sketch.c:128-219). Appends to out.
void sketch(const char *seq, int len, int w, int k, uint32_t rid, const Bloom *bloom, std::vector<m128> &out, bool hpc = false);      // hpc: MM_I_HPC, src/sketch.c:152-163

struct RefSeq { std::string name; uint64_t offset; uint32_t len; };

struct JuncIntv { int32_t st, en, strand; };     // mm_idx_intv1_t (src/minimap.h): an annotated intron [st, en) and the strand of its transcript

struct Index {
	int k = 15, w = 50, flag = 0;
	std::vector<RefSeq> seq;                // mm_idx_seq_t
	std::vector<uint32_t> S;                // 4 bits per base, 8 per word (mm_seq4_set, src/mmpriv.h:29)
	uint64_t total_len = 0;
	// open-addressing table: slot i holds hkey[i] (UINT64_MAX = empty) and hval[i] = first<<32 | count into P
	int hbits = 0;
	std::vector<uint64_t> hkey, hval;
	std::vector<uint64_t> P;                // rid<<32 | lastPos<<1 | strand, ascending within a key
	Bloom bloom;
	uint64_t n_minimizers = 0, n_keys = 0;

	const uint64_t *get(uint64_t minier, int *n) const;            // mm_idx_get, src/index.c:88
	int getseq(uint32_t rid, uint32_t st, uint32_t en, uint8_t *out) const;   // mm_idx_getseq, src/index.c:161
	// runs of ambiguous bases (codes >= 4) in S as [begin, end) global base offsets, ascending: lets the aligner pick the kernel
	// variant without scanning the operands (a contig of plain ACGT answers in O(1))
	std::vector<std::pair<uint64_t, uint64_t>> n_runs;
	void scan_n_runs();
	bool has_n(uint32_t rid, uint32_t st, uint32_t en) const;
	// junction annotation (--junc-bed, mm_idx_bed_read / mm_idx_bed_junc, src/index.c:690-803): per contig, sorted by start
	std::vector<std::vector<JuncIntv>> I;
	bool has_junc() const { return !I.empty(); }
	int bed_junc(int32_t ctg, int32_t st, int32_t en, uint8_t *s) const;
	int32_t cal_max_occ(float f) const;                            // mm_idx_cal_max_occ, src/index.c:173-194
	static uint64_t slot_of(uint64_t key, int hbits) { return (key * 0x9E3779B97F4A7C15ULL) >> (64 - hbits); }
};

// names/seqs: reference contigs (ASCII). kmer_file: the -W list ("kmer count" lines) or empty.
// Returns 0, or -1 with err set (e.g. k-mer length mismatch, src/index.c:403-407).
// keys in (home slot, key) order -> the table by sequential linear probing (the canonical layout; the device build computes the same with a scan and
// calls this only for the few keys that wrap past the last slot). item(t, &key, &val) returns the home slot of the t-th key.
void index_table_insert(Index &ix, size_t n, const std::function<uint64_t(size_t, uint64_t*, uint64_t*)> &item);
int index_build(const IdxOpt &io, const std::vector<std::string> &names, const std::vector<std::string> &seqs,
                const std::string &kmer_file, int n_threads, Index &out, std::string &err);
int index_build_from_fasta(const IdxOpt &io, const std::string &fasta, const std::string &kmer_file, int n_threads, Index &out, std::string &err);
// `-I batch_bases` (src/main.c:193, src/index.c:289-300, 378-384): the reference reads the FASTA in mini-batches of min(50 M, batch_bases) bases
// (whole sequences, until the mini-batch reaches that size) and closes an index part once its sequences exceed batch_bases; every part is an
// index of its own (contig ids start at 0). Returns the number of parts (>= 1) or -1.
int index_build_parts_from_fasta(const IdxOpt &io, const std::string &fasta, const std::string &kmer_file, int n_threads, uint64_t batch_bases,
                                 std::vector<Index> &parts, std::string &err);

// mm_mapopt_update (src/options.c:71-82): the options that depend on the index (-f as a fraction → mid_occ, the --min-occ-floor clamp)
// and the implied MM_F_SPLICE bit
struct MapOpt;
void mapopt_update(MapOpt &opt, const Index &ix);
// mm_idx_bed_read (src/index.c:756-766): BED6 intervals, or with read_junc the introns between the blocks of BED12 records; 0 / -1 (cannot open)
int index_read_bed(Index &ix, const std::string &path, bool read_junc, std::string &err);

void index_table_from_minimizers(Index &ix, std::vector<m128> &all);
// the two halves of index_build around the sketching of the contigs (so that a device can do that part): bloom filter from the -W list +
// sequence table + 4-bit packing, then (key, position) records -> table
int index_begin(const IdxOpt &io, const std::vector<std::string> &names, const std::vector<std::string> &seqs, const std::string &kmer_file, int n_threads, Index &ix, std::string &err);
// the reference's index file format (winnowmap -d): written by either program, read by either program
int index_save_mmi(const Index &ix, const std::string &path, std::string &err);
int index_load_mmi(const std::string &path, const std::string &kmer_file, Index &ix, std::string &err);

} // namespace wm
