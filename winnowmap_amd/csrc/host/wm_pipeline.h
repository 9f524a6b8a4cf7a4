// wm_pipeline.h — the file-level loop around the mapper: mm_map_file_frag's three pipeline steps (read a mini-batch, map it,
// write its records; src/map.c:1107-1224) with the steps genuinely overlapped (the reference forces them to run one at a
// time, src/map.c:1259): a reader thread parses batch i+1 and a writer thread prints batch i-1 while batch i is on the GPU.
#pragma once
#include <functional>
#include <stdio.h>
#include "wm_mapper.h"

namespace wm {

class FastxReader {                      // FASTA/FASTQ, optionally gzip (src/bseq.c, src/kseq.h record semantics)
public:
	FastxReader();
	~FastxReader();
	int open(const std::string &fn, std::string &err);
	int next_batch(int64_t max_bases, bool with_qual, std::vector<ReadIn> &out);
	void close();
private:
	struct Impl;
	Impl *p_;
};

struct FileStats { uint64_t n_reads = 0, n_bases = 0, n_batches = 0; double t_read = 0, t_map = 0, t_write = 0; };

// map_fn(batch, text, lane): maps the reads of one mini-batch IN THE GIVEN ORDER and appends their records to text; returns 0 / error.
// Each mini-batch is ordered longest read first (ties: later read first) exactly like src/map.c:1124-1143, so the output
// file equals the reference's. Two mini-batches are mapped at a time (lane 0 and 1, two threads calling map_fn concurrently; WM_MAP_LANES=1:
// one): a mapping call ramps up and drains its pipeline of dependent device calls over several hundred milliseconds, which the other lane's
// steady state covers. Records are written in input order whatever lane finishes first.
typedef std::function<int(std::vector<ReadIn> &batch, std::string &text, int lane)> MapFn;
// n_lanes: mini-batches mapped at a time (threads calling map_fn concurrently with lane = 0 .. n_lanes - 1); 0 = default_lanes().
// A caller that owns several devices passes two lanes per device (wm_map_file_multi).
int default_lanes();        // WM_MAP_LANES (1 .. 4), default 2
int map_file(const std::string &reads_path, int64_t mini_batch_bases, bool with_qual, const MapFn &map_fn, FILE *out, FileStats *st, std::string &err, int n_lanes = 0);
// the same with the ordinal of the mini-batch (0, 1, …: the same reads file read again yields the same mini-batches in the same order)
typedef std::function<int(std::vector<ReadIn> &batch, std::string &text, int lane, uint64_t batch_id)> MapFnId;
int map_file_id(const std::string &reads_path, int64_t mini_batch_bases, bool with_qual, const MapFnId &map_fn, FILE *out, FileStats *st, std::string &err, int n_lanes = 0);

// A reference indexed in several parts (`-I` smaller than the reference, `--split-prefix`; src/main.c:398-429): the reads are mapped against one
// part after the other (begin_part(j) makes part j current — upload, mapper —, map_part maps one mini-batch against it), the hits of every read
// are kept, and a last pass over the reads merges them exactly like mm_split_merge / merge_hits (src/map.c:1050-1105): part order, contig ids
// shifted by the parts before, mm_hit_sort, mm_set_parent, mm_select_sub, mm_set_sam_pri, mm_set_mapq with the largest rep_len. `dict` = the
// contigs of all parts in order (names and lengths only). The reference spills the per-part hits to <prefix>.NNNN.tmp files (src/map.c:1174-1190); here each part has an
// anonymous temporary file under $TMPDIR (one blob per mini-batch), so that memory holds one mini-batch's hits per lane, not reads x parts. --cs / --MD are refused like the reference does (src/options.c:139-141).
// The flow ONE PART AT A TIME, as the reference's main runs it (src/main.c:398-429: read / build a part, map every read against it, destroy it, next):
// add_part maps the whole reads file against the part the caller has made current (contigs = its names and lengths) and spills the hits; the part
// may be destroyed afterwards. finish is the merge pass; dict() = the contigs of every part so far (what the SAM header lists).
class SplitRun {
public:
	SplitRun(const std::string &reads_path, int64_t mini_batch_bases, const MapOpt &opt, int k, int w);
	~SplitRun();
	SplitRun(const SplitRun&) = delete; SplitRun &operator=(const SplitRun&) = delete;
	int add_part(const std::vector<RefSeq> &contigs, const std::function<int(std::vector<ReadIn> &batch, std::vector<ReadOut> &out, int lane)> &map_part, std::string &err);
	int finish(FILE *out, FileStats *st, std::string &err);
	const Index &dict() const;
	int n_parts() const;
private:
	struct Impl;
	Impl *p_;
};
// the sequences of the next index part of a reference FASTA (mm_idx_gen's reading rule with batch_size = batch_bases, src/index.c:289-300,383): 0 = no more
class IndexPartReader {
public:
	IndexPartReader();
	~IndexPartReader();
	IndexPartReader(const IndexPartReader&) = delete; IndexPartReader &operator=(const IndexPartReader&) = delete;
	int open(const std::string &fasta, std::string &err);
	int next(uint64_t batch_bases, std::vector<std::string> &names, std::vector<std::string> &seqs);
private:
	struct Impl;
	Impl *p_;
};
// (every part given up front: the form of rounds 3-4, kept for callers that hold the parts anyway)
struct SplitPart { int n_seq; };
int map_file_split(const std::string &reads_path, int64_t mini_batch_bases, const MapOpt &opt, int k, const Index &dict, const std::vector<SplitPart> &parts,
                   const std::function<int(int part)> &begin_part,
                   const std::function<int(int part, std::vector<ReadIn> &batch, std::vector<ReadOut> &out, int lane)> &map_part,
                   FILE *out, FileStats *st, std::string &err);

} // namespace wm
