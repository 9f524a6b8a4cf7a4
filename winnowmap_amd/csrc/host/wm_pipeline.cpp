// wm_pipeline.cpp — see wm_pipeline.h
#include "wm_pipeline.h"
#include "wm_hit.h"
#include "wm_format.h"
#include <thread>
#include <mutex>
#include <condition_variable>
#include <algorithm>
#include <chrono>
#include <memory>
#include <atomic>
#include <map>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>
#include <unistd.h>

namespace wm {

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// a one-slot mailbox between two pipeline stages
template <class T> struct Slot {
	std::mutex mu; std::condition_variable cv; std::unique_ptr<T> item; bool closed = false;
	void put(std::unique_ptr<T> x) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !item; }); item = std::move(x); cv.notify_all(); }
	std::unique_ptr<T> take() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return item || closed; }); std::unique_ptr<T> x = std::move(item); cv.notify_all(); return x; }
	void close() { std::unique_lock<std::mutex> lk(mu); closed = true; cv.notify_all(); }
};
}

int default_lanes()
{
	const char *le = getenv("WM_MAP_LANES");                                 // mini-batches in flight per mapper: 1 .. 4, default 2
	const int n = le ? atoi(le) : 2;
	return n < 1 ? 1 : n > 4 ? 4 : n;
}

int map_file(const std::string &reads_path, int64_t mini_batch_bases, bool with_qual, const MapFn &map_fn, FILE *out, FileStats *st, std::string &err, int n_lanes_arg)
{
	return map_file_id(reads_path, mini_batch_bases, with_qual, [&](std::vector<ReadIn> &b, std::string &t, int lane, uint64_t) { return map_fn(b, t, lane); }, out, st, err, n_lanes_arg);
}

int map_file_id(const std::string &reads_path, int64_t mini_batch_bases, bool with_qual, const MapFnId &map_fn, FILE *out, FileStats *st, std::string &err, int n_lanes_arg)
{
	FastxReader rd;
	if (rd.open(reads_path, err) < 0) return -1;
	if (mini_batch_bases <= 0) mini_batch_bases = 1000000000;               // src/options.c:50
	typedef std::vector<ReadIn> Batch;
	struct Item { uint64_t id; Batch reads; };
	Slot<Item> to_map;
	FileStats fs;
	std::mutex fs_mu;
	std::atomic<int> rc(0);
	std::atomic<bool> stop(false), io_error(false);
	// finished mini-batches wait here for their turn: the file lists them in input order
	std::mutex omu; std::condition_variable ocv;
	std::map<uint64_t, std::unique_ptr<std::string>> done;
	const int n_lanes = n_lanes_arg > 0 ? n_lanes_arg : default_lanes();
	uint64_t next_out = 0; int lanes_running = n_lanes;                 // (set BEFORE the writer starts: it leaves when no lane is running and nothing is queued)
	std::thread reader([&]() {
		uint64_t id = 0;
		for (;;) {
			if (stop.load()) break;                                            // mapper or writer failed: stop parsing the input
			const double t0 = now_s();
			std::unique_ptr<Batch> b(new Batch());
			if (rd.next_batch(mini_batch_bases, with_qual, *b) == 0) break;
			// longest read first, ties by higher input index (std::greater on (length, index), src/map.c:1124-1143)
			std::vector<std::pair<int, int>> key(b->size());
			for (size_t i = 0; i < b->size(); ++i) key[i] = std::make_pair((int)(*b)[i].seq.size(), (int)i);
			std::sort(key.begin(), key.end(), std::greater<std::pair<int, int>>());
			std::unique_ptr<Item> s(new Item());
			s->id = id++; s->reads.resize(b->size());
			for (size_t i = 0; i < key.size(); ++i) s->reads[i] = std::move((*b)[key[i].second]);
			fs.t_read += now_s() - t0;
			to_map.put(std::move(s));
		}
		to_map.close();
	});
	std::thread writer([&]() {
		for (;;) {
			std::unique_ptr<std::string> t;
			{
				std::unique_lock<std::mutex> lk(omu);
				ocv.wait(lk, [&] { return done.count(next_out) || (lanes_running == 0 && done.empty()) || (lanes_running == 0 && rc.load() != 0); });
				auto it = done.find(next_out);
				if (it == done.end()) break;
				t = std::move(it->second); done.erase(it); ++next_out;
				ocv.notify_all();
			}
			const double t0 = now_s();
			// a full disk / closed pipe must not yield a silently truncated file (the reference aborts in mm_err_puts)
			if (out && !io_error.load() && !t->empty() && fwrite(t->data(), 1, t->size(), out) != t->size()) { io_error = true; stop = true; }
			fs.t_write += now_s() - t0;
		}
		if (out && (fflush(out) != 0 || ferror(out))) io_error = true;
	});
	auto lane_fn = [&](int lane) {
		for (;;) {
			std::unique_ptr<Item> b = to_map.take();
			if (!b) break;
			const double t0 = now_s();
			std::unique_ptr<std::string> text(new std::string());
			int r = 0;
			if (rc.load() == 0 && !io_error.load()) {                         // after an error: drain what the reader already queued
				try { r = map_fn(b->reads, *text, lane, b->id); }
				catch (const std::exception &e) { r = -1; std::lock_guard<std::mutex> lk(fs_mu); if (err.empty()) err = std::string("mapping failed: ") + e.what(); }
				catch (...) { r = -1; }                                       // (a lane is a thread of its own: nothing may escape it)
			}
			if (r != 0) { int z = 0; rc.compare_exchange_strong(z, r); stop = true; std::lock_guard<std::mutex> lk(omu); ocv.notify_all(); }
			uint64_t nb = 0;
			for (const ReadIn &x : b->reads) nb += x.seq.size();
			{ std::lock_guard<std::mutex> lk(fs_mu); fs.t_map += now_s() - t0; fs.n_batches += 1; fs.n_reads += b->reads.size(); fs.n_bases += nb; }
			if (rc.load() == 0) {
				std::unique_lock<std::mutex> lk(omu);
				ocv.wait(lk, [&] { return b->id < next_out + (uint64_t)n_lanes || rc.load() != 0; });          // (at most one finished text per lane waits for the writer)
				done[b->id] = std::move(text);
				ocv.notify_all();
			}
		}
		std::lock_guard<std::mutex> lk(omu);
		--lanes_running;
		ocv.notify_all();
	};
	std::vector<std::thread> lanes;
	for (int l = 1; l < n_lanes; ++l) lanes.emplace_back(lane_fn, l);
	lane_fn(0);
	for (std::thread &t : lanes) t.join();
	reader.join();
	writer.join();
	if (st) *st = fs;
	int ret = rc.load();
	if (ret && err.empty()) err = "mapping failed";
	else if (io_error.load()) { err = "write error on the output file"; ret = -2; }
	return ret;
}

namespace {
// hits of one index part on disk: blobs of (n reads; per read rep_len, frag_gap, n regs; per reg the fixed fields + CIGAR), keyed by mini-batch
struct HitSpill {
	int fd = -1; uint64_t end = 0; std::string error;
	std::mutex mu; std::map<uint64_t, std::pair<uint64_t, uint64_t>> at;      // mini-batch -> (offset, bytes)
	~HitSpill() { if (fd >= 0) close(fd); }
	bool open(std::string &err)
	{
		const char *td = getenv("TMPDIR");
		std::string path = std::string(td && td[0] ? td : "/tmp") + "/wm_split_XXXXXX";
		fd = mkstemp(&path[0]);
		if (fd < 0) { err = "cannot create a temporary file for the hits of an index part in " + path; return false; }
		unlink(path.c_str());                                               // anonymous: gone with the descriptor, whatever happens
		return true;
	}
	static void w32(std::string &b, uint32_t v) { b.append((const char*)&v, 4); }
	bool put(uint64_t id, const std::vector<ReadOut> &o)
	{
		const size_t fixed = offsetof(Reg, cigar);
		std::string b;
		w32(b, (uint32_t)o.size());
		for (const ReadOut &r : o) {
			w32(b, (uint32_t)r.rep_len); w32(b, (uint32_t)r.frag_gap); w32(b, (uint32_t)r.regs.size());
			for (const Reg &g : r.regs) {
				b.append((const char*)&g, fixed);
				w32(b, (uint32_t)g.cigar.size());
				b.append((const char*)g.cigar.data(), g.cigar.size() * 4);
			}
		}
		std::lock_guard<std::mutex> lk(mu);
		size_t done = 0;
		while (done < b.size()) {
			const ssize_t k = pwrite(fd, b.data() + done, b.size() - done, (off_t)(end + done));
			if (k <= 0) { error = "write error on the temporary file of an index part's hits"; return false; }
			done += (size_t)k;
		}
		at[id] = std::make_pair(end, (uint64_t)b.size());
		end += b.size();
		return true;
	}
	bool get(uint64_t id, std::vector<ReadOut> &o)
	{
		std::pair<uint64_t, uint64_t> w;
		{ std::lock_guard<std::mutex> lk(mu); auto it = at.find(id); if (it == at.end()) return false; w = it->second; }
		std::string b(w.second, '\0');
		size_t done = 0;
		while (done < b.size()) {
			const ssize_t k = pread(fd, &b[done], b.size() - done, (off_t)(w.first + done));
			if (k <= 0) return false;
			done += (size_t)k;
		}
		const size_t fixed = offsetof(Reg, cigar);
		size_t p = 0;
		auto r32 = [&](uint32_t &v) { if (p + 4 > b.size()) return false; memcpy(&v, &b[p], 4); p += 4; return true; };
		uint32_t n;
		if (!r32(n)) return false;
		o.assign(n, ReadOut());
		for (ReadOut &r : o) {
			uint32_t rl, fg, nr;
			if (!r32(rl) || !r32(fg) || !r32(nr)) return false;
			r.rep_len = (int)rl; r.frag_gap = (int)fg; r.regs.resize(nr);
			for (Reg &g : r.regs) {
				uint32_t nc;
				if (p + fixed > b.size()) return false;
				memcpy((void*)&g, &b[p], fixed); p += fixed;
				if (!r32(nc) || p + (size_t)nc * 4 > b.size()) return false;
				g.cigar.resize(nc);
				if (nc) memcpy(g.cigar.data(), &b[p], (size_t)nc * 4);
				p += (size_t)nc * 4;
			}
		}
		return p == b.size();
	}
};
}

// ---- a reference indexed in parts, one part at a time ----
struct SplitRun::Impl {
	std::string reads_path; int64_t mini_batch_bases; MapOpt opt; int k; bool with_qual;
	std::vector<std::unique_ptr<HitSpill>> spill;
	std::vector<int> n_seq;
	Index dict;
	FileStats fs_all;
};
SplitRun::SplitRun(const std::string &reads_path, int64_t mini_batch_bases, const MapOpt &opt, int k, int w) : p_(new Impl())
{
	p_->reads_path = reads_path; p_->mini_batch_bases = mini_batch_bases; p_->opt = opt; p_->k = k; p_->with_qual = (opt.flag & F_OUT_SAM) != 0;
	p_->dict.k = k; p_->dict.w = w;
}
SplitRun::~SplitRun() { delete p_; }
const Index &SplitRun::dict() const { return p_->dict; }
int SplitRun::n_parts() const { return (int)p_->spill.size(); }

int SplitRun::add_part(const std::vector<RefSeq> &contigs, const std::function<int(std::vector<ReadIn> &batch, std::vector<ReadOut> &out, int lane)> &map_part, std::string &err)
{
	if (p_->opt.flag & (F_OUT_CS | F_OUT_MD)) { err = "--cs or --MD doesn't work with a reference indexed in parts"; return -1; }      // src/options.c:139-141
	// The hits of one index part — one ReadOut per read, CIGARs included — go to an anonymous temporary file per part, one blob per mini-batch, as the
	// reference spills them to <prefix>.NNNN.tmp (src/map.c:1174-1190, read back in merge_hits, src/map.c:1050-1105): this flow exists for references too big for one index, and reads x parts hits
	// do not belong in RAM. In memory: where each blob lies. $TMPDIR, else /tmp.
	std::unique_ptr<HitSpill> sp(new HitSpill());
	if (!sp->open(err)) return -1;
	HitSpill &spill = *sp;
	FileStats fs;
	const int rc = map_file_id(p_->reads_path, p_->mini_batch_bases, p_->with_qual, [&](std::vector<ReadIn> &batch, std::string &, int lane, uint64_t id) {
		std::vector<ReadOut> o;
		const int r = map_part(batch, o, lane);
		if (r) return r;
		return spill.put(id, o) ? 0 : -1;
	}, 0, &fs, err);
	if (rc) { if (err.empty() || err == "mapping failed") err = spill.error.empty() ? err : spill.error; return rc; }
	p_->fs_all.t_read += fs.t_read; p_->fs_all.t_map += fs.t_map;
	for (const RefSeq &r : contigs) p_->dict.seq.push_back(r);
	p_->n_seq.push_back((int)contigs.size());
	p_->spill.push_back(std::move(sp));
	return 0;
}

// the merge pass (merge_hits, src/map.c:1050-1105)
int SplitRun::finish(FILE *out, FileStats *st, std::string &err)
{
	const int n_parts = (int)p_->spill.size();
	const MapOpt &opt = p_->opt;
	const int k = p_->k;
	std::vector<std::unique_ptr<HitSpill>> &spill = p_->spill;
	std::vector<int> rid_shift(n_parts > 0 ? n_parts : 1, 0);
	for (int j = 1; j < n_parts; ++j) rid_shift[j] = rid_shift[j - 1] + p_->n_seq[j - 1];
	FileStats fs;
	std::string merge_err;
	const int rc = map_file_id(p_->reads_path, p_->mini_batch_bases, p_->with_qual, [&](std::vector<ReadIn> &batch, std::string &text, int, uint64_t id) {
		std::vector<std::vector<ReadOut>> ph(n_parts);                    // this mini-batch's hits, part by part
		for (int j = 0; j < n_parts; ++j)
			if (!spill[j]->get(id, ph[j]) || ph[j].size() != batch.size()) {
				if (merge_err.empty()) merge_err = !spill[j]->error.empty() ? spill[j]->error : "the hits of index part " + std::to_string(j) + " for mini-batch " + std::to_string(id) + " could not be read back from their temporary file";
				return -1;
			}
		for (size_t i = 0; i < batch.size(); ++i) {
			ReadOut m;
			for (int j = 0; j < n_parts; ++j) {
				ReadOut &p = ph[j][i];
				if (p.rep_len > m.rep_len) m.rep_len = p.rep_len;
				if (j == 0) m.frag_gap = p.frag_gap;
				for (Reg &r : p.regs) { r.rid += rid_shift[j]; m.regs.push_back(std::move(r)); }
				std::vector<Reg>().swap(p.regs);
			}
			hit_sort(m.regs);
			set_parent(opt.mask_level, opt.mask_len, m.regs, opt.a * 2 + opt.b, (opt.flag & F_HARD_MLEVEL) != 0);
			if (!(opt.flag & F_ALL_CHAINS)) {
				select_sub(opt.pri_ratio, k * 2, opt.best_n, m.regs);
				set_sam_pri(m.regs);
			}
			set_mapq(m.regs, opt.min_chain_score, opt.a, m.rep_len, (opt.flag & F_SR) != 0);
			m.rep_len = 0;               // merge_hits keeps the merged rep_len in a local (src/map.c:1067) and never stores s->rep_len[k]: the records carry rl:i:0
			write_read(text, p_->dict, batch[i], m, opt.flag);
		}
		return 0;
	}, out, &fs, err);
	if (rc && !merge_err.empty() && (err.empty() || err == "mapping failed")) err = merge_err;
	if (st) { *st = fs; st->t_map += p_->fs_all.t_map; st->t_read += p_->fs_all.t_read; }
	return rc;
}

int map_file_split(const std::string &reads_path, int64_t mini_batch_bases, const MapOpt &opt, int k, const Index &dict, const std::vector<SplitPart> &parts,
                   const std::function<int(int part)> &begin_part,
                   const std::function<int(int part, std::vector<ReadIn> &batch, std::vector<ReadOut> &out, int lane)> &map_part,
                   FILE *out, FileStats *st, std::string &err)
{
	if (opt.flag & (F_OUT_CS | F_OUT_MD)) { err = "--cs or --MD doesn't work with a reference indexed in parts"; return -1; }      // src/options.c:139-141
	SplitRun run(reads_path, mini_batch_bases, opt, k, dict.w);
	size_t at = 0;
	for (int j = 0; j < (int)parts.size(); ++j) {
		if (begin_part(j)) { err = "cannot set up index part " + std::to_string(j); return -1; }
		std::vector<RefSeq> contigs(dict.seq.begin() + at, dict.seq.begin() + at + parts[j].n_seq);
		at += parts[j].n_seq;
		const int rc = run.add_part(contigs, [&](std::vector<ReadIn> &batch, std::vector<ReadOut> &o, int lane) { return map_part(j, batch, o, lane); }, err);
		if (rc) return rc;
	}
	return run.finish(out, st, err);
}

// ---- the parts of a reference FASTA, built when their turn comes (mm_idx_reader_read, src/index.c:655-676 with mm_idx_gen's reading rule, :289-300) ----
struct IndexPartReader::Impl { FastxReader rd; bool open = false, eof = false; };
IndexPartReader::IndexPartReader() : p_(new Impl()) {}
IndexPartReader::~IndexPartReader() { delete p_; }
int IndexPartReader::open(const std::string &fasta, std::string &err)
{
	if (p_->rd.open(fasta, err) < 0) return -1;
	p_->open = true; p_->eof = false;
	return 0;
}
int IndexPartReader::next(uint64_t batch_bases, std::vector<std::string> &names, std::vector<std::string> &seqs)
{
	names.clear(); seqs.clear();
	if (!p_->open || p_->eof) return 0;
	const uint64_t mini = batch_bases < 50000000ULL ? batch_bases : 50000000ULL;      // mm_idx_gen: min(mini_batch_size, batch_size), src/index.c:383
	uint64_t sum = 0;
	while (sum <= batch_bases) {                                        // step 0 of the reference's pipeline: another mini-batch unless sum_len > batch_size (:295)
		std::vector<ReadIn> chunk;
		if (p_->rd.next_batch((int64_t)mini, false, chunk) == 0) { p_->eof = true; break; }      // mm_bseq_read: sequences until the chunk reaches `mini` bases
		for (ReadIn &r : chunk) { sum += r.seq.size(); names.push_back(std::move(r.name)); seqs.push_back(std::move(r.seq)); }
	}
	return (int)seqs.size();
}

} // namespace wm
