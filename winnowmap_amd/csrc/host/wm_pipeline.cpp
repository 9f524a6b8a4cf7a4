// wm_pipeline.cpp — see wm_pipeline.h
#include "wm_pipeline.h"
#include <thread>
#include <mutex>
#include <condition_variable>
#include <algorithm>
#include <chrono>
#include <memory>
#include <atomic>

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

int map_file(const std::string &reads_path, int64_t mini_batch_bases, bool with_qual, const MapFn &map_fn, FILE *out, FileStats *st, std::string &err)
{
	FastxReader rd;
	if (rd.open(reads_path, err) < 0) return -1;
	if (mini_batch_bases <= 0) mini_batch_bases = 1000000000;               // src/options.c:50
	typedef std::vector<ReadIn> Batch;
	Slot<Batch> to_map;
	Slot<std::string> to_write;
	FileStats fs;
	int rc = 0;
	std::atomic<bool> stop(false), io_error(false);
	std::thread reader([&]() {
		for (;;) {
			if (stop.load()) break;                                            // mapper or writer failed: stop parsing the input
			const double t0 = now_s();
			std::unique_ptr<Batch> b(new Batch());
			if (rd.next_batch(mini_batch_bases, with_qual, *b) == 0) break;
			// longest read first, ties by higher input index (std::greater on (length, index), src/map.c:1124-1143)
			std::vector<std::pair<int, int>> key(b->size());
			for (size_t i = 0; i < b->size(); ++i) key[i] = std::make_pair((int)(*b)[i].seq.size(), (int)i);
			std::sort(key.begin(), key.end(), std::greater<std::pair<int, int>>());
			std::unique_ptr<Batch> s(new Batch(b->size()));
			for (size_t i = 0; i < key.size(); ++i) (*s)[i] = std::move((*b)[key[i].second]);
			fs.t_read += now_s() - t0;
			to_map.put(std::move(s));
		}
		to_map.close();
	});
	std::thread writer([&]() {
		for (;;) {
			std::unique_ptr<std::string> t = to_write.take();
			if (!t) break;
			const double t0 = now_s();
			// a full disk / closed pipe must not yield a silently truncated file (the reference aborts in mm_err_puts)
			if (!io_error.load() && !t->empty() && fwrite(t->data(), 1, t->size(), out) != t->size()) { io_error = true; stop = true; }
			fs.t_write += now_s() - t0;
		}
		if (fflush(out) != 0 || ferror(out)) io_error = true;
	});
	for (;;) {
		std::unique_ptr<Batch> b = to_map.take();
		if (!b) break;
		const double t0 = now_s();
		std::unique_ptr<std::string> text(new std::string());
		if (rc == 0 && !io_error.load()) rc = map_fn(*b, *text);          // after an error: drain what the reader already queued
		if (rc != 0) stop = true;
		fs.t_map += now_s() - t0;
		fs.n_batches += 1; fs.n_reads += b->size();
		for (const ReadIn &r : *b) fs.n_bases += r.seq.size();
		if (rc == 0) to_write.put(std::move(text));
	}
	to_write.close();
	reader.join();
	writer.join();
	if (st) *st = fs;
	if (rc) err = "mapping failed";
	else if (io_error.load()) { err = "write error on the output file"; rc = -2; }
	return rc;
}

} // namespace wm
