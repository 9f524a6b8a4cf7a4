// tests/host_harness/prof_main.cpp — TEST INFRASTRUCTURE ONLY: host-glue profiler.
// Maps a read set with the product's host mapper twice. Pass 1 runs the oracle-backed device operations and RECORDS the result of every
// window / alignment request under a key made of the request's content; pass 2..N REPLAY the recorded results (any number of worker threads, any
// batching: the key does not depend on the order), so their CPU time — and the sampling profile taken over them (tools/sprof) — is the host glue
// alone: fibers, hit / align logic, request handling. That is the part of the product a GPU box's few host cores have to carry.
//   tests/host_harness/prof.sh            (build + run; SPROF=1: with the sampling profiler, symbolised report)
#include "harness.cpp"
#include <chrono>
#include <unordered_map>
#include <sys/resource.h>
#include <signal.h>
#include <malloc.h>

static inline uint64_t mix64(uint64_t h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); h *= 0xff51afd7ed558ccdull; return h ^ (h >> 29); }
static uint64_t hash_bytes(const void *p, size_t n, uint64_t h)
{
	const uint8_t *b = (const uint8_t*)p;
	size_t i = 0;
	for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, b + i, 8); h = mix64(h, v); }
	uint64_t v = 0;
	if (i < n) memcpy(&v, b + i, n - i);
	return mix64(h, v ^ n);
}

struct ReplayOps : DeviceOps {
	OracleOps *inner = 0;
	bool replay = false;
	double latency_ms = 0;                  // replay: a batched call sleeps this long (the device's part), so that the hub's batching sees realistic queues
	struct WR { std::vector<m128> a; std::vector<uint64_t> u; int rep_len, n_anchors; };
	struct KR { wm_ksw_result_t ez; std::vector<uint32_t> cigar; bool has_zd; wm_zd_t zd; };
	std::unordered_map<uint64_t, WR> wr;
	std::unordered_map<uint64_t, KR> kr;
	std::mutex mu;
	std::atomic<uint64_t> n_calls[2], n_reqs[2], miss{0};
	ReplayOps() { for (int i = 0; i < 2; ++i) n_calls[i] = 0, n_reqs[i] = 0; }
	int max_inflight() const override { return 6; }
	bool waits_asleep() const override { return replay; }
	bool load_reads(const uint8_t *codes, size_t n, int slot, int64_t *base) override { return inner->load_reads(codes, n, slot, base); }
	static uint64_t key(const WindowReq &r)
	{
		uint64_t h = mix64(1, (uint64_t)r.len);
		h = r.dev_off >= 0 ? mix64(h, (uint64_t)r.dev_off) : r.len > 0 ? hash_bytes(r.seq, (size_t)r.len, h) : h;
		h = hash_bytes(r.pre.data(), r.pre.size() * sizeof(m128), h);
		const int p[9] = { r.max_occ, r.max_dist_x, r.min_dist_x, r.max_dist_y, r.bw, r.max_skip, r.max_iter, r.min_cnt, r.min_sc };
		return hash_bytes(p, sizeof(p), mix64(h, (uint64_t)r.flag));
	}
	static uint64_t key(const KswReq &r)
	{
		const int64_t p[12] = { r.qwin_off, r.qwin_len, r.q_pos, r.rid, r.t_pos, r.ql, r.tl, r.step, r.w, r.zdrop, r.end_bonus, r.flag };
		return hash_bytes(p, sizeof(p), 2);
	}
	void nap() { if (latency_ms > 0) std::this_thread::sleep_for(std::chrono::duration<double, std::milli>(latency_ms)); }
	void sketch_batch(int w, int k, std::vector<SketchReq*> &reqs) override { inner->sketch_batch(w, k, reqs); }
	void seed_batch(std::vector<SeedReq*> &reqs) override { inner->seed_batch(reqs); }
	void chain_batch(std::vector<ChainReq*> &reqs) override { inner->chain_batch(reqs); }
	void exts2_batch(const wm_ksw_score_t &sc, int noncan, int junc_bonus, std::vector<KswReq*> &reqs) override { inner->exts2_batch(sc, noncan, junc_bonus, reqs); }
	void window_batch(int w, int k, std::vector<WindowReq*> &reqs) override
	{
		++n_calls[0]; n_reqs[0] += reqs.size();
		if (!replay) {
			std::vector<uint64_t> keys;
			for (WindowReq *r : reqs) keys.push_back(key(*r));          // (before the call: `pre` is consumed)
			DeviceOps::window_batch(w, k, reqs);
			std::lock_guard<std::mutex> lk(mu);
			for (size_t i = 0; i < reqs.size(); ++i) wr[keys[i]] = WR{ reqs[i]->a, reqs[i]->u, reqs[i]->rep_len, reqs[i]->n_anchors };
			return;
		}
		nap();
		for (WindowReq *r : reqs) {
			auto it = wr.find(key(*r));
			if (it == wr.end()) { ++miss; continue; }
			r->a = it->second.a; r->u = it->second.u; r->rep_len = it->second.rep_len; r->n_anchors = it->second.n_anchors;
		}
	}
	void ksw_batch(const wm_ksw_score_t &sc, std::vector<KswReq*> &reqs) override
	{
		++n_calls[1]; n_reqs[1] += reqs.size();
		if (!replay) {
			inner->ksw_batch(sc, reqs);
			std::lock_guard<std::mutex> lk(mu);
			for (KswReq *r : reqs) kr[key(*r)] = KR{ r->ez, r->cigar, r->has_zd, r->zd };
			return;
		}
		nap();
		for (KswReq *r : reqs) {
			auto it = kr.find(key(*r));
			if (it == kr.end()) { ++miss; continue; }
			r->ez = it->second.ez; r->cigar = it->second.cigar; r->has_zd = it->second.has_zd; r->zd = it->second.zd;
		}
	}
};

static double cpu_now() { rusage ru; getrusage(RUSAGE_SELF, &ru); return ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec); }

int main(int argc, char **argv)
{
	if (!getenv("NO_MALLOPT")) { mallopt(M_TOP_PAD, 64 << 20); mallopt(M_TRIM_THRESHOLD, 0x7fffffff); mallopt(M_MMAP_THRESHOLD, 32 << 20); }   // (what libwmgpu.so sets for the product)
	if (argc < 4) { fprintf(stderr, "usage: prof_main ref.fa rep.txt reads.fa [n_reads] [preset]   (env: THREADS, REPLAYS, LATENCY_MS, FORMAT=1)\n"); return 1; }
	Harness *h = (Harness*)h_index_build(argv[1], argv[2], 15, 50, 8);
	std::vector<std::string> names, seqs; std::string err;
	if (read_fastx(argv[3], names, seqs, 0, 0, err) < 0) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
	const size_t n = argc > 4 && atoi(argv[4]) > 0 ? std::min<size_t>(seqs.size(), atoi(argv[4])) : seqs.size();
	IdxOpt io; MapOpt mo;
	set_preset(0, io, mo); set_preset(argc > 5 ? argv[5] : "map-ont", io, mo);
	mo.flag |= 0x4 | 0x20;
	mapopt_update(mo, h->idx);
	OracleOps ops; ops.idx = &h->idx; ops.bloom = h->bloom; ops.opt = &mo;
	ReplayOps rp; rp.inner = &ops;
	std::vector<ReadIn> reads(n);
	uint64_t bases = 0;
	for (size_t i = 0; i < n; ++i) { reads[i].name = names[i]; reads[i].seq = seqs[i]; bases += seqs[i].size(); }
	std::vector<ReadOut> out;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const int threads = getenv("THREADS") ? atoi(getenv("THREADS")) : 1;
	double t0 = now();
	map_batch(h->idx, mo, &rp, reads, out, 0, 8);
	const double t_rec = now() - t0;
	fprintf(stderr, "record pass: %zu reads, %.1f s, %zu window + %zu ksw results kept\n", n, t_rec, rp.wr.size(), rp.kr.size());
	{   // what the glue has to digest per read
		uint64_t nu = 0, na = 0, nc = 0, cells = 0; size_t big = 0;
		for (auto &kv : rp.wr) { nu += kv.second.u.size(); na += kv.second.a.size(); big = std::max(big, kv.second.u.size()); }
		for (auto &kv : rp.kr) nc += kv.second.cigar.size();
		fprintf(stderr, "per read: %.1f chains (largest window %zu), %.0f chained anchors, %.0f CIGAR ops returned by the alignments\n", (double)nu / n, big, (double)na / n, (double)nc / n);
	}
	rp.replay = true;
	if (prof_on()) { std::lock_guard<std::mutex> g(prof_mutex()); for (ProfSlot &sl : prof_slots()) sl.ms = 0, sl.n = 0; }      // (region timers: the replays only)
	rp.latency_ms = getenv("LATENCY_MS") ? atof(getenv("LATENCY_MS")) : 0;
	const bool do_format = getenv("FORMAT") && atoi(getenv("FORMAT"));
	const int n_replays = getenv("REPLAYS") ? atoi(getenv("REPLAYS")) : 3;
	for (int i = 0; i < 2; ++i) rp.n_calls[i] = 0, rp.n_reqs[i] = 0;
	std::vector<ReadOut> out2;
	std::vector<std::string> texts(n);
	const std::function<void(size_t)> fmt = [&](size_t i) { texts[i].clear(); write_read(texts[i], h->idx, reads[i], out2[i], mo.flag); };
	if (getenv("SPROF_MARK")) raise(SIGUSR2);          // (tools/sprof: samples are kept from here on)
	const double c0 = cpu_now();
	t0 = now();
	for (int rep = 0; rep < n_replays; ++rep) { out2.clear(); map_batch(h->idx, mo, &rp, reads, out2, 0, threads, do_format ? &fmt : 0); }
	const double t_rep = (now() - t0) / n_replays, c_rep = (cpu_now() - c0) / n_replays;
	size_t nh = 0, nh1 = 0; for (auto &o : out2) nh += o.regs.size(); for (auto &o : out) nh1 += o.regs.size();
	fprintf(stderr, "REPLAY (host glue only, %d worker thread(s), %d pass(es)): wall %.3f s, CPU %.3f s = %.1f us CPU per read = %.2f CPU-s per Gbase | hits %zu (record pass %zu) misses %llu\n",
	        threads, n_replays, t_rep, c_rep, c_rep * 1e6 / n, c_rep / (bases * 1e-9), nh, nh1, (unsigned long long)rp.miss.load());
	fprintf(stderr, "calls per pass: window %.0f ksw %.0f | requests per read: window %.1f ksw %.1f\n", (double)rp.n_calls[0] / n_replays, (double)rp.n_calls[1] / n_replays,
	        (double)rp.n_reqs[0] / n_replays / n, (double)rp.n_reqs[1] / n_replays / n);
	if (prof_on()) prof_report(stderr);
	return nh == nh1 && rp.miss.load() == 0 ? 0 : 2;
}
