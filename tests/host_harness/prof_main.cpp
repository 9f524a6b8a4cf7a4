// tests/host_harness/prof_main.cpp — TEST INFRASTRUCTURE ONLY: host-glue profiler.
// Maps a read set twice with the product's host mapper: pass 1 runs the oracle-backed device operations and RECORDS every
// result in request order; pass 2 REPLAYS the recorded results, so its run time (and its gprof profile, when built with -pg)
// is the host glue alone — fibers, hit/align logic, request packing — which is what bounds throughput on a GPU box whose
// container has few host cores. Build + run: tests/host_harness/prof.sh
#include "harness.cpp"
#include <chrono>

struct ReplayOps : DeviceOps {
	OracleOps *inner = 0;
	bool replay = false;
	std::vector<std::vector<m128>> sk, sd_a, ch_a; std::vector<int> sd_rep; std::vector<std::vector<uint64_t>> ch_u;
	struct KR { wm_ksw_result_t ez; std::vector<uint32_t> cigar; };
	std::vector<KR> kr;
	size_t i_sk = 0, i_sd = 0, i_ch = 0, i_kr = 0;
	uint64_t n_calls[4] = {0, 0, 0, 0}, n_reqs[4] = {0, 0, 0, 0}, bytes_q = 0;
	void sketch_batch(int w, int k, std::vector<SketchReq*> &reqs) override
	{
		++n_calls[0]; n_reqs[0] += reqs.size();
		if (!replay) { inner->sketch_batch(w, k, reqs); for (SketchReq *r : reqs) sk.push_back(r->mini); return; }
		for (SketchReq *r : reqs) r->mini = sk[i_sk++];
	}
	void seed_batch(std::vector<SeedReq*> &reqs) override
	{
		++n_calls[1]; n_reqs[1] += reqs.size();
		if (!replay) { inner->seed_batch(reqs); for (SeedReq *r : reqs) { sd_a.push_back(r->a); sd_rep.push_back(r->rep_len); } return; }
		for (SeedReq *r : reqs) { r->a = sd_a[i_sd]; r->rep_len = sd_rep[i_sd++]; }
	}
	void chain_batch(std::vector<ChainReq*> &reqs) override
	{
		++n_calls[2]; n_reqs[2] += reqs.size();
		if (!replay) { inner->chain_batch(reqs); for (ChainReq *r : reqs) { ch_a.push_back(r->a); ch_u.push_back(r->u); } return; }
		for (ChainReq *r : reqs) { r->a = ch_a[i_ch]; r->u = ch_u[i_ch++]; }
	}
	void ksw_batch(const wm_ksw_score_t &sc, std::vector<KswReq*> &reqs) override
	{
		++n_calls[3]; n_reqs[3] += reqs.size();
		if (!replay) { inner->ksw_batch(sc, reqs); for (KswReq *r : reqs) { kr.push_back(KR{ r->ez, r->cigar }); } return; }
		for (KswReq *r : reqs) { r->ez = kr[i_kr].ez; r->cigar = kr[i_kr++].cigar; bytes_q += r->ql + r->tl; }
	}
};

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: prof_main ref.fa rep.txt reads.fa [n_reads] [preset]\n"); return 1; }
	Harness *h = (Harness*)h_index_build(argv[1], argv[2], 15, 50, 8);
	std::vector<std::string> names, seqs; std::string err;
	if (read_fastx(argv[3], names, seqs, 0, 0, err) < 0) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
	const size_t n = argc > 4 && atoi(argv[4]) > 0 ? std::min<size_t>(seqs.size(), atoi(argv[4])) : seqs.size();
	IdxOpt io; MapOpt mo;
	set_preset(0, io, mo); set_preset(argc > 5 ? argv[5] : "map-ont", io, mo);
	mo.flag |= 0x4 | 0x20;
	OracleOps ops; ops.idx = &h->idx; ops.bloom = h->bloom; ops.opt = &mo;
	ReplayOps rp; rp.inner = &ops;
	std::vector<ReadIn> reads(n);
	for (size_t i = 0; i < n; ++i) { reads[i].name = names[i]; reads[i].seq = seqs[i]; }
	std::vector<ReadOut> out;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t0 = now();
	map_batch(h->idx, mo, &rp, reads, out, 0, 1);
	const double t_rec = now() - t0;
	rp.replay = true;
	memset(rp.n_calls, 0, sizeof(rp.n_calls)); memset(rp.n_reqs, 0, sizeof(rp.n_reqs));
	t0 = now();
	std::vector<ReadOut> out2;
	const int n_replays = getenv("REPLAYS") ? atoi(getenv("REPLAYS")) : 1;        // (more replays = more samples of the glue in a gprof profile)
	std::vector<double> each;
	for (int rep = 0; rep < n_replays; ++rep) { const double r0 = now(); rp.i_sk = rp.i_sd = rp.i_ch = rp.i_kr = 0; out2.clear(); map_batch(h->idx, mo, &rp, reads, out2, 0, 1); each.push_back(now() - r0); }
	std::sort(each.begin(), each.end());
	const double t_rep = each[each.size() / 2];                                    // median replay (min: see below)
	fprintf(stderr, "replays %d: min %.3f ms/read, median %.3f ms/read\n", n_replays, each[0] * 1e3 / n, t_rep * 1e3 / n);
	size_t nh = 0; for (auto &o : out2) nh += o.regs.size();
	fprintf(stderr, "reads %zu  record pass %.2f s  REPLAY (host glue only) %.3f s = %.3f ms/read  hits %zu\n", n, t_rec, t_rep, t_rep * 1e3 / n, nh);
	fprintf(stderr, "flushes: sketch %llu seed %llu chain %llu ksw %llu | requests per read: sketch %.1f seed %.1f chain %.1f ksw %.1f | ksw seq bytes/read %.0f\n",
	        (unsigned long long)rp.n_calls[0], (unsigned long long)rp.n_calls[1], (unsigned long long)rp.n_calls[2], (unsigned long long)rp.n_calls[3],
	        (double)rp.n_reqs[0] / n, (double)rp.n_reqs[1] / n, (double)rp.n_reqs[2] / n, (double)rp.n_reqs[3] / n, (double)rp.bytes_q / n);
	if (prof_on()) prof_report(stderr);
	return 0;
}
