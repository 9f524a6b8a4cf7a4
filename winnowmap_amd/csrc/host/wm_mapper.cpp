// wm_mapper.cpp — see wm_mapper.h.
#include "wm_mapper.h"
#include <thread>
#include <memory>
#include "wm_hit.h"
#include "wm_align.h"
#include "wm_fiber.h"
#include <math.h>
#include <algorithm>
#include <tuple>

namespace wm {

namespace {

struct Segment {                       // result of mapping one (sub)sequence: the body shared by stage 1, stage 2 and the fallback
	std::vector<Reg> regs;
	std::vector<m128> a;
	int rep_len = 0;
};

inline uint32_t read_hash(const char *qname, int qlen, int seed)
{   // src/map.c:355-357
	uint32_t h = qname ? x31_hash_string(qname) : 0;
	h ^= wang_hash32((uint32_t)qlen) + wang_hash32((uint32_t)seed);
	return wang_hash32(h);
}

// One window from its codes to aligned regions: sketch → seed → [anchors handed in] → chain in ONE device call (WindowReq), then region
// generation + alignment + MAPQ on the chains (src/map.c:69-84, 222-254, 375-430 and :880-933)
void window_and_align(Scheduler &sch, const Index &idx, const MapOpt &o, float gap_scale, const uint8_t *seq_codes, int64_t seq_dev_off, bool sketch_it,
                      const uint8_t *codes, int64_t dev_off, int qlen, uint32_t hash, std::vector<m128> &&pre, int *rep_len_io, Segment &out, int *frag_gap)
{
	const int max_gap_qry = o.max_gap;
	int max_gap_ref;
	if (o.max_gap_ref > 0) max_gap_ref = o.max_gap_ref;
	else if (o.max_frag_len > 0) { max_gap_ref = o.max_frag_len - qlen; if (max_gap_ref < o.max_gap) max_gap_ref = o.max_gap; }
	else max_gap_ref = o.max_gap;
	const int min_gap_ref = o.min_gap_ref < max_gap_ref ? o.min_gap_ref : max_gap_ref;
	if (frag_gap) *frag_gap = max_gap_ref;

	WindowReq wr;
	if (sketch_it) { wr.seq = seq_codes; wr.len = qlen; wr.dev_off = seq_dev_off; }
	wr.pre = std::move(pre);
	wr.max_occ = o.mid_occ; wr.flag = o.flag;
	wr.max_dist_x = max_gap_ref; wr.min_dist_x = min_gap_ref; wr.max_dist_y = max_gap_qry; wr.bw = o.bw;
	wr.max_skip = o.max_chain_skip; wr.max_iter = o.max_chain_iter; wr.min_cnt = o.min_cnt; wr.min_sc = o.min_chain_score;
	wr.gap_scale = gap_scale;
	wr.is_cdna = (o.flag & F_SPLICE) != 0;                              // is_splice of src/map.c:282
	if (wr.len > 0 || !wr.pre.empty()) sch.window(wr);
	if (sketch_it) *rep_len_io = wr.rep_len;
	const int rep_len = *rep_len_io;
	out.a = std::move(wr.a);
	out.rep_len = rep_len;
	// mm_gen_regs, then chain_post (src/map.c:256-265): set_parent + select_sub fused with the region generation (wm_hit.h)
	if (!(o.flag & F_ALL_CHAINS)) {
		out.regs = gen_regs_select(hash, qlen, (int)wr.u.size(), wr.u.data(), out.a.data(), o.mask_level, o.mask_len, o.a * 2 + o.b, (o.flag & F_HARD_MLEVEL) != 0,
		                           o.pri_ratio, idx.k * 2, o.best_n);
		if (!(o.flag & (F_SPLICE | F_SR | F_NO_LJOIN))) join_long(o, qlen, out.regs, out.a.data());
	} else out.regs = gen_regs(hash, qlen, (int)wr.u.size(), wr.u.data(), out.a.data());
	// align_regs (src/map.c:267-277)
	if (o.flag & F_CIGAR) {
		align_skeleton(sch, o, idx, qlen, codes, dev_off, out.regs, out.a.data());
		if (!(o.flag & F_ALL_CHAINS)) {
			set_parent(o.mask_level, o.mask_len, out.regs, o.a * 2 + o.b, (o.flag & F_HARD_MLEVEL) != 0);
			select_sub(o.pri_ratio, idx.k * 2, o.best_n, out.regs);
			set_sam_pri(out.regs);
		}
	}
	set_mapq(out.regs, o.min_chain_score, o.a, rep_len, 0);
}

struct ReadTask {
	const ReadIn *in = 0;
	ReadOut *out = 0;
	const uint8_t *codes = 0;                    // 0..4 codes of the read (inside the mini-batch's code buffer)
	int64_t dev_off = -1;                        // where they live on the device (DeviceOps::load_reads), -1 = not resident
	int qlen = 0;
	std::vector<std::vector<m128>> collect;     // MCAS anchors per suffix position (collect_a, src/map.c:296)
	std::vector<uint8_t> mapped;                 // seqMapped, src/map.c:310
	int pending = 0;
};

// one stage-1 start position: grow the window to the right, then to the left, until a confident alignment appears
void stage1_position(Scheduler &sch, const Index &idx, const MapOpt &opt, const MapOpt &o2, ReadTask &T, int sub_begin, int suffix_id)
{
	const int L = T.qlen;
	const char *qname = T.in->name.c_str();
	bool found = false;
	for (int sub_len = o2.minPrefixLength; sub_len <= o2.maxPrefixLength; sub_len = (int)(sub_len * o2.prefixIncrementFactor)) {
		for (int dir = 0; dir < 2; ++dir) {             // 0: bases to the right of sub_begin, 1: to the left (src/map.c:346, :518)
			const int start = dir == 0 ? sub_begin : sub_begin - sub_len + 1;
			if (dir == 0 ? (sub_begin + sub_len > L) : (start < 0)) continue;
			Segment S;
			int rep_len = 0;
			const int64_t dev = T.dev_off >= 0 ? T.dev_off + start : -1;
			window_and_align(sch, idx, o2, opt.chain_gap_scale, T.codes + start, dev, true, T.codes + start, dev, sub_len, read_hash(qname, sub_len, o2.seed), std::vector<m128>(), &rep_len, S, 0);
			for (const Reg &r : S.regs) {
				if ((int)r.mapq >= o2.min_mapq && r.blen >= o2.min_qcov * sub_len && r.cnt > 0) {
					found = true;
					std::vector<m128> &dst = T.collect[suffix_id];
					dst.resize(r.cnt);
					// shift the anchors of this chain from window to read coordinates (src/map.c:482-496, :655-668)
					const uint64_t fwd_shift = (uint64_t)start;
					const uint64_t rev_shift = dir == 0 ? (uint64_t)(L - sub_begin - sub_len) : (uint64_t)((L - 1) - sub_begin);
					for (int i = 0; i < r.cnt; ++i) {
						m128 t = S.a[i + r.as];
						t.y += (t.x >> 63) ? rev_shift : fwd_shift;
						dst[i] = t;
					}
					for (int i = start; i < start + sub_len; ++i) T.mapped[i] = 1;
					break;
				}
			}
			if (found || S.regs.empty()) goto done;
		}
	}
done:
	return;
}

void stage2(Scheduler &sch, const Index &idx, const MapOpt &opt, ReadTask &T)
{
	const int L = T.qlen;
	const char *qname = T.in->name.c_str();
	MapOpt o3 = opt;                                                   // src/map.c:709-717
	o3.zdrop_inv = std::min(opt.zdrop_inv, opt.stage2_zdrop_inv);
	o3.bw = std::max(opt.bw, opt.stage2_bw);
	o3.max_gap = std::max(opt.max_gap, opt.stage2_max_gap);
	const uint32_t hash = read_hash(qname, L, o3.seed);
	std::vector<m128> a;
	int rep_len = 0;   // NB: the reference leaves this uninitialised on the pure-MCAS path (src/map.c:281); 0 is our defined value
	{ WM_PROF("map.stage2_merge");
	for (const auto &v : T.collect) a.insert(a.end(), v.begin(), v.end());
	if (!a.empty()) {                                                  // merge, dedup, order (src/map.c:739-781)
		std::sort(a.begin(), a.end(), [](const m128 &p, const m128 &q) { return std::tie(p.x, p.y) < std::tie(q.x, q.y); });
		a.erase(std::unique(a.begin(), a.end(), [](const m128 &p, const m128 &q) { return p.x == q.x && p.y == q.y; }), a.end());
		radix_sort_128x(a.data(), a.data() + a.size());
		if ((int)a.size() < o3.min_cnt) a.clear();
	} }
	size_t unmapped = 0;
	for (int i = 0; i < L; ++i) unmapped += T.mapped[i] == 0;
	Segment S;
	int frag_gap = 0;
	if (!a.empty() && unmapped > 0) {                                  // seeds from the stretches stage 1 left unmapped join the collected anchors (:786-846)
		std::vector<uint8_t> masked(T.codes, T.codes + L);
		for (int i = 0; i < L; ++i) if (T.mapped[i]) masked[i] = 4;
		window_and_align(sch, idx, o3, opt.chain_gap_scale, masked.data(), -1, true, T.codes, T.dev_off, L, hash, std::move(a), &rep_len, S, &frag_gap);   // (a masked copy: not resident)
	} else if (a.empty()) {                                            // plain minimap2-style mapping with the user's options (:849-865)
		o3 = opt;
		window_and_align(sch, idx, o3, opt.chain_gap_scale, T.codes, T.dev_off, true, T.codes, T.dev_off, L, hash, std::vector<m128>(), &rep_len, S, &frag_gap);
	} else                                                             // the collected anchors cover the read: chain them as they are
		window_and_align(sch, idx, o3, opt.chain_gap_scale, 0, -1, false, T.codes, T.dev_off, L, hash, std::move(a), &rep_len, S, &frag_gap), T.out->rep_len_defined = false;
	T.out->regs = std::move(S.regs);
	T.out->rep_len = rep_len;
	T.out->frag_gap = frag_gap;
}

} // namespace

namespace {
struct Worker;
// create the fibers of one read on scheduler `sch` (stage-1 positions, then stage 2; src/map.c:304-341); returns false if the read needs no work.
// `on_done` runs (inside the last fiber of the read) when the read is finished.
bool spawn_read(Scheduler &sch, const Index &idx, const MapOpt &opt, const MapOpt &o2, ReadTask &T, std::function<void()> on_done)
{
	if (T.qlen == 0) return false;
	if (opt.max_qlen > 0 && T.qlen > opt.max_qlen) return false;
	const int off = o2.suffixSampleOffset;
	const int n_pos = 1 + (int)ceil(T.qlen * 1.0 / off);
	T.collect.assign(n_pos, std::vector<m128>());
	T.mapped.assign(T.qlen, 0);
	ReadTask *tp = &T;
	Scheduler *sp = &sch;
	const Index *ip = &idx; const MapOpt *op = &opt, *o2p = &o2;
	auto finish = [sp, ip, op, tp, on_done]() {
		stage2(*sp, *ip, *op, *tp);
		// the read is done: drop its scratch (the window keeps thousands of reads in flight)
		std::vector<std::vector<m128>>().swap(tp->collect); std::vector<uint8_t>().swap(tp->mapped);
		on_done();
	};
	if (o2.SVaware && T.qlen >= o2.SVawareMinReadLength) {
		std::vector<std::pair<int, int>> pos;                           // (sub_begin, suffix_id), src/map.c:334-341
		for (int sb = 0; sb < T.qlen + off - 1; sb += off) {
			const int sid = sb / off;
			int b = sb;
			if (b >= T.qlen) b = T.qlen - 1;
			pos.push_back(std::make_pair(b, sid));
			if (b != sb) break;
		}
		T.pending = (int)pos.size();
		for (auto p : pos)
			sch.spawn([sp, ip, op, o2p, tp, p, finish]() {
				stage1_position(*sp, *ip, *op, *o2p, *tp, p.first, p.second);
				if (--tp->pending == 0) sp->spawn(finish);
			});
	} else sch.spawn(finish);
	return true;
}
} // namespace

void map_batch(const Index &idx, const MapOpt &opt, DeviceOps *ops, const std::vector<ReadIn> &reads, std::vector<ReadOut> &out, MapStats *stats, int n_threads,
               const std::function<void(size_t)> *on_read_done, int slot)
{
	out.assign(reads.size(), ReadOut());
	wm_ksw_score_t sc;
	sc.match = (int8_t)opt.a; sc.mismatch = (int8_t)-abs(opt.b); sc.sc_ambi = (int8_t)-abs(opt.sc_ambi);
	sc.q = (int8_t)opt.q; sc.e = (int8_t)opt.e; sc.q2 = (int8_t)opt.q2; sc.e2 = (int8_t)opt.e2;
	MapOpt o2 = opt;                                                   // stage-1 options (src/map.c:300-302)
	o2.best_n = std::max(5, o2.best_n);
	std::vector<ReadTask> tasks(reads.size());
	// n_threads = host cores to use. Workers inside a batched device call sleep, so up to max_inflight() more workers keep the cores busy
	const int extra = getenv("WM_EXTRA_WORKERS") ? atoi(getenv("WM_EXTRA_WORKERS")) : (ops->waits_asleep() && n_threads > 1 ? ops->max_inflight() : 0);
	const int T = (n_threads < 1 ? 1 : n_threads) + (extra > 0 ? extra : 0);
	// the 0..4 codes of the whole mini-batch, back to back (seq_nt4_table, src/sketch.c:19-36): encoded once, handed to the device once
	std::vector<uint64_t> code_off(reads.size() + 1, 0);
	for (size_t i = 0; i < reads.size(); ++i) code_off[i + 1] = code_off[i] + reads[i].seq.size();
	std::unique_ptr<uint8_t[]> codes_all(new uint8_t[code_off[reads.size()] + 1]);
	parallel_for(T, reads.size(), [&](size_t i) {
		uint8_t *d = codes_all.get() + code_off[i];
		const std::string &s = reads[i].seq;
		for (size_t j = 0; j < s.size(); ++j) d[j] = nt4_table[(uint8_t)s[j]];
	});
	int64_t dev_base = 0;
	const bool resident = ops->load_reads(codes_all.get(), (size_t)code_off[reads.size()], slot, &dev_base);
	struct Release { DeviceOps *o; int s; ~Release() { o->release_reads(s); } } release_guard{ ops, slot };
	for (size_t i = 0; i < reads.size(); ++i) {
		tasks[i].in = &reads[i]; tasks[i].out = &out[i]; tasks[i].qlen = (int)reads[i].seq.size();
		tasks[i].codes = codes_all.get() + code_off[i]; tasks[i].dev_off = resident ? dev_base + (int64_t)code_off[i] : -1;
	}
	// reads in flight over all workers (WM_INFLIGHT): large enough for full device batches at every stage, small enough that the
	// stages overlap instead of running in lock-step phases
	const long inflight_env = getenv("WM_INFLIGHT") ? atol(getenv("WM_INFLIGHT")) : 0;
	const size_t window = std::max<size_t>(1, (size_t)(inflight_env > 0 ? inflight_env : 16384) / (size_t)T);
	Hub hub(ops, sc, idx.w, idx.k);
	hub.n_workers = T;
	hub.max_sw_mat = opt.max_sw_mat;
	hub.splice = (opt.flag & F_SPLICE) != 0; hub.noncan = opt.noncan; hub.junc_bonus = opt.junc_bonus;
	std::vector<std::unique_ptr<Scheduler>> sch(T);
	for (int t = 0; t < T; ++t) sch[t].reset(new Scheduler(&hub, t));
	// worker t owns reads t, t+T, ...: it admits `window` of them and one more whenever one finishes
	struct Feed { size_t next; };
	std::vector<Feed> feed(T);
	for (int t = 0; t < T; ++t) {
		feed[t].next = (size_t)t;
		size_t mine = 0;
		for (size_t i = t; i < tasks.size(); i += T) ++mine;
		sch[t]->hold((int64_t)mine);
	}
	std::function<void(int)> admit = [&](int t) {
		for (;;) {                                                       // reads that need no work are skipped
			const size_t i = feed[t].next;
			if (i >= tasks.size()) return;
			feed[t].next += (size_t)T;
			const bool spawned = spawn_read(*sch[t], idx, opt, o2, tasks[i], [&admit, t, i, on_read_done]() { if (on_read_done) (*on_read_done)(i); admit(t); });
			sch[t]->release(1);
			if (spawned) return;
			if (on_read_done) (*on_read_done)(i);
		}
	};
	ErrorSink sink;                                                      // this call's internal errors (two calls may share the process)
	auto work = [&](int t) {
		ErrorSink *const prev = tl_error_sink();
		tl_error_sink() = &sink;
		for (size_t k = 0; k < window; ++k) admit(t);
		sch[t]->run();
		tl_error_sink() = prev;
	};
	std::vector<std::thread> th;
	for (int t = 1; t < T; ++t) th.emplace_back(work, t);
	work(0);
	for (auto &x : th) x.join();
	if (!sink.msg.empty()) {
		if (stats) stats->internal_error = sink.msg;
		else put_internal_error(sink.msg);
	}
	if (stats) {
		for (int op = 0; op < OP_N; ++op) stats->n_flush += hub.n_batches[op];
		stats->n_ksw += hub.n_reqs[OP_KSW] + hub.n_reqs[OP_KSW_HEAVY] + hub.n_reqs[OP_KSW_HUGE]; stats->n_chain += hub.n_reqs[OP_CHAIN]; stats->n_seed += hub.n_reqs[OP_SEED];
		stats->n_sketch += hub.n_reqs[OP_SKETCH] + hub.n_reqs[OP_WINDOW] + hub.n_reqs[OP_WINDOW_BIG];           // (windows: every window is sketched, seeded and chained in one call)
		if (getenv("WM_TRACE")) { for (auto &e : hub.site_cpu) fprintf(stderr, "[site] %-28s %8.2f s CPU\n", e.first, e.second); }
		stats->cpu_fiber += hub.cpu_fiber; stats->wall_idle += hub.wall_idle; stats->cpu_help += hub.cpu_help;
		stats->wall_fiber += hub.wall_fiber; stats->wall_lock += hub.wall_lock; stats->wall_total += hub.wall_total;
		for (int op = 0; op < OP_N; ++op) {           // (the heavy alignment queues are reported with the ksw operation, the fused window call in the first slot)
			const int o = op == OP_WINDOW || op == OP_WINDOW_BIG ? 0 : op >= OP_KSW_HEAVY ? OP_KSW : op;
			stats->n_batches[o] += hub.n_batches[op]; stats->cpu_op[o] += hub.cpu_op[op]; stats->wall_op[o] += hub.wall_op[op];
		}
	}
}

} // namespace wm
