// wm_fiber.h — cooperative fibers + batching scheduler.
//
// The reference maps one read per thread with deeply data-dependent control flow (src/map.c:279-974: MCAS windows,
// MAPQ-gated retries, chain splitting). Rewriting that as explicit state machines is error-prone, so each unit of
// work (one stage-1 window position, one stage-2 pass) runs as a FIBER whose code reads sequentially; whenever it
// needs a device operation it enqueues the request and yields. When every fiber is blocked the scheduler flushes the
// queues as ONE batched device call per operation type and resumes the waiters. Thousands of reads in flight give
// the kernels their batch sizes; the host code stays a straight restatement of the reference's semantics.
#pragma once
#include <ucontext.h>
#include <functional>
#include <deque>
#include <vector>
#include <memory>
#include <stdlib.h>
#include <stdio.h>
#include <chrono>
#include <pthread.h>
#include "wm_ops.h"

namespace wm {

class Scheduler;

// A TEAM of schedulers, one per host thread, that share their device batches: every member runs its own fibers (the
// host glue of its reads) in parallel with the others; when all members have nothing runnable they meet at a barrier,
// member 0 issues ONE batched device call per operation type for the whole team, and everybody resumes its waiters.
// Host work scales with the threads while the kernels still see the batch of the whole team.
struct SchedTeam {
	explicit SchedTeam(int n) : n_(n) { pthread_barrier_init(&bar_, 0, (unsigned)n); }
	~SchedTeam() { pthread_barrier_destroy(&bar_); }
	int n_;
	pthread_barrier_t bar_;
	std::vector<Scheduler*> members;
	bool done = false;
	ParallelExec exec;                                             // members 1.. lend themselves to member 0 during the device phase
	uint64_t round = 0;
};

class Scheduler {
public:
	Scheduler(DeviceOps *ops, const wm_ksw_score_t &sc, int w, int k, SchedTeam *team = 0, int rank = 0) : ops_(ops), sc_(sc), w_(w), k_(k), team_(team), rank_(rank) {}
	~Scheduler() { for (Fiber *f : pool_) { free(f->stack); delete f; } }

	void spawn(std::function<void()> fn)
	{
		Fiber *f;
		if (!pool_.empty()) { f = pool_.back(); pool_.pop_back(); }
		else { f = new Fiber(); f->stack = (char*)malloc(kStack); }
		f->fn = std::move(fn); f->done = false;
		getcontext(&f->ctx);
		f->ctx.uc_stack.ss_sp = f->stack; f->ctx.uc_stack.ss_size = kStack; f->ctx.uc_link = &main_;
		const uintptr_t p = (uintptr_t)f;
		makecontext(&f->ctx, (void (*)())&Scheduler::entry, 3, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32), 0);
		f->owner = this;
		ready_.push_back(f);
		++live_;
	}

	// run until every fiber (of the whole team, if there is one) has finished
	void run()
	{
		static const bool trace2 = getenv("WM_TRACE2") != 0;
		for (;;) {
			const auto tb0 = std::chrono::steady_clock::now();
			size_t n_run = 0;
			while (!ready_.empty()) {
				cur_ = ready_.front(); ready_.pop_front();
				swapcontext(&main_, &cur_->ctx);
				if (cur_->done) { cur_->fn = nullptr; pool_.push_back(cur_); --live_; }
				cur_ = 0;
				++n_run;
			}
			if (trace2 && n_run) fprintf(stderr, "[member %d] ran %zu fiber slices in %.2f ms\n", rank_, n_run, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
			if (!team_) {
				if (live_ == 0) break;
				flush();
			} else if (!team_round()) break;
		}
	}

	// ---- called from inside a fiber ----
	void sketch(SketchReq &r) { q_sketch_.push_back(&r); wait(w_sketch_); }
	void seed(SeedReq &r) { q_seed_.push_back(&r); wait(w_seed_); }
	void chain(ChainReq &r) { q_chain_.push_back(&r); wait(w_chain_); }
	void ksw(std::vector<KswReq> &rs) { if (rs.empty()) return; for (KswReq &r : rs) q_ksw_.push_back(&r); wait(w_ksw_); }

	uint64_t n_flush = 0, n_ksw_jobs = 0, n_chain_jobs = 0, n_sketch_jobs = 0, n_seed_jobs = 0;

private:
	struct Fiber { ucontext_t ctx; char *stack; std::function<void()> fn; bool done; Scheduler *owner; };
	static constexpr size_t kStack = 256 * 1024;
	static void entry(unsigned lo, unsigned hi, unsigned)
	{
		Fiber *f = (Fiber*)((uintptr_t)lo | (uintptr_t)hi << 32);
		f->fn();
		f->done = true;                     // uc_link returns to the scheduler
	}
	void wait(std::vector<Fiber*> &w) { Fiber *me = cur_; w.push_back(me); swapcontext(&me->ctx, &main_); }
	void wake(std::vector<Fiber*> &w) { for (Fiber *f : w) ready_.push_back(f); w.clear(); }
	bool team_round()
	{
		pthread_barrier_wait(&team_->bar_);                            // every member is blocked or finished
		const uint64_t round = ++my_round_;
		if (rank_ == 0) {
			size_t live = 0;
			for (Scheduler *m : team_->members) live += m->live_;
			team_->done = live == 0;
			if (!team_->done) {                                          // gather everybody's requests into this member's queues
				for (Scheduler *m : team_->members) {
					if (m == this) continue;
					q_sketch_.insert(q_sketch_.end(), m->q_sketch_.begin(), m->q_sketch_.end()); m->q_sketch_.clear();
					q_seed_.insert(q_seed_.end(), m->q_seed_.begin(), m->q_seed_.end()); m->q_seed_.clear();
					q_chain_.insert(q_chain_.end(), m->q_chain_.begin(), m->q_chain_.end()); m->q_chain_.clear();
					q_ksw_.insert(q_ksw_.end(), m->q_ksw_.begin(), m->q_ksw_.end()); m->q_ksw_.clear();
				}
				team_->exec.wait_servers(team_->n_ - 1);
				tl_parallel_exec() = &team_->exec;
				flush();
				tl_parallel_exec() = 0;
			}
			team_->exec.close(round);
		} else
			team_->exec.serve(round);                                    // help with the host side of the batched calls until they are done
		pthread_barrier_wait(&team_->bar_);                            // results are in the requests
		if (team_->done) return false;
		if (rank_ != 0) { wake(w_sketch_); wake(w_seed_); wake(w_chain_); wake(w_ksw_); }
		return true;
	}
	void flush()
	{
		++n_flush;
		static const bool trace = getenv("WM_TRACE") != 0;
		auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		const double t0 = now();
		const size_t n0 = q_sketch_.size(), n1 = q_seed_.size(), n2 = q_chain_.size(), n3 = q_ksw_.size();
		if (!q_sketch_.empty()) { n_sketch_jobs += q_sketch_.size(); ops_->sketch_batch(w_, k_, q_sketch_); q_sketch_.clear(); wake(w_sketch_); }
		const double t1 = now();
		if (!q_seed_.empty()) { n_seed_jobs += q_seed_.size(); ops_->seed_batch(q_seed_); q_seed_.clear(); wake(w_seed_); }
		const double t2 = now();
		if (!q_chain_.empty()) { n_chain_jobs += q_chain_.size(); ops_->chain_batch(q_chain_); q_chain_.clear(); wake(w_chain_); }
		const double t3 = now();
		if (!q_ksw_.empty()) { n_ksw_jobs += q_ksw_.size(); ops_->ksw_batch(sc_, q_ksw_); q_ksw_.clear(); wake(w_ksw_); }
		const double t4 = now();
		if (trace) fprintf(stderr, "[flush %3llu] host %.1f ms | sketch %zu: %.1f ms | seed %zu: %.1f ms | chain %zu: %.1f ms | ksw %zu: %.1f ms\n",
		                   (unsigned long long)n_flush, t0 - t_last_, n0, t1 - t0, n1, t2 - t1, n2, t3 - t2, n3, t4 - t3);
		t_last_ = now();
	}
	double t_last_ = 0;
	DeviceOps *ops_;
	wm_ksw_score_t sc_;
	int w_, k_;
	SchedTeam *team_;
	int rank_;
	uint64_t my_round_ = 0;
	ucontext_t main_;
	Fiber *cur_ = 0;
	size_t live_ = 0;
	std::deque<Fiber*> ready_;
	std::vector<Fiber*> pool_;
	std::vector<SketchReq*> q_sketch_; std::vector<SeedReq*> q_seed_; std::vector<ChainReq*> q_chain_; std::vector<KswReq*> q_ksw_;
	std::vector<Fiber*> w_sketch_, w_seed_, w_chain_, w_ksw_;
};

} // namespace wm
