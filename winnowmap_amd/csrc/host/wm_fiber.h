// wm_fiber.h — cooperative fibers + the batching hub.
//
// The reference maps one read per thread with deeply data-dependent control flow (src/map.c:279-974: MCAS windows,
// MAPQ-gated retries, chain splitting). Rewriting that as explicit state machines is error-prone, so each unit of
// work (one stage-1 window position, one stage-2 pass) runs as a FIBER whose code reads sequentially; whenever it
// needs a device operation it files the request and yields.
//
// Execution model (one Hub per mapping call, T worker threads, no barriers):
//   * every worker owns the fibers of its reads and runs them until all of them wait for a device result;
//   * requests collect in the worker's local queues and are published to the Hub when the worker runs dry;
//   * a worker with nothing runnable becomes a DISPATCHER: it takes the whole pending queue of one operation type and
//     issues it as ONE batched device call (DeviceOps::*_batch — the call itself sleeps while the kernels run); several
//     dispatchers can be in flight at once (different operations, or the same one: DeviceOps::max_inflight()), so the
//     device always has queued work while the other workers keep running host glue;
//   * when a batch returns, the fibers that waited for it are handed back to their owners' inboxes.
// Queues grow while the device is busy and are taken whole, so batch sizes regulate themselves. Reads are admitted in a
// sliding window (the worker spawns a new read when one of its reads finishes), which keeps every stage of the path —
// sketch, seed, chain, align — in demand at the same time instead of in lock-step phases.
// Results do not depend on how requests are batched (every job is independent), so the output is deterministic.
#pragma once
#if !defined(__x86_64__)
#include <ucontext.h>
#endif
#include <functional>
#include <exception>
#include <deque>
#include <vector>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <stdlib.h>
#include <stdio.h>
#include <chrono>
#include <time.h>
#include "wm_ops.h"

namespace wm {

class Scheduler;
// Context switch. glibc's swapcontext saves and restores the signal mask with a system call on every switch (~200 ns; a read makes ~150
// switches); fibers never touch signal masks, so on x86-64 the switch is the textbook one: push the callee-saved registers, swap the stack
// pointers, pop, return. Other targets keep ucontext.
#if defined(__x86_64__)
extern "C" void wm_fiber_switch(void **save_sp, void *load_sp);
__asm__(
	".pushsection .text\n"
	".p2align 4\n"
	".globl wm_fiber_switch\n"
	".type wm_fiber_switch,@function\n"
	"wm_fiber_switch:\n"
	"	pushq %rbp\n	pushq %rbx\n	pushq %r12\n	pushq %r13\n	pushq %r14\n	pushq %r15\n"
	"	movq %rsp, (%rdi)\n"
	"	movq %rsi, %rsp\n"
	"	popq %r15\n	popq %r14\n	popq %r13\n	popq %r12\n	popq %rbx\n	popq %rbp\n"
	"	ret\n"
	".size wm_fiber_switch,.-wm_fiber_switch\n"
	".popsection\n");
struct FiberCtx { void *sp; };
#else
struct FiberCtx { ucontext_t uc; };
#endif
struct Fiber { FiberCtx ctx; char *stack; std::function<void()> fn; bool done; Scheduler *owner; int pending; };   // pending: batches this fiber still waits for
// OP_KSW_HEAVY: alignments whose single-wave run time is long (many rows x many lanes). They get their own queue so that the tail of a
// batch of heavy jobs (tens of milliseconds for ONE alignment) never delays the bulk of short gap fills.
// OP_KSW_HUGE: the few alignments that run for a large fraction of a second (unbanded fills across structural variants): a third queue,
// so that they do not hold back the heavy ones either.
// OP_WINDOW: one MCAS window / stage-2 pass, sketch → seed → sort → chain → extraction in one device call (WindowReq). The per-stage operations
// (OP_SKETCH / OP_SEED / OP_CHAIN) remain for callers that want a single stage.
// OP_WINDOW_BIG: the windows that can carry giant anchor sets — a stage-2 pass (handed-in anchors, or the whole read) — in a queue of their own: a
// window call lasts as long as its largest job (one 10^5-anchor sort + chain: ~25 ms), and the thousands of 1-kb stage-1 windows of a batch, which
// finish in one small kernel, would wait for it (round 4 timeline: gpurun_out/r04f).
enum { OP_SKETCH = 0, OP_SEED = 1, OP_CHAIN = 2, OP_KSW = 3, OP_KSW_HEAVY = 4, OP_KSW_HUGE = 5, OP_WINDOW = 6, OP_WINDOW_BIG = 7, OP_N = 8 };
inline double thread_cpu_s() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
inline double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// a host-side loop of a batched call that idle workers help with (ParHook, wm_core.h)
struct HelpTask { const std::function<void(size_t)> *fn; size_t n, chunk; std::atomic<size_t> next; int helpers; const char *site; std::exception_ptr err; std::mutex err_mu; };

struct Hub {
	Hub(DeviceOps *ops_, const wm_ksw_score_t &sc_, int w_, int k_) : ops(ops_), sc(sc_), w(w_), k(k_) { max_inflight = ops->max_inflight(); if (max_inflight < 1) max_inflight = 1; read_env(); par.H = this; }
	DeviceOps *ops;
	wm_ksw_score_t sc;
	int w, k;
	int max_inflight;                       // batched device calls that may run concurrently (device contexts of the ops object)
	bool splice = false; int noncan = 0, junc_bonus = 0;     // MM_F_SPLICE: alignments go through DeviceOps::exts2_batch with these (src/align.c:326-327)
	int64_t max_sw_mat = 0;                 // mm_mapopt_t::max_sw_mat (--cap-sw-mat): pairs with more DP cells are not aligned (src/align.c:323-325)
	std::mutex mu;
	std::condition_variable cv;
	// published requests and the fibers waiting for them (taken whole by a dispatcher)
	std::vector<SketchReq*> q_sketch; std::vector<SeedReq*> q_seed; std::vector<ChainReq*> q_chain; std::vector<KswReq*> q_ksw, q_kswh, q_kswx; std::vector<WindowReq*> q_window, q_windowb;
	std::vector<Fiber*> waiters[OP_N];
	int inflight = 0, inflight_op[OP_N] = {0, 0, 0, 0, 0, 0, 0, 0};
	int n_workers = 1, n_idle = 0;          // workers of the mapping call; workers asleep with nothing runnable
	std::vector<HelpTask*> help;            // loops of running batched calls that still have chunks to hand out
	double cpu_help = 0;                    // CPU seconds idle workers spent inside such loops
	std::vector<std::pair<const char*, double>> site_cpu;   // ... and the CPU seconds of every labelled loop (WM_SITE), dispatcher + helpers
	void add_site(const char *site, double s) { if (!site) site = "(unlabelled)"; for (auto &e : site_cpu) if (e.first == site) { e.second += s; return; } site_cpu.emplace_back(site, s); }
	// an exception in the loop body (staging buffers throw bad_alloc) is kept — the first one — and rethrown by the call's owner after every helper
	// has left; the remaining chunks are skipped (a helper thread must never unwind into std::terminate, nor leave `helpers` raised)
	static void help_work(HelpTask &t)
	{
		try {
			for (;;) { const size_t b = t.next.fetch_add(t.chunk); if (b >= t.n) break; const size_t e = b + t.chunk < t.n ? b + t.chunk : t.n; for (size_t i = b; i < e; ++i) (*t.fn)(i); }
		} catch (...) {
			std::lock_guard<std::mutex> lk(t.err_mu);
			if (!t.err) t.err = std::current_exception();
			t.next.store(t.n);
		}
	}
	// the hook a dispatching worker installs around its batched call: the loop is shared with whoever is idle
	struct Par : ParHook {
		Hub *H;
		void run(size_t n, const std::function<void(size_t)> &fn) override
		{
			HelpTask t;
			t.fn = &fn; t.n = n; t.next.store(0); t.helpers = 0; t.site = tl_par_site(); tl_par_site() = 0;
			t.chunk = n / (size_t)(4 * (H->n_workers > 0 ? H->n_workers : 1));
			if (t.chunk < 16) t.chunk = 16;
			if (tl_par_chunk()) { t.chunk = tl_par_chunk(); tl_par_chunk() = 0; }         // (coarse tasks: parallel_tasks)
			{ std::lock_guard<std::mutex> lk(H->mu); H->help.push_back(&t); }
			H->cv.notify_all();
			const double c0 = thread_cpu_s();
			help_work(t);
			const double dc = thread_cpu_s() - c0;
			std::unique_lock<std::mutex> lk(H->mu);
			H->add_site(t.site, dc);
			for (size_t i = 0; i < H->help.size(); ++i) if (H->help[i] == &t) { H->help.erase(H->help.begin() + i); break; }
			while (t.helpers > 0) H->cv.wait(lk);          // (helpers announce themselves and leave under the hub mutex)
			lk.unlock();
			if (t.err) std::rethrow_exception(t.err);      // (the task is deregistered and nobody refers to this frame any more)
		}
	} par;
	std::atomic<int64_t> live{0};           // fibers alive + reads not yet admitted, over all workers: 0 = the mapping call is finished
	uint64_t n_batches[OP_N] = {0, 0, 0, 0, 0, 0, 0, 0}, n_reqs[OP_N] = {0, 0, 0, 0, 0, 0, 0, 0};
	// where the host time goes (seconds, summed over the workers): CPU time running fibers, CPU and wall time inside the batched
	// calls per operation, wall time asleep waiting for results
	double cpu_fiber = 0, cpu_op[OP_N] = {0, 0, 0, 0, 0, 0, 0, 0}, wall_op[OP_N] = {0, 0, 0, 0, 0, 0, 0, 0}, wall_idle = 0;
	double wall_fiber = 0, wall_lock = 0, wall_total = 0;     // wall time running fibers (>> cpu_fiber: the workers are being descheduled), waiting for the hub mutex, inside run()
	size_t pending(int op) const { return op == OP_SKETCH ? q_sketch.size() : op == OP_SEED ? q_seed.size() : op == OP_CHAIN ? q_chain.size() : op == OP_KSW ? q_ksw.size() : op == OP_KSW_HEAVY ? q_kswh.size() : op == OP_KSW_HUGE ? q_kswx.size() : op == OP_WINDOW ? q_window.size() : q_windowb.size(); }
	// scheduling knobs (environment, read once per mapping call): a further concurrent batch of an operation that is already in flight is
	// issued when at least min_more[op] requests are pending and fewer than max_op[op] batches of it are running
	size_t min_more[OP_N] = { 4096, 4096, 4096, 24576, 2048, 64, 4096, 1024 };
	int max_op[OP_N] = { 2, 2, 2, 3, 2, 1, 3, 2 };
	// a batch is worth its fixed cost — a kernel launch lasts at least as long as its longest job, and the device runs only so many
	// kernels at once — when it is large: an operation is issued when min_batch[op] requests are pending or the oldest has waited
	// max_wait_ms[op], whichever comes first (0 / 0 = at once)
	// (the huge queue: 512 jobs / 250 ms since round 5 — a launch of stripe-pipelined jobs lasts as long as its longest job whatever their number, so
	// fewer, fuller launches: +3..9 % on config 2 together with the sixteen-wavefront routing, profiles/r05_sched.txt; before: 256 / 100 ms)
	size_t min_batch[OP_N] = { 8192, 8192, 8192, 98304, 12288, 512, 8192, 2048 };
	double max_wait_ms[OP_N] = { 15, 15, 15, 30, 60, 250, 15, 30 };
	double first_pending[OP_N] = { 0, 0, 0, 0, 0, 0, 0, 0 };          // wall_s() when the queue of the operation last became non-empty
	long heavy_units = 8192;                // rows x 128-lane register pairs above which an alignment goes to the heavy queue (0 = no heavy queue)
	long huge_units = 131072;               // ... and to the queue of the few very long ones (0 = none)
	void read_env()
	{
		static const char *nm[OP_N] = { "SKETCH", "SEED", "CHAIN", "KSW", "KSWH", "KSWX", "WINDOW", "WINDOWB" };
		for (int op = 0; op < OP_N; ++op) {
			char key[64];
			snprintf(key, sizeof(key), "WM_%s_MIN_MORE", nm[op]); if (getenv(key)) min_more[op] = (size_t)atol(getenv(key));
			snprintf(key, sizeof(key), "WM_%s_MAX", nm[op]); if (getenv(key)) max_op[op] = atoi(getenv(key));
			snprintf(key, sizeof(key), "WM_%s_MIN_BATCH", nm[op]); if (getenv(key)) min_batch[op] = (size_t)atol(getenv(key));
			snprintf(key, sizeof(key), "WM_%s_MAX_WAIT_MS", nm[op]); if (getenv(key)) max_wait_ms[op] = atof(getenv(key));
		}
		if (getenv("WM_KSW_HEAVY_UNITS")) heavy_units = atol(getenv("WM_KSW_HEAVY_UNITS"));
		if (getenv("WM_KSW_HUGE_UNITS")) huge_units = atol(getenv("WM_KSW_HUGE_UNITS"));
	}
};

class Scheduler {
public:
	explicit Scheduler(Hub *hub, int rank = 0) : hub_(hub), rank_(rank) {}
	~Scheduler() { for (Fiber *f : pool_) delete f; for (char *sl : slabs_) free(sl); }

	void spawn(std::function<void()> fn)
	{
		Fiber *f;
		if (!pool_.empty()) { f = pool_.back(); pool_.pop_back(); }
		else {           // stacks come from slabs of 64 (one mapping per slab: a window of 10^4..10^5 reads x ~10 fibers would otherwise need that many mappings)
			if (slab_left_ == 0) { slabs_.push_back((char*)malloc(kStack * kSlab)); slab_left_ = kSlab; }
			f = new Fiber(); f->stack = slabs_.back() + kStack * (size_t)(kSlab - slab_left_--);
		}
		f->fn = std::move(fn); f->done = false;
#if defined(__x86_64__)
		{   // a fresh stack that "returns" into the trampoline: six zeroed callee-saved registers, then the entry address; the slot above it
			// keeps the stack 16-byte aligned at the trampoline's entry as the ABI expects after a call
			void **top = (void**)(((uintptr_t)f->stack + kStack) & ~(uintptr_t)15);
			top[-1] = 0;
			top[-2] = (void*)&Scheduler::trampoline;
			for (int i = 3; i <= 8; ++i) top[-i] = 0;
			f->ctx.sp = (void*)(top - 8);
		}
#else
		getcontext(&f->ctx.uc);
		f->ctx.uc.uc_stack.ss_sp = f->stack; f->ctx.uc.uc_stack.ss_size = kStack; f->ctx.uc.uc_link = &main_.uc;
		const uintptr_t p = (uintptr_t)f;
		makecontext(&f->ctx.uc, (void (*)())&Scheduler::entry, 3, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32), 0);
#endif
		f->owner = this;
		ready_.push_back(f);
		hub_->live.fetch_add(1);
	}
	// work that is not a fiber yet (reads waiting for admission) also keeps the mapping call alive
	void hold(int64_t n) { hub_->live.fetch_add(n); }
	void release(int64_t n) { hub_->live.fetch_sub(n); }

	// run until the whole mapping call (all workers) has finished
	void run()
	{
		Hub &H = *hub_;
		double cpu_fiber = 0, wall_idle = 0, wall_fiber = 0, wall_lock = 0;
		const double run_w0 = wall_s();
#if defined(__x86_64__)
		tl_running() = this;
#endif
		for (;;) {
			const double c0 = thread_cpu_s(), fw0 = wall_s();
			while (!ready_.empty()) {
				cur_ = ready_.front(); ready_.pop_front();
				switch_to(main_, cur_->ctx);
				if (cur_->done) { cur_->fn = nullptr; pool_.push_back(cur_); H.live.fetch_sub(1); }
				cur_ = 0;
			}
			cpu_fiber += thread_cpu_s() - c0;
			const double lw0 = wall_s();
			wall_fiber += lw0 - fw0;
			std::unique_lock<std::mutex> lk(H.mu);
			wall_lock += wall_s() - lw0;
			publish_locked();
			for (;;) {
				if (!inbox_.empty()) { for (Fiber *f : inbox_) ready_.push_back(f); inbox_.clear(); break; }
				if (H.live.load() == 0) { H.cpu_fiber += cpu_fiber; H.wall_idle += wall_idle; H.wall_fiber += wall_fiber; H.wall_lock += wall_lock; H.wall_total += wall_s() - run_w0; H.cv.notify_all(); return; }
				double wake_in = 1e9;
				const int op = pick_locked(&wake_in);
				if (op >= 0) { dispatch(op, lk); continue; }      // (returns with the lock held again)
				{   // nothing to issue: lend a hand to a running batched call's host-side loop
					HelpTask *ht = 0;
					for (HelpTask *t : H.help) if (t->next.load() < t->n) { ht = t; break; }
					if (ht) {
						++ht->helpers;
						lk.unlock();
						const double hc0 = thread_cpu_s();
						Hub::help_work(*ht);
						const double hc = thread_cpu_s() - hc0;
						lk.lock();
						H.cpu_help += hc; H.add_site(ht->site, hc);
						if (--ht->helpers == 0) H.cv.notify_all();
						continue;
					}
				}
				const double w0 = wall_s();
				++H.n_idle;
				if (wake_in > 0.5) wake_in = 0.5;                  // (also re-checks the drain condition)
				H.cv.wait_for(lk, std::chrono::duration<double>(wake_in < 2e-4 ? 2e-4 : wake_in));
				--H.n_idle;
				wall_idle += wall_s() - w0;
			}
		}
	}

	// ---- called from inside a fiber ----
	void sketch(SketchReq &r) { l_sketch_.push_back(&r); wait(1 << OP_SKETCH); }
	void seed(SeedReq &r) { l_seed_.push_back(&r); wait(1 << OP_SEED); }
	void chain(ChainReq &r) { l_chain_.push_back(&r); wait(1 << OP_CHAIN); }
	void window(WindowReq &r)
	{
		// WM_WINDOW_BIG_LEN=8192 switches the second queue on (stage-1 windows are at most maxPrefixLength long: 8 000 for map-ont). Off by default: measured
		// neutral (0.242 vs 0.247 Gbp/s, profiles/r04g_sched_sweep.txt) — even a window call of a dozen requests takes 50-90 ms, waiting for a device context
		static const int big_len = getenv("WM_WINDOW_BIG_LEN") ? atoi(getenv("WM_WINDOW_BIG_LEN")) : -1;
		const bool big = big_len >= 0 && (!r.pre.empty() || r.len > big_len);
		(big ? l_windowb_ : l_window_).push_back(&r);
		wait(1 << (big ? OP_WINDOW_BIG : OP_WINDOW));
	}
	void ksw(std::vector<KswReq> &rs)
	{
		if (rs.empty()) return;
		int ops = 0;
		const long hu = hub_->heavy_units, xu = hub_->huge_units;
		for (KswReq &r : rs) {
			if (hub_->max_sw_mat > 0 && (int64_t)r.tlen() * r.qlen() > hub_->max_sw_mat) {      // --cap-sw-mat (src/align.c:323-325): not aligned, reported as z-dropped
				r.ez = wm_ksw_result_t(); r.ez.max_q = r.ez.max_t = r.ez.mqe_t = r.ez.mte_q = -1; r.ez.score = r.ez.mqe = r.ez.mte = -0x40000000; r.ez.zdropped = 1;
				r.cigar.clear();
				continue;
			}
			const long un = ksw_units(r);
			const int op = hub_->splice ? OP_KSW : xu > 0 && un > xu ? OP_KSW_HUGE : hu > 0 && un > hu ? OP_KSW_HEAVY : OP_KSW;   // (splice mode: one queue, exts2 has no band to classify by)
			(op == OP_KSW_HUGE ? l_kswx_ : op == OP_KSW_HEAVY ? l_kswh_ : l_ksw_).push_back(&r);
			ops |= 1 << op;
		}
		if (ops) wait(ops);
	}
	// serial work of one alignment on its wavefront: anti-diagonals x 128-lane register pairs of the band hull
	static long ksw_units(const KswReq &r)
	{
		const long ql = (long)r.qlen(), tl = (long)r.tlen();
		long w = r.w < 0 ? (ql > tl ? ql : tl) : r.w, n = ql < tl ? ql : tl;
		if (n > w + 1) n = w + 1;
		return (ql + tl) * ((n + 127) / 128 + 1);
	}

private:
	static constexpr size_t kStack = 256 * 1024;
	static constexpr int kSlab = 64;
#if defined(__x86_64__)
	static Scheduler *&tl_running() { static thread_local Scheduler *s = 0; return s; }
	static void switch_to(FiberCtx &from, FiberCtx &to) { wm_fiber_switch(&from.sp, to.sp); }
	static void trampoline()                // first activation of a fiber: runs on the fiber's own stack, never returns
	{
		Scheduler *self = tl_running();
		Fiber *f = self->cur_;
		f->fn();
		f->done = true;
		self = tl_running();                // (the fiber may have been resumed by the same worker only: owners do not migrate)
		switch_to(f->ctx, self->main_);
		__builtin_trap();
	}
#else
	static void switch_to(FiberCtx &from, FiberCtx &to) { swapcontext(&from.uc, &to.uc); }
	static void entry(unsigned lo, unsigned hi, unsigned)
	{
		Fiber *f = (Fiber*)((uintptr_t)lo | (uintptr_t)hi << 32);
		f->fn();
		f->done = true;                     // uc_link returns to the scheduler
	}
#endif
	void wait(int ops)                      // ops: bit set of the operations this fiber filed requests for
	{
		Fiber *me = cur_;
		me->pending = 0;
		for (int op = 0; op < OP_N; ++op) if (ops >> op & 1) { l_wait_[op].push_back(me); ++me->pending; }
		switch_to(me->ctx, main_);
	}
	void publish_locked()
	{
		Hub &H = *hub_;
		bool any = false;
		const double now = wall_s();
		for (int op = 0; op < OP_N; ++op) if (H.pending(op) == 0) H.first_pending[op] = now;     // (stamps queues that are about to become non-empty)
		if (!l_sketch_.empty()) { H.q_sketch.insert(H.q_sketch.end(), l_sketch_.begin(), l_sketch_.end()); l_sketch_.clear(); any = true; }
		if (!l_seed_.empty()) { H.q_seed.insert(H.q_seed.end(), l_seed_.begin(), l_seed_.end()); l_seed_.clear(); any = true; }
		if (!l_chain_.empty()) { H.q_chain.insert(H.q_chain.end(), l_chain_.begin(), l_chain_.end()); l_chain_.clear(); any = true; }
		if (!l_ksw_.empty()) { H.q_ksw.insert(H.q_ksw.end(), l_ksw_.begin(), l_ksw_.end()); l_ksw_.clear(); any = true; }
		if (!l_kswh_.empty()) { H.q_kswh.insert(H.q_kswh.end(), l_kswh_.begin(), l_kswh_.end()); l_kswh_.clear(); any = true; }
		if (!l_kswx_.empty()) { H.q_kswx.insert(H.q_kswx.end(), l_kswx_.begin(), l_kswx_.end()); l_kswx_.clear(); any = true; }
		if (!l_window_.empty()) { H.q_window.insert(H.q_window.end(), l_window_.begin(), l_window_.end()); l_window_.clear(); any = true; }
		if (!l_windowb_.empty()) { H.q_windowb.insert(H.q_windowb.end(), l_windowb_.begin(), l_windowb_.end()); l_windowb_.clear(); any = true; }
		for (int op = 0; op < OP_N; ++op)
			if (!l_wait_[op].empty()) { H.waiters[op].insert(H.waiters[op].end(), l_wait_[op].begin(), l_wait_[op].end()); l_wait_[op].clear(); }
		if (any) H.cv.notify_all();          // somebody idle may want to dispatch what was just published
	}
	// which operation this idle worker should issue now (-1: none). An operation with nothing in flight goes first (keeps every stage
	// of the path moving); a further concurrent batch of the same operation is only worth its fixed cost when the queue is large.
	int pick_locked(double *wake_in_s)
	{
		Hub &H = *hub_;
		*wake_in_s = 1e9;
		if (H.inflight >= H.max_inflight) return -1;
		const double now = wall_s();
		// nothing running anywhere and nobody able to produce more requests: whatever is pending must go now
		const bool drain = H.inflight == 0 && H.n_idle + 1 >= H.n_workers;
		int best = -1;
		static const int stage_order[OP_N] = { OP_KSW_HUGE, OP_KSW_HEAVY, OP_KSW, OP_WINDOW_BIG, OP_CHAIN, OP_SEED, OP_SKETCH, OP_WINDOW };
		for (int oi = 0; oi < OP_N; ++oi) {                   // later stages first: finishing reads frees their memory and admits new ones
			const int op = stage_order[oi];
			const size_t n = H.pending(op);
			if (n == 0) continue;
			const double waited_ms = (now - H.first_pending[op]) * 1e3;
			const bool ripe = drain || n >= H.min_batch[op] || waited_ms >= H.max_wait_ms[op];
			if (!ripe) { const double left = (H.max_wait_ms[op] - waited_ms) * 1e-3; if (left < *wake_in_s) *wake_in_s = left; continue; }
			if (H.inflight_op[op] == 0) return op;
			if (n >= H.min_more[op] && H.inflight_op[op] < H.max_op[op] && best < 0) best = op;
		}
		return best;
	}
	void dispatch(int op, std::unique_lock<std::mutex> &lk)
	{
		Hub &H = *hub_;
		std::vector<Fiber*> waiters;
		waiters.swap(H.waiters[op]);
		std::vector<SketchReq*> a; std::vector<SeedReq*> b; std::vector<ChainReq*> c; std::vector<KswReq*> d; std::vector<WindowReq*> wq;
		size_t n = 0;
		if (op == OP_SKETCH) { a.swap(H.q_sketch); n = a.size(); } else if (op == OP_SEED) { b.swap(H.q_seed); n = b.size(); }
		else if (op == OP_CHAIN) { c.swap(H.q_chain); n = c.size(); } else if (op == OP_KSW) { d.swap(H.q_ksw); n = d.size(); } else if (op == OP_KSW_HEAVY) { d.swap(H.q_kswh); n = d.size(); }
		else if (op == OP_KSW_HUGE) { d.swap(H.q_kswx); n = d.size(); } else if (op == OP_WINDOW) { wq.swap(H.q_window); n = wq.size(); } else { wq.swap(H.q_windowb); n = wq.size(); }
		++H.inflight; ++H.inflight_op[op]; ++H.n_batches[op]; H.n_reqs[op] += n;
		lk.unlock();
		static const bool trace = getenv("WM_TRACE") != 0;
		const auto t0 = std::chrono::steady_clock::now();
		const double c0 = thread_cpu_s(), w0 = wall_s();
		ParHook *const prev_hook = tl_par_hook();
		tl_par_hook() = &H.par;
		try {
			if (op == OP_SKETCH) H.ops->sketch_batch(H.w, H.k, a);
			else if (op == OP_SEED) H.ops->seed_batch(b);
			else if (op == OP_CHAIN) H.ops->chain_batch(c);
			else if (op == OP_WINDOW || op == OP_WINDOW_BIG) H.ops->window_batch(H.w, H.k, wq);
			else if (H.splice) H.ops->exts2_batch(H.sc, H.noncan, H.junc_bonus, d);
			else H.ops->ksw_batch(H.sc, d);
		} catch (const std::exception &e) {          // the waiters below are released in any case (their requests keep their zero-initialised results)
			note_internal_error(e.what(), __FILE__, __LINE__);
		} catch (...) {
			note_internal_error("exception in a batched device call", __FILE__, __LINE__);
		}
		if (trace) fprintf(stderr, "[batch] worker %2d %s n=%zu %.1f ms\n", rank_, op == OP_SKETCH ? "sketch" : op == OP_SEED ? "seed" : op == OP_CHAIN ? "chain" : op == OP_KSW ? "ksw" : op == OP_KSW_HEAVY ? "ksw-heavy" : op == OP_KSW_HUGE ? "ksw-huge" : op == OP_WINDOW_BIG ? "window-big" : "window", n,
		                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
		tl_par_hook() = prev_hook;
		const double dc = thread_cpu_s() - c0, dw = wall_s() - w0;
		lk.lock();
		H.cpu_op[op] += dc; H.wall_op[op] += dw;
		--H.inflight; --H.inflight_op[op];
		for (Fiber *f : waiters)
			if (--f->pending == 0) f->owner->inbox_.push_back(f);      // (counters and inboxes are protected by the hub mutex)
		H.cv.notify_all();
	}

	Hub *hub_;
	int rank_;
	FiberCtx main_;
	Fiber *cur_ = 0;
	std::deque<Fiber*> ready_;
	std::vector<Fiber*> pool_;
	std::vector<char*> slabs_; int slab_left_ = 0;
	std::vector<Fiber*> inbox_;             // fibers whose results arrived (filled by dispatchers under the hub mutex)
	std::vector<SketchReq*> l_sketch_; std::vector<SeedReq*> l_seed_; std::vector<ChainReq*> l_chain_; std::vector<KswReq*> l_ksw_, l_kswh_, l_kswx_; std::vector<WindowReq*> l_window_, l_windowb_;
	std::vector<Fiber*> l_wait_[OP_N];
};

} // namespace wm
