// tools/sprof/sprof.cpp — a minimal sampling profiler for the host side of a run (LD_PRELOAD=tools/sprof/libsprof.so): SIGPROF every millisecond of
// process CPU time, the interrupted thread's program counter is stored; at exit the samples are written as "<module path> <offset>" lines to
// $SPROF_OUT.<pid> (default sprof.txt.<pid>). tools/sprof/resolve.py turns them into self time per function with the modules' symbol tables (nm).
// No perf / gdb in this image; this is what says where the host's CPU seconds go (DESIGN.md "Host budget").
#define _GNU_SOURCE 1
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>
#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <vector>
#include <string>
#include <map>

namespace {
const size_t CAP = 1u << 20;
const int NSTK = 40;                     // stack words kept per sample: finish() keeps those that point into code = the callers, without unwind tables
void **g_pc = 0;
void **g_stk = 0;
std::atomic<size_t> g_n(0);
void on_prof(int, siginfo_t *, void *uc_)
{
	ucontext_t *uc = (ucontext_t*)uc_;
	const size_t i = g_n.fetch_add(1, std::memory_order_relaxed);
	if (i < CAP) {
		g_pc[i] = (void*)uc->uc_mcontext.gregs[REG_RIP];
		void **sp = (void**)uc->uc_mcontext.gregs[REG_RSP];
		for (int k = 0; k < NSTK; ++k) g_stk[i * NSTK + k] = sp[k];          // (the interrupted thread's own stack: mapped well beyond 40 words above rsp)
	}
}
void on_mark(int) { g_n.store(0, std::memory_order_relaxed); }      // SIGUSR2 (raise(SIGUSR2) in the profiled program): forget what was sampled so far
struct Mod { uintptr_t lo, hi, base; std::string path; };
std::vector<Mod> g_mods;
int on_phdr(struct dl_phdr_info *info, size_t, void *)
{
	for (int i = 0; i < info->dlpi_phnum; ++i) {
		const ElfW(Phdr) &ph = info->dlpi_phdr[i];
		if (ph.p_type != PT_LOAD || !(ph.p_flags & PF_X)) continue;
		Mod m; m.base = info->dlpi_addr; m.lo = info->dlpi_addr + ph.p_vaddr; m.hi = m.lo + ph.p_memsz;
		m.path = info->dlpi_name && info->dlpi_name[0] ? info->dlpi_name : "[main]";
		g_mods.push_back(m);
	}
	return 0;
}
void finish()
{
	struct itimerval off; memset(&off, 0, sizeof(off));
	setitimer(ITIMER_PROF, &off, 0);
	const char *out = getenv("SPROF_OUT");
	size_t n0 = g_n.load();
	if (n0 < 50) return;                                    // (helper processes of the command line: nothing to say)
	char path[4096];
	snprintf(path, sizeof(path), "%s.%d", out ? out : "sprof.txt", (int)getpid());
	FILE *f = fopen(path, "w");
	if (!f) return;
	dl_iterate_phdr(on_phdr, 0);
	size_t n = g_n.load(); if (n > CAP) n = CAP;
	std::map<std::pair<std::string, uintptr_t>, size_t> cnt;
	for (size_t i = 0; i < n; ++i) {
		const uintptr_t pc = (uintptr_t)g_pc[i];
		const Mod *hit = 0;
		for (const Mod &m : g_mods) if (pc >= m.lo && pc < m.hi) { hit = &m; break; }
		if (hit) ++cnt[std::make_pair(hit->path, pc - hit->base)]; else ++cnt[std::make_pair(std::string("[unknown]"), pc)];
	}
	fprintf(f, "# samples %zu (one per tick of process CPU time: 1 ms asked, the kernel's HZ decides)\n", n);
	{   // the resolved addresses of libc's IFUNC routines (memcpy and friends live under local symbols that nm -D does not list: without these
		// hints their samples are booked on whatever exported symbol happens to precede them)
		static const char *names[] = { "memcpy", "memmove", "memset", "memcmp", "memchr", "strlen", "strcmp", "strchr", "malloc", "free", "realloc", "calloc", "mprotect", "madvise", "brk", "sbrk", "mmap", "munmap", 0 };
		for (int i = 0; names[i]; ++i) {
			const uintptr_t a = (uintptr_t)dlsym(RTLD_DEFAULT, names[i]);
			if (!a) continue;
			for (const Mod &m : g_mods) if (a >= m.lo && a < m.hi) { fprintf(f, "#sym %s %zx %s\n", m.path.c_str(), (size_t)(a - m.base), names[i]); break; }
		}
		char exe[4096]; const ssize_t k = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
		if (k > 0) { exe[k] = 0; fprintf(f, "#main %s\n", exe); }
	}
	for (const auto &kv : cnt) fprintf(f, "%s %zx %zu\n", kv.first.first.c_str(), (size_t)kv.first.second, kv.second);
	// caller chains ("#stk <module> <offset> | <module> <offset> ..."): per sample the pc and the first stack words that point into executable code
	std::map<std::string, size_t> chains;
	for (size_t i = 0; i < n; ++i) {
		std::string key;
		char buf[4200];
		int kept = 0;
		for (int k = -1; k < NSTK && kept < 6; ++k) {
			const uintptr_t v = k < 0 ? (uintptr_t)g_pc[i] : (uintptr_t)g_stk[i * NSTK + k];
			for (const Mod &m : g_mods) if (v >= m.lo && v < m.hi) { snprintf(buf, sizeof(buf), "%s%s %zx", kept ? " | " : "", m.path.c_str(), (size_t)(v - m.base)); key += buf; ++kept; break; }
		}
		++chains[key];
	}
	for (const auto &kv : chains) fprintf(f, "#stk %zu %s\n", kv.second, kv.first.c_str());
	fclose(f);
}
__attribute__((constructor)) void start()
{
	if (getenv("SPROF_OFF")) return;
	g_pc = (void**)calloc(CAP, sizeof(void*));
	g_stk = (void**)calloc(CAP * NSTK, sizeof(void*));
	struct sigaction sa; memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
	sigaction(SIGPROF, &sa, 0);
	struct sigaction sm; memset(&sm, 0, sizeof(sm));
	sm.sa_handler = on_mark; sm.sa_flags = SA_RESTART;
	sigaction(SIGUSR2, &sm, 0);
	struct itimerval it; it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 1000; it.it_value = it.it_interval;
	setitimer(ITIMER_PROF, &it, 0);
	atexit(finish);
}
}
