// wm_rt.hip — libwmgpu.so, runtime unit: errors, device contexts (stream + arena + pinned slab + events), process-wide defaults. No kernels here.
// There is no CPU compute path in this library: every entry point needs a HIP device (wm_ctx_create fails with WM_ENODEV without one).
#include "wm_rt.h"

// ======================================================================================================
// host
// ======================================================================================================
static thread_local char g_err[512] = "";
const char *wm_err_text() { return g_err; }
void wm_err_clear() { g_err[0] = 0; }
int set_err(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}


void *pin_take(wm_ctx_s *c, size_t bytes, size_t *mark)
{
	if (!c->pin) {
		if (c->pin_bytes == (size_t)-1) return 0;                        // allocation failed before: stay pageable
		const size_t want = (size_t)(getenv("WM_PINNED_MB") ? atoll(getenv("WM_PINNED_MB")) : 3072) << 20;
		if (want == 0 || hipHostMalloc((void**)&c->pin, want, hipHostMallocDefault) != hipSuccess) { c->pin = 0; c->pin_bytes = (size_t)-1; (void)hipGetLastError(); return 0; }
		c->pin_bytes = want; c->pin_used = 0;
	}
	const size_t off = (c->pin_used + 255) & ~(size_t)255;
	if (off + bytes > c->pin_bytes) return 0;
	*mark = c->pin_used;
	c->pin_used = off + bytes;
	return c->pin + off;
}
void pin_release(wm_ctx_s *c, size_t mark) { c->pin_used = mark; }


extern "C" const char *wm_last_error(void) { return g_err; }
#ifndef WM_BUILD_DEFINES
#define WM_BUILD_DEFINES ""
#endif
extern "C" const char *wm_build_defines(void) { return WM_BUILD_DEFINES; }      // kernel-variant defines this library was compiled with (winnowmap_amd/build.py)

// wait for everything queued on the context's stream WITHOUT burning a host core: hipStreamSynchronize — and, on the GPU boxes, also
// hipEventSynchronize on a blocking-sync event (thread CPU time == wall time inside the batched calls, profiles/r02c_bench_hub.json) —
// spin for as long as the kernels run, and the container's CPU quota is the scarce resource. So: record an event, poll it, sleep in
// between (50 us doubling to 1 ms; the batches take tens of milliseconds). WM_SPIN_SYNC=1 restores hipStreamSynchronize (A/B).
hipError_t ctx_sync(wm_ctx_s *c)
{
	static const bool spin = getenv("WM_SPIN_SYNC") != 0;
	if (spin) return hipStreamSynchronize(c->stream);
	hipError_t e = hipEventRecord(c->sync_ev, c->stream);
	if (e != hipSuccess) return e;
	int us = 50;
	for (;;) {
		e = hipEventQuery(c->sync_ev);
		if (e != hipErrorNotReady) return e;
		std::this_thread::sleep_for(std::chrono::microseconds(us));
		if (us < 1000) us *= 2;
	}
}

// ROCm maps HIP streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and reads it when the runtime initialises: a library
// constructor sets the default the mapper is tuned for (6 contexts + 14 side streams) before any HIP call of this process can have happened
// through this library; a value given by the user wins. (Python callers get the same default from winnowmap_amd/__init__.py.)
// (this is the one setting that has to happen at load time — the HIP runtime reads the variable when it initialises. WM_NO_PROCESS_DEFAULTS=1
// leaves the process alone; include/wm_gpu.h documents both process-wide settings.)
// Round 6: no library constructor any more — loading the library changes nothing in the process. The default is set by the first
// wm_ctx_create / wm_device_count of the process, immediately before this library's first HIP call; if the host program has already initialised the HIP
// runtime by then (it read the variable at that moment), the setting is simply too late and the program's own environment rules.
static void wm_default_hw_queues()
{
	static std::once_flag once;
	std::call_once(once, [] { if (!getenv("WM_NO_PROCESS_DEFAULTS")) setenv("GPU_MAX_HW_QUEUES", "20", 0); });
}

// The mapping calls allocate and free their per-call tables (tens of MB per batched call, from 16+ worker threads) at a rate at which glibc's defaults
// turn into system calls: a worker's malloc arena grows in 128-KB steps (one mprotect each), gives the memory back as soon as it is free, deletes and
// re-creates its 64-MB heaps, and serves anything above the mmap threshold by mmap / munmap. The sampling profile of a bench run had 58 % of the
// host's CPU samples inside mprotect (profiles/r04l_host_sampling_profile.txt) — with the address-space lock held, i.e. with every other thread's page
// faults waiting. Keep the memory instead: grow in 64-MB steps, never trim, allocate up to 32 MB from the arenas: -15 % host CPU, +9 % throughput in one
// GPU call (profiles/r04m_malloc_tuning.txt). Process-wide, like GPU_MAX_HW_QUEUES; WM_MALLOPT=0 or any MALLOC_* tunable of the caller's own wins.
// Applied when the first mapper of the process is created (not at load time: a program that only links the library for its batched operations keeps
// glibc's defaults — ADVICE r4); WM_MALLOPT=0 / WM_NO_PROCESS_DEFAULTS=1 switch it off.
void wm_default_malloc()
{
	static std::once_flag once;
	std::call_once(once, [] {
		const char *off = getenv("WM_MALLOPT");
		if ((off && atoi(off) == 0) || getenv("WM_NO_PROCESS_DEFAULTS")) return;
		if (!getenv("MALLOC_TOP_PAD_")) mallopt(M_TOP_PAD, 64 << 20);
		if (!getenv("MALLOC_TRIM_THRESHOLD_")) mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
		if (!getenv("MALLOC_MMAP_THRESHOLD_")) mallopt(M_MMAP_THRESHOLD, 32 << 20);
	});
}

// one recorded event per device: the zero of the interval clock above (hipEventElapsedTime works between events of different streams)
hipEvent_t device_base_event(int device)
{
	static std::mutex mu;
	static std::vector<hipEvent_t> ev;
	std::lock_guard<std::mutex> lk(mu);
	if ((int)ev.size() <= device) ev.resize(device + 1, (hipEvent_t)0);
	if (!ev[device]) {
		hipEvent_t e;
		if (hipSetDevice(device) != hipSuccess || hipEventCreate(&e) != hipSuccess) return 0;
		if (hipEventRecord(e, 0) != hipSuccess || hipEventSynchronize(e) != hipSuccess) { hipEventDestroy(e); return 0; }
		ev[device] = e;
	}
	return ev[device];
}

// how the mapper's side-stream pool of P streams is divided among light | heavy | huge ksw calls (WM_SIDE_SPLIT=light,heavy; the rest = huge)
void side_split(int P, int *light, int *heavy)
{
	int a = (P * 3 + 3) / 7, h = (P * 2 + 3) / 7;           // 14 streams: 6 | 4 | 4 (profiles/r04g_sched_sweep.txt); 10: 4 | 3 | 3
	if (const char *e = getenv("WM_SIDE_SPLIT")) sscanf(e, "%d,%d", &a, &h);
	*light = std::max(0, a); *heavy = std::max(0, h);
}

extern "C" int wm_device_count(void)
{
	wm_default_hw_queues();
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" int wm_ctx_create(int device, size_t arena_bytes, wm_ctx_t **out)
{
	int n = 0;
	*out = 0;
	wm_default_hw_queues();
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_err(WM_ENODEV, "no HIP device visible (libwmgpu has no CPU fallback)");
	if (device < 0 || device >= n) return set_err(WM_EINVAL, "device %d out of range (%d visible)", device, n);
	HIPCHK(hipSetDevice(device));
	wm_ctx_t *c = new wm_ctx_t();
	c->device = device;
	if (arena_bytes == 0) {
		size_t fr = 0, tot = 0;
		HIPCHK(hipMemGetInfo(&fr, &tot));
		arena_bytes = fr / 4 < ((size_t)24 << 30) ? fr / 4 : ((size_t)24 << 30);
	}
	c->arena_bytes = arena_bytes;
	HIPCHK(hipMalloc((void**)&c->arena, arena_bytes));
	HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	for (int i = 0; i < 4; ++i) c->kstream[i] = 0;
	c->side_pool = 0; c->n_side_pool = 0; c->side_next = 0;
	for (int i = 0; i < 3; ++i) c->owned_next[i] = 0;
	for (int i = 0; i < 5; ++i) HIPCHK(hipEventCreateWithFlags(&c->kev[i], hipEventDisableTiming));
	for (int k = 0; k < WM_KSW_NCLASS; ++k) { HIPCHK(hipEventCreate(&c->cev[k][0])); HIPCHK(hipEventCreate(&c->cev[k][1])); c->k_ms[k] = 0; c->k_cells[k] = c->k_launches[k] = 0; }
	for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&c->ev[i]));
	HIPCHK(hipEventCreateWithFlags(&c->sync_ev, hipEventBlockingSync | hipEventDisableTiming));
	c->arena_used = 0; c->last_ms = 0; c->aux_ms = 0; c->host_threads = 1; c->pin = 0; c->pin_bytes = c->pin_used = 0; c->have_index = false; c->owns_index = false; c->d_hkey = c->d_hval = c->d_P = 0; c->d_bloom = 0;
	c->owns_filter = false;
	c->pin_small = 0;
	if (hipHostMalloc((void**)&c->pin_small, 256, hipHostMallocDefault) != hipSuccess) { c->pin_small = 0; (void)hipGetLastError(); }
	c->d_S = 0; c->d_reads = 0; c->d_reads_nm = 0; c->reads_bytes = c->reads_cap = 0; c->owns_reads = false;
	*out = c;
	return WM_OK;
}

extern "C" void wm_ctx_destroy(wm_ctx_t *c)
{
	if (!c) return;
	hipSetDevice(c->device);
	hipStreamSynchronize(c->stream);
	for (int i = 0; i < 4; ++i) hipEventDestroy(c->ev[i]);
	hipEventDestroy(c->sync_ev);
	hipStreamDestroy(c->stream);
	if (c->pin) hipHostFree(c->pin);
	if (c->pin_small) hipHostFree(c->pin_small);
	for (int i = 0; i < 4; ++i) if (c->kstream[i]) hipStreamDestroy(c->kstream[i]);
	for (hipStream_t st : c->owned_pool) hipStreamDestroy(st);
	for (int i = 0; i < 5; ++i) hipEventDestroy(c->kev[i]);
	for (int k = 0; k < WM_KSW_NCLASS; ++k) { hipEventDestroy(c->cev[k][0]); hipEventDestroy(c->cev[k][1]); }
	hipFree(c->arena);
	if (c->have_index && c->owns_index) { hipFree(c->d_hkey); hipFree(c->d_hval); hipFree(c->d_P); hipFree(c->d_bloom); hipFree(c->d_S); }
	if (c->d_reads && c->owns_reads) hipFree(c->d_reads);
	if (!c->have_index && c->owns_filter && c->d_bloom) hipFree(c->d_bloom);
	delete c;
}

extern "C" float wm_last_kernel_ms(const wm_ctx_t *c) { return c ? c->last_ms : 0.f; }
extern "C" int wm_ctx_device(const wm_ctx_t *c) { return c ? c->device : -1; }

void *arena_take(wm_ctx_t *c, size_t bytes)
{
	size_t a = (c->arena_used + 255) & ~(size_t)255;
	if (a + bytes > c->arena_bytes) return 0;
	c->arena_used = a + bytes;
	return c->arena + a;
}


