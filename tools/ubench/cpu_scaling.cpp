// cpu_scaling.cpp — what the host really offers: aggregate throughput of N threads for (a) pure integer compute and
// (b) malloc/free of mapper-sized buffers (64 KB - 4 MB: glibc serves the large ones with mmap/munmap -> TLB shootdowns).
// Build: g++ -O2 -pthread -o cpu_scaling cpu_scaling.cpp
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <thread>
#include <vector>
#include <chrono>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static uint64_t compute(uint64_t n, uint64_t seed) { uint64_t x = seed | 1; for (uint64_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; } return x; }
static uint64_t mallocs(int n, uint64_t seed)
{
	uint64_t x = seed | 1, s = 0;
	for (int i = 0; i < n; ++i) {
		x ^= x << 13; x ^= x >> 7; x ^= x << 17;
		const size_t sz = (size_t)65536 << (x % 7);
		char *p = (char*)malloc(sz);
		for (size_t k = 0; k < sz; k += 4096) p[k] = (char)k;      // touch every page
		s += p[sz / 2];
		free(p);
	}
	return s;
}
int main(int argc, char **argv)
{
	const int maxt = argc > 1 ? atoi(argv[1]) : 256;
	std::atomic<uint64_t> sink(0);
	printf("%8s %14s %10s | %14s %10s\n", "threads", "compute Mit/s", "speedup", "malloc k/s", "speedup");
	double c1 = 0, m1 = 0;
	for (int t = 1; t <= maxt; t = t < 16 ? t * 2 : t < 64 ? t + 16 : t * 2) {
		const uint64_t n = 60000000;
		double t0 = now();
		{ std::vector<std::thread> th; for (int i = 0; i < t; ++i) th.emplace_back([&, i] { sink += compute(n, i + 1); }); for (auto &x : th) x.join(); }
		const double ct = now() - t0, crate = (double)n * t / ct / 1e6;
		const int nm = 3000;
		t0 = now();
		{ std::vector<std::thread> th; for (int i = 0; i < t; ++i) th.emplace_back([&, i] { sink += mallocs(nm, i + 1); }); for (auto &x : th) x.join(); }
		const double mt = now() - t0, mrate = (double)nm * t / mt / 1e3;
		if (t == 1) c1 = crate, m1 = mrate;
		printf("%8d %14.1f %10.2f | %14.1f %10.2f\n", t, crate, crate / c1, mrate, mrate / m1);
		fflush(stdout);
	}
	return (int)(sink.load() & 1);
}
