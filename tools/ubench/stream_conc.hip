// stream_conc.hip — how many kernels from different HIP streams run at the same time on one MI355X?
// K streams each get one launch of a single-wave kernel that spins for ~10 ms; if all K overlap the wall time stays ~10 ms,
// if the device runs at most C at once it grows like ceil(K / C) * 10 ms. Run with GPU_MAX_HW_QUEUES=4 (default) and =32.
// build: hipcc --offload-arch=gfx950 -O2 -o stream_conc stream_conc.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
__global__ void spin_kernel(long long ticks, int *out)
{
	const long long t0 = wall_clock64();
	long long t = t0;
	int acc = 0;
	while (t - t0 < ticks) { t = wall_clock64(); ++acc; }
	if (out) out[blockIdx.x] = acc;
}
int main()
{
	int rate_khz = 0;
	hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
	const long long ticks = (long long)rate_khz * 10;            // 10 ms
	printf("GPU_MAX_HW_QUEUES=%s wall clock %d kHz\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)", rate_khz);
	const int KMAX = 48;
	std::vector<hipStream_t> st(KMAX);
	for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	int *d; hipMalloc(&d, 4096 * 4);
	hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], ticks / 10, d);
	hipDeviceSynchronize();
	for (int blocks : {1, 64}) {
		for (int K : {1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 48}) {
			auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < K; ++i) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, st[i], ticks, d);
			hipDeviceSynchronize();
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			printf("blocks/kernel %3d  streams %2d  wall %7.2f ms  -> ~%.1f kernels at once\n", blocks, K, ms, K * 10.0 / ms);
		}
	}
	return 0;
}
