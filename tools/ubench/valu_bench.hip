// valu_bench.hip — pure-VALU issue-rate micro-benchmark for gfx950 (settles the ceiling used in DESIGN.md):
// how many wave64 lane-ops per second the chip sustains for the integer instructions the ksw cell is made of.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_bench valu_bench.hip ; run: ./valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REP16(x) x x x x x x x x x x x x x x x x

// OP: asm text with %0 (in/out accumulator) and %1 (second operand). ILP independent accumulators per thread.
#define DEF_KERNEL(NAME, ASM2)                                                                          \
template <int ILP> __global__ __launch_bounds__(64) void NAME(int iters, int *out, int seed)            \
{                                                                                                       \
	int a[ILP];                                                                                         \
	int b = seed + threadIdx.x, c = seed * 3 + 1;                                                       \
	_Pragma("unroll") for (int i = 0; i < ILP; ++i) a[i] = seed + i + threadIdx.x;                      \
	for (int it = 0; it < iters; ++it) {                                                                \
		REP16(_Pragma("unroll") for (int i = 0; i < ILP; ++i) asm volatile(ASM2 : "+v"(a[i]) : "v"(b), "v"(c));) \
	}                                                                                                   \
	int s = 0;                                                                                          \
	_Pragma("unroll") for (int i = 0; i < ILP; ++i) s ^= a[i];                                          \
	if (s == 0x12345678) out[0] = s;                                                                    \
}

DEF_KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1")
DEF_KERNEL(k_max_i32, "v_max_i32 %0, %0, %1")
DEF_KERNEL(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
DEF_KERNEL(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEF_KERNEL(k_pk_sub_u16, "v_pk_sub_u16 %0, %0, %1")
DEF_KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEF_KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
DEF_KERNEL(k_pk_lshl_b16, "v_pk_lshlrev_b16 %0, 1, %0")
DEF_KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_cmp_cnd, "v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF_KERNEL(k_dpp_shr1, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL(k_dpp_rowshr1, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL(k_add_dpp, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
DEF_KERNEL(k_readlane, "v_readlane_b32 s20, %0, 15\n v_add_u32 %0, s20, %0")
DEF_KERNEL(k_bperm, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
DEF_KERNEL(k_sdwa_add, "v_add_u32_sdwa %0, %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3")

template <class K> static double run(K kern, int waves_per_simd, int iters, int ilp, int ops_per_asm, int *d_out, int n_cu)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int grid = n_cu * 4 * waves_per_simd;
	hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, 16, d_out, 1);
	hipDeviceSynchronize();
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, iters, d_out, rep + 2);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	const double lane_ops = (double)grid * 64 * (double)iters * 16 * ilp * ops_per_asm;
	hipEventDestroy(e0); hipEventDestroy(e1);
	return lane_ops / (best * 1e-3) / 1e12;       // T lane-ops/s
}

#define ROW(NAME, OPS) do { \
	printf("%-18s", #NAME); \
	for (int w = 1; w <= 8; w *= 2) { \
		printf("  w%d: ilp1 %6.2f ilp4 %6.2f", w, run(NAME<1>, w, iters, 1, OPS, d_out, n_cu), run(NAME<4>, w, iters, 4, OPS, d_out, n_cu)); } \
	printf("\n"); fflush(stdout); } while (0)

int main(int argc, char **argv)
{
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
	const int n_cu = p.multiProcessorCount;
	int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	printf("device %s, %d CUs, clock %d kHz; numbers are T lane-ops/s (wave64 instr x 64 lanes), grid = CUs x 4 SIMDs x w waves\n", p.name, n_cu, clk);
	printf("reference points: 16 lanes/SIMD/clk -> %.1f T/s, 32 lanes/SIMD/clk -> %.1f T/s at the reported clock\n", n_cu * 4 * 16 * (clk * 1e3) / 1e12, n_cu * 4 * 32 * (clk * 1e3) / 1e12);
	int *d_out; hipMalloc(&d_out, 64);
	const int iters = argc > 1 ? atoi(argv[1]) : 2048;
	ROW(k_add_u32, 1); ROW(k_sub_u32, 1); ROW(k_max_i32, 1); ROW(k_max3_i32, 1); ROW(k_add3_u32, 1); ROW(k_and_or, 1);
	ROW(k_pk_add_u16, 1); ROW(k_pk_sub_u16, 1); ROW(k_pk_max_i16, 1); ROW(k_pk_min_i16, 1); ROW(k_pk_lshl_b16, 1); ROW(k_pk_mad_u16, 1);
	ROW(k_perm, 1); ROW(k_cndmask, 1); ROW(k_cmp_cnd, 2); ROW(k_sad_u8, 1); ROW(k_sdwa_add, 1);
	ROW(k_dpp_shr1, 1); ROW(k_dpp_rowshr1, 1); ROW(k_add_dpp, 1); ROW(k_readlane, 2); ROW(k_bperm, 1);
	return 0;
}
