// Diagnostic microbenchmark: cycles per fp32 MFMA (16x16x4 vs 32x32x2), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* cyc, int iters, float av, float bv) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = (f32x4){0, 0, 0, 0};
    float a = av + threadIdx.x, b = bv + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 16 / NACC; ++s)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* cyc, int iters, float av, float bv) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    float a = av + threadIdx.x, b = bv + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 8 / NACC; ++s)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    unsigned long long h[256];
    const int iters = 2000;
    auto rep = [&](const char* name, int per_iter, int grid) {
        hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
        printf("%-28s grid %3d: %.1f cycles / MFMA\n", name, grid, s / grid / iters / per_iter);
    };
    for (int grid : {1, 64, 256}) {
        hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); hipDeviceSynchronize(); rep("16x16x4 4 acc", 16, grid);
        hipLaunchKernelGGL(k16<2>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); hipDeviceSynchronize(); rep("16x16x4 2 acc", 16, grid);
        hipLaunchKernelGGL(k16<1>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); hipDeviceSynchronize(); rep("16x16x4 1 acc", 16, grid);
        hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); hipDeviceSynchronize(); rep("32x32x2 2 acc", 8, grid);
        hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f); hipDeviceSynchronize(); rep("32x32x2 1 acc", 8, grid);
    }
    return 0;
}
