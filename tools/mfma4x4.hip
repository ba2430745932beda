// Diagnostic: lane layout, A-broadcast (cbsz/abid) and rate of v_mfma_f32_4x4x1_16b_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_layout(float* out, int mode) {
    const int l = threadIdx.x;
    float a = (float)(l + 1), b = 1000.f * (float)(l + 1);
    f32x4 c = {0, 0, 0, 0};
    f32x4 d;
    if (mode == 0) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    else d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);      // cbsz=4: A of block 0 broadcast to all 16
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[4];
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x, b = threadIdx.x * 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[t], 4, 0, 0);
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 65536 * 4); hipMalloc(&cyc, 256 * 8);
    float h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out, mode);
        hipMemcpy(h, out, 256 * 4, hipMemcpyDeviceToHost);
        printf("mode %d (D[lane][reg] / 1000 = a_src * b_src):\n", mode);
        for (int l : {0, 1, 2, 3, 4, 5, 9, 63}) printf("  lane %2d: %g %g %g %g\n", l, h[l*4]/1000, h[l*4+1]/1000, h[l*4+2]/1000, h[l*4+3]/1000);
    }
    unsigned long long hc[256];
    for (int grid : {1, 256}) {
        hipLaunchKernelGGL(k_rate, dim3(grid), dim3(256), 0, 0, out, cyc, 2000);
        hipMemcpy(hc, cyc, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; ++i) s += hc[i];
        printf("4x4x1 16b, 4 acc, grid %d: %.2f cycles / MFMA\n", grid, s / grid / 2000 / 16);
    }
    return 0;
}
