// Probe: what the kernel-argument fetch costs at wave start, and whether kernarg preloading
// (-mllvm -amdgpu-kernarg-preload-count=N: the first N dwords of arguments arrive in SGPRs with the wave) is honoured
// by this box's firmware.  Build twice:
//   hipcc -O3 --offload-arch=gfx950 tools/probes/kernarg_preload_probe.hip -o probe_plain
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 tools/probes/kernarg_preload_probe.hip -o probe_preload
// Each wave stamps s_memtime at entry and again once a load THROUGH a pointer argument has returned (the buffer is
// small and L2-warm after the first launch; a fresh kernarg slot per launch is not).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void probe(const float* x, unsigned long long* out, int n) {
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const float v = x[(blockIdx.x * 256 + threadIdx.x) % n];
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(v) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    const int n = 1 << 16, G = 256;
    float* x; unsigned long long* out;
    hipMalloc(&x, n * 4); hipMalloc(&out, G * 8); hipMemset(x, 0, n * 4);
    std::vector<unsigned long long> h(G);
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, x, out, n);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("launch %d: entry -> first load returned: median %llu cycles (min %llu, max %llu)\n", rep, h[G / 2], h[0], h[G - 1]);
    }
    return 0;
}
