// How long does a kernel wait for its BY-VALUE argument block?  (round 5: with HIP_FORCE_DEV_KERNARG=0 -- kernel arguments in host
// memory -- k_tower4 takes 15.4 instead of 13.4 us, k_wgrad_adam 9.2 instead of 8.2; the default on this ROCm keeps them in
// device memory.  What does the device-memory fetch still cost, and does a field of a persistent device buffer come faster?)
//   k_struct   reads one float of a 512-B struct passed by value (an s_load from the kernarg segment: fresh memory every launch)
//   k_buffer   reads the same float from a persistent device buffer (pointer = first argument), touched by the previous launch
// each workgroup's wave 0 times the dependent scalar load with s_memtime; a dummy kernel that writes other memory runs in between
// (the step kernels alternate).  hipcc --offload-arch=gfx950 -O3 tools/probes/kernarg_latency_probe.hip -o /tmp/ka_probe && /tmp/ka_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

struct Big { float f[128]; };
constexpr int WG = 256, REPS = 200;

__global__ __launch_bounds__(256) void k_struct(unsigned long long* cycles, float* sink, const Big a) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const float v = a.f[96];
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(v) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (v == 12345.f) sink[0] = v;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_buffer(const Big* a, unsigned long long* cycles, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const float v = a->f[96];
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(v) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (v == 12345.f) sink[0] = v;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_other(float* scratch) { scratch[(size_t)blockIdx.x * 256 + threadIdx.x] += 1.f; }

static void report(const char* name, std::vector<unsigned long long>& all) {
    std::sort(all.begin(), all.end());
    printf("%-52s median %6llu  p10 %6llu  p90 %6llu cycles (s_memtime)\n", name, all[all.size() / 2], all[all.size() / 10],
           all[all.size() * 9 / 10]);
}

int main() {
    unsigned long long* cyc;
    float *sink, *scratch;
    Big* dev;
    hipMalloc(&cyc, WG * sizeof(unsigned long long));
    hipMalloc(&sink, 16);
    hipMalloc(&scratch, (size_t)WG * 256 * 4);
    hipMalloc(&dev, sizeof(Big));
    hipMemset(scratch, 0, (size_t)WG * 256 * 4);
    Big h;
    for (int i = 0; i < 128; ++i) h.f[i] = (float)i;
    hipMemcpy(dev, &h, sizeof(Big), hipMemcpyHostToDevice);
    std::vector<unsigned long long> host(WG), all;
    for (int mode = 0; mode < 2; ++mode) {
        all.clear();
        for (int r = 0; r < REPS; ++r) {
            h.f[96] = (float)r;
            hipLaunchKernelGGL(k_other, dim3(WG), dim3(256), 0, 0, scratch);
            if (mode == 0) hipLaunchKernelGGL(k_struct, dim3(WG), dim3(256), 0, 0, cyc, sink, h);
            else hipLaunchKernelGGL(k_buffer, dim3(WG), dim3(256), 0, 0, dev, cyc, sink);
            hipMemcpy(host.data(), cyc, WG * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            if (r >= 10) all.insert(all.end(), host.begin(), host.end());
        }
        report(mode == 0 ? "field of a 512-B by-value struct (kernarg segment)" : "field of a persistent device buffer", all);
    }
    return 0;
}
