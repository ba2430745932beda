// Does what one kernel leaves in an XCD's L2 survive the kernel boundary?  ("cold starts": every
// tower launch starts with a round trip to the infinity cache / HBM for the batch rows of its step; workgroup i of step t
// and workgroup i of step t + 1 land on the same XCD under round-robin placement, so step t could TOUCH the rows step
// t + 1 will need -- if the lines are still there when the next kernel starts.)
//   kernel A  workgroup b loads its 4 KB slice of `buf` (touch), or nothing (cold), or the slice of workgroup b + 1
//             (another XCD touched it)
//   kernel B  workgroup b times one dependent round of loads over its slice (s_memtime, one float4 per thread)
// Between A and B: an ordinary kernel boundary on one stream (what the step kernels have).  The buffer is 256 x 4 KB per
// repetition and every repetition uses fresh addresses (nothing is warm by accident).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l2_survive_probe.hip -o /tmp/l2_probe && /tmp/l2_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WG = 256, SLICE_F4 = 256, REPS = 20;

__global__ __launch_bounds__(256) void k_touch(f32x4* buf, int mode, float* sink) {
    if (mode == 0) return;                                      // cold: nobody touches
    const int b = (mode == 1 || mode == 3 || mode == 5) ? blockIdx.x : (blockIdx.x + 1) % WG;   // own slice / the neighbour's (another XCD)
    f32x4* p = buf + (size_t)b * SLICE_F4 + threadIdx.x;
    if (mode <= 2) {
        const f32x4 v = *p;
        if (v[0] == 12345.f) sink[0] = v[1];                    // keep the load
    } else if (mode <= 4) {                                     // 3 / 4: WRITTEN with write-through stores (what the step kernels' workspace stores are)
        const f32x4 v = (f32x4){1.f, 2.f, 3.f, (float)b};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    } else {                                                    // 5 / 6: plain (write-back) stores
        *p = (f32x4){1.f, 2.f, 3.f, (float)b};
    }
}
__global__ __launch_bounds__(256) void k_timed(const f32x4* buf, unsigned long long* cycles, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const f32x4 v = buf[(size_t)blockIdx.x * SLICE_F4 + threadIdx.x];
    float s = v[0] + v[1] + v[2] + v[3];
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(s) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.f) sink[1] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
// a kernel in between that writes other memory (the step has k_wgrad_adam between two towers)
__global__ __launch_bounds__(256) void k_other(float* scratch) { scratch[(size_t)blockIdx.x * 256 + threadIdx.x] += 1.f; }

int main() {
    f32x4* buf;
    unsigned long long* cyc;
    float *sink, *scratch;
    const size_t per_rep = (size_t)WG * SLICE_F4;
    hipMalloc(&buf, per_rep * REPS * 7 * 2 * sizeof(f32x4));
    hipMemset(buf, 0, per_rep * REPS * 7 * 2 * sizeof(f32x4));
    hipMalloc(&cyc, WG * sizeof(unsigned long long));
    hipMalloc(&sink, 16);
    hipMalloc(&scratch, WG * 256 * sizeof(float));
    hipMemset(scratch, 0, WG * 256 * sizeof(float));
    // push the memset's own lines out of the 8 x 4 MB of L2 (the infinity cache keeps what it keeps: the step's rows were
    // written a few launches earlier too)
    float* flush;
    hipMalloc(&flush, (size_t)256 << 20);
    hipMemset(flush, 0, (size_t)256 << 20);
    hipDeviceSynchronize();
    const char* names[7] = {"cold (nobody touched the slice)", "read by the SAME workgroup id in the kernel before",
                            "read by workgroup id + 1 (another XCD) in the kernel before",
                            "WRITTEN (sc1 write-through) by the same workgroup id", "WRITTEN (sc1 write-through) by workgroup id + 1",
                            "WRITTEN (plain stores) by the same workgroup id", "WRITTEN (plain stores) by workgroup id + 1"};
    for (int between = 0; between < 2; ++between) {
        printf(between ? "-- with another kernel between the two (touch, k_other, timed)\n" : "-- back to back (touch, timed)\n");
        for (int mode = 0; mode < 7; ++mode) {
            std::vector<unsigned long long> med;
            for (int r = 0; r < REPS; ++r) {
                f32x4* p = buf + ((size_t)(between * 7 + mode) * REPS + r) * per_rep;
                hipLaunchKernelGGL(k_touch, dim3(WG), dim3(256), 0, 0, p, mode, sink);
                if (between) hipLaunchKernelGGL(k_other, dim3(WG), dim3(256), 0, 0, scratch);
                hipLaunchKernelGGL(k_timed, dim3(WG), dim3(256), 0, 0, p, cyc, sink);
                std::vector<unsigned long long> h(WG);
                hipMemcpy(h.data(), cyc, WG * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.end());
                med.push_back(h[WG / 2]);
            }
            std::sort(med.begin(), med.end());
            printf("  %-66s median %5llu cycles (min %llu, max %llu over %d repetitions of the per-launch median)\n", names[mode],
                   med[REPS / 2], med.front(), med.back(), REPS);
        }
    }
    return 0;
}
