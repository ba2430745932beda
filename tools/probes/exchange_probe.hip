// What an activation exchange between the CUs of one XCD costs on this part -- the price of the "column split" of the tower
// (DESIGN.md section 6): G workgroups (one per CU, same XCD by round-robin placement: blockIdx % 8 equal) own a slice of a
// layer's output each, publish it (16-B sc1 write-through stores, drained, one agent-scope arrival per workgroup) and then
// read everybody else's slices (sc1 loads) -- per layer of the forward and backward chain, five to six times per step.
//   G = 4   16 rows x 256 columns per cluster, 4 KB per member  (every CU streams 1/4 of the weights)
//   G = 32  128 rows x 256 columns per cluster, 4 KB per member (every CU streams 1/32 of the weights)
// Measured with s_memtime around every round on every workgroup; checks every word that arrives.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/probes/exchange_probe.hip -o /tmp/exchange_probe && /tmp/exchange_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SLICE_F4 = 256;      // 4 KB per member = 256 float4 = one per thread
constexpr int ROUNDS = 24;

__device__ __forceinline__ void st_sc1(f32x4* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 ld_sc1(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned ld_u32_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int G>
__global__ __launch_bounds__(256) void k_exchange(f32x4* buf, unsigned* counters, unsigned long long* cycles, unsigned* bad,
                                                  unsigned long long* clk) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int xcd = b & 7, slot = b >> 3;              // 32 workgroups share an XCD under round-robin placement
    const int cluster = xcd * (32 / G) + slot / G, member = slot % G;
    __shared__ unsigned ok;
    unsigned long long t_all = 0;
    unsigned wrong = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        f32x4* mine = buf + ((size_t)(cluster * 2 + (r & 1)) * G + member) * SLICE_F4;
        const float tag = (float)(cluster * 1000 + member * 10 + r);
        st_sc1(mine + tid, (f32x4){tag, (float)tid, tag + 1.f, (float)r});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* cnt = counters + (size_t)cluster * ROUNDS + r;
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (ld_u32_sc1(cnt) < (unsigned)G && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            ok = spins < (1 << 22);
        }
        __syncthreads();
        // read every other member's slice: (G - 1) float4 per thread, all in flight
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int m0 = 0; m0 < G; m0 += 8) {
            f32x4 v[8];
            const f32x4* p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = buf + ((size_t)(cluster * 2 + (r & 1)) * G + (m0 + u) % G) * SLICE_F4 + tid;
            // (eight loads in flight and their wait in ONE statement: the compiler must not touch the results earlier)
            asm volatile(
                "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
                "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                : "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = (m0 + u) % G;
                if (m0 + u < G) {
                    const float want = (float)(cluster * 1000 + m * 10 + r);
                    if (v[u][0] != want || v[u][1] != (float)tid || v[u][3] != (float)r) ++wrong;
                    acc += v[u];
                }
            }
        }
        if (acc[0] == -1.f) buf[0] = acc;          // keep the loads
        __syncthreads();
        t_all += __builtin_amdgcn_s_memtime() - t0;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (wrong || !ok) atomicAdd(bad, wrong + (ok ? 0u : 1u));
    if (tid == 0) {
        cycles[b] = t_all / ROUNDS;
        clk[2 * b] = c1 - c0;
        clk[2 * b + 1] = r1 - r0;
    }
}

template <int G>
void run(const char* what) {
    f32x4* buf;
    unsigned *cnt, *bad;
    unsigned long long *cyc, *clk;
    const int clusters = 256 / G;
    hipMalloc(&buf, (size_t)clusters * 2 * G * SLICE_F4 * sizeof(f32x4));
    hipMalloc(&cnt, (size_t)clusters * ROUNDS * sizeof(unsigned));
    hipMalloc(&bad, sizeof(unsigned));
    hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    hipMalloc(&clk, 512 * sizeof(unsigned long long));
    std::vector<double> us;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(cnt, 0, (size_t)clusters * ROUNDS * sizeof(unsigned));
        hipMemset(bad, 0, sizeof(unsigned));
        hipLaunchKernelGGL(k_exchange<G>, dim3(256), dim3(256), 0, 0, buf, cnt, cyc, bad, clk);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256), hc(512);
        unsigned hb = 0;
        hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), clk, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipMemcpy(&hb, bad, sizeof(unsigned), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double ghz = (double)hc[0] / ((double)hc[1] * 10.0) ;      // s_memrealtime ticks at 100 MHz
        if (rep) us.push_back((double)h[128] / (ghz * 1e3));
        printf("%s rep %d: median %llu cycles / exchange (min %llu, max %llu), clock %.2f GHz, wrong words %u\n", what, rep, h[128],
               h[0], h[255], ghz, hb);
    }
    std::sort(us.begin(), us.end());
    printf("%s: %.2f us per exchange (median of 4 launches, %d rounds each)\n", what, us[us.size() / 2], ROUNDS);
    hipFree(buf); hipFree(cnt); hipFree(bad); hipFree(cyc); hipFree(clk);
}

int main() {
    run<4>("G=4  (16 KB per cluster)");
    run<32>("G=32 (128 KB per cluster)");
    return 0;
}
