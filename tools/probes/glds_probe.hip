// Probe: LDS-DMA (global_load_lds_dword) into a 160 KB LDS allocation with a per-row rotated image; checks the
// bytes above 64 KB land where expected and times the 128 KB stream and the two read patterns k_tower4 would use.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/glds_probe.hip -o /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))
constexpr int K = 256, N = 128, BASE = 7488;   // floats ahead of the image (as in the tower's LDS map)
__global__ __launch_bounds__(512) void probe(const float* W, float* out, unsigned long long* stamps, int reps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* img = smem + BASE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long t0, t1, t2, t3;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const int k = 32 * w + kk + (lane >> 5);                 // two rows per instruction
            const float* src = W + k * N + 4 * ((lane & 31) ^ (k & 31));
            __builtin_amdgcn_global_load_lds(GLBP(src), LDSP(img + (32 * w + kk) * N), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    // forward pattern: lane l reads W[k][l], W[k][l + 64]
    float s0 = 0.f, s1 = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 32 * w + kk;
            const f32x2 v = *reinterpret_cast<const f32x2*>(img + k * N + 4 * ((lane >> 1) ^ (k & 31)) + 2 * (lane & 1));
            s0 += v[0];
            s1 += v[1];
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    // backward pattern: lane l reads W[l + 64 t][c], t < 4
    float q = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int cq = 4 * w; cq < 4 * w + 4; ++cq) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = lane + 64 * t;
                const f32x4 v = *reinterpret_cast<const f32x4*>(img + r * N + 4 * (cq ^ (r & 31)));
                q += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3)::"memory");
    // check: column sums over this wave's k rows (forward) and row sums over its columns (backward)
    out[(blockIdx.x * 8 + w) * 192 + lane] = s0;
    out[(blockIdx.x * 8 + w) * 192 + 64 + lane] = s1;
    out[(blockIdx.x * 8 + w) * 192 + 128 + lane] = q;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = t1 - t0;
        stamps[blockIdx.x * 4 + 1] = t2 - t1;
        stamps[blockIdx.x * 4 + 2] = t3 - t2;
    }
}
int main() {
    std::vector<float> hW(K * N);
    for (int i = 0; i < K * N; ++i) hW[i] = (float)((i * 2654435761u) >> 20) / 4096.f;
    float *dW, *dout; unsigned long long* dst;
    const int G = 256;
    hipMalloc(&dW, K * N * 4); hipMalloc(&dout, G * 8 * 192 * 4); hipMalloc(&dst, G * 4 * 8);
    hipMemcpy(dW, hW.data(), K * N * 4, hipMemcpyHostToDevice);
    const size_t lds = (BASE + K * N) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("lds %zu bytes, attr %s\n", lds, hipGetErrorString(e));
    for (int reps : {1, 1, 4}) {
        hipLaunchKernelGGL(probe, dim3(G), dim3(512), lds, 0, dW, dout, dst, reps);
        e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("launch: %s\n", hipGetErrorString(e)); return 1; }
        std::vector<float> o(G * 8 * 192); std::vector<unsigned long long> st(G * 4);
        hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int b = 0; b < G; b += 85) for (int w = 0; w < 8; ++w) for (int l = 0; l < 64; ++l) {
            float s0 = 0, s1 = 0, q = 0;
            for (int r = 0; r < reps; ++r) {
                for (int kk = 0; kk < 32; ++kk) { s0 += hW[(32 * w + kk) * N + 2 * l]; s1 += hW[(32 * w + kk) * N + 2 * l + 1]; }
                for (int cq = 4 * w; cq < 4 * w + 4; ++cq) for (int t = 0; t < 4; ++t) { const float* v = &hW[(l + 64 * t) * N + 4 * cq]; q += (v[0] + v[1]) + (v[2] + v[3]); }
            }
            const float* p = &o[(b * 8 + w) * 192];
            if (p[l] != s0 || p[64 + l] != s1 || p[128 + l] != q) ++bad;
        }
        double a[3] = {0, 0, 0};
        for (int b = 0; b < G; ++b) for (int j = 0; j < 3; ++j) a[j] += (double)st[b * 4 + j] / G;
        printf("reps %d: mismatches %d; cycles/rep: stream %.0f, fwd reads %.0f, bwd reads %.0f\n", reps, bad, a[0] / reps, a[1] / reps, a[2] / reps);
    }
    return 0;
}
