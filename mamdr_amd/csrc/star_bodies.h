// Device bodies of Star kernels that can ride in another kernel's launch (see emb_bodies.h for the pattern).
#pragma once
#include "mamdr_kernels.h"

namespace mamdr {

constexpr int STAR_DM_LANES = 16;              // chunk lanes per column (fixed: it defines the summation order)

// column sums of dx[:, 256:384] (the domain-table row gradient) from the per-chunk partials: thread (column cl,
// lane j) sums chunks j, j + 16, ... (eight loads in flight), lane 0 then adds the 16 lanes in lane order.  COLS
// columns per workgroup (COLS x 16 threads): the split over workgroups does not touch the per-column order.
// sh = [16][COLS] floats of LDS.
template <int COLS>
__device__ __forceinline__ void star_dm_final_body(const StarPnBwdArgs& a, int bx, float* sh) {
    const int cl = threadIdx.x & (COLS - 1), j = threadIdx.x / COLS;
    const int k = bx * COLS + cl;
    float g = 0.f;
    for (int ch0 = j; ch0 < a.n_chunks; ch0 += 8 * STAR_DM_LANES) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = a.dmpart[(size_t)min(ch0 + u * STAR_DM_LANES, a.n_chunks - 1) * EMB + k];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ch0 + u * STAR_DM_LANES < a.n_chunks) g += t[u];
    }
    sh[j * COLS + cl] = g;
    __syncthreads();
    if (j != 0) return;
    for (int q = 1; q < STAR_DM_LANES; ++q) g += sh[q * COLS + cl];
    a.dmsum[k] = g;
}

}  // namespace mamdr
