// Device bodies of Star kernels that can ride in another kernel's launch (see emb_bodies.h for the pattern).
#pragma once
#include "mamdr_kernels.h"

namespace mamdr {

constexpr int STAR_DM_LANES = 16;              // chunk lanes per column (fixed: it defines the summation order)

// PartitionedNorm's backward, one element: d x = coef * t with t = (g - s1 / B) - xhat * (s2 / B).  The roundings are
// spelled out -- ONE fused multiply-add for t, and the column sum of d x taken as fma(coef, t, sum) -- because that is
// what the compiler's contraction has made of k_star_pnb_apply since round 2, and every site that forms d x
// (k_star_pnb_apply, the table workgroups of k_wgrad_reduce, star_pnb_dom_body) must round alike.
__device__ __forceinline__ float pn_bwd_t(float g, float m1, float xh, float m2) {
#pragma clang fp contract(off)
    return __builtin_fmaf(-m2, xh, g - m1);
}

// The domain columns (256..383) of PartitionedNorm's backward without a pass over the other 256: the per-chunk column
// sums of d x[:, 256:384] -- k_star_pnb_apply's arithmetic and order for those columns; d x itself has no other reader
// (the table rows take theirs inside k_emb_reduce, EmbStepArgs::pn_sums).  256 threads = two chunks x 128 columns.
__device__ __forceinline__ void star_pnb_dom_body(const StarPnBwdArgs& a, int bx) {
#pragma clang fp contract(off)
    const int kk = threadIdx.x & (EMB - 1), ch = bx * 2 + (int)(threadIdx.x >> 7);
    if (ch >= a.n_chunks) return;
    const int c = 2 * EMB + kk;
    const int r0 = ch * STAR_CHUNK;
    const int nb = min(STAR_CHUNK, a.rows - r0);
    const float m1 = a.means[c], m2 = a.means[XDIM + c];
    const float coef = a.pn[4 * XDIM + c];
    const float xh = (a.dm_row[kk] - a.pn[2 * XDIM + c]) * a.pn[3 * XDIM + c];
    float g[STAR_CHUNK];
#pragma unroll
    for (int r = 0; r < STAR_CHUNK; ++r) g[r] = a.dxe[(size_t)(r0 + min(r, nb - 1)) * XDIM + c];
    float colsum = 0.f;
#pragma unroll
    for (int r = 0; r < STAR_CHUNK; ++r)
        if (r < nb) colsum = __builtin_fmaf(coef, pn_bwd_t(g[r], m1, xh, m2), colsum);
    a.dmpart[(size_t)ch * EMB + kk] = colsum;
}

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
