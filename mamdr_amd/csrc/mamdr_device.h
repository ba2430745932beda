// Device-side helpers shared by the MAMDR gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mamdr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- fixed architecture of the hot path (model_zoo/DeepCTR/deepctr.py:95-136 with
//      the BASELINE configs: three 128-d embeddings -> 384 -> 256 -> 128 -> 64 -> 1)
constexpr int EMB = 128;
constexpr int XDIM = 3 * EMB;   // 384
constexpr int H1 = 256;
constexpr int H2 = 128;
constexpr int H3 = 64;
constexpr int ACT_LD = XDIM + H1 + H2 + H3;   // 832: [x | h1 | h2 | h3] per row
constexpr int DZ_LD = H1 + H2 + H3;           // 448: [dz1 | dz2 | dz3] per row
constexpr int TILE_ROWS = 16;                 // batch rows per workgroup in the step kernel
// transposed copies of W1 / W2 for the 4-row tower's backward layers (workspace, kept by k_update)
constexpr int W1T_OFF = 0;                    // [128][256]
constexpr int W2T_OFF = H1 * H2;              // [64][128]
constexpr int W0T_OFF = H1 * H2 + H2 * H3;    // [256][256]: W0[0:256, :]^T (user | item rows), trainable tables only
constexpr int WT_FLOATS = H1 * H2 + H2 * H3 + 2 * EMB * H1;

// dense block of the flat trainable vector, relative to the domain table start
// (DeepFM appends its 1-d linear table of the domain feature, `ld`, behind the global bias)
// (uncertainty weighting appends one trainable scalar per domain, `lv`, at the very end)
// (PNN appends the three rows of its first kernel that the field pairs' inner products feed, `wx` [3][H1], at the end:
// W0 itself keeps its [XDIM][H1] place, so every contraction of the mlp tower reads the same addresses)
struct DenseLayout {
    int dm, w0, w1, w2, b0, b1, b2, wo, gb, ld, ld_count, lv, lv_count, wx, wx_count, count, alloc;
    __host__ __device__ static DenseLayout make(int n_domain, bool deepfm = false, bool uncertainty = false, bool pnn = false) {
        DenseLayout L;
        L.dm = 0;
        L.w0 = n_domain * EMB;
        L.w1 = L.w0 + XDIM * H1;
        L.w2 = L.w1 + H1 * H2;
        L.b0 = L.w2 + H2 * H3;
        L.b1 = L.b0 + H1;
        L.b2 = L.b1 + H2;
        L.wo = L.b2 + H3;
        L.gb = L.wo + H3;
        L.ld = L.gb + 1;
        L.ld_count = deepfm ? n_domain : 0;
        L.lv = L.ld + L.ld_count;
        L.lv_count = uncertainty ? n_domain : 0;
        L.wx = (L.lv + L.lv_count + 3) & ~3;        // (16-B aligned rows)
        L.wx_count = pnn ? 3 * H1 : 0;
        L.count = pnn ? L.wx + L.wx_count : L.lv + L.lv_count;
        L.alloc = (L.count + 3) & ~3;
        return L;
    }
};

// Star tower (model_zoo/Star): block of the flat trainable vector behind the (optional) tables.
// The meta parameters of the reference's name filter ("emb", "kernel_shared", "bias_shared",
// config/Taobao-10/star_taobao.json:37-41) come first, so theta / phi are a prefix of the vector.
struct StarLayout {
    int dm, ws[3], bs[3], n_meta;
    int pgs, pbs, pgd, pbd;        // PartitionedNorm gamma / beta: shared [384], specific [D][384]
    int wd[3], bd[3];              // specific kernels [D][in*out] and biases [D][out]
    int wo, gb, count, alloc;
    __host__ __device__ static int ksize(int l) { return l == 0 ? XDIM * H1 : (l == 1 ? H1 * H2 : H2 * H3); }
    __host__ __device__ static int bsize(int l) { return l == 0 ? H1 : (l == 1 ? H2 : H3); }
    __host__ __device__ static StarLayout make(int n_domain) {
        StarLayout S;
        int o = 0;
        S.dm = o; o += n_domain * EMB;
        for (int l = 0; l < 3; ++l) { S.ws[l] = o; o += ksize(l); }
        for (int l = 0; l < 3; ++l) { S.bs[l] = o; o += bsize(l); }
        S.n_meta = o;
        S.pgs = o; o += XDIM;
        S.pbs = o; o += XDIM;
        S.pgd = o; o += n_domain * XDIM;
        S.pbd = o; o += n_domain * XDIM;
        for (int l = 0; l < 3; ++l) { S.wd[l] = o; o += n_domain * ksize(l); }
        for (int l = 0; l < 3; ++l) { S.bd[l] = o; o += n_domain * bsize(l); }
        S.wo = o; o += H3;
        S.gb = o; o += 1;
        S.count = o;
        S.alloc = (o + 3) & ~3;
        return S;
    }
};
// non-trainable PartitionedNorm state per domain (partitioned_norm.py:71-87 + the zero-debias slots
// of TF 1.12's assign_moving_average): [mov_mean | mov_var | biased_mean | biased_var] each [D][384], steps [D]
struct StarAuxLayout {
    int mov_mean, mov_var, biased_mean, biased_var, steps, count;
    __host__ __device__ static StarAuxLayout make(int n_domain) {
        StarAuxLayout A;
        A.mov_mean = 0;
        A.mov_var = n_domain * XDIM;
        A.biased_mean = 2 * n_domain * XDIM;
        A.biased_var = 3 * n_domain * XDIM;
        A.steps = 4 * n_domain * XDIM;
        A.count = (A.steps + n_domain + 3) & ~3;
        return A;
    }
};
constexpr float PN_EPS = 1e-3f;
constexpr float PN_MOMENTUM = 0.99f;
constexpr int STAR_CHUNK = 16;        // batch rows per partial of the column statistics
constexpr int PN_XDOM_OFF = 5 * XDIM;    // the batch's normalised domain-embedding row (what every sample's x[256:384] is)
constexpr int PN_WS_FLOATS = 5 * XDIM + EMB;   // scale | shift | mean | inv | coef = gamma_eff * inv | xdom

// ---- counter-based dropout stream (restated in oracle/rng.py)
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline uint32_t dropout_layer_key(uint32_t seed, uint32_t step, uint32_t layer) {
    uint32_t k0 = fmix32(seed + 0x9E3779B9u * (step + 1u));
    return fmix32(k0 ^ (0x85EBCA6Bu * (layer + 1u)));
}
__device__ __forceinline__ uint32_t mamdr_dropout_u32(uint32_t key, uint32_t elem) {
    return fmix32(key + 0x9E3779B9u * elem);
}

#ifdef MAMDR_ABLATE_MFMA   // diagnostic builds only: keep the operands live, skip the matrix op
#define MAMDR_MFMA16(a, b, c) ((c) + (f32x4){(a), (b), (a), (b)})
#else
#define MAMDR_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif
#define MAMDR_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

}  // namespace mamdr
