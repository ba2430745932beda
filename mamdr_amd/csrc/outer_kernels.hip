// Outer (meta) parameter updates as HBM-bound elementwise kernels for gfx950.
//
// The reference does these in numpy on the host (model_zoo/domain_negotiation.py:118-123,
// reptile.py:127-142, mamdr.py:168-196, specific_base_model.py:164-172); every
// arithmetic step there is a separately rounded fp32 op.  This file is compiled
// with -ffp-contract=off and uses the __f*_rn intrinsics so that no multiply-add is
// fused: results match numpy bit-for-bit.  16 B per lane, grid-stride, <= 2048 blocks.
#include "mamdr_kernels.h"

namespace mamdr {

namespace {
constexpr int BLOCK = 256;
inline int grid_for(int64_t n4) {
    int64_t b = (n4 + BLOCK - 1) / BLOCK;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// Apply F elementwise over n floats: float4 body + scalar tail, pointers 16-B aligned
// (flat vectors are torch allocations) -- checked by the caller.
template <typename F>
__global__ __launch_bounds__(BLOCK) void k_elementwise(int64_t n, F f) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) f.vec(i);
    const int64_t tail0 = n4 << 2;
    for (int64_t i = tail0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) f.one(i);
}

struct Interp {   // dst += (a - b) * scale
    float* dst; const float* a; const float* b; float scale;
    __device__ __forceinline__ float op(float d, float x, float y) const {
        return __fadd_rn(d, __fmul_rn(__fsub_rn(x, y), scale));
    }
    __device__ __forceinline__ void vec(int64_t i) const {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
        const f32x4 y = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 d = reinterpret_cast<f32x4*>(dst)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = op(d[c], x[c], y[c]);
        reinterpret_cast<f32x4*>(dst)[i] = d;
    }
    __device__ __forceinline__ void one(int64_t i) const { dst[i] = op(dst[i], a[i], b[i]); }
};

template <int MODE>
struct Merge {    // dst = t + p | t * p
    float* dst; const float* t; const float* p;
    __device__ __forceinline__ float op(float x, float y) const {
        return MODE == 0 ? __fadd_rn(x, y) : __fmul_rn(x, y);
    }
    __device__ __forceinline__ void vec(int64_t i) const {
        const f32x4 x = reinterpret_cast<const f32x4*>(t)[i];
        const f32x4 y = reinterpret_cast<const f32x4*>(p)[i];
        f32x4 d;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = op(x[c], y[c]);
        reinterpret_cast<f32x4*>(dst)[i] = d;
    }
    __device__ __forceinline__ void one(int64_t i) const { dst[i] = op(t[i], p[i]); }
};

struct Sub {      // dst = a - b
    float* dst; const float* a; const float* b;
    __device__ __forceinline__ void vec(int64_t i) const {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
        const f32x4 y = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 d;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = __fsub_rn(x[c], y[c]);
        reinterpret_cast<f32x4*>(dst)[i] = d;
    }
    __device__ __forceinline__ void one(int64_t i) const { dst[i] = __fsub_rn(a[i], b[i]); }
};

// K.moving_average_update(ag, g, momentum) with TF 1.12's zero-debiased moving average (maml.py:219-220):
//   biased -= (biased - value) * decay;  unbiased -= unbiased - biased / denom      (denom = 1 - (1 - decay)^step, host)
struct MovingAverage {
    float* unbiased; float* biased; const float* value; float decay, denom;
    __device__ __forceinline__ void op(float& u, float& b, float g) const {
        b = __fsub_rn(b, __fmul_rn(__fsub_rn(b, g), decay));
        u = __fsub_rn(u, __fsub_rn(u, __fdiv_rn(b, denom)));
    }
    __device__ __forceinline__ void vec(int64_t i) const {
        f32x4 u = reinterpret_cast<f32x4*>(unbiased)[i];
        f32x4 b = reinterpret_cast<f32x4*>(biased)[i];
        const f32x4 g = reinterpret_cast<const f32x4*>(value)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float uc = u[c], bc = b[c];
            op(uc, bc, g[c]);
            u[c] = uc; b[c] = bc;
        }
        reinterpret_cast<f32x4*>(unbiased)[i] = u;
        reinterpret_cast<f32x4*>(biased)[i] = b;
    }
    __device__ __forceinline__ void one(int64_t i) const {
        float u = unbiased[i], b = biased[i];
        op(u, b, value[i]);
        unbiased[i] = u; biased[i] = b;
    }
};

// One DR support step in a single pass (mamdr.py:103-105 followed by the next support's :74 assignment):
//   phi += (w - merged) * gamma;  merged = theta (+|*) phi;  [w = merged]
// -- the same roundings, in the same order, as mamdr_interp + mamdr_merge + mamdr_copy one after the other.
template <int MODE, bool ASSIGN>
struct DrAdvance {
    float* phi; float* w; float* merged; const float* theta; float gamma;
    __device__ __forceinline__ void elem(float& ph, float& wv, float& mg, float th) const {
        ph = __fadd_rn(ph, __fmul_rn(__fsub_rn(wv, mg), gamma));
        mg = MODE == 0 ? __fadd_rn(th, ph) : __fmul_rn(th, ph);
        if (ASSIGN) wv = mg;
    }
    __device__ __forceinline__ void vec(int64_t i) const {
        f32x4 ph = reinterpret_cast<f32x4*>(phi)[i];
        f32x4 wv = reinterpret_cast<f32x4*>(w)[i];
        f32x4 mg = reinterpret_cast<f32x4*>(merged)[i];
        const f32x4 th = reinterpret_cast<const f32x4*>(theta)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = ph[c], b = wv[c], m = mg[c];
            elem(a, b, m, th[c]);
            ph[c] = a; wv[c] = b; mg[c] = m;
        }
        reinterpret_cast<f32x4*>(phi)[i] = ph;
        reinterpret_cast<f32x4*>(merged)[i] = mg;
        if (ASSIGN) reinterpret_cast<f32x4*>(w)[i] = wv;
    }
    __device__ __forceinline__ void one(int64_t i) const {
        float a = phi[i], b = w[i], m = merged[i];
        elem(a, b, m, theta[i]);
        phi[i] = a;
        merged[i] = m;
        if (ASSIGN) w[i] = b;
    }
};

template <bool SHARED>
struct Accumulate {   // acc += (a - b) [* shared] / divisor
    float* acc; const float* a; const float* b; const float* shared; float divisor;
    __device__ __forceinline__ float op(float d, float x, float y, float s) const {
        float g = __fsub_rn(x, y);
        if (SHARED) g = __fmul_rn(g, s);
        return __fadd_rn(d, __fdiv_rn(g, divisor));
    }
    __device__ __forceinline__ void vec(int64_t i) const {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
        const f32x4 y = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (SHARED) s = reinterpret_cast<const f32x4*>(shared)[i];
        f32x4 d = reinterpret_cast<f32x4*>(acc)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = op(d[c], x[c], y[c], s[c]);
        reinterpret_cast<f32x4*>(acc)[i] = d;
    }
    __device__ __forceinline__ void one(int64_t i) const {
        acc[i] = op(acc[i], a[i], b[i], SHARED ? shared[i] : 0.f);
    }
};

template <bool DIVIDE>
struct ApplyAcc {     // dst += acc [/ divisor] * scale; acc = 0
    float* dst; float* acc; float divisor; float scale;
    __device__ __forceinline__ float op(float d, float g) const {
        if (DIVIDE) g = __fdiv_rn(g, divisor);
        return __fadd_rn(d, __fmul_rn(g, scale));
    }
    __device__ __forceinline__ void vec(int64_t i) const {
        const f32x4 g = reinterpret_cast<f32x4*>(acc)[i];
        f32x4 d = reinterpret_cast<f32x4*>(dst)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = op(d[c], g[c]);
        reinterpret_cast<f32x4*>(dst)[i] = d;
        reinterpret_cast<f32x4*>(acc)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void one(int64_t i) const {
        dst[i] = op(dst[i], acc[i]);
        acc[i] = 0.f;
    }
};

struct AdamApply {    // TF1 ApplyAdam on a flat vector (outer optimiser of MAML)
    float* p; float* m; float* v; const float* g; float gscale, alpha, omb1, omb2, eps;
    __device__ __forceinline__ void step(float& pp, float& mm, float& vv, float gg) const {
        gg = gg * gscale;
        mm = mm + (gg - mm) * omb1;
        vv = vv + (gg * gg - vv) * omb2;
        pp = pp - (mm * alpha) / (sqrtf(vv) + eps);
    }
    __device__ __forceinline__ void vec(int64_t i) const {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = pp[c], b = mm[c], d = vv[c];
            step(a, b, d, gg[c]);
            pp[c] = a; mm[c] = b; vv[c] = d;
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    __device__ __forceinline__ void one(int64_t i) const { step(p[i], m[i], v[i], g[i]); }
};

template <typename F>
void run(int64_t n, const F& f, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_elementwise<F>, dim3(grid_for(n >> 2)), dim3(BLOCK), 0, s, n, f);
}
}  // namespace

void launch_interp(float* dst, const float* a, const float* b, float scale, int64_t n, hipStream_t s) {
    run(n, Interp{dst, a, b, scale}, s);
}
void launch_moving_average(float* unbiased, float* biased, const float* value, float decay, float denom, int64_t n,
                           hipStream_t s) {
    run(n, MovingAverage{unbiased, biased, value, decay, denom}, s);
}
void launch_dr_advance(float* phi, float* w, float* merged, const float* theta, float gamma, int mode, int assign, int64_t n,
                       hipStream_t s) {
    const dim3 grid(grid_for(n >> 2)), block(BLOCK);
    if (mode == 0 && assign) hipLaunchKernelGGL((k_elementwise<DrAdvance<0, true>>), grid, block, 0, s, n, (DrAdvance<0, true>{phi, w, merged, theta, gamma}));
    else if (mode == 0) hipLaunchKernelGGL((k_elementwise<DrAdvance<0, false>>), grid, block, 0, s, n, (DrAdvance<0, false>{phi, w, merged, theta, gamma}));
    else if (assign) hipLaunchKernelGGL((k_elementwise<DrAdvance<1, true>>), grid, block, 0, s, n, (DrAdvance<1, true>{phi, w, merged, theta, gamma}));
    else hipLaunchKernelGGL((k_elementwise<DrAdvance<1, false>>), grid, block, 0, s, n, (DrAdvance<1, false>{phi, w, merged, theta, gamma}));
}
// the same with a domain-table step of the k_wgrad_adam path still pending (DmStep): the lanes that own the table's
// elements materialise it first -- live p / m / v := dm_step4, as k_dm_finish would have -- and go on with the value
// they have just written (one launch instead of k_dm_finish + this one; the same bits)
template <int MODE, bool ASSIGN>
struct DrAdvanceDm {
    DrAdvance<MODE, ASSIGN> op;
    DmStep q;
    float* live_m; float* live_v;      // the domain table's Adam slots
    int64_t dm_off4; int dm_n4;        // its float4 range inside op.w
    __device__ __forceinline__ void vec(int64_t i) const {
        if (i >= dm_off4 && i < dm_off4 + dm_n4) {
            const int k = (int)(i - dm_off4);
            f32x4 p, m, v;
            dm_step4(q, k / (EMB / 4), k % (EMB / 4), p, m, v);
            reinterpret_cast<f32x4*>(op.w)[i] = p;
            reinterpret_cast<f32x4*>(live_m)[k] = m;
            reinterpret_cast<f32x4*>(live_v)[k] = v;
        }
        op.vec(i);
    }
    __device__ __forceinline__ void one(int64_t i) const { op.one(i); }
};
void launch_dr_advance_dm(float* phi, float* w, float* merged, const float* theta, float gamma, int mode, int assign, int64_t n,
                          const DmStep& q, float* live_m, float* live_v, int64_t dm_off4, int dm_n4, hipStream_t s) {
    const dim3 grid(grid_for(n >> 2)), block(BLOCK);
    if (mode == 0 && assign) hipLaunchKernelGGL((k_elementwise<DrAdvanceDm<0, true>>), grid, block, 0, s, n, (DrAdvanceDm<0, true>{{phi, w, merged, theta, gamma}, q, live_m, live_v, dm_off4, dm_n4}));
    else if (mode == 0) hipLaunchKernelGGL((k_elementwise<DrAdvanceDm<0, false>>), grid, block, 0, s, n, (DrAdvanceDm<0, false>{{phi, w, merged, theta, gamma}, q, live_m, live_v, dm_off4, dm_n4}));
    else if (assign) hipLaunchKernelGGL((k_elementwise<DrAdvanceDm<1, true>>), grid, block, 0, s, n, (DrAdvanceDm<1, true>{{phi, w, merged, theta, gamma}, q, live_m, live_v, dm_off4, dm_n4}));
    else hipLaunchKernelGGL((k_elementwise<DrAdvanceDm<1, false>>), grid, block, 0, s, n, (DrAdvanceDm<1, false>{{phi, w, merged, theta, gamma}, q, live_m, live_v, dm_off4, dm_n4}));
}
void launch_merge(float* dst, const float* t, const float* p, int mode, int64_t n, hipStream_t s) {
    if (mode == 0) run(n, Merge<0>{dst, t, p}, s);
    else run(n, Merge<1>{dst, t, p}, s);
}
void launch_sub(float* dst, const float* a, const float* b, int64_t n, hipStream_t s) {
    run(n, Sub{dst, a, b}, s);
}
void launch_accumulate(float* acc, const float* a, const float* b, const float* shared, float divisor, int64_t n,
                       hipStream_t s) {
    if (shared) run(n, Accumulate<true>{acc, a, b, shared, divisor}, s);
    else run(n, Accumulate<false>{acc, a, b, nullptr, divisor}, s);
}
void launch_apply_accumulated(float* dst, float* acc, float divisor, float scale, int64_t n, hipStream_t s) {
    if (divisor > 0.f) run(n, ApplyAcc<true>{dst, acc, divisor, scale}, s);
    else run(n, ApplyAcc<false>{dst, acc, 1.f, scale}, s);
}

void launch_adam_apply(float* p, float* m, float* v, const float* g, float gscale, float alpha, float omb1, float omb2,
                       float eps, int64_t n, hipStream_t s) {
    run(n, AdamApply{p, m, v, g, gscale, alpha, omb1, omb2, eps}, s);
}

// ------------------------------------------------------------------ PCGrad projection
// The reference does this in numpy on the host (model_zoo/pcgrad.py:152-160): per tensor and per slice along
// the last axis, d = sum(cur * aux); where d > 0: aux -= (d / ||cur||) * cur; cur += aux.  numpy reduces with
// its pairwise summation (8 running sums over blocks of <= 128 elements, halves above that); this kernel walks
// one slice per thread in exactly that order, so the result matches numpy bit for bit (tested against vectors
// produced by the reference's own method).  Slices are short (<= 256 elements) and few outside the tables.
namespace {
template <typename F>
__device__ float np_pairwise_sum(const F& f, int i0, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, f(i0 + i));
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = f(i0 + j);
        int i = 8;
        for (; i < n - (n % 8); i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], f(i0 + i + j));
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, f(i0 + i));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise_sum(f, i0, n2), np_pairwise_sum(f, i0 + n2, n - n2));
}
}  // namespace

__global__ __launch_bounds__(256) void k_pcgrad(const PcgArgs a) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.row_start[a.n_seg]) return;
    int sgm = 0;
    while (row >= a.row_start[sgm + 1]) ++sgm;
    const int n = a.cols[sgm];
    const int64_t base = a.off[sgm] + (row - a.row_start[sgm]) * n;
    float* cur = a.fin + base;
    float* aux = a.aux + base;
    const float dot = __fadd_rn(0.f, np_pairwise_sum([&](int i) { return __fmul_rn(cur[i], aux[i]); }, 0, n));
    if (dot > 0.f) {
        const float nrm2 = __fadd_rn(0.f, np_pairwise_sum([&](int i) { return __fmul_rn(cur[i], cur[i]); }, 0, n));
        // sqrtf and '/' are the correctly rounded IEEE operations here (HIP's __fsqrt_rn is the native
        // approximation unless OCML_BASIC_ROUNDED_OPERATIONS is defined)
        const float q = dot / sqrtf(nrm2);
        for (int i = 0; i < n; ++i) aux[i] = __fsub_rn(aux[i], __fmul_rn(q, cur[i]));
    }
    for (int i = 0; i < n; ++i) cur[i] = __fadd_rn(cur[i], aux[i]);
}
void launch_pcgrad(const PcgArgs& a, hipStream_t s) {
    const int64_t rows = a.row_start[a.n_seg];
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_pcgrad, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace mamdr
