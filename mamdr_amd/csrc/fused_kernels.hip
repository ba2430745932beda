// k_wgrad_adam: the weight gradients AND the optimiser step of the dense block in ONE launch (mlp tower,
// frozen user / item tables), output-stationary: every workgroup owns a 16 x 32 tile of one weight matrix,
// contracts over ALL batch rows of the step, and steps its own 512 parameters -- no split-K slabs in HBM, no
// second kernel to reduce them (k_wgrad + k_update: 7.9 + 5.5 us per step at 1,024 rows).
//
//   tile workgroup   8 waves split the batch rows (the reduction index); a wave keeps its whole share of both
//                    operands in flight (a 32-deep register ring: 4 rows per slot, 4-byte loads of the 16
//                    activation columns, 8-byte loads of the 32 gradient columns) and feeds two
//                    v_mfma_f32_16x16x4_f32 per slot; the 8 partial tiles are summed through LDS in wave order
//                    (fixed order: bitwise reproducible); thread (m, n) then applies TF1 Adam / SGD / accumulate
//                    to element (m, n) and keeps k_tower4's transposed W1 / W2 copies current.
//   S workgroups     8 workgroups, one per 32 columns of dz1: S = onehot(domain)^T dz1 for their columns (the
//                    same contraction with a synthesised A operand), from which follow, by linearity (the rows
//                    256..383 of x are the domain-embedding row of the sample's domain):
//                      db0[c]        = sum_d S[d][c]
//                      dW0[256+r][c] = sum_d Dm[d][r] S[d][c]          (stepped here)
//                      pdm[blk][d][r] = sum_{c in blk} S[d][c] W0[256+r][c]   (partial domain-table gradient)
//                    against PRE-update snapshots of Dm and W0[256:384] made by the tower kernel of the step.
//   domain table     g = sum_blk pdm[blk] + 2 l2 Dm: applied by the NEXT step's tower kernel (DmStep, mamdr_kernels.h);
//                    k_dm_finish materialises the last step of a call.
//
// Replaces the weight-gradient and ApplyAdam ops behind `model.train_on_batch`
// (model_zoo/DeepCTR/deepctr.py:54-60; call sites model_zoo/mamdr.py:54,86,97).
#include <hip/hip_ext.h>

#include "mamdr_kernels.h"

namespace mamdr {

constexpr int FZ_THREADS = 512;
constexpr int FZ_WAVES = 8;
#ifndef MAMDR_FZ_RING
#define MAMDR_FZ_RING 32
#endif
constexpr int FZ_RING = MAMDR_FZ_RING;       // slots (of 4 batch rows) in flight per wave (diagnostic builds may override)
constexpr int FZ_SBLK = DM_PARTS;                   // S workgroups = 8-column blocks of dz1 (the optimiser step of their
                                              // 128 x 8 block of W0[256:384] is the long part: 2 elements per thread)
constexpr int FZ_SC = 8;                      // columns per S workgroup
constexpr int FZ_OUTB = 2;                    // output-unit workgroups = 32-column blocks of h3
constexpr int FZ_T0 = (2 * EMB / 16) * (H1 / 32);     // 128 tiles of dW0[0:256, :]
constexpr int FZ_T1 = (H1 / 16) * (H2 / 32);          // 64 tiles of dW1
constexpr int FZ_T2 = (H2 / 16) * (H3 / 32);          // 16 tiles of dW2
constexpr int FZ_TILES = FZ_T0 + FZ_T1 + FZ_T2;       // 208

#ifdef MAMDR_STAMPS   // diagnostic build only (tools/stamp_fused.py)
#define FZSTAMP(k)                                                                            \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (k)] = t_;                \
    } while (0)
#define FZREAL(k)                                                                             \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (k)] = t_;                \
    } while (0)
#else
#define FZSTAMP(k) do { } while (0)
#define FZREAL(k) do { } while (0)
#endif

__device__ __forceinline__ void fz_opt(const FusedArgs& a, float g, float& p, float& m, float& v) {
    if (a.optimizer == 0) {
        m = m + (g - m) * a.omb1;
        v = v + (g * g - v) * a.omb2;
        p = p - (m * a.alpha) / (sqrtf(v) + a.eps);
    } else if (a.optimizer == 1) {
        p = p - g * a.alpha;
    } else {
        m = m + g;
    }
}

#ifdef FZ_SC1_STORES          // diagnostic builds: write-through parameter stores
#define FZ_ST(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define FZ_ST(ptr, val) (*(ptr) = (val))
#endif
__device__ __forceinline__ void fz_store(const FusedArgs& a, int e, float p, float m, float v) {
    if (a.optimizer == 2) {
        FZ_ST(&a.m[e], m);
        return;
    }
    if (a.optimizer == 0) {
        FZ_ST(&a.m[e], m);
        FZ_ST(&a.v[e], v);
    }
    FZ_ST(&a.p[e], p);
}

// this wave's share of the 4-row slots [0, n_slots): contiguous, in wave order
__device__ __forceinline__ void fz_share(int n_slots, int w, int& s0, int& s1) {
    const int per = (n_slots + FZ_WAVES - 1) / FZ_WAVES;
    s0 = min(w * per, n_slots);
    s1 = min(s0 + per, n_slots);
}

// ---- dense tile: acc[t] += A[rows, 16 cols]^T . B[rows, cols 2 j + t], rows = 4-row slots s0..s1 of the batch.
// MFMA operands: a lane (i = lane & 15, k = lane >> 4) = A[4 s + k][a0 + i]; b lane (j, k) = B[4 s + k][b0 + 2 j + t];
// D register r of lane l = (row 4 (l >> 4) + r, col l & 15) of the 16 x 16 product.
// Straight-line passes over the ring (no branch between a slot's MFMAs and its reload: the hardware retires
// loads in order, and the compiler only counts them -- vmcnt(2 RING - 2) instead of vmcnt(0) -- inside a basic
// block): every pass but the last reloads each slot for the pass after it (the address clamped to the last
// slot, which only the pass before the last can reach), the last pass masks the slots beyond the share.
// Addresses are (wave-uniform row pointer) + (per-lane 32-bit offset): the row pointer advances on the scalar unit
// and the loads take the saddr form -- no vector arithmetic per load (the 64-bit multiply-adds of a per-lane
// pointer cost more issue cycles per slot than its two MFMAs).
template <bool CSUM>
__device__ __forceinline__ void fz_dense(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                         int s0, int s1, f32x4 (&acc)[2], f32x2& csum) {
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const int n = s1 - s0;                       // wave-uniform (scalar)
    if (n <= 0) return;
    const unsigned aoff = (unsigned)(kq * lda + j), boff = (unsigned)(kq * ldb + 2 * j);
    const float* arow = A + (size_t)(4 * s0) * lda;          // uniform
    const float* brow = B + (size_t)(4 * s0) * ldb;
    const size_t astep = (size_t)4 * lda, bstep = (size_t)4 * ldb;
    const int last = n - 1;
    float ra[FZ_RING];
    f32x2 rb[FZ_RING];
#pragma unroll
    for (int u = 0; u < FZ_RING; ++u) {
#ifdef FZ_ABLATE_LOADS       // diagnostic builds only
        ra[u] = (float)(u + lane);
        rb[u] = (f32x2){(float)u, (float)lane};
#else
        const int idx = min(u, last);
        ra[u] = (arow + idx * astep)[aoff];
        rb[u] = *reinterpret_cast<const f32x2*>(brow + idx * bstep + boff);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    const int passes = (n + FZ_RING - 1) / FZ_RING;
    int base = 0;
    for (int pass = 0; pass + 1 < passes; ++pass, base += FZ_RING) {
#pragma unroll
        for (int u = 0; u < FZ_RING; ++u) {
            acc[0] = MAMDR_MFMA16(ra[u], rb[u][0], acc[0]);
            acc[1] = MAMDR_MFMA16(ra[u], rb[u][1], acc[1]);
            if (CSUM) csum += rb[u];
#ifndef FZ_ABLATE_LOADS
            const int idx = min(base + u + FZ_RING, last);
            ra[u] = (arow + idx * astep)[aoff];
            rb[u] = *reinterpret_cast<const f32x2*>(brow + idx * bstep + boff);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < FZ_RING; ++u) {
        const float keep = (base + u < n) ? 1.0f : 0.0f;      // scalar condition
        const f32x2 b2 = rb[u] * keep;
        acc[0] = MAMDR_MFMA16(ra[u], b2[0], acc[0]);
        acc[1] = MAMDR_MFMA16(ra[u], b2[1], acc[1]);
        if (CSUM) csum += b2;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- one-hot tile: acc[mt] += onehot(domain)[rows, 16 mt .. 16 mt + 15]^T . dz1[rows, c0 + j]   (8 columns)
template <int MT>
__device__ __forceinline__ void fz_onehot(const int32_t* __restrict__ domrow, const float* __restrict__ B, int ldb,
                                          int s0, int s1, f32x4 (&acc)[4]) {
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const int n = s1 - s0;
    if (n <= 0) return;
    const unsigned boff = (unsigned)(kq * ldb + (j & (FZ_SC - 1)));     // (columns j >= 8 repeat the block: never stored)
    const int32_t* drow = domrow + 4 * s0;                    // uniform
    const float* brow = B + (size_t)(4 * s0) * ldb;
    const size_t bstep = (size_t)4 * ldb;
    const int last = n - 1;
    int rd[FZ_RING];
    float rb[FZ_RING];
#pragma unroll
    for (int u = 0; u < FZ_RING; ++u) {
        const int idx = min(u, last);
        rd[u] = (drow + idx * 4)[(unsigned)kq];
        rb[u] = (brow + idx * bstep)[boff];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int passes = (n + FZ_RING - 1) / FZ_RING;
    int base = 0;
    for (int pass = 0; pass + 1 < passes; ++pass, base += FZ_RING) {
#pragma unroll
        for (int u = 0; u < FZ_RING; ++u) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float av = (rd[u] == 16 * mt + j) ? 1.0f : 0.0f;
                acc[mt] = MAMDR_MFMA16(av, rb[u], acc[mt]);
            }
            const int idx = min(base + u + FZ_RING, last);
            rd[u] = (drow + idx * 4)[(unsigned)kq];
            rb[u] = (brow + idx * bstep)[boff];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < FZ_RING; ++u) {
        const int dsel = (base + u < n) ? rd[u] : -1;          // beyond the share: matches no domain
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float av = (dsel == 16 * mt + j) ? 1.0f : 0.0f;
            acc[mt] = MAMDR_MFMA16(av, rb[u], acc[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// LDS: red[w][16][32] partial tiles (+ [w][32] column sums behind them)
constexpr int FZ_RED = FZ_WAVES * 16 * 32;            // 4096 floats
constexpr int FZ_CS = FZ_RED;                         // [8][32]
constexpr int FZ_LDS_TILE = FZ_RED + FZ_WAVES * 32;   // 4352 floats
// S workgroup: red[w][MT*16][8] partials, then s_blk[16 MT][8]
static int fz_lds_floats(int n_domain) {
    const int mt = (n_domain + 15) / 16;
    const int s = FZ_WAVES * 16 * mt * FZ_SC + 16 * mt * FZ_SC;
    return s > FZ_LDS_TILE ? s : FZ_LDS_TILE;
}

__device__ __forceinline__ void fz_tile_body(const FusedArgs& a, int t, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform: shares and trip counts stay scalar
    // tile -> operands / destination.  Workgroup b runs on XCD b % 8 (round-robin dispatch; a speed assumption
    // only), and the 8 L2s do not share data: the tiles are dealt so that XCD x = (xb, xa) only reads a quarter of
    // the activation columns (xa) and a half of the gradient columns (xb) of every matrix -- 12.7 MB leave the
    // infinity cache per launch at 1,024 rows instead of 24 MB with one gradient block per XCD.
    const int x = t & 7, xa = x & 3, xb = x >> 2, li = t >> 3;      // li: 0..25 inside the XCD
    const float* A;
    int lda, b_off, dst, ldn, ablk, gemm;
    if (li < 16) {
        gemm = 0;
        ablk = 4 * xa + (li & 3);
        const int bblk = 4 * xb + (li >> 2);
        A = a.xa + 16 * ablk;
        lda = a.xa_ld;
        b_off = 32 * bblk;
        ldn = H1;
        dst = a.L.w0 + 16 * ablk * H1 + 32 * bblk;
    } else if (li < 24) {
        gemm = 1;
        const int u = li - 16;
        ablk = 4 * xa + (u & 3);
        const int bblk = 2 * xb + (u >> 2);
        A = a.acts + XDIM + 16 * ablk;
        lda = ACT_LD;
        b_off = H1 + 32 * bblk;
        ldn = H2;
        dst = a.L.w1 + 16 * ablk * H2 + 32 * bblk;
    } else {
        gemm = 2;
        ablk = 2 * xa + (li - 24);
        A = a.acts + XDIM + H1 + 16 * ablk;
        lda = ACT_LD;
        b_off = H1 + H2 + 32 * xb;
        ldn = H3;
        dst = a.L.w2 + 16 * ablk * H3 + 32 * xb;
    }
    // this thread's parameter and slots: requested before the contraction
    const int em = tid >> 5, en = tid & 31;
    const int e = dst + em * ldn + en;
    const float p0 = a.p[e];
    const float m0 = a.optimizer == 1 ? 0.f : a.m[e];
    const float v0 = a.optimizer == 0 ? a.v[e] : 0.f;
    // the first row block of dW1 / dW2 also owns the bias of its 32 columns (column sums of dz)
    const bool bias_tile = gemm > 0 && ablk == 0;
    int be = 0;
    float bp0 = 0.f, bm0 = 0.f, bv0 = 0.f;
    if (bias_tile) {                                // uniform branch; every lane loads (see fz_s_body)
        be = (gemm == 1 ? a.L.b1 + (b_off - H1) : a.L.b2 + (b_off - H1 - H2)) + (tid & 31);
        bp0 = a.p[be];
        bm0 = a.optimizer == 1 ? 0.f : a.m[be];
        bv0 = a.optimizer == 0 ? a.v[be] : 0.f;
    }
    int s0, s1;
    fz_share(a.rows_pad / 4, w, s0, s1);
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x2 csum = (f32x2){0.f, 0.f};
    FZSTAMP(1);
    if (bias_tile) fz_dense<true>(A, lda, a.dz + b_off, DZ_LD, s0, s1, acc, csum);
    else fz_dense<false>(A, lda, a.dz + b_off, DZ_LD, s0, s1, acc, csum);
    // partial tile of this wave -> LDS [w][m][n], n = 2 j + t
#pragma unroll
    for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x2*>(lds + (w * 16 + 4 * kq + r) * 32 + 2 * j) = (f32x2){acc[0][r], acc[1][r]};
    if (bias_tile) {
        // column sums: the four k lanes of a column pair, in lane order
        csum[0] += __shfl_xor(csum[0], 16);
        csum[1] += __shfl_xor(csum[1], 16);
        csum[0] += __shfl_xor(csum[0], 32);
        csum[1] += __shfl_xor(csum[1], 32);
        if (lane < 16) *reinterpret_cast<f32x2*>(lds + FZ_CS + w * 32 + 2 * j) = csum;
    }
    FZSTAMP(2);
    __syncthreads();
    FZSTAMP(3);
    float g = lds[em * 32 + en];
#pragma unroll
    for (int ww = 1; ww < FZ_WAVES; ++ww) g += lds[(ww * 16 + em) * 32 + en];
    float p = p0, m = m0, v = v0;
    fz_opt(a, g, p, m, v);
    fz_store(a, e, p, m, v);
    if (a.wT && a.optimizer != 2) {          // k_tower4's transposed copies of W1 / W2
        if (gemm == 2) {
            const int row = (e - a.L.w2) / H3, col = (e - a.L.w2) - row * H3;
            FZ_ST(&a.wT[W2T_OFF + col * H2 + row], p);
        } else if (gemm == 1) {
            const int row = (e - a.L.w1) / H2, col = (e - a.L.w1) - row * H2;
            FZ_ST(&a.wT[W1T_OFF + col * H1 + row], p);
        }
    }
    if (bias_tile && tid < 32) {
        float gb_ = lds[FZ_CS + tid];
#pragma unroll
        for (int ww = 1; ww < FZ_WAVES; ++ww) gb_ += lds[FZ_CS + ww * 32 + tid];
        float bp = bp0, bm = bm0, bv = bv0;
        fz_opt(a, gb_, bp, bm, bv);
        fz_store(a, be, bp, bm, bv);
    }
    FZSTAMP(4);
}

template <int MT>
__device__ __forceinline__ void fz_s_contract(const FusedArgs& a, int c0, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int s0, s1;
    fz_share(a.rows_pad / 4, w, s0, s1);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    fz_onehot<MT>(a.domrow, a.dz + c0, DZ_LD, s0, s1, acc);
    if (j < FZ_SC) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[(w * (16 * MT) + 16 * mt + 4 * kq + r) * FZ_SC + j] = acc[mt][r];
    }
}

__device__ __forceinline__ void fz_s_body(const FusedArgs& a, int blk, float* lds) {
    const int tid = threadIdx.x;
    const int D = a.n_domain, c0 = FZ_SC * blk;
    const int MT = (D + 15) / 16;
    // W0[256 + r][c0 + 2 q], [.. + 1]: this thread's 2 parameters and slots, requested first
    const int r = tid >> 2, q = tid & 3;
    const int e0 = a.L.w0 + (2 * EMB + r) * H1 + c0 + 2 * q;
    const f32x2 p0 = *reinterpret_cast<const f32x2*>(a.p + e0);
    const f32x2 m0 = a.optimizer == 1 ? (f32x2){0.f, 0.f} : *reinterpret_cast<const f32x2*>(a.m + e0);
    const f32x2 v0 = a.optimizer == 0 ? *reinterpret_cast<const f32x2*>(a.v + e0) : (f32x2){0.f, 0.f};
    // pre-update W0[256 + c'][c0 .. c0 + 7] for the partial domain-table gradient: thread (c' = tid & 127, dq = tid >> 7)
    const int cp = tid & 127, dq = tid >> 7;
    f32x4 wsn[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) wsn[k] = *reinterpret_cast<const f32x4*>(a.w0dom_snap + (size_t)cp * H1 + c0 + 4 * k);
    // Dm[d][r] (pre-update) of every (padded) domain: requested before the contraction too
    float xd[64];
#pragma unroll
    for (int c16 = 0; c16 < 4; ++c16) {
        if (c16 < MT) {                                     // uniform: one batch of 16 loads per 16 domains
#pragma unroll
            for (int d = 0; d < 16; ++d) xd[16 * c16 + d] = a.dm_snap[min(16 * c16 + d, D - 1) * EMB + r];
        } else {
#pragma unroll
            for (int d = 0; d < 16; ++d) xd[16 * c16 + d] = 0.f;
        }
    }
    // b0[c0 + tid]  (every lane loads: a divergent branch around loads makes the compiler drain ALL loads at its end)
    const int be0 = a.L.b0 + c0 + (tid & (FZ_SC - 1));
    const float bp0 = a.p[be0];
    const float bm0 = a.optimizer == 1 ? 0.f : a.m[be0];
    const float bv0 = a.optimizer == 0 ? a.v[be0] : 0.f;
    FZSTAMP(1);
    if (MT == 1) fz_s_contract<1>(a, c0, lds);
    else if (MT == 2) fz_s_contract<2>(a, c0, lds);
    else if (MT == 3) fz_s_contract<3>(a, c0, lds);
    else fz_s_contract<4>(a, c0, lds);
    FZSTAMP(2);
    __syncthreads();
    FZSTAMP(3);
    // S block [16 MT][8]: sum of the 8 wave partials in wave order
    float* sb = lds + FZ_WAVES * 16 * MT * FZ_SC;
    for (int idx = tid; idx < 16 * MT * FZ_SC; idx += FZ_THREADS) {
        float s = lds[idx];
#pragma unroll
        for (int ww = 1; ww < FZ_WAVES; ++ww) s += lds[ww * (16 * MT) * FZ_SC + idx];
        sb[idx] = s;
    }
    __syncthreads();
    // (a) dW0[256 + r][c0 + 2 q + k] = sum_d Dm[d][r] S[d][2 q + k]   (Dm: pre-update snapshot)
    {
        // (rows D .. 16 MT - 1 of the S block are exactly zero -- no sample carries such a domain -- so the loop
        // runs over the padded block without a branch; the clamped Dm values they meet are finite)
        f32x2 g = (f32x2){0.f, 0.f};
#pragma unroll
        for (int c16 = 0; c16 < 4; ++c16) {
            if (c16 < MT) {                                 // uniform
                f32x2 sa[16];
#pragma unroll
                for (int d = 0; d < 16; ++d) sa[d] = *reinterpret_cast<const f32x2*>(sb + (16 * c16 + d) * FZ_SC + 2 * q);
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    g[0] = fmaf(xd[16 * c16 + d], sa[d][0], g[0]);
                    g[1] = fmaf(xd[16 * c16 + d], sa[d][1], g[1]);
                }
            }
        }
        f32x2 p = p0, m = m0, v = v0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float pc = p[k], mc = m[k], vc = v[k];
            fz_opt(a, g[k], pc, mc, vc);
            p[k] = pc;
            m[k] = mc;
            v[k] = vc;
        }
        if (a.optimizer == 2) {
            *reinterpret_cast<f32x2*>(a.m + e0) = m;
        } else {
            if (a.optimizer == 0) {
                *reinterpret_cast<f32x2*>(a.m + e0) = m;
                *reinterpret_cast<f32x2*>(a.v + e0) = v;
            }
            *reinterpret_cast<f32x2*>(a.p + e0) = p;
        }
    }
    // (b) pdm[blk][d][c'] = sum_k S[d][k] W0[256 + c'][c0 + k]   (W0: pre-update snapshot)
    for (int d = dq; d < 16 * MT; d += 4) {
        const f32x4 s0v = *reinterpret_cast<const f32x4*>(sb + d * FZ_SC);
        const f32x4 s1v = *reinterpret_cast<const f32x4*>(sb + d * FZ_SC + 4);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fmaf(s0v[k], wsn[0][k], s);
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fmaf(s1v[k], wsn[1][k], s);
        if (d < D) a.pdm[((size_t)blk * D + d) * EMB + cp] = s;
    }
    // (c) db0[c0 + tid] = sum_d S[d][tid]   (all lanes compute, 8 store)
    {
        float g = 0.f;
        for (int d0 = 0; d0 < 16 * MT; d0 += 16) {         // 16 independent LDS reads, then the sum in row order
            float t[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) t[d] = sb[(d0 + d) * FZ_SC + (tid & (FZ_SC - 1))];
#pragma unroll
            for (int d = 0; d < 16; ++d) g += t[d];
        }
        float p = bp0, m = bm0, v = bv0;
        fz_opt(a, g, p, m, v);
        if (tid < FZ_SC) fz_store(a, be0, p, m, v);
    }
    FZSTAMP(4);
}

// output unit: dwo[c] = sum_b h3[b][c] dlogit[b], dgb = sum_b dlogit[b].  The same ring on the matrix unit, two
// workgroups of 32 columns of h3: one 8-byte load per lane (lane j: columns 2 j, 2 j + 1; component i feeds MFMA i,
// whose output row m is column 2 m + i), B carries dlogit in column 0 only.
__device__ __forceinline__ void fz_out_body(const FusedArgs& a, int ob, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int s0, s1;
    fz_share(a.rows_pad / 4, w, s0, s1);
    const int n = s1 - s0;
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    float sg = 0.f;
    FZSTAMP(1);
    if (n > 0) {
        const unsigned aoff = (unsigned)(kq * ACT_LD + 2 * j);
        const float* arow = a.acts + (size_t)(4 * s0) * ACT_LD + XDIM + H1 + H2 + 32 * ob;      // uniform
        const float* drow = a.dlogit + 4 * s0;
        const int last = n - 1;
        f32x2 ra[FZ_RING];
        float rd[FZ_RING];
#pragma unroll
        for (int u = 0; u < FZ_RING; ++u) {
            const int idx = min(u, last);
            ra[u] = *reinterpret_cast<const f32x2*>(arow + (size_t)idx * 4 * ACT_LD + aoff);
            rd[u] = (drow + idx * 4)[(unsigned)kq];
        }
        __builtin_amdgcn_sched_barrier(0);
        const int passes = (n + FZ_RING - 1) / FZ_RING;
        int base = 0;
        for (int pass = 0; pass + 1 < passes; ++pass, base += FZ_RING) {
#pragma unroll
            for (int u = 0; u < FZ_RING; ++u) {
                const float bv = j == 0 ? rd[u] : 0.f;
                acc[0] = MAMDR_MFMA16(ra[u][0], bv, acc[0]);
                acc[1] = MAMDR_MFMA16(ra[u][1], bv, acc[1]);
                sg += bv;
                const int idx = min(base + u + FZ_RING, last);
                ra[u] = *reinterpret_cast<const f32x2*>(arow + (size_t)idx * 4 * ACT_LD + aoff);
                rd[u] = (drow + idx * 4)[(unsigned)kq];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < FZ_RING; ++u) {
            const float bv = (j == 0 && base + u < n) ? rd[u] : 0.f;
            acc[0] = MAMDR_MFMA16(ra[u][0], bv, acc[0]);
            acc[1] = MAMDR_MFMA16(ra[u][1], bv, acc[1]);
            sg += bv;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // output column 0 lives in the lanes with j == 0: D register r of MFMA i = h3 column 32 ob + 2 (4 kq + r) + i
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x2*>(lds + w * 32 + 2 * (4 * kq + r)) = (f32x2){acc[0][r], acc[1][r]};
    }
    sg += __shfl_xor(sg, 16);          // the four k lanes of column 0 (the other lanes hold 0)
    sg += __shfl_xor(sg, 32);
    if (lane == 0) lds[256 + w] = sg;
    FZSTAMP(2);
    __syncthreads();
    FZSTAMP(3);
    if (tid < 32) {
        float g = lds[tid];
#pragma unroll
        for (int ww = 1; ww < FZ_WAVES; ++ww) g += lds[ww * 32 + tid];
        const int e = a.L.wo + 32 * ob + tid;
        float p = a.p[e], m = a.optimizer == 1 ? 0.f : a.m[e], v = a.optimizer == 0 ? a.v[e] : 0.f;
        fz_opt(a, g, p, m, v);
        fz_store(a, e, p, m, v);
    } else if (tid == 64 && ob == 0) {
        float g = lds[256];
#pragma unroll
        for (int ww = 1; ww < FZ_WAVES; ++ww) g += lds[256 + ww];
        const int e = a.L.gb;
        float p = a.p[e], m = a.optimizer == 1 ? 0.f : a.m[e], v = a.optimizer == 0 ? a.v[e] : 0.f;
        fz_opt(a, g, p, m, v);
        fz_store(a, e, p, m, v);
    }
    FZSTAMP(4);
}

// loss of the step = mean BCE + l2 (sum of squares of the three embedding tables), the oracle's reg_loss order
__device__ __forceinline__ void fz_loss_body(const FusedArgs& a, float* lds) {
    const int tid = threadIdx.x;
    float ss = 0.f, ls = 0.f;
    for (int e = tid; e < a.n_domain * EMB; e += FZ_THREADS) ss = fmaf(a.p[e], a.p[e], ss);
    for (int e = tid; e < a.n_loss_tiles; e += FZ_THREADS) ls += a.loss_part[e];
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o);
        ls += __shfl_xor(ls, o);
    }
    if ((tid & 63) == 0) {
        lds[tid >> 6] = ss;
        lds[8 + (tid >> 6)] = ls;
    }
    __syncthreads();
    if (tid == 0) {
        float sst = 0.f, lst = 0.f;
        for (int ww = 0; ww < FZ_WAVES; ++ww) {
            sst += lds[ww];
            lst += lds[8 + ww];
        }
        const float reg = a.l2_emb * a.frozen_sumsq[0] + a.l2_emb * a.frozen_sumsq[1] + a.l2_emb * sst;
        a.loss_out[0] = lst / (float)a.rows + reg;
    }
}

// rider block: its 8 waves touch the rows of 8 four-row tiles of the NEXT tower launch, all of them tiles whose workgroup
// will run on THIS block's XCD (tile t -> XCD t mod 8).  One word per 128-B line of a tile's 4 KB, the tile's first domain /
// label words; nothing is written (the sink keeps the loads alive).
__device__ __forceinline__ void fz_prefetch_body(const FusedArgs& a, const int b, const int first_rider) {
    if (b < first_rider) return;            // (the blocks that pad the grid to a multiple of 8 in front of the riders)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int xcd = b & 7, round = (b - first_rider) >> 3;
    const int tile = xcd + 8 * (FZ_WAVES * round + w);
    if (tile >= a.pf_tiles) return;
    const float* base = a.pf_x + (size_t)tile * 4 * (2 * EMB);
    float t = 0.f;
    if (lane < 32) t = base[lane * 32];
    else if (lane == 32) t = (float)a.pf_dom[4 * tile];
    else if (lane == 33) t = a.pf_lab[4 * tile];
    if (t == 1.2345678e30f) a.pf_sink[0] = t;
}

// grid: [0, 32) S workgroups (the longest chains first), [32, 240) tiles, 240 / 241 output unit, 242 loss (optional),
// then the prefetch riders
__global__ __launch_bounds__(FZ_THREADS) void k_wgrad_adam(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = (int)blockIdx.x;
    FZSTAMP(0);
    FZREAL(5);
    const int n_main = FZ_SBLK + FZ_TILES + FZ_OUTB + (a.loss_out ? 1 : 0);
    if (b < FZ_SBLK) fz_s_body(a, b, lds);
    else if (b < FZ_SBLK + FZ_TILES) fz_tile_body(a, b - FZ_SBLK, lds);
    else if (b < FZ_SBLK + FZ_TILES + FZ_OUTB) fz_out_body(a, b - FZ_SBLK - FZ_TILES, lds);
    else if (b < n_main) fz_loss_body(a, lds);
    else fz_prefetch_body(a, b, (n_main + 7) & ~7);
    FZREAL(6);
}

void launch_wgrad_adam(const FusedArgs& a, hipStream_t s) {
    int grid = FZ_SBLK + FZ_TILES + FZ_OUTB + (a.loss_out ? 1 : 0);
    if (a.pf_tiles > 0) {
        // riders start at the next multiple of 8 (block b runs on XCD b mod 8: `round` counts whole rounds over the XCDs)
        const int first = (grid + 7) & ~7;
        const int per_xcd = (a.pf_tiles + 7) / 8;                      // tiles per XCD
        const int rounds = (per_xcd + FZ_WAVES - 1) / FZ_WAVES;        // 8 tiles (one per wave) per block and XCD
        grid = first + 8 * rounds;
    }
    const size_t lds = (size_t)fz_lds_floats(a.n_domain) * sizeof(float);
    if (lds > 65536) {          // 49..64 domains: 72 KB of partial S tiles (raised once: thread-safe static initialiser, to the
                                // size of the largest domain count this path takes)
        static const bool big_lds_set = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_adam),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize,
                                                            (int)(fz_lds_floats(64) * sizeof(float))) == hipSuccess;
        (void)big_lds_set;
    }
    MAMDR_LAUNCH(k_wgrad_adam, dim3(grid), dim3(FZ_THREADS), lds, s, a);
}

// ---- domain table: the pending step of the last step of a call (DmStep, mamdr_kernels.h), one float4 per thread
__global__ __launch_bounds__(256) void k_dm_finish(const DmStep q, float* live_p, float* live_m, float* live_v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= q.n_domain * (EMB / 4)) return;
    const int d = i / (EMB / 4), c4 = i - d * (EMB / 4);
    f32x4 p, m, v;
    dm_step4(q, d, c4, p, m, v);
    const size_t row = (size_t)d * EMB + 4 * c4;
    if (q.optimizer != 2) *reinterpret_cast<f32x4*>(live_p + row) = p;
    if (q.optimizer != 1) *reinterpret_cast<f32x4*>(live_m + row) = m;
    if (q.optimizer == 0) *reinterpret_cast<f32x4*>(live_v + row) = v;
}

void launch_dm_finish(const DmStep& q, float* live_p, float* live_m, float* live_v, hipStream_t s) {
    MAMDR_LAUNCH(k_dm_finish, dim3((q.n_domain * (EMB / 4) + 255) / 256), dim3(256), 0, s, q, live_p, live_m, live_v);
}

// ---- k_pass_prep: the rows of a whole call resolved and gathered ONCE (frozen tables: the rows do not depend on
// the weights).  One wave per position: every lane resolves the position (the same addresses: broadcast loads),
// lane l copies float4 l of the 1-KB [user | item] row.  Replaces, per step, the tower kernel's chain of three
// dependent loads (perm -> uid / pid -> table rows) and its copy of x into the activation workspace.
// position i of a pass (one wave): see k_pass_prep
template <typename P>
__device__ __forceinline__ void pass_prep_row(const P& a, const float* user_tab, const float* item_tab, int n_user, int n_item,
                                              int n_domain, int64_t pos0, float* xpre, int32_t* pdom, float* plabel, int64_t i) {
    const int lane = threadIdx.x & 63;
    if (i >= a.n) {
        // 16 more rows: k_wgrad_adam contracts whole 16-row tiles counted from EVERY step's own first row (against
        // zero gradients for the padding rows, but 0 x garbage must stay 0); with a batch size that is no multiple of
        // 16 the last step's tile ends up to 15 rows past the call's last row, wherever that row sits
        // (and carry the pass's domain: a tower tile compares all of its rows' domains with the caller's)
        if (i < a.n + 16) {
            *reinterpret_cast<f32x4*>(xpre + (size_t)i * (2 * EMB) + 4 * lane) = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (lane == 0) {
                pdom[i] = a.pad_dom;
                plabel[i] = 0.f;
            }
        }
        return;
    }
    int64_t src = a.perm ? (int64_t)a.perm[pos0 + i] : pos0 + i;
    src = src < 0 ? 0 : (src >= a.n_rows_split ? a.n_rows_split - 1 : src);
    int u = a.uid[src], it = a.pid[src];
    u = u < 0 ? 0 : (u >= n_user ? n_user - 1 : u);
    it = it < 0 ? 0 : (it >= n_item ? n_item - 1 : it);
    const float* row = lane < 32 ? user_tab + (size_t)u * EMB + 4 * lane : item_tab + (size_t)it * EMB + 4 * (lane - 32);
    const f32x4 v = *reinterpret_cast<const f32x4*>(row);
#ifdef MAMDR_PREP_PLAIN_STORES      // (A/B: plain stores, so that the rows may stay in the infinity cache for the towers)
    *reinterpret_cast<f32x4*>(xpre + (size_t)i * (2 * EMB) + 4 * lane) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(xpre + (size_t)i * (2 * EMB) + 4 * lane));
#endif
    if (lane == 0) {
        int d = a.dom[src];
        pdom[i] = d < 0 ? 0 : (d >= n_domain ? n_domain - 1 : d);
        plabel[i] = a.label[src];
    }
}
__global__ __launch_bounds__(256) void k_pass_prep(const PassPrepArgs a) {
    if ((int)blockIdx.x >= a.n_prep_wgs) {       // the call's transposed weight copies (k_transpose_w) in the same launch
        transpose_w_elem(a.tw_dense, a.tw_L, a.tw_wT, ((int)blockIdx.x - a.n_prep_wgs) * 256 + (int)threadIdx.x);
        return;
    }
    pass_prep_row(a, a.user_tab, a.item_tab, a.n_user, a.n_item, a.n_domain, a.pos0, a.xpre, a.pdom, a.plabel,
                  (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
}
// the passes of several upcoming calls in one launch (mamdr_pregather_passes): the same rows, bit for bit
__global__ __launch_bounds__(256) void k_pass_prep_multi(const PassPrepMultiArgs a) {
    int k = 0;
    while (k + 1 < a.n_pass && (int)blockIdx.x >= a.wg_end[k]) ++k;      // (uniform)
    const int wg0 = k ? a.wg_end[k - 1] : 0;
    const PassPrepMultiArgs::Pass& p = a.p[k];
    pass_prep_row(p, a.user_tab, a.item_tab, a.n_user, a.n_item, a.n_domain, (int64_t)0, a.xpre + (size_t)p.out_off * (2 * EMB),
                  a.pdom + p.out_off, a.plabel + p.out_off, (int64_t)((int)blockIdx.x - wg0) * 4 + (threadIdx.x >> 6));
}
void launch_pass_prep_multi(const PassPrepMultiArgs& a, hipStream_t s) {
    if (a.n_pass <= 0) return;
    MAMDR_LAUNCH(k_pass_prep_multi, dim3((unsigned)a.wg_end[a.n_pass - 1]), dim3(256), 0, s, a);
}
void launch_pass_prep(const PassPrepArgs& a0, hipStream_t s) {
    if (a0.n <= 0) return;
    PassPrepArgs a = a0;
    a.n_prep_wgs = (int)((a.n + 16 + 3) / 4);
    MAMDR_LAUNCH(k_pass_prep, dim3((unsigned)(a.n_prep_wgs + (a.tw_wT ? TRANSPOSE_WGS : 0))), dim3(256), 0, s, a);
}

}  // namespace mamdr
