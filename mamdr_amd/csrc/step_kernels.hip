// Inner-step kernels of the MAMDR hot path for gfx950 (MI355X).
//
//   k_tower<TRAIN>  one workgroup = 16 batch rows: embedding gather (coalesced 512-B
//                   rows staged in LDS) -> 384-256-128-64-1 MLP on fp32 MFMA
//                   (v_mfma_f32_16x16x4_f32, weights streamed from L2 with 16-B
//                   loads, activations in LDS) -> sigmoid/BCE -> backward activation
//                   chain (dz3, dz2, dz1, d domain-embedding row).
//   k_wgrad         all weight-gradient contractions (K = batch rows) on
//                   v_mfma_f32_32x32x2_f32, split-K over row groups, one partial
//                   slab per group, fixed summation order (no float atomics).
//   k_update        slab reduction + regulariser gradient + TF1 Adam / SGD.
//
// Replaces `model.train_on_batch` / `model.evaluate` of the compiled Keras model
// (model_zoo/DeepCTR/deepctr.py:54-60,118-136; call sites model_zoo/mamdr.py:54,86,97).
#include <hip/hip_ext.h>

#include <cstring>

#include "emb_bodies.h"
#include "star_bodies.h"

namespace mamdr {

// ------------------------------------------------------------------ LDS map (floats)
constexpr int XS_LD = XDIM + 4;   // 388: row stride = 4 (mod 64) banks -> b128 reads spread
constexpr int H1_LD = H1 + 4;     // 260
constexpr int H2_LD = H2 + 4;     // 132
constexpr int H3_LD = H3 + 4;     // 68
// The x tile is dead after layer 0; its region is reused by the backward chain:
// dz3 (dead once dz2 exists) and dz1 share the front, dz2 sits behind them.
constexpr int XS_OFF = 0;
constexpr int DZ1S_OFF = XS_OFF;
constexpr int DZ3S_OFF = XS_OFF;
constexpr int DZ2S_OFF = XS_OFF + TILE_ROWS * H1_LD;                // 4160
constexpr int XREGION = DZ2S_OFF + TILE_ROWS * H2_LD;               // 6272 >= 16*388
constexpr int H1S_OFF = XS_OFF + XREGION;                           // 6272
constexpr int H2S_OFF = H1S_OFF + TILE_ROWS * H1_LD;                // 10432
constexpr int H3S_OFF = H2S_OFF + TILE_ROWS * H2_LD;                // 12544
constexpr int ROWI_OFF = H3S_OFF + TILE_ROWS * H3_LD;               // 13632
constexpr int LDS_FLOATS = ROWI_OFF + 12 * TILE_ROWS;               // 13824 floats = 55,296 B (rowf [64, 112): PNN's inner products)
static_assert(XREGION >= TILE_ROWS * XS_LD, "x tile must fit in its region");
static_assert(LDS_FLOATS * 4 <= 65536, "stay under the 64 KiB dynamic-LDS default");

size_t tower_lds_bytes() { return LDS_FLOATS * sizeof(float); }

template <int N> struct VecT;
template <> struct VecT<4> { typedef f32x4 type; };
template <> struct VecT<2> { typedef f32x2 type; };
template <> struct VecT<1> { typedef float type; };

// Workspace stores (activations, gradients, gradient slabs): write-through (agent-scope relaxed atomic store =
// global_store ... sc1) under -DMAMDR_WS_SC1 -- see tower4_kernels.hip: dirty lines left in the L2s are written back
// BETWEEN the kernels.  4-byte pieces only (wider vectors are stored piecewise).
#ifdef MAMDR_WS_PLAIN
#define WS_STORE1(ptr, val) (*(ptr) = (val))
#else
#define WS_STORE1(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
template <typename V>
__device__ __forceinline__ void ws_store(float* p, const V& v) {
    if constexpr (sizeof(V) == 4) {
        WS_STORE1(p, v);
    } else {
#pragma unroll
        for (int t = 0; t < (int)(sizeof(V) / 4); ++t) WS_STORE1(p + t, v[t]);
    }
}

constexpr int TOWER_THREADS = 512;   // 8 waves = 2 per SIMD: one wave's epilogue / waits overlap the other's MFMAs

template <int TPW>
__device__ __forceinline__ void load_b_rows(float (&b)[4][TPW], const float* __restrict__ p, int ld) {
    typedef typename VecT<TPW>::type V;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        V v = *reinterpret_cast<const V*>(p + (size_t)s * ld);
        if constexpr (TPW == 1) {
            b[s][0] = v;
        } else {
#pragma unroll
            for (int t = 0; t < TPW; ++t) b[s][t] = v[t];
        }
    }
}

// 16-deep chunk of the forward contraction.  With one tile per wave a single accumulator
// would chain on the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32, so even / odd
// sub-steps use two chains (acc[0], odd).
template <int TPW>
__device__ __forceinline__ void mfma_fwd_chunk(f32x4 (&acc)[TPW], f32x4& odd, const f32x4 a, const float (&b)[4][TPW]) {
    if constexpr (TPW == 1) {
        acc[0] = MAMDR_MFMA16(a[0], b[0][0], acc[0]);
        odd = MAMDR_MFMA16(a[1], b[1][0], odd);
        acc[0] = MAMDR_MFMA16(a[2], b[2][0], acc[0]);
        odd = MAMDR_MFMA16(a[3], b[3][0], odd);
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = MAMDR_MFMA16(a[s], b[s][t], acc[t]);
    }
}

// ---- forward layer: H[16 x N] = relu(A[16 x K] . W[K x N] + bias) (* dropout).
// NW waves split the N columns (16 TPW each); lane (j = lane & 15) owns TPW consecutive
// columns, so one 4/8/16-byte load per k row feeds TPW MFMAs, and the k index inside each
// 16-deep chunk is permuted identically for A and B (slot k' of sub-step s is
// k = kk0 + 4 k' + s), which turns the A fragment into one ds_read_b128.
// The weight stream is independent of the activations: a PF-deep register ring keeps PF
// chunks in flight from L2, pinned with sched_barrier (the scheduler otherwise sinks the
// loads to their use and serialises on vmcnt(0)); the first PF chunks and the bias (rewritten
// by k_update every step = an L2 miss) are requested one phase early by `prefetch`.
template <int K, int N, int PF, int NW>
struct FwdW {
    static constexpr int TPW = N / (16 * NW);
    static constexpr int NC = K / 16;
    static_assert(TPW >= 1 && N == 16 * NW * TPW, "columns must split evenly over the active waves");
    float b[PF][4][TPW];
    float bias[TPW];
    // (only register arrays live in the struct: scalar members captured through the `mid`
    // lambdas would be kept in scratch memory)
    static __device__ __forceinline__ bool active() { return (int)(threadIdx.x >> 6) < NW; }
    static __device__ __forceinline__ int ncol() {
        return (int)(threadIdx.x >> 6) * (16 * TPW) + TPW * (int)(threadIdx.x & 15);
    }
    static __device__ __forceinline__ const float* wptr(const float* __restrict__ W) {
        return W + (size_t)(4 * ((threadIdx.x & 63) >> 4)) * N + ncol();
    }
    __device__ __forceinline__ void prefetch(const float* __restrict__ W, const float* __restrict__ bias_ptr) {
        if (!active()) return;
        const int ncol = this->ncol();
        const float* wp = wptr(W);
#pragma unroll
        for (int t = 0; t < TPW; ++t) bias[t] = bias_ptr[ncol + t];
#pragma unroll
        for (int c = 0; c < PF && c < NC; ++c) load_b_rows<TPW>(b[c], wp + (size_t)(16 * c) * N, N);
        __builtin_amdgcn_sched_barrier(0);
    }
};

// `mid()` runs between the K loop and the epilogue: the caller requests the NEXT layer's
// weights there, so that those loads are older than this epilogue's global stores (vmcnt
// retires in issue order: a load queued behind the stores would wait for all of them).
// `extra(row, col)` (optional) is added to the pre-activation: PNN's three inner-product inputs times their rows of W0
struct NoExtra {
    static constexpr bool on = false;
    __device__ __forceinline__ float operator()(int, int) const { return 0.f; }
};
struct PnnExtra {
    static constexpr bool on = true;
    const float* ip;           // LDS [TILE_ROWS][3]
    const float* wx;           // global [3][H1]
    bool act;                  // (uniform) this launch is a PNN tower
    __device__ __forceinline__ float operator()(int row, int col) const {
        if (!act) return 0.f;
        return (ip[3 * row] * wx[col] + ip[3 * row + 1] * wx[H1 + col]) + ip[3 * row + 2] * wx[2 * H1 + col];
    }
};
template <int K, int N, int LDA, int LDO, bool TRAIN, int PF, int NW, typename Mid, typename Extra = NoExtra>
__device__ __forceinline__ void fwd_layer(FwdW<K, N, PF, NW>& fw, const float* __restrict__ W, const float* As,
                                          float* Os, float* gout,
                                          uint32_t key, uint32_t thresh, float scale, bool use_dropout, int row0,
                                          Mid mid, Extra extra = Extra()) {
    constexpr int TPW = FwdW<K, N, PF, NW>::TPW;
    constexpr int NC = K / 16;
    static_assert(NC % PF == 0, "chunk count must be a multiple of the ring depth");
    if (!FwdW<K, N, PF, NW>::active()) {
        mid();
        return;
    }
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    const int ncol = FwdW<K, N, PF, NW>::ncol();
    const float* wp = FwdW<K, N, PF, NW>::wptr(W);
    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float bv = fw.bias[t];
        acc[t] = (f32x4){bv, bv, bv, bv};
    }
    f32x4 odd = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ap = As + j * LDA + 4 * kq;
    f32x4 a_cur = *reinterpret_cast<const f32x4*>(ap);
#pragma unroll 1
    for (int c0 = 0; c0 < NC - PF; c0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            // A fragment of the NEXT chunk is requested before this chunk's MFMAs
            const f32x4 a_next = *reinterpret_cast<const f32x4*>(ap + 16 * (c0 + u + 1));
            __builtin_amdgcn_sched_barrier(0);
            mfma_fwd_chunk<TPW>(acc, odd, a_cur, fw.b[u]);
#ifndef MAMDR_ABLATE_LOADS   // diagnostic builds only: time the loop without its weight stream
            load_b_rows<TPW>(fw.b[u], wp + (size_t)(16 * (c0 + u + PF)) * N, N);
#endif
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_next;
        }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        f32x4 a_next = a_cur;
        if (u + 1 < PF) a_next = *reinterpret_cast<const f32x4*>(ap + 16 * (NC - PF + u + 1));
        __builtin_amdgcn_sched_barrier(0);
        mfma_fwd_chunk<TPW>(acc, odd, a_cur, fw.b[u]);
        __builtin_amdgcn_sched_barrier(0);
        a_cur = a_next;
    }
    if constexpr (TPW == 1) acc[0] += odd;
    mid();
    typedef typename VecT<TPW>::type V;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;
        float h[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            float z = acc[t][r];
            if constexpr (Extra::on) z += extra(row, ncol + t);
            z = fmaxf(z, 0.0f);
#ifndef MAMDR_ABLATE_HASH
            if (TRAIN && use_dropout) {
                const uint32_t u = mamdr_dropout_u32(key, (uint32_t)(row0 + row) * (uint32_t)N + (uint32_t)(ncol + t));
                z = (u >= thresh) ? z * scale : 0.0f;
            }
#endif
            h[t] = z;
        }
        V v;
        if constexpr (TPW == 1) {
            v = h[0];
        } else {
#pragma unroll
            for (int t = 0; t < TPW; ++t) v[t] = h[t];
        }
        *reinterpret_cast<V*>(Os + row * LDO + ncol) = v;
#ifndef MAMDR_ABLATE_STORES
        if (TRAIN) ws_store<V>(gout + (size_t)row * ACT_LD + ncol, v);
#endif
    }
}

// ---- backward layer: dH[16 x N] = dZ[16 x K] . W^T, W stored [N][LDW] with the reduction
// index contiguous.  Chunks are 32 deep: lane (j, k') loads the 32 B at row (n0 + 16 t + j),
// k = kk0 + 8 k' .. +7 as two 16-B loads, so the four k' lanes of a row consume one whole
// 128-B line; sub-step s of the chunk uses component s (slot k' <-> k = kk0 + 8 k' + s), the
// same permutation as the A fragment (two ds_read_b128).
template <int K, int N, int LDW, int PF, int NW>
struct BwdW {
    static constexpr int TPW = N / (16 * NW);
    static constexpr int NC = K / 32;
    static_assert(TPW >= 1 && N == 16 * NW * TPW, "columns must split evenly over the active waves");
    f32x4 b[PF][TPW][2];
    static __device__ __forceinline__ bool active() { return (int)(threadIdx.x >> 6) < NW; }
    static __device__ __forceinline__ int nbase() { return (int)(threadIdx.x >> 6) * (16 * TPW); }
    static __device__ __forceinline__ const float* wptr(const float* __restrict__ W) {
        return W + (size_t)(nbase() + (int)(threadIdx.x & 15)) * LDW + 8 * ((threadIdx.x & 63) >> 4);
    }
    __device__ __forceinline__ void load(int slot, int c, const float* wp) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const float* q = wp + (size_t)(16 * t) * LDW + 32 * c;
            b[slot][t][0] = *reinterpret_cast<const f32x4*>(q);
            b[slot][t][1] = *reinterpret_cast<const f32x4*>(q + 4);
        }
    }
    __device__ __forceinline__ void prefetch(const float* __restrict__ W) {
        if (!active()) return;
        const float* wp = wptr(W);
#pragma unroll
        for (int c = 0; c < PF && c < NC; ++c) load(c, c, wp);
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <int TPW>
__device__ __forceinline__ void mfma_bwd_chunk(f32x4 (&acc)[TPW], f32x4& odd, const f32x4 a0, const f32x4 a1,
                                               const f32x4 (&b)[TPW][2]) {
    if constexpr (TPW == 1) {
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
            acc[0] = MAMDR_MFMA16(a0[s], b[0][0][s], acc[0]);
            odd = MAMDR_MFMA16(a0[s + 1], b[0][0][s + 1], odd);
        }
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
            acc[0] = MAMDR_MFMA16(a1[s], b[0][1][s], acc[0]);
            odd = MAMDR_MFMA16(a1[s + 1], b[0][1][s + 1], odd);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = MAMDR_MFMA16(a0[s], b[t][0][s], acc[t]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = MAMDR_MFMA16(a1[s], b[t][1][s], acc[t]);
    }
}

template <int K, int N, int LDW, int LDA, int PF, int NW, typename Mid, typename Epi>
__device__ __forceinline__ void bwd_layer(BwdW<K, N, LDW, PF, NW>& bw, const float* __restrict__ W, const float* As,
                                          Mid mid, Epi epi) {
    constexpr int TPW = BwdW<K, N, LDW, PF, NW>::TPW;
    constexpr int NC = K / 32;
    static_assert(NC % PF == 0, "chunk count must be a multiple of the ring depth");
    if (!BwdW<K, N, LDW, PF, NW>::active()) {
        mid();
        return;
    }
    const float* wp = BwdW<K, N, LDW, PF, NW>::wptr(W);
    const int nbase = BwdW<K, N, LDW, PF, NW>::nbase();
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 odd = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ap = As + j * LDA + 8 * kq;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
#pragma unroll 1
    for (int c0 = 0; c0 < NC - PF; c0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const f32x4 n0 = *reinterpret_cast<const f32x4*>(ap + 32 * (c0 + u + 1));
            const f32x4 n1 = *reinterpret_cast<const f32x4*>(ap + 32 * (c0 + u + 1) + 4);
            __builtin_amdgcn_sched_barrier(0);
            mfma_bwd_chunk<TPW>(acc, odd, a0, a1, bw.b[u]);
#ifndef MAMDR_ABLATE_LOADS
            bw.load(u, c0 + u + PF, wp);
#endif
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0;
            a1 = n1;
        }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        f32x4 n0 = a0, n1 = a1;
        if (u + 1 < PF) {
            n0 = *reinterpret_cast<const f32x4*>(ap + 32 * (NC - PF + u + 1));
            n1 = *reinterpret_cast<const f32x4*>(ap + 32 * (NC - PF + u + 1) + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_bwd_chunk<TPW>(acc, odd, a0, a1, bw.b[u]);
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0;
        a1 = n1;
    }
    if constexpr (TPW == 1) acc[0] += odd;
    mid();
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(4 * kq + r, nbase + 16 * t + j, acc[t][r]);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// row bookkeeping + embedding gather of one 16-row tile into LDS (and optionally global)
// perm[position of row tid & 15], requested by every lane before anything else (see k_tower: loads retire in order)
__device__ __forceinline__ int early_perm(const TowerArgs& a, int r0) {
    if (!a.perm) return 0;
    const int64_t pc = a.row_base + min(r0 + (int)(threadIdx.x & (TILE_ROWS - 1)), max(a.rows - 1, 0));
    return a.perm[pc];
}
// pre-gathered pass (k_pass_prep): the tile's sixteen [user | item] rows (2 float4 per lane), domains and labels sit
// at known addresses -- requested before anything else, no dependent chain
struct PreTile {
    f32x4 x[2];
    int dom;
    float lab;
};
__device__ __forceinline__ void early_pre(const TowerArgs& a, int r0, PreTile& t) {
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + TOWER_THREADS * u;                 // 16 rows x 64 float4
        const int rr = min(r0 + (e >> 6), max(a.rows - 1, 0));
        t.x[u] = *reinterpret_cast<const f32x4*>(a.xpre + (size_t)rr * (2 * EMB) + 4 * (e & 63));
    }
    const int rb = min(r0 + (tid & (TILE_ROWS - 1)), max(a.rows - 1, 0));
    t.dom = a.pdom[rb];
    t.lab = a.plabel[rb];
}
// (PRE, the pre-gathered tile and the early permutation entry travel as a template flag / reference / value: handed over as
// `pre ? &pt : nullptr` and `&perm_src` they lived in scratch -- 96 B per lane in the one instance that used them, round 5)
template <bool PRE = false>
__device__ __forceinline__ void gather_tile(const TowerArgs& a, float* smem, int r0, float* gx, int gx_ld, const bool dmw,
                                            const bool has_perm_src, const int perm_src_v, const PreTile& pt) {
    int* rowi = reinterpret_cast<int*>(smem + ROWI_OFF);
    float* rowf = smem + ROWI_OFF + 4 * TILE_ROWS;
    const int tid = threadIdx.x;
    constexpr bool pre = PRE;
    if (pre) {
        if (tid < TILE_ROWS) {
            rowi[tid] = 0;
            rowi[TILE_ROWS + tid] = 0;
            rowi[2 * TILE_ROWS + tid] = pt.dom;
            rowi[3 * TILE_ROWS + tid] = (r0 + tid) < a.rows ? 1 : 0;
            rowf[tid] = pt.lab;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + TOWER_THREADS * u, row = e >> 6;
            f32x4 v = pt.x[u];
            if (r0 + row >= a.rows) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(smem + XS_OFF + row * XS_LD + 4 * (e & 63)) = v;
        }
    } else if (tid < TILE_ROWS) {
        const bool valid = (r0 + tid) < a.rows;
        int64_t pos = a.row_base + r0 + tid;
        int64_t src = 0;
        if (valid) {
            src = a.perm ? (has_perm_src ? (int64_t)perm_src_v : (int64_t)a.perm[pos]) : pos;
            if (src < 0) src = 0;
            if (src >= a.n_rows_split) src = a.n_rows_split - 1;
        }
        rowi[tid] = clampi(a.uid[src], 0, a.n_user - 1);
        rowi[TILE_ROWS + tid] = clampi(a.pid[src], 0, a.n_item - 1);
        rowi[2 * TILE_ROWS + tid] = clampi(a.dom[src], 0, a.n_domain - 1);
        rowi[3 * TILE_ROWS + tid] = valid ? 1 : 0;
        rowf[tid] = a.label[src];
    }
    __syncthreads();
    // 16 rows x 96 float4: 32 consecutive lanes read one 512-B embedding row.  The loads of a
    // thread are independent and issued back to back (indices are clamped, so the loads of
    // padding rows are harmless and are zeroed afterwards).
    float* xs = smem + XS_OFF;
    constexpr int PER = TILE_ROWS * (XDIM / 4) / TOWER_THREADS;   // 3
    // domain-table step still pending (DmStep, mamdr_kernels.h): the domain rows as that step leaves them, requested
    // beside the table rows (waves 4..7).  One domain per batch is the rule; if the tile's rows carry several, every
    // lane of a domain segment works alone.  (LDS scratch: the h1 region, unused until layer 0's epilogue.)
    const bool pend = dmw && a.dms.snap != nullptr;
    const int d0 = rowi[2 * TILE_ROWS];
    bool same = pend;
    if (pend) {
#pragma unroll
        for (int r = 1; r < TILE_ROWS; ++r) same = same && rowi[2 * TILE_ROWS + r] == d0;
    }
    float* parts = smem + H1S_OFF;
    DmTile dmt_;
    const DmTile* dmt = &dmt_;
    if (dmw) dm_tile_begin(a, (int)blockIdx.x, d0, dmt_);
    f32x4 v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + TOWER_THREADS * u;
        const int row = e / (XDIM / 4), c4 = e - row * (XDIM / 4);
        const int seg = c4 >> 5, off = (c4 & 31) * 4;
        if (pend && seg == 2) {
            v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!same) {
                f32x4 mn, vn;
                dm_step4(a.dms, rowi[2 * TILE_ROWS + row], c4 & 31, v[u], mn, vn);
            }
            continue;
        }
        if (pre && seg < 2) {          // already in LDS
            v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const float* base = seg == 0 ? a.user_tab : (seg == 1 ? a.item_tab : a.dense + a.L.dm);
        v[u] = *reinterpret_cast<const f32x4*>(base + (size_t)rowi[seg * TILE_ROWS + row] * EMB + off);
    }
    // Star: PartitionedNorm folded into the gather as a per-column affine, y = x * scale + shift with
    // separately rounded multiply and add (nn.batch_normalization, partitioned_norm.py:172-174)
    const float* aff = a.pn_aff;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + TOWER_THREADS * u;
        const int row = e / (XDIM / 4), c4 = e - row * (XDIM / 4);
        if (aff) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + c4 * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + XDIM + c4 * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = __fadd_rn(__fmul_rn(v[u][k], sc[k]), sh[k]);
        }
        if (same && (c4 >> 5) == 2) continue;        // filled behind the barrier below
        if (pre && (c4 >> 5) < 2) continue;          // pre-gathered pass: in LDS already, not copied to the workspace
        if (!rowi[3 * TILE_ROWS + row]) v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(xs + row * XS_LD + c4 * 4) = v[u];
        if (gx) *reinterpret_cast<f32x4*>(gx + (size_t)row * gx_ld + c4 * 4) = v[u];
    }
    if (dmw) {
        dm_tile_stage(a, (int)blockIdx.x, *dmt, parts);
        if (pend) __syncthreads();
        if (tid >= 256 && tid < 256 + EMB) {          // waves 4, 5: one column of the batch's domain row each
            if (same) {
                const int c = tid - 256;
                const float pn = dm_tile_reader(a, c, parts, *dmt);
#pragma unroll
                for (int rr = 0; rr < TILE_ROWS; ++rr) {
                    const float val = rowi[3 * TILE_ROWS + rr] ? pn : 0.f;
                    xs[rr * XS_LD + 2 * EMB + c] = val;
                    if (gx) gx[(size_t)rr * gx_ld + 2 * EMB + c] = val;
                }
            }
        } else if (tid >= 256 + EMB) {                // waves 6, 7: the row this workgroup writes back
            dm_tile_writer(a, (int)blockIdx.x, (int)gridDim.x, tid - 256 - EMB, parts, *dmt);
        }
    }
    __syncthreads();
}

// Diagnostic build only (-DMAMDR_STAMPS, tools/stamp_tower.py): per-phase s_memtime stamps
// of wave 0 into a buffer nothing else reads.  The production library has no stamp code.
#ifdef MAMDR_STAMPS
#define STAMP(k)                                                                              \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 16 + (k)] = t_;               \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// ring depths: forward in 16-deep chunks, backward in 32-deep chunks
// ring depths in chunks (diagnostic builds may override them)
#ifndef MAMDR_PF0
#define MAMDR_PF0 4
#endif
#ifndef MAMDR_PF1
#define MAMDR_PF1 4
#endif
#ifndef MAMDR_PF2
#define MAMDR_PF2 4
#endif
#ifndef MAMDR_PFB1
#define MAMDR_PFB1 2
#endif
constexpr int PF0 = MAMDR_PF0, PF1 = MAMDR_PF1, PF2 = MAMDR_PF2;
constexpr int PFB2 = 2, PFB1 = MAMDR_PFB1, PFB0 = 4;

// DX: the user / item tables are trainable, so every row also needs d loss / d [user | item]
// embedding = dz1 . W0[0:256, :]^T (the frozen-table path replaces that contraction by linearity).
// FM: DeepFM tower: logit += sum_f w_f[id_f] + sum_k (u i + u d + i d)_k.
// DXW: width of the input gradient written to a.dxe: 0 none, 256 = [user | item] (trainable tables),
// 384 = all three fields (Star: PartitionedNorm's backward needs d loss / d normalised input).
// FZ: the k_wgrad_adam path's duties compiled in (pre-gathered passes, the pending domain-table step, the W0 snapshot)
template <bool TRAIN, int DXW, bool FM, bool FZ = false>
__global__ __launch_bounds__(TOWER_THREADS) void k_tower(const int32_t* __restrict__ k_perm, const int64_t k_row_base,
                                                        const float* __restrict__ k_w0, const float* __restrict__ k_b0,
                                                        const int k_rows, const TowerArgs a) {
    // (leading scalar arguments = what the prologue's first loads need, preloaded into SGPRs with the wave -- see
    // k_tower4 -- `a` carries the same values)
    static_assert(!FZ || (TRAIN && DXW == 0 && !FM), "k_wgrad_adam serves the frozen-table mlp tower");
    constexpr bool DX = DXW > 0;
    constexpr int DXN = DX ? DXW : 2 * EMB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    const int n_tiles = (int)gridDim.x;
    const int r0 = tile * TILE_ROWS;
    int* rowi = reinterpret_cast<int*>(smem + ROWI_OFF);
    // [0,16) label, [16,32) DeepFM fm + linear term, [32,40) per-wave loss, [48,64) DeepFM dlogit
    float* rowf = smem + ROWI_OFF + 4 * TILE_ROWS;
    float* acts_t = TRAIN ? a.acts + (size_t)r0 * ACT_LD : nullptr;

    const float* P = a.dense;
    // layer-0 weights and the output-unit parameters do not depend on the gather: request them first
    FwdW<XDIM, H1, PF0, 8> fw0;
    FwdW<H1, H2, PF1, 8> fw1;
    // EARLY2 (training): layer 2's weights -- all of a wave's share, 32 registers -- are requested
    // one layer early, behind layer 0's K loop, and the backward copy of W2 behind layer 1's: the two short layers
    // otherwise start on a cold miss each (stamps at 4,096 rows: layer 2 6.3 K -> 4.8 K cycles, output / loss
    // 4.0 K -> 1.8 K; k_tower 26.2 -> 24.6 us).  A W1 image in LDS on top of this (as in k_tower4; built and measured:
    // requests at kernel start, gather + 2.5 K cycles; spread over layer 0's K loop, + 6.5 K) gained nothing here --
    // with 16 rows per tile layers 0 and 1 are paced by their MFMAs and epilogues, not by the weight stream.
    // (the whole share only where registers allow: the 384-wide Star variant stays under 128 for two workgroups per CU)
    constexpr bool EARLY2 = TRAIN && DXW <= 2 * EMB;
    // (the 384-wide Star variant: the same request points at the normal ring depth -- still 120 VGPRs; 65.0 -> 64.0 us)
    constexpr bool EARLY2S = TRAIN && DXW > 2 * EMB;
    FwdW<H2, H3, EARLY2 ? H2 / 16 : PF2, 4> fw2;
    BwdW<H3, H2, H3, PFB2, 8> bw2;
    BwdW<H2, H1, H2, PFB1, 8> bw1;
    // (the 384-wide variant keeps a 2-deep ring: 120 instead of 168 VGPRs, so that two workgroups share a CU)
    constexpr int PFB0V = DXN > 2 * EMB ? 2 : PFB0;
    BwdW<H1, DXN, H1, PFB0V, 8> bw0;
    STAMP(0);
    // the k_wgrad_adam path (pre-gathered passes, the pending domain-table step) exists for the frozen-table mlp tower
    // only and is an instance of its own: compiled into every instance, its registers cost the 384-wide Star variant
    // the second co-resident workgroup per CU (202 instead of <= 128 VGPRs: k_tower<384> 64.9 -> 72 us at 8,192 rows)
    constexpr bool FUSED_OK = FZ;
    const bool pre = FUSED_OK && a.xpre != nullptr;
    PreTile pt;
    if (pre) early_pre(a, r0, pt);
    int perm_src = 0;
    if (!pre && k_perm) perm_src = k_perm[k_row_base + min(r0 + (int)(threadIdx.x & (TILE_ROWS - 1)), max(k_rows - 1, 0))];
    __builtin_amdgcn_sched_barrier(0);
    fw0.prefetch(k_w0, k_b0);
    if (FUSED_OK) tower_snapshots(a, TOWER_THREADS, n_tiles);
    const bool dmw = FUSED_OK && a.dm_snap_out != nullptr;   // k_wgrad_adam path: domain-table duty (DmStep)
    const f32x2 wo_reg = *reinterpret_cast<const f32x2*>(P + a.L.wo + (tid & 31) * 2);
    const float gb_reg = P[a.L.gb];

    if (FUSED_OK && pre) gather_tile<true>(a, smem, r0, acts_t, ACT_LD, dmw, false, 0, pt);
    else gather_tile<false>(a, smem, r0, acts_t, ACT_LD, dmw, true, perm_src, pt);
    STAMP(1);
    // DeepFM: thread (i, part) owns columns 4 part .. +3 of row i's three fields.  u + i stays in
    // registers for the domain-table gradient (the x tile is overwritten by the backward chain).
    f32x4 fm_ui = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (FM) {
        const int i = tid >> 5, part = tid & 31;
        const float* xr = smem + XS_OFF + i * XS_LD + 4 * part;
        const f32x4 u = *reinterpret_cast<const f32x4*>(xr);
        const f32x4 it = *reinterpret_cast<const f32x4*>(xr + EMB);
        const f32x4 d = *reinterpret_cast<const f32x4*>(xr + 2 * EMB);
        const float fmw = a.deepfm == 1 ? 1.0f : 0.0f;    // 2 = WDL: linear tables only, no FM term
        fm_ui = fmw * (u + it);
        if (a.deepfm == 4) {
            // NFM (forward only here, as PNN): the DNN's input is the bi-interaction u i + (u + i) d; it takes the domain
            // field's place in the tile (rows 0..255 of W0 are zero), the linear tables' logit is formed below as WDL's
            *reinterpret_cast<f32x4*>(smem + XS_OFF + i * XS_LD + 2 * EMB + 4 * part) = u * it + (u + it) * d;
        }
        if (a.deepfm == 3) {
            // PNN (forward only here: its training steps run on the four-row tower): ip = <u,i> <u,d> <i,d> of the row
            const f32x4 p0 = u * it, p1 = u * d, p2 = it * d;
            float s0 = (p0[0] + p0[1]) + (p0[2] + p0[3]), s1 = (p1[0] + p1[1]) + (p1[2] + p1[3]),
                  s2 = (p2[0] + p2[1]) + (p2[2] + p2[3]);
            for (int o = 1; o < 32; o <<= 1) {
                s0 += __shfl_xor(s0, o);
                s1 += __shfl_xor(s1, o);
                s2 += __shfl_xor(s2, o);
            }
            if (part == 0) {
                rowf[4 * TILE_ROWS + 3 * i] = s0;
                rowf[4 * TILE_ROWS + 3 * i + 1] = s1;
                rowf[4 * TILE_ROWS + 3 * i + 2] = s2;
                rowf[TILE_ROWS + i] = 0.f;             // no linear / FM logit in this tower
            }
        } else {
            const f32x4 t = u * it + (u + it) * d;
            float s = fmw * ((t[0] + t[1]) + (t[2] + t[3]));
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 16);
            if (part == 0) {
                float lin = P[a.L.ld + rowi[2 * TILE_ROWS + i]];
                if (a.lin_user) lin = (a.lin_user[rowi[i]] + a.lin_item[rowi[TILE_ROWS + i]]) + lin;
                rowf[TILE_ROWS + i] = s + lin;
            }
        }
        __syncthreads();       // (PNN: layer 0's epilogue reads every row's inner products)
    }

    const float scale = a.use_dropout ? a.keep_scale : 1.0f;
    const int row0 = r0;   // row index inside the batch seeds the dropout stream
    const uint32_t key0 = TRAIN ? dropout_layer_key(a.seed, a.step, 0) : 0u;
    const uint32_t key1 = TRAIN ? dropout_layer_key(a.seed, a.step, 1) : 0u;
    const uint32_t key2 = TRAIN ? dropout_layer_key(a.seed, a.step, 2) : 0u;

    auto mid0 = [&]() {
        fw1.prefetch(P + a.L.w1, P + a.L.b1);
        if (EARLY2 || EARLY2S) fw2.prefetch(P + a.L.w2, P + a.L.b2);
    };
    if constexpr (FM)
        fwd_layer<XDIM, H1, XS_LD, H1_LD, TRAIN>(fw0, P + a.L.w0, smem + XS_OFF, smem + H1S_OFF, TRAIN ? acts_t + XDIM : nullptr,
                                                 key0, a.drop_thresh, scale, a.use_dropout != 0, row0, mid0,
                                                 PnnExtra{rowf + 4 * TILE_ROWS, P + a.L.wx, a.deepfm == 3});
    else
        fwd_layer<XDIM, H1, XS_LD, H1_LD, TRAIN>(fw0, P + a.L.w0, smem + XS_OFF, smem + H1S_OFF, TRAIN ? acts_t + XDIM : nullptr,
                                                 key0, a.drop_thresh, scale, a.use_dropout != 0, row0, mid0);
    STAMP(2);
    __syncthreads();
    fwd_layer<H1, H2, H1_LD, H2_LD, TRAIN>(fw1, P + a.L.w1, smem + H1S_OFF, smem + H2S_OFF,
                                           TRAIN ? acts_t + XDIM + H1 : nullptr, key1, a.drop_thresh, scale,
                                           a.use_dropout != 0, row0,
                                           [&]() {
                                               if (EARLY2 || EARLY2S) bw2.prefetch(P + a.L.w2);
                                               else fw2.prefetch(P + a.L.w2, P + a.L.b2);
                                           });
    STAMP(3);
    __syncthreads();
    // the backward weights are requested ahead of layer 2's epilogue stores
    fwd_layer<H2, H3, H2_LD, H3_LD, TRAIN>(fw2, P + a.L.w2, smem + H2S_OFF, smem + H3S_OFF,
                                           TRAIN ? acts_t + XDIM + H1 + H2 : nullptr, key2, a.drop_thresh, scale,
                                           a.use_dropout != 0, row0, [&]() {
                                               if (TRAIN) {
                                                   if (!EARLY2 && !EARLY2S) bw2.prefetch(P + a.L.w2);
                                                   bw1.prefetch(P + a.L.w1);
                                               }
                                           });
    STAMP(4);
    __syncthreads();

    // ---- output unit, sigmoid, Keras BCE and dz3, all 512 threads: thread (i, part) owns
    // 2 of the 64 hidden units of row i; the 32 lanes of a row reduce by xor-shuffles.
    float* dz_t = TRAIN ? a.dz + (size_t)r0 * DZ_LD : nullptr;
    {
        const int i = tid >> 5, part = tid & 31, n2 = part * 2;
        const f32x2 wo = wo_reg;
        const f32x2 h = *reinterpret_cast<const f32x2*>(smem + H3S_OFF + i * H3_LD + n2);
        float s = fmaf(h[1], wo[1], h[0] * wo[0]);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        float logit = s + gb_reg;
        if (FM) logit += rowf[TILE_ROWS + i];
        float p;
        if (logit >= 0.f) {
            p = 1.0f / (1.0f + __expf(-logit));
        } else {
            const float ez = __expf(logit);
            p = ez / (1.0f + ez);
        }
        const bool valid = rowi[3 * TILE_ROWS + i] != 0;
        const float y = rowf[i];
        const float lo = 1e-7f, hi = 1.0f - 1e-7f;
        const float pc = fminf(fmaxf(p, lo), hi);
        const float zc = __logf(pc / (1.0f - pc));
        float loss = fmaxf(zc, 0.f) - zc * y + __logf(1.0f + __expf(-fabsf(zc)));
        if (!valid || part != 0) loss = 0.f;
        if (TRAIN) {
            const float inside = (p >= lo && p <= hi) ? 1.0f : 0.0f;
            float dl = valid ? ((p - y) * inside) / (float)a.rows : 0.0f;
            if (a.uw_off >= 0) {       // uncertainty weighting: d loss / d logit scales by 1 / var^2
                const float var = P[a.uw_off];
                dl *= 1.0f / (var * var);
            }
            if (part == 0) {
                a.dlogit[r0 + i] = dl;
                a.domrow[r0 + i] = rowi[2 * TILE_ROWS + i];
                if (DX) {
                    a.urow[r0 + i] = valid ? rowi[i] : -1;
                    a.irow[r0 + i] = valid ? rowi[TILE_ROWS + i] : -1;
                    if (valid && a.map_u) {   // representative of a table row = its smallest batch position (exact)
                        atomicMin(a.map_u + rowi[i], r0 + i);
                        atomicMin(a.map_i + rowi[TILE_ROWS + i], r0 + i);
                    }
                }
            }
            // gate = relu'(z) * dropout mask / keep = (h_post > 0) * scale
            f32x2 d;
#pragma unroll
            for (int c = 0; c < 2; ++c) d[c] = (h[c] > 0.f) ? (dl * wo[c]) * scale : 0.f;
            *reinterpret_cast<f32x2*>(smem + DZ3S_OFF + i * H3_LD + n2) = d;
            ws_store<f32x2>(dz_t + (size_t)i * DZ_LD + H1 + H2 + n2, d);
            if (FM) {
                // d fm / d domain embedding = u + i: per-row term of the domain-table gradient
                *reinterpret_cast<f32x4*>(a.fmq + (size_t)(r0 + i) * EMB + 4 * part) = dl * fm_ui;
                if (DX && part == 0) rowf[3 * TILE_ROWS + i] = dl;
            }
        } else if (part == 0 && valid) {
            // AUC bin = number of thresholds strictly below p (utils/metrics_utils.py:309: pred > thr)
            int blo = 0, bhi = 500;
            while (blo < bhi) {
                const int mid = (blo + bhi) >> 1;
                if (a.thresholds[mid] < p) blo = mid + 1; else bhi = mid;
            }
            atomicAdd(a.hist + (y != 0.f ? 501 : 0) + blo, 1u);
            if (a.pred_out) a.pred_out[a.row_base + r0 + i] = p;
        }
        // tile loss: the 2 rows of this wave sit in lanes 0 and 32 (others hold 0)
        loss += __shfl_xor(loss, 32);
        if (lane == 0) rowf[2 * TILE_ROWS + w] = loss;
    }
    STAMP(5);
    __syncthreads();
    if (tid == 0) {
        const float* lp = rowf + 2 * TILE_ROWS;
        a.loss_part[tile] = (((lp[0] + lp[1]) + (lp[2] + lp[3])) + ((lp[4] + lp[5]) + (lp[6] + lp[7])));
    }
    if (!TRAIN) return;
    STAMP(6);
    {
        float* dzs = smem + DZ2S_OFF;
        const float* hs = smem + H2S_OFF;
        bwd_layer<H3, H2, H3, H3_LD>(bw2, P + a.L.w2, smem + DZ3S_OFF, [&]() { if (DX) bw0.prefetch(P + a.L.w0); },
                                     [&](int row, int col, float v) {
            const float d = (hs[row * H2_LD + col] > 0.f) ? v * scale : 0.f;
            dzs[row * H2_LD + col] = d;
#ifndef MAMDR_ABLATE_STORES
            WS_STORE1(&dz_t[(size_t)row * DZ_LD + H1 + col], d);
#endif
        });
    }
    STAMP(7);
    __syncthreads();
    {
        // dz1 is only stored: the gradient of the domain-embedding rows follows from it by
        // linearity in k_wgrad / k_update (dDm = onehot(domain)^T dz1 . W0[256:384,:]^T), so no
        // per-row contraction with W0 is needed while the user / item tables are frozen.
        const float* hs = smem + H1S_OFF;
        float* dzs = smem + DZ1S_OFF;
        bwd_layer<H2, H1, H2, H2_LD>(bw1, P + a.L.w1, smem + DZ2S_OFF, []() {}, [&](int row, int col, float v) {
            const float d = (hs[row * H1_LD + col] > 0.f) ? v * scale : 0.f;
            if (DX) dzs[row * H1_LD + col] = d;
#ifndef MAMDR_ABLATE_STORES
            WS_STORE1(&dz_t[(size_t)row * DZ_LD + col], d);
#endif
        });
    }
    STAMP(8);
    if (DX) {
        __syncthreads();
        float* dxe_t = a.dxe + (size_t)r0 * DXN;
        bwd_layer<H1, DXN, H1, H1_LD>(bw0, P + a.L.w0, smem + DZ1S_OFF, []() {}, [&](int row, int col, float v) {
            if (FM) {
                // d fm / d e_f = sum of the other two fields (x re-read from the activation workspace)
                const float* xr = acts_t + (size_t)row * ACT_LD;
                const int k = col & (EMB - 1);
                const float other = (col < EMB ? xr[EMB + k] : xr[k]) + xr[2 * EMB + k];
                if (a.deepfm == 1) v = fmaf(rowf[3 * TILE_ROWS + row], other, v);
            }
            dxe_t[(size_t)row * DXN + col] = v;
        });
        if constexpr (DXW == XDIM) {
            // Star: PartitionedNorm's backward starts with the column sums s1 = sum dxn, s2 = sum dxn * xhat over the
            // batch.  The tile's share is formed here, in k_star_pnb_partial's arithmetic and row order (one thread per
            // column, rows 0..15 in order: the same bits), on the d x the workgroup has just written (L1 / L2) and the raw
            // table rows its gather read -- a launch (5.2 us + its boundary) less per step.
            if (a.pn_part) {
                __syncthreads();
                if (tid < XDIM) {
                    const int c = tid, seg = c >> 7, kk = c & (EMB - 1);
                    const int nb = min(TILE_ROWS, a.rows - r0);
                    const float mean = a.pn_aff[2 * XDIM + c], inv = a.pn_aff[3 * XDIM + c];
                    float g[TILE_ROWS], x[TILE_ROWS];
#pragma unroll
                    for (int r = 0; r < TILE_ROWS; ++r) {
                        const int rr = min(r, nb - 1);
                        g[r] = dxe_t[(size_t)rr * DXN + c];
                        const float* rowp = seg == 0 ? a.user_tab + (size_t)rowi[rr] * EMB
                                                     : (seg == 1 ? a.item_tab + (size_t)rowi[TILE_ROWS + rr] * EMB
                                                                 : P + a.L.dm + (size_t)rowi[2 * TILE_ROWS + rr] * EMB);
                        x[r] = rowp[kk];
                    }
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int r = 0; r < TILE_ROWS; ++r) {
                        if (r < nb) {
                            const float xh = (x[r] - mean) * inv;
                            s1 += g[r];
                            s2 += g[r] * xh;
                        }
                    }
                    a.pn_part[(size_t)tile * 2 * XDIM + c] = s1;
                    a.pn_part[(size_t)tile * 2 * XDIM + XDIM + c] = s2;
                }
            }
        }
    }
    STAMP(9);
}

void launch_tower_train(const TowerArgs& a, hipStream_t s) {
    const int tiles = (a.rows + TILE_ROWS - 1) / TILE_ROWS;
    const dim3 grid(tiles), block(TOWER_THREADS);
    const size_t lds = tower_lds_bytes();
    if (a.deepfm) {
        if (a.dxe) MAMDR_LAUNCH((k_tower<true, 256, true>), grid, block, lds, s, a.perm, a.row_base, a.dense + a.L.w0, a.dense + a.L.b0, a.rows, a);
        else MAMDR_LAUNCH((k_tower<true, 0, true>), grid, block, lds, s, a.perm, a.row_base, a.dense + a.L.w0, a.dense + a.L.b0, a.rows, a);
    } else if (a.dxe && a.dx_ld == XDIM) {
        MAMDR_LAUNCH((k_tower<true, 384, false>), grid, block, lds, s, a.perm, a.row_base, a.dense + a.L.w0, a.dense + a.L.b0, a.rows, a);
    } else if (a.dxe) {
        MAMDR_LAUNCH((k_tower<true, 256, false>), grid, block, lds, s, a.perm, a.row_base, a.dense + a.L.w0, a.dense + a.L.b0, a.rows, a);
    } else if (a.xpre || a.dm_snap_out || a.dms.snap || a.w0dom_snap) {
        MAMDR_LAUNCH((k_tower<true, 0, false, true>), grid, block, lds, s, a.perm, a.row_base, a.dense + a.L.w0, a.dense + a.L.b0, a.rows, a);
    } else {
        MAMDR_LAUNCH((k_tower<true, 0, false>), grid, block, lds, s, a.perm, a.row_base, a.dense + a.L.w0, a.dense + a.L.b0, a.rows, a);
    }
}
void launch_tower_eval(const TowerArgs& a, hipStream_t s) {
    const int tiles = (a.rows + TILE_ROWS - 1) / TILE_ROWS;
    if (a.deepfm)
        hipLaunchKernelGGL((k_tower<false, 0, true>), dim3(tiles), dim3(TOWER_THREADS), tower_lds_bytes(), s, a.perm, a.row_base, a.dense + a.L.w0,
                           a.dense + a.L.b0, a.rows, a);
    else
        hipLaunchKernelGGL((k_tower<false, 0, false>), dim3(tiles), dim3(TOWER_THREADS), tower_lds_bytes(), s, a.perm, a.row_base, a.dense + a.L.w0,
                           a.dense + a.L.b0, a.rows, a);
}

// ------------------------------------------------------------------ standalone gather
__global__ __launch_bounds__(TOWER_THREADS) void k_gather(const TowerArgs a, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int r0 = blockIdx.x * TILE_ROWS;
    {
        PreTile none;
        gather_tile<false>(a, smem, r0, nullptr, 0, false, false, 0, none);
    }
    const float* xs = smem + XS_OFF;
    for (int e = threadIdx.x; e < TILE_ROWS * (XDIM / 4); e += TOWER_THREADS) {
        const int row = e / (XDIM / 4), c4 = e - row * (XDIM / 4);
        if (r0 + row < a.rows)
            *reinterpret_cast<f32x4*>(out + (size_t)(r0 + row) * XDIM + c4 * 4) =
                *reinterpret_cast<const f32x4*>(xs + row * XS_LD + c4 * 4);
    }
}
void launch_gather(const TowerArgs& a, float* out, hipStream_t s) {
    const int tiles = (a.rows + TILE_ROWS - 1) / TILE_ROWS;
    MAMDR_LAUNCH(k_gather, dim3(tiles), dim3(TOWER_THREADS), tower_lds_bytes(), s, a, out);
}

// ------------------------------------------------------------------ eval loss epilogue
// Keras evaluate: mean over batches of (batch-mean BCE + regularisers)
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    // fixed-order tree: wave shuffle then 4 partials in order
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// regularisers of the reported loss: l2_emb on the three embedding tables (frozen or not)
// and, for DeepFM, l2_lin on the three linear tables (A.8); same order as the oracle's reg_loss
__device__ __forceinline__ float reg_terms(const float* dense, int dm_count, float l2_emb, const float* frozen_sumsq,
                                           int ld_off, int ld_count, float l2_lin, float* red) {
    float ss = 0.f;
    for (int e = threadIdx.x; e < dm_count; e += 256) ss = fmaf(dense[e], dense[e], ss);
    ss = block_sum_256(ss, red);
    float reg = l2_emb * frozen_sumsq[0] + l2_emb * frozen_sumsq[1] + l2_emb * ss;
    if (ld_count > 0) {
        float sl = 0.f;
        for (int e = threadIdx.x; e < ld_count; e += 256) sl = fmaf(dense[ld_off + e], dense[ld_off + e], sl);
        sl = block_sum_256(sl, red);
        reg = ((reg + l2_lin * frozen_sumsq[2]) + l2_lin * frozen_sumsq[3]) + l2_lin * sl;
    }
    return reg;
}

__global__ __launch_bounds__(256) void k_eval_finish(const EvalFinishArgs a) {
    __shared__ float red[4];
    const float reg = reg_terms(a.dense, a.dm_count, a.l2_emb, a.frozen_sumsq, a.ld_off, a.ld_count, a.l2_lin, red);
    const int64_t n_rows = a.n_rows;
    const int batch = a.batch;
    const int64_t n_batches = (n_rows + batch - 1) / batch;
    const int tiles_per_batch = batch / TILE_ROWS;
    const int64_t n_tiles = (n_rows + TILE_ROWS - 1) / TILE_ROWS;
    float acc = 0.f;
    for (int64_t b = threadIdx.x; b < n_batches; b += 256) {
        const int64_t t0 = b * tiles_per_batch;
        const int64_t t1 = (t0 + tiles_per_batch < n_tiles) ? t0 + tiles_per_batch : n_tiles;
        float s = 0.f;
        for (int64_t t = t0; t < t1; ++t) s += a.loss_part[t];
        const int64_t rows_b = (b == n_batches - 1) ? (n_rows - b * (int64_t)batch) : batch;
        acc += s / (float)rows_b + reg;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) a.loss_out[0] = acc / (float)n_batches;
}
void launch_eval_finish(const EvalFinishArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_eval_finish, dim3(1), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ sum of squares (frozen tables)
__global__ __launch_bounds__(256) void k_sumsq_part(const float* x, int64_t n, float* partials) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        s = fmaf(x[e], x[e], s);
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_sumsq_final(const float* partials, int n, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) s += partials[e];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = s;
}
void launch_sumsq(const float* x, int64_t n, float* partials, float* out, hipStream_t s) {
    const int blocks = 1024;
    hipLaunchKernelGGL(k_sumsq_part, dim3(blocks), dim3(256), 0, s, x, n, partials);
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, s, partials, blocks, out);
}

// ------------------------------------------------------------------ weight gradients
// One workgroup = one 32x32 output tile x one row group; its 4 waves split the group's
// rows and are summed through LDS in wave order.  Operands are fetched straight from
// the row-major activation / gradient workspaces: lane (c = lane & 31, kk = lane >> 5)
// reads element c of row b + kk, i.e. one full 128-B line per half wave.  The loop is
// latency-bound (two 4-byte loads per MFMA), so loads run one 16-row block ahead of the
// MFMAs in a second register set, pinned with sched_barrier.
template <int AK>
__device__ __forceinline__ float fetch_a(const WgradArgs& g, const TileDesc& t, int b, int c) {
    if (AK == 0) return g.acts[(size_t)b * ACT_LD + t.a_off + c];
    if (AK == 1) return c == 0 ? 1.0f : 0.0f;
    if (AK == 3) return c < 4 ? g.ipbuf[(size_t)b * 4 + c] : 0.0f;      // PNN: the rows' inner products (column 3 is zero)
    return (g.domrow[b] == t.a_off + c) ? 1.0f : 0.0f;
}
template <int BK>
__device__ __forceinline__ float fetch_b(const WgradArgs& g, const TileDesc& t, int b, int c) {
    if (BK == 0) return g.dz[(size_t)b * DZ_LD + t.b_off + c];
    if (BK == 2) return g.fmq[(size_t)b * EMB + t.b_off + c];
    return c == 0 ? g.dlogit[b] : 0.0f;
}

template <int AK, int BK>
__device__ __forceinline__ void wgrad_load16(const WgradArgs& g, const TileDesc& t, int b, int kk, int c,
                                             float (&av)[8], float (&bv)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        av[u] = fetch_a<AK>(g, t, b + 2 * u + kk, c);
        bv[u] = fetch_b<BK>(g, t, b + 2 * u + kk, c);
    }
}

template <int AK, int BK>
__device__ __forceinline__ void wgrad_rows(const WgradArgs& g, const TileDesc& t, int b0, int b1, f32x16& acc) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, kk = lane >> 5;
    int b = b0;
    if (b + 16 <= b1) {
        float a0[8], v0[8], a1[8], v1[8];
        wgrad_load16<AK, BK>(g, t, b, kk, c, a0, v0);
        __builtin_amdgcn_sched_barrier(0);
        // two register sets alternate: while one feeds the MFMAs the other is in flight
        for (; b + 48 <= b1; b += 32) {
            wgrad_load16<AK, BK>(g, t, b + 16, kk, c, a1, v1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = MAMDR_MFMA32(a0[u], v0[u], acc);
            wgrad_load16<AK, BK>(g, t, b + 32, kk, c, a0, v0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = MAMDR_MFMA32(a1[u], v1[u], acc);
        }
        if (b + 32 <= b1) {
            wgrad_load16<AK, BK>(g, t, b + 16, kk, c, a1, v1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = MAMDR_MFMA32(a0[u], v0[u], acc);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = MAMDR_MFMA32(a1[u], v1[u], acc);
            b += 32;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = MAMDR_MFMA32(a0[u], v0[u], acc);
            b += 16;
        }
    }
    for (; b + 2 <= b1; b += 2) {
        const float av = fetch_a<AK>(g, t, b + kk, c);
        const float bv = fetch_b<BK>(g, t, b + kk, c);
        acc = MAMDR_MFMA32(av, bv, acc);
    }
}

constexpr int W0DOM_FLOAT4 = EMB * H1 / 4;               // rows 256..383 of W0
constexpr int W0DOM_COPY_WGS = W0DOM_FLOAT4 / 256;        // 32


#ifdef MAMDR_STAMPS
#define WSTAMP(k)                                                                             \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (g.stamps && threadIdx.x == 0 && blockIdx.x < 1024) g.stamps[blockIdx.x * 8 + (k)] = t_; \
    } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif

// ---- 64x64 tiles of acts^T dz: the four waves own the 2x2 32x32 quadrants of the tile and share the
// operands through LDS.  Rows are staged in chunks of 32 (16-B global loads, 16 lanes per 256-B row slice,
// a 4-deep register ring ahead of the MFMAs); LDS rows are 96 floats apart so that the two half-waves of
// an operand read (rows 2i and 2i+1) hit disjoint banks.
constexpr int WG_KC = 32;          // rows per staged chunk
constexpr int WG_LD = 96;          // LDS row stride (floats)
constexpr int WG_PF = 4;           // chunks in flight
constexpr int WG_BUF = WG_KC * WG_LD;          // one operand buffer
static_assert(4 * WG_BUF >= 4 * 1024, "the 32x32 path's reduction buffer aliases the staging buffers");

__device__ __forceinline__ void wgrad_big(const WgradArgs& g, const TileDesc& t, int gb0, int gb1, float* lds, int grp) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int c = lane & 31, kk = lane >> 5;
    const int lr = tid >> 4, lc4 = (tid & 15) * 4;          // staging: rows lr and lr + 16, columns lc4..+3
    const int n_chunks = (gb1 - gb0 + WG_KC - 1) / WG_KC;
    f32x4 ra[WG_PF][2], rb[WG_PF][2];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int slot, int ch) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = gb0 + ch * WG_KC + lr + 16 * h;
            const bool ok = ch < n_chunks && b < gb1;
            ra[slot][h] = ok ? *reinterpret_cast<const f32x4*>(g.acts + (size_t)b * ACT_LD + t.a_off + lc4) : zero4;
            rb[slot][h] = ok ? *reinterpret_cast<const f32x4*>(g.dz + (size_t)b * DZ_LD + t.b_off + lc4) : zero4;
        }
    };
#pragma unroll
    for (int s = 0; s < WG_PF; ++s) issue(s, s);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ch0 = 0; ch0 < n_chunks; ch0 += WG_PF) {
#pragma unroll
        for (int u = 0; u < WG_PF; ++u) {
            const int ch = ch0 + u;
            if (ch >= n_chunks) break;                 // uniform
            float* As = lds + (ch & 1) * 2 * WG_BUF;
            float* Bs = As + WG_BUF;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<f32x4*>(As + (lr + 16 * h) * WG_LD + lc4) = ra[u][h];
                *reinterpret_cast<f32x4*>(Bs + (lr + 16 * h) * WG_LD + lc4) = rb[u][h];
            }
            __syncthreads();
            issue(u, ch + WG_PF);
            __builtin_amdgcn_sched_barrier(0);
            const float* ap = As + kk * WG_LD + wm * 32 + c;
            const float* bp = Bs + kk * WG_LD + wn * 32 + c;
#pragma unroll
            for (int i = 0; i < WG_KC / 2; ++i) acc = MAMDR_MFMA32(ap[2 * i * WG_LD], bp[2 * i * WG_LD], acc);
        }
    }
    WSTAMP(2);
    // D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* dst = g.slabs + (size_t)grp * g.slab_ld + t.dst_off +
                 (size_t)(wm * 32) * t.dst_ld + wn * 32;
    const int rb4 = 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) WS_STORE1(&dst[(size_t)((r & 3) + 8 * (r >> 2) + rb4) * t.dst_ld + c], acc[r]);
    WSTAMP(3);
    WSTAMP(4);
}

// (a body: the kernel also hosts k_emb_reduce's workgroups in k_wgrad_reduce; bid = workgroup index, red = 4 WG_BUF floats of LDS)
__device__ __forceinline__ void wgrad_body(const WgradArgs& g, const int bid, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n_work = g.n_tiles * g.n_groups;
    if (bid > n_work) {
        // ---- snapshots (pre-update) for k_update: W0[256:384, :] for the domain-table gradient and the
        // domain table for dW0[256:384, :]
        const int e = (bid - n_work - 1) * 256 + tid;
        if (e < W0DOM_FLOAT4) {
            reinterpret_cast<f32x4*>(g.w0dom_copy)[e] = reinterpret_cast<const f32x4*>(g.w0dom)[e];
        } else if (g.dm_copy && e - W0DOM_FLOAT4 < g.dm_count / 4) {
            reinterpret_cast<f32x4*>(g.dm_copy)[e - W0DOM_FLOAT4] = reinterpret_cast<const f32x4*>(g.dense)[e - W0DOM_FLOAT4];
        }
        return;
    }
    if (bid == n_work) {
        // ---- one extra workgroup: loss of the step = mean BCE + regularisers
        if (g.loss_out == nullptr && g.lv_count == 0) return;
        float* r4 = red;
        const float reg = reg_terms(g.dense, g.dm_count, g.l2_emb, g.frozen_sumsq, g.ld_off, g.ld_count, g.l2_lin, r4);
        float ls = 0.f;
        for (int e = tid; e < g.n_loss_tiles; e += 256) ls += g.loss_part[e];
        ls = block_sum_256(ls, r4);
        const float mean_bce = ls / (float)g.rows;
        if (g.lv_count > 0) {
            // uncertainty weighting: loss = mean(BCE) / var^2 + log var + regularisers;
            // d loss / d var = -2 mean(BCE) / var^3 + 1 / var for the batch's domain, 0 for the others
            // (written for every domain: slab 0 is the only slab that carries this region)
            const float var = g.dense[g.lv_off + g.uw_d];
            if (tid < g.lv_count)
                g.slabs[g.lv_off + tid] = tid == g.uw_d ? (-2.0f * mean_bce / (var * var * var) + 1.0f / var) : 0.f;
            if (tid == 0 && g.loss_out) g.loss_out[0] = (1.0f / (var * var)) * mean_bce + __logf(var) + reg;
            return;
        }
        if (tid == 0) g.loss_out[0] = mean_bce + reg;
        return;
    }
    WSTAMP(0);
    // XCD-aware mapping: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own
    // 4 MB L2), so the row group is the fast index: every XCD then touches the activation / gradient
    // rows of only n_groups/8 (or one of n_groups) groups and its tiles' re-reads of them hit its L2.
    const int grp = bid % g.n_groups, tile = bid / g.n_groups;
    const TileDesc t = g.tiles[tile];
    WSTAMP(1);
    const int gb0 = grp * g.rows_per_group;
    const int gb1 = min(gb0 + g.rows_per_group, g.rows_pad);
    if (t.big) {
        wgrad_big(g, t, gb0, gb1, red, grp);
        return;
    }
    // split the group's rows over the 4 waves in multiples of 2
    const int span = gb1 > gb0 ? gb1 - gb0 : 0;
    const int per = ((span + 7) / 8) * 2;
    const int b0 = min(gb0 + w * per, gb1), b1 = min(b0 + per, gb1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (t.a_kind == 0 && t.b_kind == 0) wgrad_rows<0, 0>(g, t, b0, b1, acc);
    else if (t.a_kind == 1 && t.b_kind == 0) wgrad_rows<1, 0>(g, t, b0, b1, acc);
    else if (t.a_kind == 0 && t.b_kind == 1) wgrad_rows<0, 1>(g, t, b0, b1, acc);
    else if (t.a_kind == 1 && t.b_kind == 1) wgrad_rows<1, 1>(g, t, b0, b1, acc);
    else if (t.a_kind == 3) wgrad_rows<3, 0>(g, t, b0, b1, acc);      // PNN: dW0x = ip^T dz1
    else if (t.b_kind == 0) wgrad_rows<2, 0>(g, t, b0, b1, acc);
    else if (t.b_kind == 1) wgrad_rows<2, 1>(g, t, b0, b1, acc);    // DeepFM: linear domain table
    else wgrad_rows<2, 2>(g, t, b0, b1, acc);                        // DeepFM: S2 = onehot(domain)^T fmq
    WSTAMP(2);
    // D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    {
        const int col = lane & 31, rb = 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) red[w * 1024 + ((r & 3) + 8 * (r >> 2) + rb) * 32 + col] = acc[r];
    }
    __syncthreads();
    WSTAMP(3);
    float* slab = g.slabs + (size_t)grp * g.slab_ld;
#pragma unroll
    for (int e = tid; e < 1024; e += 256) {
        const int row = e >> 5, col = e & 31;
        if (row < t.m_valid && col < t.n_valid) {
            const float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
            WS_STORE1(&slab[t.dst_off + row * t.dst_ld + col], v);
        }
    }
    WSTAMP(4);
}
// (leading scalar arguments: preloaded into SGPRs with the wave -- see k_tower4 -- so that the tile descriptor and the
// first operand rows are requested before the argument block has been fetched; `g0` carries the same values)
#define WGRAD_EARLY_PARAMS                                                                                             \
    const float *__restrict__ k_acts, const float *__restrict__ k_dz, const TileDesc *__restrict__ k_tiles,            \
        const int k_n_tiles, const int k_rows_pad, const int k_n_groups, const int k_rows_per_group
#define WGRAD_EARLY_ARGS(a) (a).acts, (a).dz, (a).tiles, (a).n_tiles, (a).rows_pad, (a).n_groups, (a).rows_per_group
#define WGRAD_EARLY_APPLY(g, g0)                                                                                       \
    WgradArgs g = g0;                                                                                                  \
    g.acts = k_acts; g.dz = k_dz; g.tiles = k_tiles; g.n_tiles = k_n_tiles; g.rows_pad = k_rows_pad;                  \
    g.n_groups = k_n_groups; g.rows_per_group = k_rows_per_group
__global__ __launch_bounds__(256) void k_wgrad(WGRAD_EARLY_PARAMS, const WgradArgs g0) {
    __shared__ __attribute__((aligned(16))) float red[4 * WG_BUF];
    WGRAD_EARLY_APPLY(g, g0);
    wgrad_body(g, (int)blockIdx.x, red);
}
// k_wgrad and k_emb_reduce only need the tower's outputs and write disjoint state: one launch, the table
// workgroups behind the weight-gradient ones (they share CUs instead of queueing behind each other)
// ... and the NEXT step's k_emb_rows (n_rows workgroups): it writes the other half of the row-id / map double
// buffer, so this step's reduction still sees its own maps
// ... and (Star tower) the domain-row column sums of PartitionedNorm's backward, n_dm workgroups of 16 columns
__global__ __launch_bounds__(256) void k_wgrad_reduce(WGRAD_EARLY_PARAMS, const WgradArgs g0, const EmbStepArgs e,
                                                      const int n_wgrad, const EmbRowsArgs nr, const int n_rows,
                                                      const StarPnBwdArgs sd, const int n_dm) {
    WGRAD_EARLY_APPLY(g, g0);
    __shared__ __attribute__((aligned(16))) float red[4 * WG_BUF];
    int bid = (int)blockIdx.x;
    if (bid < n_dm) {
        if (sd.fused == 2) star_pnb_dom_body(sd, bid);      // the partials themselves (no k_star_pnb_apply before this launch)
        else star_dm_final_body<16>(sd, bid, red);
        return;
    }
    bid -= n_dm;
    if (bid < n_wgrad) {
        wgrad_body(g, bid, red);
        return;
    }
    const int nb = (e.rows + 7) / 8, idx = bid - n_wgrad;
    if (idx < 2 * nb) emb_reduce_body(e, idx % nb, idx / nb, reinterpret_cast<uint16_t(*)[RED_CAP]>(red));
    else emb_rows_body(nr, idx - 2 * nb);
}
#ifdef MAMDR_WGRAD8     // (a rejected experiment, profiles/r05_ab_wgrad_pairs.txt: in diagnostic builds only, tools/build_variant.sh x -DMAMDR_WGRAD8)
// ---- k_wgrad8 (round 5): one workgroup of EIGHT waves = one tile x TWO adjacent row groups.  Waves 0..3 run row group
// 2 q, waves 4..7 row group 2 q + 1, each half with its own staging buffers; the two partial tiles are added through LDS
// (group 2 q first) and ONE slab per pair is written: half the slab bytes for k_update -- which is bound by what it pulls
// over the fabric -- with the same eight waves per CU that two co-resident four-wave workgroups give.  (Fewer, longer
// four-wave workgroups lose that: profiles/r05_ab_groups_taobao30.txt.)  Frozen-table slab path only.
__device__ __forceinline__ void wgrad8_big(const WgradArgs& g, const TileDesc& t, int gb0, int gb1, int n_chunks, float* lds,
                                            float* xch, int pair, int half) {
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int c = lane & 31, kk = lane >> 5;
    const int lr = tid >> 4, lc4 = (tid & 15) * 4;
    f32x4 ra[WG_PF][2], rb[WG_PF][2];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int slot, int ch) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = gb0 + ch * WG_KC + lr + 16 * h;
            const bool ok = ch < n_chunks && b < gb1;
            ra[slot][h] = ok ? *reinterpret_cast<const f32x4*>(g.acts + (size_t)b * ACT_LD + t.a_off + lc4) : zero4;
            rb[slot][h] = ok ? *reinterpret_cast<const f32x4*>(g.dz + (size_t)b * DZ_LD + t.b_off + lc4) : zero4;
        }
    };
#pragma unroll
    for (int s = 0; s < WG_PF; ++s) issue(s, s);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // (n_chunks is the same for both halves -- a short or missing last group stages zeros -- so that the barriers pair up)
    for (int ch0 = 0; ch0 < n_chunks; ch0 += WG_PF) {
#pragma unroll
        for (int u = 0; u < WG_PF; ++u) {
            const int ch = ch0 + u;
            if (ch >= n_chunks) break;                 // uniform
            float* As = lds + (ch & 1) * 2 * WG_BUF;
            float* Bs = As + WG_BUF;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<f32x4*>(As + (lr + 16 * h) * WG_LD + lc4) = ra[u][h];
                *reinterpret_cast<f32x4*>(Bs + (lr + 16 * h) * WG_LD + lc4) = rb[u][h];
            }
            __syncthreads();
            issue(u, ch + WG_PF);
            __builtin_amdgcn_sched_barrier(0);
            const float* ap = As + kk * WG_LD + wm * 32 + c;
            const float* bp = Bs + kk * WG_LD + wn * 32 + c;
#pragma unroll
            for (int i = 0; i < WG_KC / 2; ++i) acc = MAMDR_MFMA32(ap[2 * i * WG_LD], bp[2 * i * WG_LD], acc);
        }
    }
    __syncthreads();                // every wave is past its last staged chunk: the exchange area may be written
    if (half == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[r * 256 + tid] = acc[r];
    }
    __syncthreads();
    if (half == 0) {
        float* dst = g.slabs + (size_t)pair * g.slab_ld + t.dst_off + (size_t)(wm * 32) * t.dst_ld + wn * 32;
        const int rb4 = 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            WS_STORE1(&dst[(size_t)((r & 3) + 8 * (r >> 2) + rb4) * t.dst_ld + c], acc[r] + xch[r * 256 + tid]);
    }
}

// lds: 2 x 4 WG_BUF floats (96 KB, dynamic)
__device__ __forceinline__ void wgrad8_body(const WgradArgs& g, const int bid, float* lds) {
    const int n_pairs = (g.n_groups + 1) / 2;
    const int n_work = g.n_tiles * n_pairs;
    if (bid >= n_work) {
        // the loss workgroup and the snapshot workgroups: the first four waves, as in k_wgrad (same indices behind n_work)
        if (threadIdx.x >= 256) return;
        WgradArgs g1 = g;
        g1.n_groups = n_pairs;       // (wgrad_body derives n_work = n_tiles * n_groups: the same boundary)
        wgrad_body(g1, bid, lds);
        return;
    }
    const int half = (int)threadIdx.x >> 8;
    const int pair = bid % n_pairs, tile = bid / n_pairs;
    const TileDesc t = g.tiles[tile];
    const int grp = 2 * pair + half;
    const int gb0 = min(grp * g.rows_per_group, g.rows_pad);
    const int gb1 = grp < g.n_groups ? min(gb0 + g.rows_per_group, g.rows_pad) : gb0;
    float* xch = lds + 4 * WG_BUF;          // (the second half's staging area, free once its chunks are consumed)
    if (t.big) {
        wgrad8_big(g, t, gb0, gb1, (g.rows_per_group + WG_KC - 1) / WG_KC, lds + half * 4 * WG_BUF, xch, pair, half);
        return;
    }
    // small tiles: the pair's rows [first group's start, second group's end) over the 8 waves in multiples of 2
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int p0 = min(2 * pair * g.rows_per_group, g.rows_pad), p1 = min(p0 + 2 * g.rows_per_group, g.rows_pad);
    const int span = p1 > p0 ? p1 - p0 : 0;
    const int per = ((span + 15) / 16) * 2;
    const int b0 = min(p0 + w * per, p1), b1 = min(b0 + per, p1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (t.a_kind == 0 && t.b_kind == 0) wgrad_rows<0, 0>(g, t, b0, b1, acc);
    else if (t.a_kind == 1 && t.b_kind == 0) wgrad_rows<1, 0>(g, t, b0, b1, acc);
    else if (t.a_kind == 0 && t.b_kind == 1) wgrad_rows<0, 1>(g, t, b0, b1, acc);
    else if (t.a_kind == 1 && t.b_kind == 1) wgrad_rows<1, 1>(g, t, b0, b1, acc);
    else if (t.a_kind == 3) wgrad_rows<3, 0>(g, t, b0, b1, acc);
    else if (t.b_kind == 0) wgrad_rows<2, 0>(g, t, b0, b1, acc);
    else if (t.b_kind == 1) wgrad_rows<2, 1>(g, t, b0, b1, acc);
    else wgrad_rows<2, 2>(g, t, b0, b1, acc);
    {
        const int col = lane & 31, rb = 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[w * 1024 + ((r & 3) + 8 * (r >> 2) + rb) * 32 + col] = acc[r];
    }
    __syncthreads();
    float* slab = g.slabs + (size_t)pair * g.slab_ld;
    for (int e = tid; e < 1024; e += 512) {
        const int row = e >> 5, col = e & 31;
        if (row < t.m_valid && col < t.n_valid) {
            float v = lds[e];
#pragma unroll
            for (int ww = 1; ww < 8; ++ww) v += lds[ww * 1024 + e];
            WS_STORE1(&slab[t.dst_off + row * t.dst_ld + col], v);
        }
    }
}
static int wgrad8_blocks(const WgradArgs& a) {
    const int dm_wgs = a.dm_copy ? (a.dm_count / 4 + 255) / 256 : 0;
    return a.n_tiles * ((a.n_groups + 1) / 2) + 1 + W0DOM_COPY_WGS + dm_wgs;
}
#endif
static int wgrad_blocks(const WgradArgs& a) {
    const int dm_wgs = a.dm_copy ? (a.dm_count / 4 + 255) / 256 : 0;
    return a.n_tiles * a.n_groups + 1 + W0DOM_COPY_WGS + dm_wgs;
}
void launch_wgrad_reduce(const WgradArgs& a, const EmbStepArgs& e, const EmbRowsArgs* next_rows, const StarPnBwdArgs* star_dm,
                         hipStream_t s) {
    const int n_wgrad = wgrad_blocks(a);
    EmbRowsArgs nr;
    memset(&nr, 0, sizeof(nr));
    if (next_rows) nr = *next_rows;
    const int n_rows = next_rows ? (nr.rows_pad + 255) / 256 : 0;
    StarPnBwdArgs sd;
    memset(&sd, 0, sizeof(sd));
    if (star_dm) sd = *star_dm;
    const int n_dm = star_dm ? (sd.fused == 2 ? (sd.n_chunks + 1) / 2 : EMB / 16) : 0;
    MAMDR_LAUNCH(k_wgrad_reduce, dim3(n_dm + n_wgrad + 2 * ((e.rows + 7) / 8) + n_rows), dim3(256), 0, s, WGRAD_EARLY_ARGS(a), a, e, n_wgrad,
                       nr, n_rows, sd, n_dm);
}
__global__ void k_wgrad_pf(WGRAD_EARLY_PARAMS, const WgradArgs g0, const GatherPf pf, const int n_wgrad, const int n_pad);
#ifdef MAMDR_WGRAD8
__global__ void k_wgrad8(WGRAD_EARLY_PARAMS, const WgradArgs g0, const GatherPf pf, const int n_wgrad, const int n_pad);
// pairs of row groups per workgroup: ONE slab per pair (k_update then sums (n_groups + 1) / 2 slabs)
bool launch_wgrad_pairs(const WgradArgs& a, hipStream_t s, const GatherPf* pf) {
    const size_t lds = (size_t)8 * WG_BUF * sizeof(float);
    static const bool raised = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad8), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(8 * WG_BUF * sizeof(float))) == hipSuccess;
    if (!raised) return false;
    GatherPf none;
    memset(&none, 0, sizeof(none));
    const GatherPf& p = (pf && pf->n_tiles > 0) ? *pf : none;
    const int n_wgrad = wgrad8_blocks(a);
    const int n_pad = (n_wgrad + 7) / 8 * 8, n_riders = p.n_tiles > 0 ? ((p.n_tiles + 7) / 8 + 3) / 4 * 8 : 0;
    MAMDR_LAUNCH(k_wgrad8, dim3(n_riders ? n_pad + n_riders : n_wgrad), dim3(512), lds, s, WGRAD_EARLY_ARGS(a), a, p, n_wgrad, n_pad);
    return true;
}
#else
bool launch_wgrad_pairs(const WgradArgs&, hipStream_t, const GatherPf*) { return false; }
#endif
void launch_wgrad(const WgradArgs& a, hipStream_t s, const GatherPf* pf) {
    if (pf && pf->n_tiles > 0) {        // (the riders of the next step's gather in THIS launch: default since round 5)
        const int n_wgrad = wgrad_blocks(a);
        const int n_pad = (n_wgrad + 7) / 8 * 8, n_riders = ((pf->n_tiles + 7) / 8 + 3) / 4 * 8;
        MAMDR_LAUNCH(k_wgrad_pf, dim3(n_pad + n_riders), dim3(256), 0, s, WGRAD_EARLY_ARGS(a), a, *pf, n_wgrad, n_pad);
        return;
    }
    MAMDR_LAUNCH(k_wgrad, dim3(wgrad_blocks(a)), dim3(256), 0, s, WGRAD_EARLY_ARGS(a), a);
}

// sum of one float4 / float over the gradient slabs IN SLAB ORDER.  Round 5: ALL of up to 16 slabs' loads in flight at once (the
// slabs were written a moment ago by k_wgrad on other XCDs: every dependent batch of loads is a ~2 K-cycle trip to the
// infinity cache -- the timeline of k_update showed 5 K cycles for the two batches of eight; round 2's rolled loop paid one
// trip per slab); the additions run in slab order as before (bit-identical)
__device__ __forceinline__ f32x4 slab_sum4(const float* slabs, int n_groups, int slab_ld, size_t e) {
    // (float4: eight slabs' loads in flight -- sixteen would cost the kernel half its occupancy: 113 instead of 62 VGPRs, and the
    // kernel is bound by how many requests the chip keeps in flight, not by this thread's trips: measured on Amazon-6)
    f32x4 g = *reinterpret_cast<const f32x4*>(slabs + e);
    for (int s0 = 1; s0 < n_groups; s0 += 8) {
        f32x4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            t[k] = *reinterpret_cast<const f32x4*>(slabs + (size_t)min(s0 + k, n_groups - 1) * slab_ld + e);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + k < n_groups) g += t[k];
    }
    return g;
}
__device__ __forceinline__ float slab_sum1(const float* slabs, int n_groups, int slab_ld, size_t e) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < n_groups) t[k] = slabs[(size_t)k * slab_ld + e];
    float g = t[0];
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (k < n_groups) g += t[k];
    for (int s0 = 16; s0 < n_groups; ++s0) g += slabs[(size_t)s0 * slab_ld + e];
    return g;
}

// ------------------------------------------------------------------ slab reduce + optimiser
// TF1 ApplyAdam: m += (g - m)(1-b1); v += (g^2 - v)(1-b2);
// p -= (m * alpha) / (sqrt(v) + eps), alpha = lr sqrt(1-b2^t)/(1-b1^t) from the host.
// optimizer 2 = accumulate only: `m` points at the meta-gradient accumulator, p and v are untouched
__device__ __forceinline__ void optimizer_step(const UpdateArgs& u, float g, float& p, float& m, float& v) {
    if (u.optimizer == 0) {
        m = m + (g - m) * u.omb1;
        v = v + (g * g - v) * u.omb2;
        p = p - (m * u.alpha) / (sqrtf(v) + u.eps);
    } else if (u.optimizer == 1) {
        p = p - g * u.alpha;
    } else {
        m = m + g;
    }
}

// four consecutive elements e..e+3 of the dense block (e a multiple of 4) with gradient sum gsum: the linear
// domain table's regulariser, the optimiser, and the transposed copies k_tower4's backward layers read
__device__ __forceinline__ void apply_vec4(const UpdateArgs& u, size_t e, f32x4 gsum, f32x4 p, f32x4 m, f32x4 v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float pc = p[c], mc = m[c], vc = v[c];
        const int ec = (int)e + c;
        if (ec >= u.ld_off && ec < u.ld_off + u.ld_count) gsum[c] += u.two_l2_lin * pc;
        optimizer_step(u, gsum[c], pc, mc, vc);
        p[c] = pc;
        m[c] = mc;
        v[c] = vc;
    }
    if (u.optimizer == 2) {
        *reinterpret_cast<f32x4*>(u.m + e) = m;
        return;
    }
    if (u.optimizer == 0) {
        *reinterpret_cast<f32x4*>(u.m + e) = m;
        *reinterpret_cast<f32x4*>(u.v + e) = v;
    }
    *reinterpret_cast<f32x4*>(u.p + e) = p;
    if (u.wT) {     // keep the transposed copies used by k_tower4's backward layers current
        const int e0 = (int)e;
        if (e0 >= u.w1_off && e0 < u.w1_off + H1 * H2) {
            const int f = e0 - u.w1_off, r = f / H2, c = f - r * H2;
#pragma unroll
            for (int k = 0; k < 4; ++k) u.wT[W1T_OFF + (c + k) * H1 + r] = p[k];
        } else if (e0 >= u.w2_off && e0 < u.w2_off + H2 * H3) {
            const int f = e0 - u.w2_off, r = f / H3, c = f - r * H3;
#pragma unroll
            for (int k = 0; k < 4; ++k) u.wT[W2T_OFF + (c + k) * H2 + r] = p[k];
        } else if (u.w0t && e0 >= u.w0_off && e0 < u.w0_off + 2 * EMB * H1) {
            const int f = e0 - u.w0_off, r = f / H1, c = f - r * H1;      // W0[r][c], user | item rows
#pragma unroll
            for (int k = 0; k < 4; ++k) u.wT[W0T_OFF + (c + k) * (2 * EMB) + r] = p[k];
        }
    }
}

__device__ __forceinline__ void apply_vec4(const UpdateArgs& u, size_t e, f32x4 gsum) {
    apply_vec4(u, e, gsum, *reinterpret_cast<const f32x4*>(u.p + e), *reinterpret_cast<const f32x4*>(u.m + e),
               *reinterpret_cast<const f32x4*>(u.v + e));
}

// dW0[256:384, :] by linearity.  Those rows of x are the domain-embedding row of the sample's domain, the same
// vector for every sample of a domain, so  sum_b x[b][256 + r] dz1[b][c] = sum_d Dm[d][r] S[d][c]  with
// S = onehot(domain)^T dz1 -- which k_wgrad computes anyway for the domain-table gradient.  8 of the 34 64x64
// tiles of k_wgrad (24 % of its MFMA work) become D fmas per element here.  One workgroup per 8 columns:
// S[:, 8 columns] is summed over the slabs into LDS once, thread (r, half) then owns W0[256 + r][c0 + 4 half .. +3].
constexpr int W0LIN_COLS = 8;
constexpr int W0LIN_WGS = H1 / W0LIN_COLS;      // 32
__device__ __forceinline__ void update_w0dom_linear(const UpdateArgs& u, int wg, float* s_l) {
    const int tid = threadIdx.x, c0 = wg * W0LIN_COLS;
    const int r = tid >> 1, half = tid & 1;
    // this thread's parameters and slots, and the first eight domains' embedding values of row r, are requested
    // before the S block is reduced
    const size_t e = (size_t)u.w0_off + (size_t)(2 * EMB + r) * H1 + c0 + 4 * half;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(u.p + e);
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(u.m + e);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(u.v + e);
    // (round 5: the first 32 domains' values of this thread's table column are requested up front, beside the S block's slab
    // loads -- the 8-at-a-time prefetch paid one trip to the infinity cache per eight domains: 12.2 K cycles per workgroup)
    float x[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k] = u.dm_copy[min(k, u.n_domain - 1) * EMB + r];
    for (int idx = tid; idx < u.n_domain * W0LIN_COLS; idx += 256) {
        const int d = idx / W0LIN_COLS, cc = idx - d * W0LIN_COLS;
        s_l[idx] = slab_sum1(u.slabs, u.n_groups, u.slab_ld, (size_t)u.s_off + (size_t)d * H1 + c0 + cc);
    }
    __syncthreads();
    f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k < u.n_domain) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + k * W0LIN_COLS + 4 * half);
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = fmaf(x[k], sv[q], g[q]);
        }
    }
    for (int d = 32; d < u.n_domain; ++d) {              // (33 .. 64 domains)
        const float xd = u.dm_copy[d * EMB + r];
        const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + d * W0LIN_COLS + 4 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = fmaf(xd, sv[q], g[q]);
    }
    apply_vec4(u, e, g, p0, m0, v0);
}

#ifdef MAMDR_STAMPS
#define USTAMP(k)                                                                             \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (u.stamps && threadIdx.x == 0 && bx < 1024) u.stamps[bx * 4 + (k)] = t_;            \
    } while (0)
#else
#define USTAMP(k) do { } while (0)
#endif

constexpr int DM_CBLOCKS = 8;       // column blocks of the domain table's update: 16 columns per workgroup
__device__ __forceinline__ void update_body(const UpdateArgs& u, const int bx, float* s_l) {
    USTAMP(0);
    const int n_vec_wgs = (u.count4 - u.dm_count / 4 + 255) / 256;
    // the workgroups with the longest dependent chain come first in the grid
    const int n_lin_wgs = u.dm_copy ? W0LIN_WGS : 0;
    if (bx < n_lin_wgs) {
        update_w0dom_linear(u, bx, s_l);
        USTAMP(2);
        return;
    }
    const int bid = bx - n_lin_wgs;
    if (bid < n_vec_wgs) {
        // dense weights behind the domain table: float4 per thread
        const int e4 = u.dm_count / 4 + bid * 256 + threadIdx.x;
        if (e4 >= u.count4) return;
        const size_t e = (size_t)e4 * 4;
        // (rows 256..383 of W0 have no tiles when their gradient comes from S: update_w0dom_linear)
        if (u.dm_copy && (int)e >= u.w0_off + 2 * EMB * H1 && (int)e < u.w0_off + XDIM * H1) return;
        // (parameters and slots are requested before the slab sum is waited for: one round of misses, not two)
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(u.p + e);
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(u.m + e);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(u.v + e);
        const f32x4 gs = slab_sum4(u.slabs, u.n_groups, u.slab_ld, e);
        USTAMP(1);
        apply_vec4(u, e, gs, p0, m0, v0);
        USTAMP(2);
        return;
    }
    // domain table, one workgroup per (domain d, 16 columns c):
    //   g[d][c] = sum_k S[d][k] * W0[256 + c][k] + 2 l2 p,   S = onehot(domain)^T dz1 summed over the slabs
    // S[d][:] is summed over the slabs ONCE per workgroup (thread k owns element k; LDS), then every wave contracts it with
    // four rows of the W0 snapshot.  (One wave per element re-summed the 16 slabs of S[d][:] for each of its 128 columns:
    // 960 workgroups x 16 KB on Taobao-30.)  Same orders as that form -- slabs in sequence, four fmas per lane, the
    // xor tree over the lanes: bit-identical.
    const int blk = bid - n_vec_wgs;
    const int d = blk / DM_CBLOCKS, c0 = (blk - d * DM_CBLOCKS) * (EMB / DM_CBLOCKS);
    if (d >= u.dm_count / EMB) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int PER_WAVE = EMB / DM_CBLOCKS / 4;
    // (round 5: the wave's four elements' parameters, slots, W0 snapshot rows and S2 sums are requested BEFORE the S row's slab
    // sums are waited for -- one trip instead of five; round 4 tried the same while the 32 linearity workgroups were the
    // kernel's long pole and saw no change)
    float pe[PER_WAVE], me[PER_WAVE], ve[PER_WAVE];
    f32x4 wve[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int c = c0 + w * PER_WAVE + i, el = d * EMB + c;
        pe[i] = u.p[el];
        me[i] = u.m[el];
        ve[i] = u.v[el];
        wve[i] = *reinterpret_cast<const f32x4*>(u.w0dom_copy + (size_t)c * H1 + 4 * lane);
    }
    s_l[threadIdx.x] = slab_sum1(u.slabs, u.n_groups, u.slab_ld, (size_t)u.s_off + (size_t)d * H1 + threadIdx.x);
    __syncthreads();
    USTAMP(1);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + 4 * lane);
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int c = c0 + w * PER_WAVE + i, el = d * EMB + c;
        float p = pe[i], m = me[i], v = ve[i];
        const f32x4 wv = wve[i];
        // (DeepFM's S2 sums stay in the loop: hoisted with the rest they made k_update_lin slower on Amazon-6, 7.55 -> 8.05 us)
        const float g2 = u.s2_off ? slab_sum1(u.slabs, u.n_groups, u.slab_ld, (size_t)u.s2_off + el) : 0.f;
        float g = fmaf(sv[3], wv[3], fmaf(sv[2], wv[2], fmaf(sv[1], wv[1], sv[0] * wv[0])));
        for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o);
        if (lane == 0) {
            if (u.no_sdm) g = 0.f;
            if (u.s2_off) g += g2;
            g += u.two_l2 * p;
            optimizer_step(u, g, p, m, v);
            if (u.optimizer == 2) {
                u.m[el] = m;
                continue;
            }
            if (u.optimizer == 0) {
                u.m[el] = m;
                u.v[el] = v;
            }
            u.p[el] = p;
        }
    }
    USTAMP(2);
}

// ---- the NARROW forms (rounds 2 - 4; 62 VGPRs, eight waves per SIMD): up to 8 row groups, k_update_lin.  Sum of one float4 / float over the gradient slabs IN SLAB ORDER, eight slabs' loads in flight (the rolled
// loop paid one dependent round trip per slab: 4 at 1024 rows, 16 at 4096)
__device__ __forceinline__ f32x4 slab_sum4_n(const float* slabs, int n_groups, int slab_ld, size_t e) {
    f32x4 g = *reinterpret_cast<const f32x4*>(slabs + e);
    for (int s0 = 1; s0 < n_groups; s0 += 8) {
        f32x4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            t[k] = *reinterpret_cast<const f32x4*>(slabs + (size_t)min(s0 + k, n_groups - 1) * slab_ld + e);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + k < n_groups) g += t[k];
    }
    return g;
}
__device__ __forceinline__ float slab_sum1_n(const float* slabs, int n_groups, int slab_ld, size_t e) {
    float g = slabs[e];
    for (int s0 = 1; s0 < n_groups; s0 += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = slabs[(size_t)min(s0 + k, n_groups - 1) * slab_ld + e];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + k < n_groups) g += t[k];
    }
    return g;
}

// dW0[256:384, :] by linearity.  Those rows of x are the domain-embedding row of the sample's domain, the same
// vector for every sample of a domain, so  sum_b x[b][256 + r] dz1[b][c] = sum_d Dm[d][r] S[d][c]  with
// S = onehot(domain)^T dz1 -- which k_wgrad computes anyway for the domain-table gradient.  8 of the 34 64x64
// tiles of k_wgrad (24 % of its MFMA work) become D fmas per element here.  One workgroup per 8 columns:
// S[:, 8 columns] is summed over the slabs into LDS once, thread (r, half) then owns W0[256 + r][c0 + 4 half .. +3].
__device__ __forceinline__ void update_w0dom_linear_n(const UpdateArgs& u, int wg, float* s_l) {
    const int tid = threadIdx.x, c0 = wg * W0LIN_COLS;
    const int r = tid >> 1, half = tid & 1;
    // this thread's parameters and slots, and the first eight domains' embedding values of row r, are requested
    // before the S block is reduced
    const size_t e = (size_t)u.w0_off + (size_t)(2 * EMB + r) * H1 + c0 + 4 * half;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(u.p + e);
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(u.m + e);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(u.v + e);
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = u.dm_copy[min(k, u.n_domain - 1) * EMB + r];
    for (int idx = tid; idx < u.n_domain * W0LIN_COLS; idx += 256) {
        const int d = idx / W0LIN_COLS, cc = idx - d * W0LIN_COLS;
        s_l[idx] = slab_sum1_n(u.slabs, u.n_groups, u.slab_ld, (size_t)u.s_off + (size_t)d * H1 + c0 + cc);
    }
    __syncthreads();
    f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < u.n_domain; d0 += 8) {
        float xn[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xn[k] = u.dm_copy[min(d0 + 8 + k, u.n_domain - 1) * EMB + r];   // next eight
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (d0 + k < u.n_domain) {
                const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + (d0 + k) * W0LIN_COLS + 4 * half);
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = fmaf(x[k], sv[q], g[q]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = xn[k];
    }
    apply_vec4(u, e, g, p0, m0, v0);
}

__device__ __forceinline__ void update_body_n(const UpdateArgs& u, const int bx, float* s_l) {
    const int n_vec_wgs = (u.count4 - u.dm_count / 4 + 255) / 256;
    // the workgroups with the longest dependent chain come first in the grid
    const int n_lin_wgs = u.dm_copy ? W0LIN_WGS : 0;
    if (bx < n_lin_wgs) {
        update_w0dom_linear_n(u, bx, s_l);
        return;
    }
    const int bid = bx - n_lin_wgs;
    if (bid < n_vec_wgs) {
        // dense weights behind the domain table: float4 per thread
        const int e4 = u.dm_count / 4 + bid * 256 + threadIdx.x;
        if (e4 >= u.count4) return;
        const size_t e = (size_t)e4 * 4;
        // (rows 256..383 of W0 have no tiles when their gradient comes from S: update_w0dom_linear_n)
        if (u.dm_copy && (int)e >= u.w0_off + 2 * EMB * H1 && (int)e < u.w0_off + XDIM * H1) return;
        // (parameters and slots are requested before the slab sum is waited for: one round of misses, not two)
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(u.p + e);
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(u.m + e);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(u.v + e);
        apply_vec4(u, e, slab_sum4_n(u.slabs, u.n_groups, u.slab_ld, e), p0, m0, v0);
        return;
    }
    // domain table, one workgroup per (domain d, 16 columns c):
    //   g[d][c] = sum_k S[d][k] * W0[256 + c][k] + 2 l2 p,   S = onehot(domain)^T dz1 summed over the slabs
    // S[d][:] is summed over the slabs ONCE per workgroup (thread k owns element k; LDS), then every wave contracts it with
    // four rows of the W0 snapshot.  (One wave per element re-summed the 16 slabs of S[d][:] for each of its 128 columns:
    // 960 workgroups x 16 KB on Taobao-30.)  Same orders as that form -- slabs in sequence, four fmas per lane, the
    // xor tree over the lanes: bit-identical.
    const int blk = bid - n_vec_wgs;
    const int d = blk / DM_CBLOCKS, c0 = (blk - d * DM_CBLOCKS) * (EMB / DM_CBLOCKS);
    if (d >= u.dm_count / EMB) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    s_l[threadIdx.x] = slab_sum1_n(u.slabs, u.n_groups, u.slab_ld, (size_t)u.s_off + (size_t)d * H1 + threadIdx.x);
    __syncthreads();
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + 4 * lane);
    constexpr int PER_WAVE = EMB / DM_CBLOCKS / 4;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int c = c0 + w * PER_WAVE + i, el = d * EMB + c;
        float p = u.p[el], m = u.m[el], v = u.v[el];
        const f32x4 wv = *reinterpret_cast<const f32x4*>(u.w0dom_copy + (size_t)c * H1 + 4 * lane);
        const float g2 = u.s2_off ? slab_sum1_n(u.slabs, u.n_groups, u.slab_ld, (size_t)u.s2_off + el) : 0.f;
        float g = fmaf(sv[3], wv[3], fmaf(sv[2], wv[2], fmaf(sv[1], wv[1], sv[0] * wv[0])));
        for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o);
        if (lane == 0) {
            if (u.no_sdm) g = 0.f;
            if (u.s2_off) g += g2;
            g += u.two_l2 * p;
            optimizer_step(u, g, p, m, v);
            if (u.optimizer == 2) {
                u.m[el] = m;
                continue;
            }
            if (u.optimizer == 0) {
                u.m[el] = m;
                u.v[el] = v;
            }
            u.p[el] = p;
        }
    }
}
#define UPDATE_EARLY_PARAMS                                                                                            \
    float *__restrict__ k_p, float *__restrict__ k_m, float *__restrict__ k_v, const float *__restrict__ k_slabs,      \
        const int k_n_groups, const int k_slab_ld, const int k_count4, const int k_dm_count, const int k_optimizer
#define UPDATE_EARLY_ARGS(a) (a).p, (a).m, (a).v, (a).slabs, (a).n_groups, (a).slab_ld, (a).count4, (a).dm_count, (a).optimizer
#define UPDATE_EARLY_APPLY(u, u0)                                                                                      \
    UpdateArgs u = u0;                                                                                                 \
    u.p = k_p; u.m = k_m; u.v = k_v; u.slabs = k_slabs; u.n_groups = k_n_groups; u.slab_ld = k_slab_ld;               \
    u.count4 = k_count4; u.dm_count = k_dm_count; u.optimizer = k_optimizer
// WIDE (round 5): the forms that keep every operand of a workgroup in flight at once -- 113 VGPRs, four waves per SIMD; they
// pay where a step has more than 8 row groups (4,096-row batches: k_update 7.5 -> 7.0 us) and cost where it has 4 and the
// launch hosts many other workgroups (Amazon-6's k_update_lin: 7.55 -> 8.0+ us with them) -- an instance of its own
template <bool WIDE>
__global__ __launch_bounds__(256) void k_update(UPDATE_EARLY_PARAMS, const UpdateArgs u0) {
    __shared__ __attribute__((aligned(16))) float s_l[64 * W0LIN_COLS];     // n_domain <= 64
    UPDATE_EARLY_APPLY(u, u0);
    if constexpr (WIDE) update_body(u, (int)blockIdx.x, s_l);
    else update_body_n(u, (int)blockIdx.x, s_l);
}
// rider workgroup rb (4 waves = 4 tiles of the next step's tower, all = rb mod 8): see GatherPf
__device__ __forceinline__ void gather_prefetch_body(const GatherPf& p, const int rb) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = (rb & 7) + 8 * (4 * (rb >> 3) + w);
    if (tile >= p.n_tiles) return;
    const int r = lane & (TILE_ROWS - 1), part = lane >> 4;       // 16 rows x (user | item) x two 256-B halves
    const int rr = min(tile * TILE_ROWS + r, max(p.rows - 1, 0));
    int64_t src = p.row_base + rr;
    if (p.perm) src = p.perm[src];
    src = src < 0 ? 0 : (src >= p.n_rows_split ? p.n_rows_split - 1 : src);
    const int u = clampi(p.uid[src], 0, p.n_user - 1), it = clampi(p.pid[src], 0, p.n_item - 1);
    const float t0 = (float)p.dom[src] + p.label[src];
    const float* base = (part < 2 ? p.user_tab + (size_t)u * EMB : p.item_tab + (size_t)it * EMB) + (part & 1) * (EMB / 2);
    const float t1 = base[0] + base[32];                          // one word per 128-B line
    if (t0 + t1 == 1.2345678e30f) p.sink[0] = t1;                 // (keeps the loads)
}
__global__ __launch_bounds__(256) void k_wgrad_pf(WGRAD_EARLY_PARAMS, const WgradArgs g0, const GatherPf pf, const int n_wgrad, const int n_pad) {
    __shared__ __attribute__((aligned(16))) float red[4 * WG_BUF];
    WGRAD_EARLY_APPLY(g, g0);
    const int bid = (int)blockIdx.x;
    if (bid < n_wgrad) wgrad_body(g, bid, red);
    else if (bid >= n_pad) gather_prefetch_body(pf, bid - n_pad);
}
#ifdef MAMDR_WGRAD8
__global__ __launch_bounds__(512) void k_wgrad8(WGRAD_EARLY_PARAMS, const WgradArgs g0, const GatherPf pf, const int n_wgrad, const int n_pad) {
    extern __shared__ __attribute__((aligned(16))) float lds8[];
    WGRAD_EARLY_APPLY(g, g0);
    const int bid = (int)blockIdx.x;
    if (bid < n_wgrad) wgrad8_body(g, bid, lds8);
    else if (bid >= n_pad && threadIdx.x < 256) gather_prefetch_body(pf, bid - n_pad);
}
#endif
template <bool WIDE>
__global__ __launch_bounds__(256) void k_update_pf(UPDATE_EARLY_PARAMS, const UpdateArgs u0, const GatherPf pf, const int n_update,
                                                   const int n_pad) {
    __shared__ __attribute__((aligned(16))) float s_l[64 * W0LIN_COLS];
    UPDATE_EARLY_APPLY(u, u0);
    const int bid = (int)blockIdx.x;
    if (bid < n_update) {
        if constexpr (WIDE) update_body(u, bid, s_l);
        else update_body_n(u, bid, s_l);
    }
    else if (bid >= n_pad) gather_prefetch_body(pf, bid - n_pad);
}
// k_update and DeepFM's k_lin_sweep touch disjoint state: one launch
// ... and so does the NEXT step's k_emb_catchup (n_cu workgroups per table; its rows were resolved in the previous
// launch, k_wgrad_reduce): the rows of the next batch are brought up to this step while the dense block steps
__global__ __launch_bounds__(256) void k_update_lin(UPDATE_EARLY_PARAMS, const UpdateArgs u0, const EmbStepArgs e,
                                                    const int n_update, const int n_lin, const EmbStepArgs nc, const int n_cu) {
    __shared__ __attribute__((aligned(16))) float s_l[64 * W0LIN_COLS];
    UPDATE_EARLY_APPLY(u, u0);
    const int bid = (int)blockIdx.x;
    if (bid < 2 * n_cu) emb_catchup_body(nc, bid % n_cu, bid / n_cu);
    else if (bid < 2 * n_cu + n_update) update_body_n(u, bid - 2 * n_cu, s_l);
    else lin_sweep_body(e, bid - 2 * n_cu - n_update, n_lin);
}
static int update_blocks(const UpdateArgs& a) {
    const int n_vec_wgs = (a.count4 - a.dm_count / 4 + 255) / 256;
    return n_vec_wgs + (a.dm_count / EMB) * DM_CBLOCKS + (a.dm_copy ? W0LIN_WGS : 0);
}
void launch_update(const UpdateArgs& a, hipStream_t s, const GatherPf* pf) {
    const int n_update = update_blocks(a);
    if (pf && pf->n_tiles > 0) {
        // riders behind the update's workgroups, at block ids that are = their tiles mod 8 (the XCD of block b is b mod 8)
        const int n_pad = (n_update + 7) / 8 * 8, n_riders = ((pf->n_tiles + 7) / 8 + 3) / 4 * 8;
        if (a.n_groups > 8)
            MAMDR_LAUNCH(k_update_pf<true>, dim3(n_pad + n_riders), dim3(256), 0, s, UPDATE_EARLY_ARGS(a), a, *pf, n_update, n_pad);
        else
            MAMDR_LAUNCH(k_update_pf<false>, dim3(n_pad + n_riders), dim3(256), 0, s, UPDATE_EARLY_ARGS(a), a, *pf, n_update, n_pad);
        return;
    }
    if (a.n_groups > 8) MAMDR_LAUNCH(k_update<true>, dim3(n_update), dim3(256), 0, s, UPDATE_EARLY_ARGS(a), a);
    else MAMDR_LAUNCH(k_update<false>, dim3(n_update), dim3(256), 0, s, UPDATE_EARLY_ARGS(a), a);
}
void launch_update_lin(const UpdateArgs& a, const EmbStepArgs& e, bool lin, const EmbStepArgs* next_catchup, hipStream_t s) {
    const int64_t n_all = e.t[0].n_rows + e.t[1].n_rows;
    int64_t n_lin = lin ? (n_all + 255) / 256 : 0;
    if (n_lin > 256 * 8) n_lin = 256 * 8;
    const int n_update = update_blocks(a);
    const EmbStepArgs& nc = next_catchup ? *next_catchup : e;
    const int n_cu = next_catchup ? (nc.rows + 7) / 8 : 0;
    MAMDR_LAUNCH(k_update_lin, dim3(2 * n_cu + n_update + (int)n_lin), dim3(256), 0, s, UPDATE_EARLY_ARGS(a), a, e, n_update,
                 (int)n_lin, nc, n_cu);
}

}  // namespace mamdr
