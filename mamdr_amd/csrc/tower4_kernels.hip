// k_tower4: the training tower for SMALL batches (rows <= 2048) on gfx950.
//
// With 16-row tiles (k_tower, step_kernels.hip) a batch of 1024 rows occupies only 64 of the
// 256 CUs.  This variant gives every workgroup 4 batch rows (256 tiles at 1024 rows) and runs
// the contractions on v_mfma_f32_4x4x1_16b_f32: 16 blocks of (4 rows x 1 k) x (1 k x 4 cols),
// i.e. one instruction = 4 batch rows x 64 output columns x one k at the full fp32 MFMA rate
// (measured 9.5 cycles / instruction, tools/mfma4x4.hip).  Lane l supplies B = W[k][n0 + l]
// (a fully coalesced row of W), the A operand x[row][k] is broadcast from block 0 (cbsz = 4),
// and D register r of lane l is out[row r][n0 + l].
//
// A wave covers ALL columns of a layer with one 16/8/4-byte load per k (col = V * lane + t), so
// the 8 waves split the reduction index k instead and are summed through LDS in wave order
// (fixed order, no atomics).  The backward layers use transposed copies of W1 / W2 kept
// current by k_update, so they stream rows exactly like the forward layers.  With one workgroup per CU
// (grids of up to 256 tiles, W1L) W1 is brought into LDS ONCE by LDS-DMA while the gather's misses are
// outstanding and feeds both layer 1 and its backward contraction (see t4_w1_request).  The dropout
// stream, loss, and workspace layout are identical to k_tower; results differ only by fp32
// summation order.  Replaces the same reference call sites (model_zoo/mamdr.py:54,86,97).
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mamdr_kernels.h"

namespace mamdr {

constexpr int T4_ROWS = 4;
constexpr int T4_THREADS = 512;
constexpr int T4_WAVES = 8;
constexpr int T4_PF = 8;                 // k rows in flight per wave (first layer, input-gradient layer)
#ifndef MAMDR_T4_DEEP
#define MAMDR_T4_DEEP 8
#endif
constexpr int T4_DEEP = MAMDR_T4_DEEP;   // ring depth of the short layers (diagnostic builds may override it)

// LDS map (floats).  dz_l overwrites h_l in place (the gate is read by the thread that writes the gradient); the
// split-k partials of the 256-column contractions share a slot between waves w and w + 4 (t4_put), so that everything
// but the W1 image is 30 KB and the image (128 KB) fits beside it in the CU's 160 KB.
// Row strides of the four activation tiles: + 4 floats, so that the A fragment reads of a contraction -- lane l reads
// 16 B of row l & 3 -- fall into four different bank groups (unpadded, every stride is a multiple of 32 banks and the
// four rows collide: SQ_LDS_BANK_CONFLICT 2.9 K cycles per CU and launch at 1,024 rows).
constexpr int T4_XLD = XDIM + 4, T4_H1LD = H1 + 4, T4_H2LD = H2 + 4, T4_H3LD = H3 + 4;
constexpr int T4_XS = 0;                               // [4][384 + 4]
constexpr int T4_H1 = T4_XS + T4_ROWS * T4_XLD;        // [4][256 + 4]  h1, then dz1 (trainable tables: input of the dx contraction)
constexpr int T4_H2 = T4_H1 + T4_ROWS * T4_H1LD;       // [4][128 + 4]  h2, then dz2
constexpr int T4_DZ3 = T4_H2 + T4_ROWS * T4_H2LD;      // [4][64 + 4]
constexpr int T4_DZ2 = T4_H2;
constexpr int T4_DZ1 = T4_H1;
constexpr int T4_RED = T4_DZ3 + T4_ROWS * T4_H3LD;     // split-k partials: [8 waves][4][N <= 128] or [4 wave pairs][4][256]
constexpr int T4_RED_FLOATS = 4 * T4_ROWS * H1;
constexpr int T4_ROWI = T4_RED + T4_RED_FLOATS;
constexpr int T4_DROW = T4_ROWI + 96;                  // NFM: [4][128] the rows' domain embedding (its tile columns carry the bi-interaction)
constexpr int T4_W1S = T4_DROW + T4_ROWS * EMB;        // W1L: [256][128] image of W1, 16-B chunks XOR-swizzled per row
// (row bookkeeping: 16 ints + 80 floats -- [0,4) label [4,8) loss [8,12) FM + linear term [12,16) dlogit [16,40) per-wave
// partials of the FM term / of PNN's three inner products [48,64) PNN: ip[row][4] [64,80) PNN: d loss / d ip [row][4])
constexpr int T4_LDS_FLOATS = T4_W1S;
constexpr int T4_LDS_FLOATS_W1L = T4_W1S + H1 * H2;
static_assert(T4_LDS_FLOATS_W1L * sizeof(float) <= 160 * 1024, "LDS of a gfx950 CU");
static_assert(T4_WAVES * T4_ROWS * H2 <= T4_RED_FLOATS, "one slot per wave up to 128 columns");

size_t tower4_lds_bytes(bool w1l) { return (w1l ? T4_LDS_FLOATS_W1L : T4_LDS_FLOATS) * sizeof(float); }

// Workspace stores of the activations / gradients: WRITE-THROUGH (agent-scope relaxed atomic store = global_store
// ... sc1).  With plain stores the 3.7 MB a step writes sit dirty in the eight L2s until the kernel ends and are
// written back between this kernel and k_wgrad_adam: 3.6 us from the tower's last workgroup to the next kernel's
// first one, 1.5 us with no stores at all, 2.2 us with write-through stores (tools/stamp_wall.py; -DT4_NT_STORES,
// nontemporal: 2.6 us).  The next kernel reads these rows from other XCDs anyway.
#ifdef T4_ABLATE_STORES        // diagnostic builds
#define T4_WS_STORE(ptr, val) do { } while (0)
#elif defined(T4_NT_STORES)
#define T4_WS_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#elif defined(T4_PLAIN_STORES)
#define T4_WS_STORE(ptr, val) (*(ptr) = (val))
#else
#define T4_WS_STORE(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
#define MAMDR_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, 0, 0)

template <int V> struct Vec4T;
template <> struct Vec4T<4> { typedef f32x4 type; };
template <> struct Vec4T<2> { typedef f32x2 type; };
template <> struct Vec4T<1> { typedef float type; };

template <int V>
__device__ __forceinline__ void t4_load(float (&b)[V], const float* __restrict__ p) {
    typedef typename Vec4T<V>::type T;

    const T v = *reinterpret_cast<const T*>(p);
    if constexpr (V == 1) {
        b[0] = v;
    } else {
#pragma unroll
        for (int t = 0; t < V; ++t) b[t] = v[t];
    }
}

// weight ring of one layer: W is [K][N] row-major, this wave owns k in [w*K/8, (w+1)*K/8).
// DEPTH k rows are in flight per wave.  Stamps (tools/stamp_tower.py): the first layer streams its 393 KB per
// CU at ~60 B/clk, the L1 fill peak; holding the later layers' whole per-wave share in flight (DEPTH 32) cuts
// the L1 / bwd1 contractions from 4.6 K / 2.7 K to 2.1 K / 2.1 K cycles but lengthens the phases the loads
// now overlap by as much -- the 0.72 MB per workgroup is a bandwidth, not a latency, cost -- so 8 stays.
template <int K, int N, int DEPTH = T4_PF>
struct T4W {
    static constexpr int V = N / 64;
    static constexpr int KW = K / T4_WAVES;
    static constexpr int PF = KW < DEPTH ? KW : DEPTH;
#ifdef T4_ABLATE_W1          // diagnostic build: layer 1 / its backward without their weight streams
    static constexpr bool LOADS = !((K == H1 && N == H2) || (K == H2 && N == H1));
#else
    static constexpr bool LOADS = true;
#endif
    float b[PF][V];
    static __device__ __forceinline__ const float* wptr(const float* __restrict__ W) {
        return W + (size_t)((threadIdx.x >> 6) * KW) * N + V * (threadIdx.x & 63);
    }
    __device__ __forceinline__ void prefetch(const float* __restrict__ W) {
        const float* wp = wptr(W);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if constexpr (LOADS) t4_load<V>(b[u], wp + (size_t)u * N);
            else for (int t = 0; t < V; ++t) b[u][t] = 1.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the same ring from the UNtransposed matrix Wn [N][K] (W = Wn^T): b[u][t] = Wn[V lane + t][w KW + u] -- for each of
    // the lane's V columns the wave's whole k range is one run of KW consecutive floats (the ring holds the wave's whole
    // share: KW == PF)
    __device__ __forceinline__ void prefetch_untransposed(const float* __restrict__ Wn) {
        static_assert(KW == PF && KW % 4 == 0, "the wave's whole share in the ring");
        const float* wp = Wn + (size_t)(V * (threadIdx.x & 63)) * K + (threadIdx.x >> 6) * KW;
#pragma unroll
        for (int t = 0; t < V; ++t) {
#pragma unroll
            for (int q = 0; q < KW; q += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(wp + (size_t)t * K + q);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[q + j][t] = v[j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
};

// Split-k partials of a wave -> LDS `red` as [slot][row][column].  D register r of lane l is row r; its column is
// V*l + t (lane-major, one vector store per row) or l + 64 t (STRIDED: the LDS-fed backward contraction).
// N <= 128: slot = wave.  N = 256: waves w and w + 4 share slot w -- waves 4..7 store, barrier, waves 0..3 add their
// own partial (p_w + p_{w+4}) -- and t4_sum adds the four slots in order.
template <int N, int V, bool STRIDED>
__device__ __forceinline__ void t4_put(const f32x4 (&acc)[V], float* red) {
    typedef typename Vec4T<V>::type T;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr bool PAIRED = N > H2;
    float* slot = red + ((PAIRED ? (w & 3) : w) * T4_ROWS) * N;
    if (!PAIRED || w >= 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (STRIDED) {
#pragma unroll
                for (int t = 0; t < V; ++t) slot[r * N + lane + 64 * t] = acc[t][r];
            } else if constexpr (V == 1) {
                slot[r * N + lane] = acc[0][r];
            } else {
                T v;
#pragma unroll
                for (int t = 0; t < V; ++t) v[t] = acc[t][r];
                *reinterpret_cast<T*>(slot + r * N + V * lane) = v;
            }
        }
    }
    if constexpr (PAIRED) {
        __syncthreads();
        if (w < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (STRIDED) {
#pragma unroll
                    for (int t = 0; t < V; ++t) slot[r * N + lane + 64 * t] = acc[t][r] + slot[r * N + lane + 64 * t];
                } else {
                    T v = *reinterpret_cast<const T*>(slot + r * N + V * lane);
#pragma unroll
                    for (int t = 0; t < V; ++t) v[t] = acc[t][r] + v[t];
                    *reinterpret_cast<T*>(slot + r * N + V * lane) = v;
                }
            }
        }
    }
}

// partial[w][4][N] = A[4][k-range of wave w] . W[k-range][N]   (into LDS `red`)
template <int K, int N, int DEPTH, typename Mid>
__device__ __forceinline__ void t4_contract(T4W<K, N, DEPTH>& tw, const float* __restrict__ W, const float* As, int lda,
                                            float* red, Mid mid) {
    constexpr int V = T4W<K, N, DEPTH>::V, KW = T4W<K, N, DEPTH>::KW, PF = T4W<K, N, DEPTH>::PF;
    static_assert(KW % 4 == 0 && KW % PF == 0, "k range of a wave must be a multiple of 4 and of the ring depth");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* wp = T4W<K, N, DEPTH>::wptr(W);
    const float* ap = As + (lane & 3) * lda + w * KW;
    f32x4 acc[V];
#pragma unroll
    for (int t = 0; t < V; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int k0 = 0; k0 < KW - PF; k0 += PF) {
#pragma unroll
        for (int q = 0; q < PF; q += 4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + k0 + q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < V; ++t) acc[t] = MAMDR_MFMA4(a4[u], tw.b[q + u][t], acc[t]);
                if constexpr (T4W<K, N, DEPTH>::LOADS) t4_load<V>(tw.b[q + u], wp + (size_t)(k0 + q + u + PF) * N);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int q = 0; q < PF; q += 4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + (KW - PF) + q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < V; ++t) acc[t] = MAMDR_MFMA4(a4[u], tw.b[q + u][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
    mid();
    t4_put<N, V, false>(acc, red);
}

// ---- layer 0 (384 -> 256) with the reduction index split in TWO segments per wave: the [user | item] part first
// (k in [32 w, 32 w + 32)), then the domain part (k in [256 + 16 w, 256 + 16 w + 16)).  `midseg` runs between them:
// with a domain-table step still pending (DmStep) the domain columns of x only become known there -- the round trip
// to the step's partials and the step itself overlap the first segment's weight stream instead of standing in front
// of the whole layer.  Same 8-deep register ring; the second segment's first rows are requested behind the last
// chunk of the first, so they are in flight across `midseg`.
struct T4L0 {
    float b[T4_PF][4];
    static __device__ __forceinline__ const float* wptr1(const float* __restrict__ W) {
        return W + (size_t)((threadIdx.x >> 6) * 32) * H1 + 4 * (threadIdx.x & 63);
    }
    static __device__ __forceinline__ const float* wptr2(const float* __restrict__ W) {
        return W + (size_t)(2 * EMB + (threadIdx.x >> 6) * 16) * H1 + 4 * (threadIdx.x & 63);
    }
    // (skip1: the contraction starts at the domain segment -- NFM, whose kernel occupies rows 256..383 only)
    __device__ __forceinline__ void prefetch(const float* __restrict__ W, bool skip1 = false) {
        const float* wp = skip1 ? wptr2(W) : wptr1(W);
#pragma unroll
        for (int u = 0; u < T4_PF; ++u) t4_load<4>(b[u], wp + (size_t)u * H1);
        __builtin_amdgcn_sched_barrier(0);
    }
};
// (`acc`: zeroed by the caller -- see k_tower4, where the registers are claimed before the gather)
template <typename MidSeg, typename Mid>
__device__ __forceinline__ void t4_contract_l0(T4L0& tw, const float* __restrict__ W, const float* xs, float* red,
                                               f32x4 (&acc)[4], MidSeg midseg, Mid mid, bool skip1 = false) {
    static_assert(T4_PF == 8, "two segments of 32 and 16 rows, 8-deep ring");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* wp1 = T4L0::wptr1(W);
    const float* wp2 = T4L0::wptr2(W);
    const float* ap1 = xs + (lane & 3) * T4_XLD + 32 * w;
    const float* ap2 = xs + (lane & 3) * T4_XLD + 2 * EMB + 16 * w;
    auto chunk = [&](const float* ap, const float* next, bool reload) {
#pragma unroll
        for (int q = 0; q < T4_PF; q += 4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = MAMDR_MFMA4(a4[u], tw.b[q + u][t], acc[t]);
                if (reload) t4_load<4>(tw.b[q + u], next + (size_t)(q + u) * H1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (!skip1) {              // (uniform)
#pragma unroll 1
        for (int k0 = 0; k0 < 24; k0 += 8) chunk(ap1 + k0, wp1 + (size_t)(k0 + 8) * H1, true);
        chunk(ap1 + 24, wp2, true);                    // ... and the domain segment's first 8 rows
    }
    midseg();
    chunk(ap2, wp2 + (size_t)8 * H1, true);
    chunk(ap2 + 8, wp2, false);
    mid();
    t4_put<H1, 4, false>(acc, red);
}

// ---- W1L: W1 [256][128] in LDS.  LDS-DMA (global_load_lds_dwordx4: 16 B per lane, two rows per wave instruction, 16
// instructions per wave, no registers) writes a lane-linear image, so the swizzle sits on the SOURCE address: the 16-B
// chunk q of row r lives at chunk position q ^ (r & 31).  Requested at kernel start, the 128 KB arrive while the
// gather's misses are outstanding (probe, tools/probes/glds_probe.hip: 2.6 K cycles with all 256 workgroups
// streaming, reads of either pattern conflict-free).  Forward (layer 1): row k of the image is read as 64 x 8 B
// (lane l: columns 2 l, 2 l + 1).  Backward (dz2 . W1^T): lane l reads the chunk of rows l + 64 t holding columns
// 4 q .. 4 q + 3 -- four consecutive k of the transposed contraction -- 8 lanes hit 8 different chunk positions.
// W1 is streamed once per workgroup instead of twice (W1 and its transposed copy), and neither contraction waits on L2.
// The requests are inline asm, i.e. NOT in the compiler's vmcnt bookkeeping: a counted LDS-DMA makes hipcc wait
// vmcnt(0) at the next use of any load, which put the whole (cold: the weights were just rewritten by another
// kernel) stream in front of the gather (stamps: +2.3 K cycles).  Uncounted, they only make the compiler's counted
// waits conservative (vmcnt retires in order); issued behind every other load of the prologue they are waited for by
// nobody until t4_w1_landed() at the end of layer 0.
template <int ROWS>
__device__ __forceinline__ void t4_w1_request(const float* __restrict__ W1, float* w1s, const int row0) {
    w1_image_request<ROWS>(W1, w1s, row0);       // mamdr_kernels.h
}
// every request of this wave has landed (a workgroup barrier must follow before another wave reads the image)
__device__ __forceinline__ void t4_w1_landed() { w1_image_landed(); }

// forward: the wave's 32 rows in chunks of 8, the next chunk's reads issued before the current chunk's MFMAs
template <typename Mid>
__device__ __forceinline__ void t4_contract_w1f(const float* w1s, const float* As, float* red, Mid mid) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* ap = As + (lane & 3) * T4_H1LD + 32 * w;
    const float* wp = w1s + (32 * w) * H2 + 2 * (lane & 1);
    const int half = lane >> 1;
    f32x4 acc[2];
    acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x2 b[2][8];
    f32x4 a4[2][2];
    auto request = [&](int c, int buf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 8 * c + u;                              // = row & 31 (the wave's first row is a multiple of 32)
            b[buf][u] = *reinterpret_cast<const f32x2*>(wp + k * H2 + 4 * (half ^ k));
        }
        a4[buf][0] = *reinterpret_cast<const f32x4*>(ap + 8 * c);
        a4[buf][1] = *reinterpret_cast<const f32x4*>(ap + 8 * c + 4);
    };
    request(0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cur = c & 1;
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < 4) request(c + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = MAMDR_MFMA4(a4[cur][u >> 2][u & 3], b[cur][u][0], acc[0]);
            acc[1] = MAMDR_MFMA4(a4[cur][u >> 2][u & 3], b[cur][u][1], acc[1]);
        }
    }
    mid();
    t4_put<H2, 2, false>(acc, red);
}
// backward: the wave's 4 chunk columns (16 k of the transposed contraction), reads one chunk column ahead
template <typename Mid>
__device__ __forceinline__ void t4_contract_w1b(const float* w1s, const float* As, float* red, Mid mid) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* ap = As + (lane & 3) * T4_H2LD + 16 * w;
    const float* rp = w1s + lane * H2;
    const int sw = lane & 31;                                     // (l + 64 t) & 31
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 b[2][4], a4[2];
    auto request = [&](int c, int buf) {
        const int pos = 4 * ((4 * w + c) ^ sw);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[buf][t] = *reinterpret_cast<const f32x4*>(rp + t * 64 * H2 + pos);
        a4[buf] = *reinterpret_cast<const f32x4*>(ap + 4 * c);
    };
    request(0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cur = c & 1;
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < 4) request(c + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MAMDR_MFMA4(a4[cur][j], b[cur][t][j], acc[t]);
    }
    mid();
    t4_put<H1, 4, true>(acc, red);
}

// sum of the partials of output (row, col), slot order (8 wave slots, or 4 wave-pair slots at 256 columns: t4_put)
template <int N>
__device__ __forceinline__ float t4_sum(const float* red, int row, int col) {
    constexpr int SLOTS = N > H2 ? 4 : T4_WAVES;
    float s = red[row * N + col];
#pragma unroll
    for (int w = 1; w < SLOTS; ++w) s += red[(w * T4_ROWS + row) * N + col];
    return s;
}

#ifdef MAMDR_STAMPS   // diagnostic build only (tools/stamp_tower.py)
#define T4STAMP(k)                                                                            \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 16 + (k)] = t_;               \
    } while (0)
// the same stamp by the first lane of wave 4 (rows 256.. of the stamp buffer)
#define T4STAMP_W4(k)                                                                         \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (a.stamps && threadIdx.x == 256) a.stamps[(blockIdx.x + 256) * 16 + (k)] = t_;     \
    } while (0)
// wall-clock stamp (s_memrealtime: one 100 MHz counter for the whole device, comparable across XCDs and kernels)
#define T4REAL(k)                                                                             \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 16 + (k)] = t_;               \
    } while (0)
#else
#define T4STAMP(k) do { } while (0)
#define T4STAMP_W4(k) do { } while (0)
#define T4REAL(k) do { } while (0)
#endif

__device__ __forceinline__ int t4_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// DX: trainable user / item tables -> d loss / d [user | item] row = dz1 . W0[0:256,:]^T through the
// transposed copy W0T (kept current by k_update), row ids + representatives for the table update.
// FM: DeepFM tower (logit += linear tables + FM second-order term), as in k_tower.
// PRE: the instance for pre-gathered passes (k_pass_prep; k_wgrad_adam path) -- the perm / uid / pid / table-row gather
// is compiled out of it, and the pre-gather out of the other one
// The leading scalar arguments repeat what the prologue's first loads need (pass pointers, rows, W0, W1): with
// -mllvm -amdgpu-kernarg-preload-count (mamdr_amd/build.py) they arrive in SGPRs with the wave, and those loads go out
// without waiting for the argument block's own fetch (probe, tools/probes/kernarg_preload_probe.hip: entry -> first
// load returned 840 -> 380 cycles).  `a` carries the same values; only the prologue reads the copies.
// W2D (PRE + W1L only): the backward pass reads W2 itself instead of the transposed copy W2T -- the first tower of a call
// whose copies are stale (an instance of its own: as a run-time branch around the ring's loads it cost every launch 0.3 us)
template <bool DX, bool FM, bool W1L, bool PRE = false, bool W2D = false>
__global__ __launch_bounds__(T4_THREADS) void k_tower4(const float* __restrict__ k_xpre, const int32_t* __restrict__ k_pdom,
                                                       const float* __restrict__ k_plabel, const float* __restrict__ k_w0,
                                                       const float* __restrict__ k_w1, const int k_rows, const TowerArgs a) {
    static_assert(!PRE || (!DX && !FM), "pre-gathered passes serve the frozen-table mlp tower");
    static_assert(!W2D || (PRE && W1L), "W2 read in place: the W1-image instance of the pre-gathered tower");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    const int n_tiles = (int)gridDim.x;
    const int r0 = tile * T4_ROWS;
    int* rowi = reinterpret_cast<int*>(smem + T4_ROWI);          // [0,4) uid [4,8) pid [8,12) dom [12,16) valid
    float* rowf = smem + T4_ROWI + 16;                           // [0,4) label, [4,8) loss of the row
    float* red = smem + T4_RED;
    const float* P = a.dense;
    float* acts_t = a.acts + (size_t)r0 * ACT_LD;
    float* dz_t = a.dz + (size_t)r0 * DZ_LD;

    // weights and small parameters first: they do not depend on the gather
    T4L0 w0;
    T4W<H1, H2, T4_DEEP> w1;      // (W1L: unused, W1 is in LDS)
    T4W<H2, H3, W1L ? 16 : T4_DEEP> w2;      // W1L: the wave's whole share in flight, requested across the LDS-fed layer 1
    T4W<H3, H2> v2;      // backward: dz3 . W2^T through the transposed copy W2T [64][128]
    T4W<H2, H1, T4_DEEP> v1;      // backward: dz2 . W1^T through W1T [128][256] (W1L: unused)
    float* w1s = smem + T4_W1S;
    T4W<H1, 2 * EMB> v0; // DX: dz1 . W0[0:256,:]^T through W0T [256][256]
    // layer 0's accumulators are zeroed HERE: initialised in front of the contraction, they took the registers of the
    // gather's (divergent-branch) table loads and the compiler drained every outstanding load -- the pending
    // domain-table partials, HBM misses needed only between layer 0's segments -- before the gather-end barrier
    f32x4 acc0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(acc0[0]), "+v"(acc0[1]), "+v"(acc0[2]), "+v"(acc0[3]));
    T4REAL(10);
    T4STAMP(0);
    // the row bookkeeping's first dependent load (perm) goes out before everything else: loads retire in order, so
    // whatever is requested ahead of it (weights, domain-table partials) would be waited for together with it.
    // Every lane loads (a clamped position): a divergent branch around a load drains all loads at its end.
    int perm_src = 0;
    // pre-gathered pass (k_pass_prep): the tile's four [user | item] rows, domains and labels sit at known addresses;
    // every lane loads (clamped rows; lanes 256.. repeat rows 0..3), no dependent chain
    // (the k_wgrad_adam path's duties -- pre-gathered passes, the pending domain-table step, the W0 snapshot -- exist
    // for the frozen-table mlp tower only: compiled out of the DX / FM instances, see k_tower)
    constexpr bool FUSED_OK = !DX && !FM;
    constexpr bool pre = PRE;
    f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f};
    int pre_dom = 0;
    float pre_lab = 0.f;
    if (pre) {
        const int rr = min(r0 + ((tid >> 6) & 3), max(k_rows - 1, 0));
        xv = *reinterpret_cast<const f32x4*>(k_xpre + (size_t)rr * (2 * EMB) + 4 * (tid & 63));
        const int rb = min(r0 + (tid & 3), max(k_rows - 1, 0));
        pre_dom = k_pdom[rb];
        pre_lab = k_plabel[rb];
    } else if (a.perm) {       // uniform
        const int64_t pc = a.row_base + min(r0 + (tid & (T4_ROWS - 1)), max(a.rows - 1, 0));
        perm_src = a.perm[pc];
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool l0_dom_only = FM && a.deepfm == 4;      // NFM: rows 0..255 of W0 are zero and meet nothing the DNN reads
    w0.prefetch(k_w0, l0_dom_only);
    // (the snapshot of W0[256:384] for k_wgrad_adam is taken at the END of the kernel since round 5: here, its load -> store
    // dependency put an `s_waitcnt vmcnt(0)` into the last wave's prologue -- one full round trip for everything that wave
    // had requested, in front of its bias loads and W1-image requests -- and the whole workgroup waited for that wave at the
    // bookkeeping barrier)
#ifdef MAMDR_T4_SNAP_EARLY
    if (FUSED_OK) tower_snapshots(a, T4_THREADS, n_tiles);
#endif
    const bool dmw = FUSED_OK && a.dm_snap_out != nullptr;       // k_wgrad_adam path: domain-table duty (DmStep)
    // (the pending domain row is requested AFTER the bookkeeping, also with a pre-gathered pass: asked for here, from
    // the caller's expected domain, its loads -- misses to HBM -- delayed the x rows by 1.5 K cycles; measured)
    DmWave dmt;
    DmWaveAddr dma;
    if (dmw) dm_wave_addr(a, tile, dma);
    const int ecol = tid & 255, erow2 = tid >> 8;                // epilogue ownership for N = 256: rows erow2, erow2 + 2
    const float b0r = P[a.L.b0 + ecol];
    const float b1r = P[a.L.b1 + (tid & 127)];
    const float b2r = P[a.L.b2 + (tid & 63)];
    const float wor = P[a.L.wo + (tid & 63)];
    const float gbr = P[a.L.gb];
    __builtin_amdgcn_sched_barrier(0);
    // W1 image: behind every other load of the prologue (uncounted requests make the counted waits ahead of them
    // conservative by their number: the wait for the x rows then also covers layer 0's first ring rows and ONE request).
    // Variants measured (k_tower4 at 1,024 rows, us): requests by every wave here 14.0-14.2; by waves 4..7 only (which
    // consume nothing before layer 0) 14.3-14.6 -- they sit in the issue queue until most of the image has landed and
    // hold the bookkeeping barrier; the same without that barrier (domains by scalar loads) 14.9; counted requests
    // (__builtin_amdgcn_global_load_lds: vmcnt(0) at the first use of any load) 14.6; no image 15.1-15.3.
    if (W1L && pre) t4_w1_request<32>(k_w1, w1s, 32 * __builtin_amdgcn_readfirstlane(w));
    __builtin_amdgcn_sched_barrier(0);

    // ---- row bookkeeping + embedding gather (4 rows x 96 float4)
    if (pre) {
        if (tid < T4_ROWS) {
            const bool valid = (r0 + tid) < a.rows;
            rowi[tid] = 0;
            rowi[4 + tid] = 0;
            rowi[8 + tid] = pre_dom;
            rowi[12 + tid] = valid ? 1 : 0;
            rowf[tid] = pre_lab;
        }
        if (tid < 256) {
            const int row = tid >> 6;
            if (r0 + row >= a.rows) xv = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(smem + T4_XS + row * T4_XLD + 4 * (tid & 63)) = xv;
        }
    } else if (tid < T4_ROWS) {
        const bool valid = (r0 + tid) < a.rows;
        int64_t pos = a.row_base + r0 + tid;
        int64_t src = 0;
        if (valid) {
            src = a.perm ? (int64_t)perm_src : pos;
            if (src < 0) src = 0;
            if (src >= a.n_rows_split) src = a.n_rows_split - 1;
        }
        rowi[tid] = t4_clamp(a.uid[src], 0, a.n_user - 1);
        rowi[4 + tid] = t4_clamp(a.pid[src], 0, a.n_item - 1);
        rowi[8 + tid] = t4_clamp(a.dom[src], 0, a.n_domain - 1);
        rowi[12 + tid] = valid ? 1 : 0;
        rowf[tid] = a.label[src];
    }
    __syncthreads();
    // domain-table step still pending (DmStep): the domain rows as that step leaves them.  `same`: one domain per tile
    // AND the one the caller announced (the addresses were formed from it); otherwise every lane works alone.
    const bool pend = FUSED_OK && a.dms.snap != nullptr;
    const bool same = pend && rowi[8] == a.dm_hint && rowi[9] == rowi[8] && rowi[10] == rowi[8] && rowi[11] == rowi[8];
    T4STAMP(12);
    T4STAMP_W4(0);
    // (requested only now, beside the table rows: at kernel start these loads -- misses all the way to HBM, the
    // partials were written by the previous kernel -- sat in the CU's miss queue ahead of the bookkeeping's
    // second dependent load and delayed the whole gather)
    if (dmw) dm_wave_begin(a, tile, same, dma, dmt);
    T4STAMP_W4(1);
    // (a scalar branch around the whole block: the loads inside sit in divergent branches, at whose end the compiler
    // drains EVERY outstanding load -- including the partials just requested, a full round trip to HBM)
    if (!(pre && same) && tid < T4_ROWS * (XDIM / 4)) {
        const int row = tid / (XDIM / 4), c4 = tid - row * (XDIM / 4);
        const int seg = c4 >> 5, off = (c4 & 31) * 4;
        if (seg == 2 && pend) {
            if (!same) {
                f32x4 pn, mn, vn;
                dm_step4(a.dms, rowi[8 + row], c4 & 31, pn, mn, vn);
                if (!rowi[12 + row]) pn = (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(smem + T4_XS + row * T4_XLD + c4 * 4) = pn;
            }
        } else if (!(pre && seg < 2)) {         // (pre-gathered pass: the user / item part is in LDS already)
            const float* base = seg == 0 ? a.user_tab : (seg == 1 ? a.item_tab : a.dense + a.L.dm);
            f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)rowi[seg * 4 + row] * EMB + off);
            if (!rowi[12 + row]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(smem + T4_XS + row * T4_XLD + c4 * 4) = v;
        }
    }
    // (pending + one-domain tile: the domain columns of x are filled between layer 0's two segments, see midseg)
    // (no pre-gathered pass: the gather's dependent loads are younger than anything requested above and would wait
    // for the image as well -- it is requested only now)
    if (W1L && !pre) t4_w1_request<32>(k_w1, w1s, 32 * __builtin_amdgcn_readfirstlane(w));
    T4STAMP_W4(2);
    __syncthreads();
    const bool pnn = FM && a.deepfm == 3;              // (uniform) the inner products of the field pairs feed three more rows of W0
    float wx0 = 0.f, wx1 = 0.f, wx2 = 0.f;             // ... rows a.L.wx + {0, 1, 2} H1, this thread's column
    if (FM && !pnn) {
        // thread (row, k): FM second-order term sum_k (u i + (u + i) d), reduced over the row's two waves
        const int row = tid >> 7, k = tid & 127;
        const float* xr = smem + T4_XS + row * T4_XLD;
        const float u = xr[k], it = xr[EMB + k], dd = xr[2 * EMB + k];
        float s = a.deepfm == 1 ? u * it + (u + it) * dd : 0.f;      // 2 = WDL: no FM term
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) rowf[16 + w] = s;                // wave w = 2 row + half
        __syncthreads();
        if (tid < T4_ROWS) {
            float lin = P[a.L.ld + rowi[8 + tid]];
            if (a.lin_user) lin = (a.lin_user[rowi[tid]] + a.lin_item[rowi[4 + tid]]) + lin;
            rowf[8 + tid] = (rowf[16 + 2 * tid] + rowf[17 + 2 * tid]) + lin;
        }
        // (read in the output-unit phase, several barriers later)
    }
    const bool nfm = FM && a.deepfm == 4;              // (uniform) the DNN reads the bi-interaction of the three fields
    if (nfm) {
        // NFM (deepctr BiInteractionPooling): f = u i + (u + i) d takes the domain field's place in the tile -- rows
        // 256..383 of W0 are deepctr's [128, 256] kernel, rows 0..255 are zero -- and the domain row moves aside for the
        // backward products.  (The linear tables' logit was formed above, as WDL's.)
        const int row = tid >> 7, k = tid & 127;
        float* xr = smem + T4_XS + row * T4_XLD;
        const float u = xr[k], it = xr[EMB + k], dd = xr[2 * EMB + k];
        smem[T4_DROW + row * EMB + k] = dd;
        xr[2 * EMB + k] = u * it + (u + it) * dd;
        __syncthreads();
    }
    if (pnn) {
        // PNN (deepctr InnerProductLayer, pairs (0,1) (0,2) (1,2)): ip = <u,i> <u,d> <i,d> per row, reduced over the row's
        // two waves; kept in LDS for layer 0's epilogue and written out for the weight gradient of the three rows
        const int row = tid >> 7, k = tid & 127;
        const float* xr = smem + T4_XS + row * T4_XLD;
        const float u = xr[k], it = xr[EMB + k], dd = xr[2 * EMB + k];
        float s0 = u * it, s1 = u * dd, s2 = it * dd;
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o);
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        if (lane == 0) {
            rowf[16 + 3 * w] = s0;
            rowf[17 + 3 * w] = s1;
            rowf[18 + 3 * w] = s2;
        }
        wx0 = P[a.L.wx + ecol];
        wx1 = P[a.L.wx + H1 + ecol];
        wx2 = P[a.L.wx + 2 * H1 + ecol];
        __syncthreads();
        if (tid < 4 * T4_ROWS) {
            const int r = tid >> 2, j = tid & 3;
            const float v = j < 3 ? rowf[16 + 6 * r + j] + rowf[19 + 6 * r + j] : 0.f;
            rowf[48 + tid] = v;
            a.ipbuf[(size_t)(r0 + r) * 4 + j] = v;
            if (j == 0) rowf[8 + r] = 0.f;             // no linear / FM logit in this tower
        }
        // (read behind layer 0's barrier)
    }

    T4STAMP(1);
    const float scale = a.use_dropout ? a.keep_scale : 1.0f;
    const bool drop = a.use_dropout != 0;

    // ---- layer 0: 384 -> 256
    t4_contract_l0(w0, k_w0, smem + T4_XS, red, acc0,
                   [&]() {
                       if (!dmw) return;
                       // every wave finishes the 16 domain columns ITS second segment contracts (partials requested
                       // behind the bookkeeping: long since arrived; no LDS staging, no barrier) and writes them to all
                       // four rows of the tile -- read back by the same wave only; workgroup d < n_domain does the same
                       // for row d and writes it back (live p / m / v + the snapshot for k_wgrad_adam and the next tower)
                       T4STAMP(13);
                       const int c = 16 * w + (lane & 15);
                       if (same) {
                           float p = dmt.rp, m = dmt.rm, v = dmt.rv;
                           dm_apply1(a.dms, dm_wave_sum(dmt.r), p, m, v);
                           if (lane < 16) {
#pragma unroll
                               for (int rr = 0; rr < T4_ROWS; ++rr) smem[T4_XS + rr * T4_XLD + 2 * EMB + c] = rowi[12 + rr] ? p : 0.f;
                           }
                       }
                       T4STAMP(14);
                       if (tile < a.n_domain) {
                           float p = dmt.wp, m = dmt.wm, v = dmt.wv;
                           if (pend) dm_apply1(a.dms, dm_wave_sum(dmt.w), p, m, v);
                           if (lane < 16) dm_store_elem(a, (size_t)tile * EMB + c, pend, p, m, v);
                       }
                       // grids smaller than the domain count (tiny batches): the remaining rows, one lane chain per float4
                       if (tile + n_tiles < a.n_domain && tid < EMB / 4) {
                           for (int d = tile + n_tiles; d < a.n_domain; d += n_tiles) {
                               f32x4 p4, m4, v4;
                               const size_t row = (size_t)d * EMB + 4 * tid, plane = (size_t)a.n_domain * EMB;
                               if (pend) {
                                   dm_step4(a.dms, d, tid, p4, m4, v4);
                                   if (a.dms.optimizer != 2) *reinterpret_cast<f32x4*>(a.dm_live_p + row) = p4;
                                   if (a.dms.optimizer != 1) *reinterpret_cast<f32x4*>(a.dm_live_m + row) = m4;
                                   if (a.dms.optimizer == 0) *reinterpret_cast<f32x4*>(a.dm_live_v + row) = v4;
                               } else {
                                   p4 = *reinterpret_cast<const f32x4*>(a.dm_live_p + row);
                                   m4 = *reinterpret_cast<const f32x4*>(a.dm_live_m + row);
                                   v4 = *reinterpret_cast<const f32x4*>(a.dm_live_v + row);
                               }
                               *reinterpret_cast<f32x4*>(a.dm_snap_out + row) = p4;
                               *reinterpret_cast<f32x4*>(a.dm_snap_out + plane + row) = m4;
                               *reinterpret_cast<f32x4*>(a.dm_snap_out + 2 * plane + row) = v4;
                           }
                       }
                       T4STAMP(15);
                   },
                   [&]() {
                       if (W1L) {
                           t4_w1_landed();           // (layer 0's ring is empty: nothing else is outstanding)
                           w2.prefetch(P + a.L.w2);
                       } else {
                           w1.prefetch(P + a.L.w1);
                       }
                   },
                   l0_dom_only);
    T4STAMP(2);
    if (!pre && tid < T4_ROWS * (XDIM / 4)) {     // x tile to the workspace, behind the weight stream (k_wgrad_adam reads
                                                  // a pre-gathered pass in place)
        const int row = tid / (XDIM / 4), c4 = tid - row * (XDIM / 4);
        *reinterpret_cast<f32x4*>(acts_t + (size_t)row * ACT_LD + c4 * 4) =
            *reinterpret_cast<const f32x4*>(smem + T4_XS + row * T4_XLD + c4 * 4);
    }
    __syncthreads();
    {
        const uint32_t key = dropout_layer_key(a.seed, a.step, 0);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = erow2 + 2 * rr;
            float z = t4_sum<H1>(red, row, ecol) + b0r;
            if (pnn) z += (rowf[48 + 4 * row] * wx0 + rowf[49 + 4 * row] * wx1) + rowf[50 + 4 * row] * wx2;
            float h = fmaxf(z, 0.f);
            if (drop) {
                const uint32_t u = mamdr_dropout_u32(key, (uint32_t)(r0 + row) * (uint32_t)H1 + (uint32_t)ecol);
                h = (u >= a.drop_thresh) ? h * scale : 0.f;
            }
            smem[T4_H1 + row * T4_H1LD + ecol] = h;
            T4_WS_STORE(&acts_t[(size_t)row * ACT_LD + XDIM + ecol], h);
        }
    }
    __syncthreads();

    T4STAMP(3);
    // ---- layer 1: 256 -> 128
    if (W1L) t4_contract_w1f(w1s, smem + T4_H1, red, [&]() {
        if constexpr (W2D) v2.prefetch_untransposed(P + a.L.w2);     // W2T stale: the call's first tower
        else v2.prefetch(a.wT + W2T_OFF);
    });
    else t4_contract(w1, P + a.L.w1, smem + T4_H1, T4_H1LD, red, [&]() { w2.prefetch(P + a.L.w2); });
    __syncthreads();
    T4STAMP(4);
    {
        const uint32_t key = dropout_layer_key(a.seed, a.step, 1);
        const int row = tid >> 7, col = tid & 127;
        float h = fmaxf(t4_sum<H2>(red, row, col) + b1r, 0.f);
        if (drop) {
            const uint32_t u = mamdr_dropout_u32(key, (uint32_t)(r0 + row) * (uint32_t)H2 + (uint32_t)col);
            h = (u >= a.drop_thresh) ? h * scale : 0.f;
        }
        smem[T4_H2 + row * T4_H2LD + col] = h;
        T4_WS_STORE(&acts_t[(size_t)row * ACT_LD + XDIM + H1 + col], h);
    }
    __syncthreads();

    T4STAMP(5);
    // ---- layer 2: 128 -> 64 (the backward weights are requested behind its K loop)
    t4_contract(w2, P + a.L.w2, smem + T4_H2, T4_H2LD, red, [&]() {
        if (!W1L) {
            v2.prefetch(a.wT + W2T_OFF);
            v1.prefetch(a.wT + W1T_OFF);
        } else if (DX) {
            v0.prefetch(a.wT + W0T_OFF);
        }
    });
    __syncthreads();
    T4STAMP(6);
    // ---- epilogue of layer 2 + output unit + sigmoid / Keras BCE + dz3: wave r owns row r
    if (w < T4_ROWS) {
        const uint32_t key = dropout_layer_key(a.seed, a.step, 2);
        const int row = w, col = lane;
        float h = fmaxf(t4_sum<H3>(red, row, col) + b2r, 0.f);
        if (drop) {
            const uint32_t u = mamdr_dropout_u32(key, (uint32_t)(r0 + row) * (uint32_t)H3 + (uint32_t)col);
            h = (u >= a.drop_thresh) ? h * scale : 0.f;
        }
        T4_WS_STORE(&acts_t[(size_t)row * ACT_LD + XDIM + H1 + H2 + col], h);
        float s = h * wor;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        float logit = s + gbr;
        if (FM) logit += rowf[8 + row];
        float p;
        if (logit >= 0.f) {
            p = 1.0f / (1.0f + __expf(-logit));
        } else {
            const float ez = __expf(logit);
            p = ez / (1.0f + ez);
        }
        const bool valid = rowi[12 + row] != 0;
        const float y = rowf[row];
        const float lo = 1e-7f, hi = 1.0f - 1e-7f;
        const float pc = fminf(fmaxf(p, lo), hi);
        const float zc = __logf(pc / (1.0f - pc));
        const float loss = fmaxf(zc, 0.f) - zc * y + __logf(1.0f + __expf(-fabsf(zc)));
        const float inside = (p >= lo && p <= hi) ? 1.0f : 0.0f;
        float dl = valid ? ((p - y) * inside) / (float)a.rows : 0.0f;
        if (a.uw_off >= 0) {           // uncertainty weighting: d loss / d logit scales by 1 / var^2
            const float var = P[a.uw_off];
            dl *= 1.0f / (var * var);
        }
        if (lane == 0) {
            a.dlogit[r0 + row] = dl;
            a.domrow[r0 + row] = rowi[8 + row];
            rowf[4 + row] = valid ? loss : 0.f;
            rowf[12 + row] = dl;
            if (DX) {
                a.urow[r0 + row] = valid ? rowi[row] : -1;
                a.irow[r0 + row] = valid ? rowi[4 + row] : -1;
                if (valid && a.map_u) {   // representative of a table row = its smallest batch position (exact)
                    atomicMin(a.map_u + rowi[row], r0 + row);
                    atomicMin(a.map_i + rowi[4 + row], r0 + row);
                }
            }
        }
        const float d = (h > 0.f) ? (dl * wor) * scale : 0.f;
        smem[T4_DZ3 + row * T4_H3LD + col] = d;
        T4_WS_STORE(&dz_t[(size_t)row * DZ_LD + H1 + H2 + col], d);
    }
    __syncthreads();
    if (tid == 0) a.loss_part[tile] = (rowf[4] + rowf[5]) + (rowf[6] + rowf[7]);
    if (FM && !pnn && !nfm) {     // d fm / d domain embedding = u + i: per-row term of the domain-table gradient
        const int row = tid >> 7, k = tid & 127;
        const float* xr = smem + T4_XS + row * T4_XLD;
        a.fmq[(size_t)(r0 + row) * EMB + k] = a.deepfm == 1 ? rowf[12 + row] * (xr[k] + xr[EMB + k]) : 0.f;
    }

    T4STAMP(7);
    // ---- backward: dz2 = (dz3 . W2^T) * gate(h2)
    t4_contract(v2, a.wT + W2T_OFF, smem + T4_DZ3, T4_H3LD, red, []() {});
    __syncthreads();
    {
        const int row = tid >> 7, col = tid & 127;
        const float v = t4_sum<H2>(red, row, col);
        const float d = (smem[T4_H2 + row * T4_H2LD + col] > 0.f) ? v * scale : 0.f;
        smem[T4_DZ2 + row * T4_H2LD + col] = d;
        T4_WS_STORE(&dz_t[(size_t)row * DZ_LD + H1 + col], d);
    }
    __syncthreads();
    T4STAMP(8);
    // ---- dz1 = (dz2 . W1^T) * gate(h1); the domain-table gradient follows from dz1 by linearity
    if (W1L) t4_contract_w1b(w1s, smem + T4_DZ2, red, []() {});
    else t4_contract(v1, a.wT + W1T_OFF, smem + T4_DZ2, T4_H2LD, red, [&]() { if (DX) v0.prefetch(a.wT + W0T_OFF); });
    __syncthreads();
    float dip[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};      // PNN: this thread's share of d loss / d ip of its two rows
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = erow2 + 2 * rr;
        const float v = t4_sum<H1>(red, row, ecol);
        const float d = (smem[T4_H1 + row * T4_H1LD + ecol] > 0.f) ? v * scale : 0.f;
        T4_WS_STORE(&dz_t[(size_t)row * DZ_LD + ecol], d);
        if (DX || nfm) smem[T4_DZ1 + row * T4_H1LD + ecol] = d;
        dip[rr][0] = d * wx0;
        dip[rr][1] = d * wx1;
        dip[rr][2] = d * wx2;
    }
    T4STAMP(9);
    T4REAL(11);
#ifndef MAMDR_T4_SNAP_EARLY
    if (FUSED_OK) tower_snapshots(a, T4_THREADS, n_tiles);      // (W0's rows are L2-resident by now: layer 0 streamed them)
#endif
    if (pnn) {
        // d loss / d ip[row][j] = sum_c dz1[row][c] W0x[j][c]: wave sums, then the row's four waves (rows erow2, erow2 + 2
        // belong to waves 4 erow2 .. 4 erow2 + 3) through LDS in wave order; the inner products' chain rule then gives the
        // per-row term of the domain-table gradient, fmq = dip_ud u + dip_id i (S2 tiles of k_wgrad), and with trainable
        // tables the terms of the two table rows (added to the input gradient below)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                for (int o = 32; o > 0; o >>= 1) dip[rr][j] += __shfl_xor(dip[rr][j], o);
        __syncthreads();           // (everybody is past the sums of `red`)
        if (lane == 0) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int j = 0; j < 3; ++j) red[(w * 2 + rr) * 4 + j] = dip[rr][j];
        }
        __syncthreads();
        if (tid < 4 * T4_ROWS) {
            const int r = tid >> 2, j = tid & 3;
            // row r = erow2 + 2 rr  ->  erow2 = r & 1, rr = r >> 1; its waves: 4 erow2 .. 4 erow2 + 3
            float v = 0.f;
            if (j < 3) {
                const int w0_ = 4 * (r & 1), rr = r >> 1;
                v = ((red[((w0_ + 0) * 2 + rr) * 4 + j] + red[((w0_ + 1) * 2 + rr) * 4 + j]) + red[((w0_ + 2) * 2 + rr) * 4 + j]) +
                    red[((w0_ + 3) * 2 + rr) * 4 + j];
            }
            rowf[64 + tid] = v;
        }
        __syncthreads();
        {
            const int row = tid >> 7, k = tid & 127;
            const float* xr = smem + T4_XS + row * T4_XLD;
            a.fmq[(size_t)(r0 + row) * EMB + k] = rowf[65 + 4 * row] * xr[k] + rowf[66 + 4 * row] * xr[EMB + k];
        }
    }
    if (nfm) {
        // d loss / d f = dz1 . W0[256:384, :]^T (rows of the kernel read in place: lane l owns output columns 2 l, 2 l + 1,
        // i.e. rows 256 + 2 l (+ 1) of W0, the wave's 32 k as two halves of 16); then the bi-interaction's chain rule:
        // the per-row term of the domain-table gradient fmq = df (u + i) and, with trainable tables, the two table rows'
        // gradients df (i + d), df (u + d)
        __syncthreads();           // dz1 complete, `red` free again
        {
            const float* wn = P + a.L.w0 + (size_t)(2 * EMB + 2 * lane) * H1 + 32 * w;
            const float* ap = smem + T4_DZ1 + (lane & 3) * T4_H1LD + 32 * w;
            f32x4 acc[2];
            acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 b0[4], b1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    b0[q] = *reinterpret_cast<const f32x4*>(wn + 16 * half + 4 * q);
                    b1[q] = *reinterpret_cast<const f32x4*>(wn + H1 + 16 * half + 4 * q);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 16 * half + 4 * q);
#pragma unroll
                    for (int u_ = 0; u_ < 4; ++u_) {
                        acc[0] = MAMDR_MFMA4(a4[u_], b0[q][u_], acc[0]);
                        acc[1] = MAMDR_MFMA4(a4[u_], b1[q][u_], acc[1]);
                    }
                }
            }
            t4_put<EMB, 2, false>(acc, red);
        }
        __syncthreads();
        {
            const int row = tid >> 7, k = tid & 127;
            const float df = t4_sum<EMB>(red, row, k);
            const float* xr = smem + T4_XS + row * T4_XLD;
            const float u = xr[k], it = xr[EMB + k], dd = smem[T4_DROW + row * EMB + k];
            a.fmq[(size_t)(r0 + row) * EMB + k] = df * (u + it);
            if (DX) {
                float* dxe_t = a.dxe + (size_t)(r0 + row) * (2 * EMB);
                dxe_t[k] = df * (it + dd);
                dxe_t[EMB + k] = df * (u + dd);
            }
        }
    } else if (DX) {
        __syncthreads();           // dz1 complete, `red` free again
        t4_contract(v0, a.wT + W0T_OFF, smem + T4_DZ1, T4_H1LD, red, []() {});
        __syncthreads();
        float* dxe_t = a.dxe + (size_t)r0 * (2 * EMB);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = erow2 + 2 * rr;
            float v = t4_sum<2 * EMB>(red, row, ecol);
            if (FM) {              // d fm / d e_f = sum of the other two fields
                const float* xr = smem + T4_XS + row * T4_XLD;
                const int k = ecol & (EMB - 1);
                const float other = (ecol < EMB ? xr[EMB + k] : xr[k]) + xr[2 * EMB + k];
                if (a.deepfm == 1) v = fmaf(rowf[12 + row], other, v);
                // PNN: d <u,i> / d u = i, d <u,d> / d u = d;  d <u,i> / d i = u, d <i,d> / d i = d
                if (pnn) v += rowf[64 + 4 * row] * (ecol < EMB ? xr[EMB + k] : xr[k]) +
                              rowf[(ecol < EMB ? 65 : 66) + 4 * row] * xr[2 * EMB + k];
            }
            dxe_t[(size_t)row * (2 * EMB) + ecol] = v;
        }
    }
}

// W1L needs the whole LDS of a CU: one workgroup per CU, i.e. grids of up to one tile per CU; larger grids keep the
// streaming variant (30 KB of LDS, several workgroups per CU overlap each other's phases).
// One-time raises of the LDS limit: function-local statics with initialisers (C++11: initialised exactly once, thread-safe --
// the lanes of mamdr_amd/parallel.py reach these launchers from several host threads at once).
template <bool DX, bool FM, bool PRE, bool W2D>
static bool t4_w1l_raised() {
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower4<DX, FM, true, PRE, W2D>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)tower4_lds_bytes(true)) == hipSuccess;
    return ok;
}
// -> 0, or MAMDR_T4_E* (no abort() inside the library: mamdr_train_steps turns these into MAMDR_ESTATE / MAMDR_EHIP through
// mamdr_last_error)
template <bool DX, bool FM, bool PRE>
static int launch_tower4_inst(const TowerArgs& a, dim3 grid, bool w1l, hipStream_t s) {
    bool raised = false;           // the LDS limit of the W1L instance is raised (refused: streaming variant only)
    if (w1l) {
        raised = t4_w1l_raised<DX, FM, PRE, false>();
        if (!raised) {
            static const bool told = (fprintf(stderr, "mamdr: k_tower4 keeps streaming W1 (its LDS image was refused)\n"), true);
            (void)told;
        }
    }
    const float* w0 = a.dense + a.L.w0;
    const float* w1 = a.dense + a.L.w1;
    if (w1l && raised && PRE && a.w2_direct) {
        if constexpr (PRE) {
            if (!t4_w1l_raised<DX, FM, PRE, true>()) return T4_E_W2D_LDS;
            MAMDR_LAUNCH((k_tower4<DX, FM, true, PRE, true>), grid, dim3(T4_THREADS), tower4_lds_bytes(true), s, a.xpre, a.pdom,
                         a.plabel, w0, w1, a.rows, a);
        }
    } else if (w1l && raised)
        MAMDR_LAUNCH((k_tower4<DX, FM, true, PRE>), grid, dim3(T4_THREADS), tower4_lds_bytes(true), s, a.xpre, a.pdom, a.plabel, w0, w1,
                     a.rows, a);
    else {
        if (a.w2_direct) return T4_E_W2D_STATE;
        MAMDR_LAUNCH((k_tower4<DX, FM, false, PRE>), grid, dim3(T4_THREADS), tower4_lds_bytes(false), s, a.xpre, a.pdom, a.plabel, w0,
                     w1, a.rows, a);
    }
    return 0;
}
bool tower4_w1l_ready() {
    static const bool ok = t4_w1l_raised<false, false, true, false>() && t4_w1l_raised<false, false, true, true>();
    return ok;
}
static int t4_cu_count() {
    static const int n = []() {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            return v;
        return 1;
    }();
    return n;
}
// the predicate BOTH the launcher and the caller's w2_direct decision use (they disagreed on devices with
// fewer CUs than a batch has four-row tiles -- a CPX partition -- and the launcher aborted)
bool tower4_takes_w1l(int64_t rows, int no_w1l) {
    const int64_t tiles = ((rows + TILE_ROWS - 1) / TILE_ROWS) * (TILE_ROWS / T4_ROWS);
    return !no_w1l && tiles <= t4_cu_count();
}
int launch_tower4_train(const TowerArgs& a, hipStream_t s) {
    const int tiles = ((a.rows + TILE_ROWS - 1) / TILE_ROWS) * (TILE_ROWS / T4_ROWS);   // cover rows_pad
    const dim3 grid(tiles);
    const bool dx = a.dxe != nullptr;
    const bool w1l = tower4_takes_w1l(a.rows, a.no_w1l);
    if (a.deepfm) {
        if (dx) return launch_tower4_inst<true, true, false>(a, grid, w1l, s);
        return launch_tower4_inst<false, true, false>(a, grid, w1l, s);
    }
    if (dx) return launch_tower4_inst<true, false, false>(a, grid, w1l, s);
    if (a.xpre) return launch_tower4_inst<false, false, true>(a, grid, w1l, s);
    return launch_tower4_inst<false, false, false>(a, grid, w1l, s);
}

// W1T / W2T from the live weights (start of every mamdr_train_steps call; k_update keeps them current)
__global__ __launch_bounds__(256) void k_transpose_w(const float* dense, DenseLayout L, float* wT) {
    transpose_w_elem(dense, L, wT, (int)(blockIdx.x * 256 + threadIdx.x));
}
void launch_transpose_w(const float* dense, const DenseLayout& L, float* wT, hipStream_t s) {
    hipLaunchKernelGGL(k_transpose_w, dim3(TRANSPOSE_WGS), dim3(256), 0, s, dense, L, wT);
}

}  // namespace mamdr
