// Towers built from GENERIC dense layers: deepctr's SharedBottom / MMOE / PLE under the reference's per-domain
// compiled models (model_zoo/DeepMTLCTR/deep_mtl_ctr.py:21-96).  SURVEY.md section 8 f4: comparison baselines next
// to the hot path -- layer widths, expert counts and gate shapes vary per config (config/*/{shared_bottom,mmoe,ple}.json:
// hidden_dim [512,256,128] ... [256], 2-5 experts, PLE 3-15 specific + 2 shared experts per task), so nothing here is
// specialised to one shape the way step_kernels.hip is to 384-256-128-64.
//
// One training step of task d on a batch (what `domain_model_dict[d].fit` executes per batch, deep_mtl_ctr.py:79-80):
//   gather x = [U[uid] | I[pid] | Dm[dom]]                                   k_graph_gather   (hbm)
//   every expert task d mixes: DNN = per layer  relu(h W + b) * dropout       k_graph_gemm<0>  (mfma, 64x64 tiles)
//   gate_d: DNN, then softmax(q Wg) and the mixture sum_e gate_e expert_e     k_graph_gate_fwd (one workgroup per row)
//   tower_d: DNN;  head: sigmoid(t w + gb), Keras BCE, d loss / d logit       k_graph_head     (one wave per row)
//   backward, layer by layer: dW = in^T dz (k_graph_gemm<2>, the batch rows split over up to 16 workgroups per tile,
//   launch_wgrad), k_graph_wfinish = the partial products summed in a fixed order + db = column sums in the same launch,
//   d in = dz W^T times the producer's relu / dropout gate (k_graph_gemm<1>); the mixture's backward
//   (k_graph_gate_bwd) between tower and experts; the first layers add into d x[:, domain columns]
//   domain table: segment sum of d x over the batch's domain ids + 2 l2 Dm    k_graph_domain_grad
//   TF1 Adam (or SGD) on the two ranges of the flat vector task d's model trains: the shared block (domain table +
//   shared experts) and task d's block (its experts, gate, tower, head)       k_graph_adam     (hbm)
// Launch plan since round 4 (the per-layer launches above remain under MAMDR_GRAPH_NO_DEFER=1):
//   * forward / d-input contractions of fewer than 512 tiles of 64 x 64 run on 32 x 32 tiles with the reduction index
//     split over the workgroup's four waves (gemm_tile32, k_graph_gemm32*: 4 x the workgroups, 128-deep stages);
//   * the weight gradients are QUEUED by the backward pass and contracted in one flat grid at its end
//     (queue_wgrad / flush_wgrads, k_graph_wgrad_multi);
//   * one tail launch (k_graph_tail) finishes them (split sums, bias column sums), runs the narrow contractions
//     (head / gate kernels, PNN's rows, attention projections), the domain table's and the linear table's gradients --
//     and steps every parameter where its gradient is finished (GradSink; k_graph_adam only for the weighted loss).
// All fp32 (`v_mfma_f32_32x32x2_f32`: exact fp32 products), every reduction in a fixed order (no float atomics).
// Trainable user / item tables (the Amazon configs: no pretraining) sit at the head of the flat vector; their step is
// TF1's dense Adam over every row -- regulariser gradient 2 l2 p + the scatter-add of the batch's row gradients --
// through the table kernels of emb_kernels.hip in their per-step form (k_emb_flag / k_emb_reduce: duplicates summed
// in batch order by the row's first position; k_emb_sweep: one HBM pass over both tables).
//
// The single-output deepctr towers on the same layers (deepctr.py:33-46; ONE task serves every domain):
//   NFM      f = bi-interaction of the three fields (k_graph_feat_fwd/bwd), DNN over f, + deepctr's linear logit
//   PNN      DNN over [x | the 3 pairwise inner products] (the 3 extra kernel rows ride as a rank-3 epilogue of the GEMM)
//   CCPM     Conv2D (6,1) -> max over the fields -> Conv2D (5,1) centre tap, tanh (k_graph_ccpm_fwd/bwd), DNN over 512 features
//   AutoInt  3 x multi-head self-attention over the 3 field tokens on compact token-major buffers (projections as GEMMs
//            over 3 B token rows, the 3 x 3 attention core in k_graph_att_fwd/bwd), beside the DNN; head over [96 | DNN]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mamdr_hip.h"
#include "mamdr_kernels.h"

// every kernel launch of this engine goes through here: the count is what tools/graph_bench.py reports as launches per step
static std::atomic<long long> g_graph_launches{0};      // (contexts may be driven from several host threads: lanes)
#define GLAUNCH(...)                      \
    do {                                  \
        ++g_graph_launches;               \
        hipLaunchKernelGGL(__VA_ARGS__);  \
    } while (0)

using namespace mamdr;

namespace {

thread_local char g_gerr[512] = "";
int gfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_gerr, sizeof(g_gerr), fmt, ap);
    va_end(ap);
    return code;
}
#define GHIP(expr)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return gfail(MAMDR_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int GT = 64;      // output tile (rows and columns)
constexpr int GK = 16;      // reduction depth per staged tile
constexpr int GLD = 68;     // LDS row stride (floats): the two half-waves of an operand read hit disjoint banks
constexpr int MAX_MIX = 32; // experts one task mixes (PLE: specific + shared)

// ------------------------------------------------------------------ gather
struct GatherArgs {
    const float *user_tab, *item_tab, *dm;
    const int32_t *uid, *pid, *dom, *perm;
    const float* label;
    int64_t row_base, n_rows_split;
    int rows, rows_pad, n_user, n_item, n_domain;
    float* x;
    int ld;
    int32_t* domrow;
    float* y;
    int32_t *urow, *irow, *map_u, *map_i;      // trainable tables: row of each position, first position of each row
    const float *lin_u, *lin_i, *lin_d;         // NFM: 1-d linear tables (lin_u / lin_i null while frozen at their zero init)
    float* extra;                               // NFM: sum of the linear terms per position
    float* xt;                                  // AutoInt: compact [rows_pad][384] copy of x (token-major: row 3 b + t)
};
// one wave per batch position: lanes 0..31 copy the user row, 32..63 the item row, then lanes 0..31 the domain row
__global__ __launch_bounds__(256) void k_graph_gather(const GatherArgs a) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= a.rows_pad) return;
    float* xr = a.x + (size_t)r * a.ld;
    if (r >= a.rows) {          // padding rows: zeros in, nothing out (their d loss / d logit is zero)
        *reinterpret_cast<f32x4*>(xr + 4 * lane) = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (lane < 32) *reinterpret_cast<f32x4*>(xr + 2 * EMB + 4 * lane) = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.xt) {
            *reinterpret_cast<f32x4*>(a.xt + (size_t)r * XDIM + 4 * lane) = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (lane < 32) *reinterpret_cast<f32x4*>(a.xt + (size_t)r * XDIM + 2 * EMB + 4 * lane) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (lane == 0) {
            a.domrow[r] = -1;
            a.y[r] = 0.f;
            if (a.urow) { a.urow[r] = -1; a.irow[r] = -1; }
            if (a.extra) a.extra[r] = 0.f;
        }
        return;
    }
    int64_t src = a.perm ? (int64_t)a.perm[a.row_base + r] : a.row_base + r;
    src = src < 0 ? 0 : (src >= a.n_rows_split ? a.n_rows_split - 1 : src);
    int u = a.uid[src], it = a.pid[src], d = a.dom[src];
    u = u < 0 ? 0 : (u >= a.n_user ? a.n_user - 1 : u);
    it = it < 0 ? 0 : (it >= a.n_item ? a.n_item - 1 : it);
    d = d < 0 ? 0 : (d >= a.n_domain ? a.n_domain - 1 : d);
    const float* row = lane < 32 ? a.user_tab + (size_t)u * EMB + 4 * lane : a.item_tab + (size_t)it * EMB + 4 * (lane - 32);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(row);
    *reinterpret_cast<f32x4*>(xr + 4 * lane) = v0;
    if (a.xt) *reinterpret_cast<f32x4*>(a.xt + (size_t)r * XDIM + 4 * lane) = v0;
    if (lane < 32) {
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(a.dm + (size_t)d * EMB + 4 * lane);
        *reinterpret_cast<f32x4*>(xr + 2 * EMB + 4 * lane) = v1;
        if (a.xt) *reinterpret_cast<f32x4*>(a.xt + (size_t)r * XDIM + 2 * EMB + 4 * lane) = v1;
    }
    if (lane == 0) {
        a.domrow[r] = d;
        a.y[r] = a.label[src];
        if (a.extra) a.extra[r] = ((a.lin_u ? a.lin_u[u] : 0.f) + (a.lin_i ? a.lin_i[it] : 0.f)) + a.lin_d[d];
        if (a.urow) {       // representative of a table row = its smallest batch position (integer atomicMin: exact)
            a.urow[r] = u;
            a.irow[r] = it;
            atomicMin(a.map_u + u, r);
            atomicMin(a.map_i + it, r);
        }
    }
}

// ------------------------------------------------------------------ dense contractions
// MODE 0:  C[M x N] = A[M x K] . B[K x N]        + bias, relu, dropout           (layer forward)
// MODE 1:  C[M x N] = A[M x K] . B[N x K]^T      x the gate of gate_y, += C        (d input = dz . W^T)
// MODE 2:  C[M x N] = A[K x M]^T . B[K x N]                                        (dW = in^T . dz, K = batch rows)
// M, N multiples of 64, K a multiple of 16.  64x64 tile per workgroup, 4 waves own its 32x32 quadrants
// (`32x32x2`, A / B fragments read from LDS as one float per lane: [k][m] images, conflict-free).
struct GemmArgs {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int K;
    const float* bias; int relu;
    uint32_t drop_key, drop_thresh; float keep_scale; int n_cols; int use_dropout;
    const float* gate_y; int gate_ld; float gate_scale; int accumulate;
    // forward only: + sum_{j < n_xe} xe[row][j] * we[j][col] before the bias (PNN: the inner products feed the last three
    // rows of the first kernel, whose 384 + 3 rows are no multiple of the tile depth)
    const float* xe; int xe_ld; const float* we; int n_xe;
    // MODE 2 only: blockIdx.z owns K rows [z K, (z + 1) K) of A and B and writes its partial product at C + z zstride
    size_t zstride;
};
constexpr int MAX_GROUP = 32;
struct GroupTab {
    int n, tiles_y;
    int64_t a_off[MAX_GROUP], b_off[MAX_GROUP], c_off[MAX_GROUP];      // floats added to A / B / C
    int64_t bias_off[MAX_GROUP];                                       // ... to bias (MODE 0) or to the bias gradient (finish)
    int64_t gate_off[MAX_GROUP];                                       // ... to gate_y (MODE 1)
    uint32_t drop_key[MAX_GROUP];
};
// KCAT (MODE 1 only): C = sum over the group's members of A_e . B_e^T -- ONE contraction whose reduction index runs through
// every member (the first layers' d x, which adds up over the experts: d x = [dz_1 .. dz_E] . [W_1 .. W_E]^T)
template <int MODE, bool KCAT = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& a, const int bx, const int by, const int bz, const GroupTab* tab = nullptr) {
    __shared__ __attribute__((aligned(16))) float As[2][GK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK * GLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = by * GT, n0 = bx * GT;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    constexpr bool A_KC = MODE != 2;        // A's reduction index is the contiguous one in memory
    constexpr bool B_KC = MODE == 1;
    const int r4 = tid >> 2, k4 = (tid & 3) * 4;        // k-contiguous operand: row r4 of the tile, 4 k's
    const int kr = tid >> 4, c4 = (tid & 15) * 4;       // otherwise: k row kr, 4 columns
    f32x4 ra, rb;
    const float *A = a.A, *B = a.B;
    float* C = a.C;
    if (MODE == 2) {
        A += (size_t)bz * a.K * a.lda;
        B += (size_t)bz * a.K * a.ldb;
        C += (size_t)bz * a.zstride;
    }
    const int nk = a.K / GK;
    auto gload = [&](int kt) {
        int k0 = kt * GK;
        if (KCAT) {
            const int e = kt / nk;
            k0 = (kt - e * nk) * GK;
            A = a.A + tab->a_off[e];
            B = a.B + tab->b_off[e];
        }
        ra = A_KC ? *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r4) * a.lda + k0 + k4)
                  : *reinterpret_cast<const f32x4*>(A + (size_t)(k0 + kr) * a.lda + m0 + c4);
        rb = B_KC ? *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + r4) * a.ldb + k0 + k4)
                  : *reinterpret_cast<const f32x4*>(B + (size_t)(k0 + kr) * a.ldb + n0 + c4);
    };
    auto lstore = [&](int buf) {
        if (A_KC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) As[buf][(k4 + j) * GLD + r4] = ra[j];
        } else {
            *reinterpret_cast<f32x4*>(&As[buf][kr * GLD + c4]) = ra;
        }
        if (B_KC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs[buf][(k4 + j) * GLD + r4] = rb[j];
        } else {
            *reinterpret_cast<f32x4*>(&Bs[buf][kr * GLD + c4]) = rb;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nkt = KCAT ? nk * tab->n : nk;
    gload(0);
    lstore(0);
    __syncthreads();
    const int kk = lane >> 5, c = lane & 31;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const float* ap = &As[buf][kk * GLD + wm + c];
        const float* bp = &Bs[buf][kk * GLD + wn + c];
#pragma unroll
        for (int i = 0; i < GK / 2; ++i) acc = MAMDR_MFMA32(ap[2 * i * GLD], bp[2 * i * GLD], acc);
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }
    // D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = n0 + wn + c;
    const float bias = (MODE == 0 && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kk;
        float v = acc[r];
        if (MODE == 0) {
            for (int j = 0; j < a.n_xe; ++j) v = fmaf(a.xe[(size_t)row * a.xe_ld + j], a.we[(size_t)j * a.n_cols + col], v);
            v += bias;
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.use_dropout) {
                const uint32_t u = mamdr_dropout_u32(a.drop_key, (uint32_t)row * (uint32_t)a.n_cols + (uint32_t)col);
                v = u >= a.drop_thresh ? v * a.keep_scale : 0.f;
            }
        } else if (MODE == 1) {
            if (a.gate_y) v = a.gate_y[(size_t)row * a.gate_ld + col] > 0.f ? v * a.gate_scale : 0.f;
            if (a.accumulate) v += C[(size_t)row * a.ldc + col];
        }
        C[(size_t)row * a.ldc + col] = v;
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_graph_gemm(const GemmArgs a) {
    gemm_tile<MODE>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- the same contraction for a GROUP of problems of one shape in ONE launch: the experts a task mixes share their input
// and their layer shapes (deep_mtl_ctr.py:31-49: num_experts / specific + shared experts, every one DNN(hidden_dim)), so a
// layer of all of them is one grid -- 12 experts x 64 tiles instead of 12 launches of 64 workgroups on 256 CUs.
// MODE 0 / 1: blockIdx.z = the problem; MODE 2: blockIdx.y = problem x row tile (blockIdx.z stays the split of the rows).
__global__ __launch_bounds__(256) void k_graph_gemm_kcat(const GemmArgs a, const GroupTab t) {
    gemm_tile<1, true>(a, blockIdx.x, blockIdx.y, 0, &t);
}
template <int MODE>
__global__ __launch_bounds__(256) void k_graph_gemm_group(GemmArgs a, const GroupTab t) {
    int gi, by = blockIdx.y, bz = blockIdx.z;
    if (MODE == 2) {
        gi = (int)blockIdx.y / t.tiles_y;
        by = (int)blockIdx.y - gi * t.tiles_y;
    } else {
        gi = blockIdx.z;
        bz = 0;
    }
    a.A += t.a_off[gi];
    a.B += t.b_off[gi];
    a.C += t.c_off[gi];
    if (MODE == 0) {
        a.bias += t.bias_off[gi];
        a.drop_key = t.drop_key[gi];
    }
    if (MODE == 1 && a.gate_y) a.gate_y += t.gate_off[gi];
    gemm_tile<MODE>(a, blockIdx.x, by, bz);
}

// ---- the same contractions (MODE 0 / 1) on 32 x 32 tiles with the reduction index split over the workgroup's four waves.
// A 1,024-row layer of 512 / 256 / 128 / 64 units is 128 / 64 / 32 / 16 tiles of 64 x 64: half to a sixteenth of the 256
// CUs, each walking a serial chain of K / 2 `32x32x2` MFMAs per wave behind one staged 16-deep tile per round trip
// (12.6 us per launch on the Taobao-10 multi-task configs, profiles/r04g_*).  Here a workgroup owns a 32 x 32 tile
// (4 x the workgroups), stages T32_K = 128 reduction indices per round trip (4 + 4 float4 loads in flight per thread, the
// next stage requested before this one's MFMAs) and every wave contracts its own 32 of them into a full-tile accumulator:
// the MFMA chain per wave is K / 8 long.  The four partial tiles meet in LDS in wave order (fixed: bit-stable), and the
// epilogue runs on 4 neighbouring outputs per thread.  K needs no multiple of 128: the tail is staged as zeros.
constexpr int T32 = 32, T32_K = 128, T32_LD = 36;
template <int MODE, bool KCAT = false>
__device__ __forceinline__ void gemm_tile32(const GemmArgs& a, const int bx, const int by, const GroupTab* tab = nullptr) {
    __shared__ __attribute__((aligned(16))) float As[T32_K * T32_LD];
    __shared__ __attribute__((aligned(16))) float Bs[T32_K * T32_LD];
    static_assert(MODE == 0 || MODE == 1, "weight gradients keep the 64 x 64 tiles");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = by * T32, n0 = bx * T32;
    constexpr bool B_KC = MODE == 1;
    const int r8 = tid >> 3, q8 = tid & 7;          // k-contiguous operand: tile row r8, float4s q8 + 8 j of its 128 k's
    f32x4 ra[4], rb[4];
    const float *A = a.A, *B = a.B;
    const int nk = (a.K + T32_K - 1) / T32_K;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int kt) {
        int k0 = kt * T32_K;
        if (KCAT) {
            const int e = kt / nk;
            k0 = (kt - e * nk) * T32_K;
            A = a.A + tab->a_off[e];
            B = a.B + tab->b_off[e];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + 4 * (q8 + 8 * j);
            ra[j] = k < a.K ? *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r8) * a.lda + k) : zero;
            if (B_KC) {
                rb[j] = k < a.K ? *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + r8) * a.ldb + k) : zero;
            } else {            // B [K x N]: k row r8 + 32 j, float4 q8 of the tile's 32 columns
                const int kr = k0 + r8 + 32 * j;
                rb[j] = kr < a.K ? *reinterpret_cast<const f32x4*>(B + (size_t)kr * a.ldb + n0 + 4 * q8) : zero;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * (q8 + 8 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) As[(k + i) * T32_LD + r8] = ra[j][i];
            if (B_KC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Bs[(k + i) * T32_LD + r8] = rb[j][i];
            } else {
                *reinterpret_cast<f32x4*>(&Bs[(r8 + 32 * j) * T32_LD + 4 * q8]) = rb[j];
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nkt = KCAT ? nk * tab->n : nk;
    const int kk = lane >> 5, c = lane & 31;
    gload(0);
    for (int kt = 0; kt < nkt; ++kt) {
        lstore();
        __syncthreads();
        if (kt + 1 < nkt) gload(kt + 1);
        const float* ap = &As[(32 * w + kk) * T32_LD + c];
        const float* bp = &Bs[(32 * w + kk) * T32_LD + c];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = MAMDR_MFMA32(ap[2 * i * T32_LD], bp[2 * i * T32_LD], acc);
        __syncthreads();
    }
    // the waves' partial tiles -> LDS (D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    float* red = As;            // (the A stage is free after the loop's last barrier)
    static_assert(4 * 32 * 33 <= T32_K * T32_LD, "the partial tiles fit the A stage");
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * 33 + c] = acc[r];
    __syncthreads();
    const int row = m0 + r8, col = n0 + 4 * q8;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = red[(0 * 32 + r8) * 33 + 4 * q8 + j];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) t += red[(ww * 32 + r8) * 33 + 4 * q8 + j];
        v[j] = t;
    }
    float* cp = a.C + (size_t)row * a.ldc + col;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = v[j];
        if (MODE == 0) {
            for (int e = 0; e < a.n_xe; ++e) x = fmaf(a.xe[(size_t)row * a.xe_ld + e], a.we[(size_t)e * a.n_cols + col + j], x);
            if (a.bias) x += a.bias[col + j];
            if (a.relu) x = fmaxf(x, 0.f);
            if (a.use_dropout) {
                const uint32_t u = mamdr_dropout_u32(a.drop_key, (uint32_t)row * (uint32_t)a.n_cols + (uint32_t)(col + j));
                x = u >= a.drop_thresh ? x * a.keep_scale : 0.f;
            }
        } else {
            if (a.gate_y) x = a.gate_y[(size_t)row * a.gate_ld + col + j] > 0.f ? x * a.gate_scale : 0.f;
            if (a.accumulate) x += cp[j];
        }
        v[j] = x;
    }
    if ((reinterpret_cast<uintptr_t>(cp) & 15) == 0) {
        *reinterpret_cast<f32x4*>(cp) = v;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) cp[j] = v[j];
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_graph_gemm32(const GemmArgs a) {
    gemm_tile32<MODE>(a, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void k_graph_gemm32_kcat(const GemmArgs a, const GroupTab t) {
    gemm_tile32<1, true>(a, blockIdx.x, blockIdx.y, &t);
}
template <int MODE>
__global__ __launch_bounds__(256) void k_graph_gemm32_group(GemmArgs a, const GroupTab t) {
    const int gi = blockIdx.z;
    a.A += t.a_off[gi];
    a.B += t.b_off[gi];
    a.C += t.c_off[gi];
    if (MODE == 0) {
        a.bias += t.bias_off[gi];
        a.drop_key = t.drop_key[gi];
    }
    if (MODE == 1 && a.gate_y) a.gate_y += t.gate_off[gi];
    gemm_tile32<MODE>(a, blockIdx.x, blockIdx.y);
}

// db[n] = sum over the batch rows of dz[b][n]: 16 columns per workgroup, 16 row groups (8 loads in flight each) summed
// through LDS in a fixed order
constexpr int CS_COLS = 16, CS_GROUPS = 16;
__global__ __launch_bounds__(256) void k_graph_colsum(const float* dz, int ld, int rows, float* out, int n_valid) {
    __shared__ float red[CS_GROUPS][CS_COLS + 1];
    const int c = threadIdx.x & (CS_COLS - 1), g = threadIdx.x / CS_COLS;
    const int col = blockIdx.x * CS_COLS + c;
    float s = 0.f;
    if (col < n_valid) {
        const float* p = dz + col;
        int b = g;
        for (; b + 7 * CS_GROUPS < rows; b += 8 * CS_GROUPS) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(b + k * CS_GROUPS) * ld];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < rows; b += CS_GROUPS) s += p[(size_t)b * ld];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && col < n_valid) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CS_GROUPS; ++k) t += red[k][c];
        out[col] = t;
    }
}
// the end of a layer's weight gradient: workgroups [0, nb_red) add the split-K partial products of dW in a fixed order,
// the rest are the bias column sums above
__global__ __launch_bounds__(256) void k_graph_wfinish(const float* part, int n_split, size_t stride, int64_t n4, float* out,
                                                       int nb_red, const float* dz, int ld, int rows, float* db, int n_valid) {
    if ((int)blockIdx.x < nb_red) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= n4) return;
        f32x4 t = reinterpret_cast<const f32x4*>(part)[i];
        for (int z = 1; z < n_split; ++z) {
            const f32x4 v = reinterpret_cast<const f32x4*>(part + z * stride)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += v[k];
        }
        reinterpret_cast<f32x4*>(out)[i] = t;
        return;
    }
    __shared__ float red[CS_GROUPS][CS_COLS + 1];
    const int c = threadIdx.x & (CS_COLS - 1), g = threadIdx.x / CS_COLS;
    const int col = ((int)blockIdx.x - nb_red) * CS_COLS + c;
    float s = 0.f;
    if (col < n_valid) {
        const float* p = dz + col;
        int b = g;
        for (; b + 7 * CS_GROUPS < rows; b += 8 * CS_GROUPS) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(b + k * CS_GROUPS) * ld];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < rows; b += CS_GROUPS) s += p[(size_t)b * ld];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && col < n_valid) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CS_GROUPS; ++k) t += red[k][c];
        db[col] = t;
    }
}
// the same for a group of weight gradients (blockIdx.y = the problem): partial products at part + (y n_split + z) stride,
// outputs at out + c_off[y], bias column sums of dz + a_off[y] into db + bias_off[y]
__global__ __launch_bounds__(256) void k_graph_wfinish_group(const float* part, int n_split, size_t stride, int64_t n4, float* out,
                                                             int nb_red, const float* dz, int ld, int rows, float* db, int n_valid,
                                                             const GroupTab t) {
    const int gi = blockIdx.y;
    if ((int)blockIdx.x < nb_red) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= n4) return;
        const float* p0 = part + (size_t)gi * n_split * stride;
        f32x4 v0 = reinterpret_cast<const f32x4*>(p0)[i];
        for (int z = 1; z < n_split; ++z) {
            const f32x4 v = reinterpret_cast<const f32x4*>(p0 + z * stride)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) v0[k] += v[k];
        }
        reinterpret_cast<f32x4*>(out + t.c_off[gi])[i] = v0;
        return;
    }
    __shared__ float red[CS_GROUPS][CS_COLS + 1];
    const int c = threadIdx.x & (CS_COLS - 1), g = threadIdx.x / CS_COLS;
    const int col = ((int)blockIdx.x - nb_red) * CS_COLS + c;
    float s = 0.f;
    if (col < n_valid) {
        const float* p = dz + t.a_off[gi] + col;
        int b = g;
        for (; b + 7 * CS_GROUPS < rows; b += 8 * CS_GROUPS) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(b + k * CS_GROUPS) * ld];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < rows; b += CS_GROUPS) s += p[(size_t)b * ld];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && col < n_valid) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < CS_GROUPS; ++k) v += red[k][c];
        db[t.bias_off[gi] + col] = v;
    }
}
// ---- every weight gradient of a step in ONE launch.  dW_l = in_l^T dz_l needs nothing but the activations and the d z's,
// which stay in the workspaces until the step ends: the backward pass runs its chain of d-input contractions first and
// queues the weight gradients (host: queue_wgrad / flush_wgrads); here a flat grid walks the queue -- workgroup ->
// (problem, tile, split of the batch rows) through the prefix table -- so that 90 tiles of four layers (shared_bottom) or of
// an expert group + gate + tower share the 256 CUs instead of following one another in 2 launches per layer.
constexpr int MAX_WQ = 40;
struct WMulti {
    int n;
    int first[MAX_WQ + 1];          // first workgroup of problem p
    const float* A[MAX_WQ]; const float* B[MAX_WQ]; float* C[MAX_WQ];       // C: the partial products' base, or dW itself (split 1)
    int lda[MAX_WQ], ldb[MAX_WQ], N[MAX_WQ], tx[MAX_WQ], ty[MAX_WQ], K[MAX_WQ];     // K = batch rows per split
    int mn[MAX_WQ];                 // M x N (stride between the splits' partial products)
};
__global__ __launch_bounds__(256) void k_graph_wgrad_multi(const WMulti t) {
    int p = 0;
    while (p + 1 < t.n && (int)blockIdx.x >= t.first[p + 1]) ++p;
    int b = (int)blockIdx.x - t.first[p];
    const int tiles = t.tx[p] * t.ty[p];
    const int bz = b / tiles;
    b -= bz * tiles;
    const int by = b / t.tx[p], bx = b - by * t.tx[p];
    GemmArgs a;
    a.A = t.A[p];
    a.lda = t.lda[p];
    a.B = t.B[p];
    a.ldb = t.ldb[p];
    a.C = t.C[p];
    a.ldc = t.N[p];
    a.K = t.K[p];
    a.zstride = (size_t)t.mn[p];
    gemm_tile<2>(a, bx, by, bz);
}
// ... and their ends: per problem, the workgroups that add the splits' partial products in a fixed order, then the ones
// that sum the columns of d z into the bias gradient (k_graph_wfinish's two halves, the queue's problems side by side)
// where a finished gradient element goes: into the gradient vector (p null), or straight through the optimiser -- the
// step's k_graph_adam launch folded into the launch that finishes the gradients (the same arithmetic per element: opt_elem)
struct GradSink {
    const float* g_base;        // base of the gradient vector (element 0 = flat-vector element `table_floats`)
    float *p, *m, *v;           // parameters / slots addressed like g_base (m: the accumulator for MAMDR_OPT_ACCUMULATE)
    int optimizer;
    float alpha, omb1, omb2, eps;
};
__device__ __forceinline__ void opt_elem(const int optimizer, const float g, float& p, float& m, float& v, const float alpha,
                                         const float omb1, const float omb2, const float eps) {
    if (optimizer == MAMDR_OPT_SGD) {
        p = p - g * alpha;
    } else if (optimizer == MAMDR_OPT_ACCUMULATE) {
        m = m + g;
    } else {        // TF1 ApplyAdam: m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); p -= m alpha / (sqrt(v) + eps)
        m = m + (g - m) * omb1;
        v = v + (g * g - v) * omb2;
        p = p - (m * alpha) / (sqrtf(v) + eps);
    }
}
__device__ __forceinline__ void sink1(const GradSink& s, float* gptr, const float g) {
    if (!s.p) {
        *gptr = g;
        return;
    }
    const size_t i = (size_t)(gptr - s.g_base);
    float p = s.p[i], m = s.m[i], v = s.optimizer == MAMDR_OPT_ADAM ? s.v[i] : 0.f;
    opt_elem(s.optimizer, g, p, m, v, s.alpha, s.omb1, s.omb2, s.eps);
    if (s.optimizer != MAMDR_OPT_ACCUMULATE) s.p[i] = p;
    if (s.optimizer != MAMDR_OPT_SGD) s.m[i] = m;
    if (s.optimizer == MAMDR_OPT_ADAM) s.v[i] = v;
}
__device__ __forceinline__ void sink4(const GradSink& s, float* gptr, const f32x4 g) {
    if (!s.p) {
        *reinterpret_cast<f32x4*>(gptr) = g;
        return;
    }
    const size_t i = (size_t)(gptr - s.g_base);
    f32x4 p = *reinterpret_cast<const f32x4*>(s.p + i), m = *reinterpret_cast<const f32x4*>(s.m + i);
    f32x4 v = s.optimizer == MAMDR_OPT_ADAM ? *reinterpret_cast<const f32x4*>(s.v + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        opt_elem(s.optimizer, g[k], pk, mk, vk, s.alpha, s.omb1, s.omb2, s.eps);
        p[k] = pk; m[k] = mk; v[k] = vk;
    }
    if (s.optimizer != MAMDR_OPT_ACCUMULATE) *reinterpret_cast<f32x4*>(s.p + i) = p;
    if (s.optimizer != MAMDR_OPT_SGD) *reinterpret_cast<f32x4*>(s.m + i) = m;
    if (s.optimizer == MAMDR_OPT_ADAM) *reinterpret_cast<f32x4*>(s.v + i) = v;
}
struct WFinish {
    int n;
    int first[MAX_WQ + 1];
    const float* part[MAX_WQ]; float* out[MAX_WQ]; const float* dz[MAX_WQ]; float* db[MAX_WQ];
    int split[MAX_WQ], nb_red[MAX_WQ], N[MAX_WQ], rows[MAX_WQ], ld[MAX_WQ], mn[MAX_WQ];
};
__device__ __forceinline__ void wfinish_multi_body(const WFinish& t, const int bid, const GradSink& sk) {
    int p = 0;
    while (p + 1 < t.n && bid >= t.first[p + 1]) ++p;
    const int bi = bid - t.first[p];
    if (bi < t.nb_red[p]) {
        const int64_t i = (int64_t)bi * 256 + threadIdx.x;
        if (i >= t.mn[p] / 4) return;
        const float* part = t.part[p];
        const size_t stride = (size_t)t.mn[p];
        f32x4 v0 = reinterpret_cast<const f32x4*>(part)[i];
        for (int z = 1; z < t.split[p]; ++z) {
            const f32x4 v = reinterpret_cast<const f32x4*>(part + z * stride)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) v0[k] += v[k];
        }
        sink4(sk, t.out[p] + 4 * i, v0);
        return;
    }
    __shared__ float red[CS_GROUPS][CS_COLS + 1];
    const int c = threadIdx.x & (CS_COLS - 1), g = threadIdx.x / CS_COLS;
    const int col = (bi - t.nb_red[p]) * CS_COLS + c;
    const int rows = t.rows[p], ld = t.ld[p];
    float s = 0.f;
    if (col < t.N[p]) {
        const float* q = t.dz[p] + col;
        int b = g;
        for (; b + 7 * CS_GROUPS < rows; b += 8 * CS_GROUPS) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = q[(size_t)(b + k * CS_GROUPS) * ld];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < rows; b += CS_GROUPS) s += q[(size_t)b * ld];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && col < t.N[p]) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < CS_GROUPS; ++k) v += red[k][c];
        sink1(sk, t.db[p] + col, v);
    }
}
__global__ __launch_bounds__(256) void k_graph_wfinish_multi(const WFinish t) {
    GradSink none;
    none.p = nullptr;
    wfinish_multi_body(t, blockIdx.x, none);
}
// d x of a group's first layers: out[b][c] (+)= sum over the members of part[e][b][c], in member order
__global__ __launch_bounds__(256) void k_graph_dx_reduce(const float* part, int n_part, size_t stride, int rows, int n4_row,
                                                         float* out, int ld, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * n4_row) return;
    const int b = (int)(i / n4_row), c4 = (int)(i - (int64_t)b * n4_row);
    f32x4 v = reinterpret_cast<const f32x4*>(part)[i];
    for (int e = 1; e < n_part; ++e) {
        const f32x4 w = reinterpret_cast<const f32x4*>(part + e * stride)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += w[k];
    }
    float* o = out + (size_t)b * ld + 4 * c4;
    if (accumulate) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = w[k] + v[k];
    }
    *reinterpret_cast<f32x4*>(o) = v;
}
static void launch_colsum(hipStream_t s, const float* dz, int ld, int rows, float* out, int n_valid) {
    GLAUNCH(k_graph_colsum, dim3((n_valid + CS_COLS - 1) / CS_COLS), dim3(256), 0, s, dz, ld, rows, out, n_valid);
}

// out[j][e] = sum_b in[b][j] * d[b][e]  for narrow right-hand sides (gate kernel Wg: e < n_e <= 32; head: n_e = 1; the
// attention projections: n_e = 128): 16 outputs per workgroup, the rows split over 16 groups summed through LDS in order
__device__ __forceinline__ void small_tn_body(const float* in, int in_ld, const float* d, int d_ld, int rows, int n_j, int n_e,
                                              float* out, float* sum_out, const int bid, const int nblk, const GradSink& sk) {
    __shared__ float red[CS_GROUPS][CS_COLS + 1];
    if (sum_out && bid == nblk - 1) {       // one more workgroup: sum_out[0] = sum over the rows of d[b][0], fixed order
        float* r1 = &red[0][0];                         // (the head's d global bias beside its d kernel: one launch less)
        float s1 = 0.f;
        for (int b = threadIdx.x; b < rows; b += 256) s1 += d[(size_t)b * d_ld];
        r1[threadIdx.x] = s1;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) r1[threadIdx.x] += r1[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) sink1(sk, sum_out, r1[0]);
        return;
    }
    const int c = threadIdx.x & (CS_COLS - 1), g = threadIdx.x / CS_COLS;
    const int idx = bid * CS_COLS + c;
    float s = 0.f;
    if (idx < n_j * n_e) {
        const int j = idx / n_e, e = idx - j * n_e;
        const float *pi = in + j, *pd = d + e;
        int b = g;
        for (; b + 3 * CS_GROUPS < rows; b += 4 * CS_GROUPS) {
            float x[4], y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                x[k] = pi[(size_t)(b + k * CS_GROUPS) * in_ld];
                y[k] = pd[(size_t)(b + k * CS_GROUPS) * d_ld];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) s = fmaf(x[k], y[k], s);
        }
        for (; b < rows; b += CS_GROUPS) s = fmaf(pi[(size_t)b * in_ld], pd[(size_t)b * d_ld], s);
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && idx < n_j * n_e) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CS_GROUPS; ++k) t += red[k][c];
        sink1(sk, out + idx, t);
    }
}
__global__ __launch_bounds__(256) void k_graph_small_tn(const float* in, int in_ld, const float* d, int d_ld, int rows,
                                                        int n_j, int n_e, float* out, float* sum_out) {
    GradSink none;
    none.p = nullptr;
    small_tn_body(in, in_ld, d, d_ld, rows, n_j, n_e, out, sum_out, blockIdx.x, gridDim.x, none);
}
static void launch_small_tn(hipStream_t s, const float* in, int in_ld, const float* d, int d_ld, int rows, int n_j, int n_e,
                            float* out, float* sum_out = nullptr) {
    GLAUNCH(k_graph_small_tn, dim3((n_j * n_e + CS_COLS - 1) / CS_COLS + (sum_out ? 1 : 0)), dim3(256), 0, s, in, in_ld,
                       d, d_ld, rows, n_j, n_e, out, sum_out);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// ------------------------------------------------------------------ NFM / PNN: interactions of the three fields, one wave per row
// kind 0 (NFM): f[c] = u i + u d + i d (= 1/2 ((u+i+d)^2 - u^2 - i^2 - d^2), BiInteractionPooling), 128 columns
// kind 1 (PNN): f[0..2] = <u,i>, <u,d>, <i,d> (InnerProductLayer over the pairs (0,1), (0,2), (1,2))
// kind 2 (DeepFM): the FM second-order term  sum_k (u i + u d + i d)_k  joins the linear logit of the row (`extra`); no columns
struct FeatArgs {
    float* act; float* dact; int ld; int f_col; int rows_pad; int kind; int dx_all;
    const float* w_ip;      // PNN: rows 384..386 of the first kernel [3][n_out]
    int z_col, n_out;       // PNN: d z of the first layer
    float* extra;           // DeepFM: the per-row logit terms outside the DNN (linear part; the FM term is added here)
    const float* dlogit;    // DeepFM: d loss / d logit of the row
};
__global__ __launch_bounds__(256) void k_graph_feat_fwd(const FeatArgs a) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= a.rows_pad) return;
    float* row = a.act + (size_t)r * a.ld;
    const f32x2 u = *reinterpret_cast<const f32x2*>(row + 2 * lane), it = *reinterpret_cast<const f32x2*>(row + EMB + 2 * lane),
                d = *reinterpret_cast<const f32x2*>(row + 2 * EMB + 2 * lane);
    if (a.kind == 0) {
        f32x2 f;
#pragma unroll
        for (int k = 0; k < 2; ++k) f[k] = fmaf(it[k], d[k], fmaf(u[k], d[k], u[k] * it[k]));
        *reinterpret_cast<f32x2*>(row + a.f_col + 2 * lane) = f;
    } else if (a.kind == 2) {
        float f = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) f += fmaf(it[k], d[k], fmaf(u[k], d[k], u[k] * it[k]));
        f = wave_sum(f);
        if (lane == 0) a.extra[r] += f;
    } else {
        const float ui = wave_sum(fmaf(u[1], it[1], u[0] * it[0])), ud = wave_sum(fmaf(u[1], d[1], u[0] * d[0])),
                    id = wave_sum(fmaf(it[1], d[1], it[0] * d[0]));
        if (lane == 0) {
            row[a.f_col + 0] = ui;
            row[a.f_col + 1] = ud;
            row[a.f_col + 2] = id;
            row[a.f_col + 3] = 0.f;
        }
    }
}
// d f -> d x.  NFM: d u = df (i + d), d i = df (u + d), d d = df (u + i), WRITTEN (nothing else feeds d x).
// PNN: dip_j = sum_c dz[c] W0[384 + j][c]; d u += dip0 i + dip1 d, d i += dip0 u + dip2 d, d d += dip1 u + dip2 i, ADDED
// to what the first layer's d x = dz W0[0:384]^T left there.  User / item columns only when the tables train.
__global__ __launch_bounds__(256) void k_graph_feat_bwd(const FeatArgs a) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= a.rows_pad) return;
    const float* row = a.act + (size_t)r * a.ld;
    float* drow = a.dact + (size_t)r * a.ld;
    const f32x2 u = *reinterpret_cast<const f32x2*>(row + 2 * lane), it = *reinterpret_cast<const f32x2*>(row + EMB + 2 * lane),
                d = *reinterpret_cast<const f32x2*>(row + 2 * EMB + 2 * lane);
    f32x2 du, di, dd;
    if (a.kind == 0) {
        const f32x2 df = *reinterpret_cast<const f32x2*>(drow + a.f_col + 2 * lane);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            du[k] = df[k] * (it[k] + d[k]);
            di[k] = df[k] * (u[k] + d[k]);
            dd[k] = df[k] * (u[k] + it[k]);
        }
    } else if (a.kind == 2) {
        // DeepFM: d fm / d e_f = the sum of the other two fields' rows, ADDED to what the DNN's first layer left in d x
        const float dl = a.dlogit[r];
        const f32x2 du0 = *reinterpret_cast<const f32x2*>(drow + 2 * lane), di0 = *reinterpret_cast<const f32x2*>(drow + EMB + 2 * lane),
                    dd0 = *reinterpret_cast<const f32x2*>(drow + 2 * EMB + 2 * lane);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            du[k] = du0[k] + dl * (it[k] + d[k]);
            di[k] = di0[k] + dl * (u[k] + d[k]);
            dd[k] = dd0[k] + dl * (u[k] + it[k]);
        }
    } else {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
        for (int c = lane; c < a.n_out; c += 64) {
            const float z = drow[a.z_col + c];
            p0 = fmaf(z, a.w_ip[c], p0);
            p1 = fmaf(z, a.w_ip[a.n_out + c], p1);
            p2 = fmaf(z, a.w_ip[2 * a.n_out + c], p2);
        }
        p0 = wave_sum(p0);
        p1 = wave_sum(p1);
        p2 = wave_sum(p2);
        const f32x2 du0 = *reinterpret_cast<const f32x2*>(drow + 2 * lane), di0 = *reinterpret_cast<const f32x2*>(drow + EMB + 2 * lane),
                    dd0 = *reinterpret_cast<const f32x2*>(drow + 2 * EMB + 2 * lane);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            du[k] = du0[k] + (p0 * it[k] + p1 * d[k]);
            di[k] = di0[k] + (p0 * u[k] + p2 * d[k]);
            dd[k] = dd0[k] + (p1 * u[k] + p2 * it[k]);
        }
    }
    if (a.dx_all) {
        *reinterpret_cast<f32x2*>(drow + 2 * lane) = du;
        *reinterpret_cast<f32x2*>(drow + EMB + 2 * lane) = di;
    }
    *reinterpret_cast<f32x2*>(drow + 2 * EMB + 2 * lane) = dd;
}
// NFM's linear domain table: g[d] = sum over the batch rows of domain d of d loss / d logit  +  2 l2_lin w[d]
// one workgroup per domain: rows strided over the 256 threads, LDS tree in a fixed order
__device__ __forceinline__ void lin_domain_grad_body(const float* dlogit, const int32_t* domrow, int rows, const float* w, float two_l2,
                                                     float* g, const int d, const GradSink& sk) {
    __shared__ float red[256];
    float s = 0.f;
    for (int b = threadIdx.x; b < rows; b += 256) s += domrow[b] == d ? dlogit[b] : 0.f;
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sink1(sk, g + d, red[0] + two_l2 * w[d]);
}
__global__ __launch_bounds__(256) void k_graph_lin_domain_grad(const float* dlogit, const int32_t* domrow, int rows, const float* w,
                                                               float two_l2, int n_domain, float* g) {
    GradSink none;
    none.p = nullptr;
    lin_domain_grad_body(dlogit, domrow, rows, w, two_l2, g, blockIdx.x, none);
}
static void launch_lin_domain_grad(hipStream_t s, const float* dlogit, const int32_t* domrow, int rows, const float* w, float two_l2,
                                   int n_domain, float* g) {
    GLAUNCH(k_graph_lin_domain_grad, dim3(n_domain), dim3(256), 0, s, dlogit, domrow, rows, w, two_l2, n_domain, g);
}

// ------------------------------------------------------------------ CCPM: convolutions over the FIELD axis, one wave per row
// (deepctr CCPM with the three fields of this model, conv_kernel_width (6, 5), conv_filters (4, 4) -- oracle/fmnets.py:
// Conv2D((6, 1), 'same', tanh) over [3 fields x 128], maximum over the fields (KMaxPooling, k = 1), Conv2D((5, 1)) on the
// one row left = its centre tap, tanh; features [128 x 4]).  conv = [w1 6x4 | b1 4 | w2 4x4 (in, out) | b2 4] = 48 floats.
struct CcpmArgs {
    float* act; float* dact; int ld; int f_col, cg_col; int rows_pad; int dx_all;
    const float* conv;
};
// a1 of (position, filter) for the three raw values of one embedding column, its maximum and where it sits
__device__ __forceinline__ void ccpm_unit(const float* __restrict__ cv, const float (&x)[3], float (&m1)[4], int (&arg)[4],
                                          float (&a2)[4]) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        float best = -3.0e38f;
        int at = 0;
#pragma unroll
        for (int pos = 0; pos < 3; ++pos) {
            float pre = 0.f;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int src = pos + t - 2;        // TF 'same' for an even kernel: 2 taps before, 3 after
                if (src >= 0 && src < 3) pre += x[src] * cv[t * 4 + f];
            }
            const float a = tanhf(pre + cv[24 + f]);
            if (a > best) { best = a; at = pos; }  // (the first maximum wins a tie, as numpy's argmax)
        }
        m1[f] = best;
        arg[f] = at;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float pre = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) pre += m1[f] * cv[28 + f * 4 + o];
        a2[o] = tanhf(pre + cv[44 + o]);
    }
}
__global__ __launch_bounds__(256) void k_graph_ccpm_fwd(const CcpmArgs a) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= a.rows_pad) return;
    float* row = a.act + (size_t)r * a.ld;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = 2 * lane + q;
        const float x[3] = {row[e], row[EMB + e], row[2 * EMB + e]};
        float m1[4], a2[4];
        int arg[4];
        ccpm_unit(a.conv, x, m1, arg, a2);
        *reinterpret_cast<f32x4*>(row + a.f_col + 4 * e) = (f32x4){a2[0], a2[1], a2[2], a2[3]};
    }
}
// d features -> d x and the row's share of the 48 convolution gradients (summed over the batch by k_graph_colsum)
__global__ __launch_bounds__(256) void k_graph_ccpm_bwd(const CcpmArgs a) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= a.rows_pad) return;
    const float* row = a.act + (size_t)r * a.ld;
    float* drow = a.dact + (size_t)r * a.ld;
    const float* cv = a.conv;
    float gc[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) gc[k] = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = 2 * lane + q;
        const float x[3] = {row[e], row[EMB + e], row[2 * EMB + e]};
        float m1[4], a2[4], dx[3] = {0.f, 0.f, 0.f};
        int arg[4];
        ccpm_unit(cv, x, m1, arg, a2);
        const f32x4 df = *reinterpret_cast<const f32x4*>(drow + a.f_col + 4 * e);
        float dm1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float dp2 = df[o] * (1.0f - a2[o] * a2[o]);
            gc[44 + o] += dp2;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                gc[28 + f * 4 + o] += m1[f] * dp2;
                dm1[f] += dp2 * cv[28 + f * 4 + o];
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float dp1 = dm1[f] * (1.0f - m1[f] * m1[f]);      // m1 = a1 at its maximum
            gc[24 + f] += dp1;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
#pragma unroll
                for (int pos = 0; pos < 3; ++pos) {
                    const int src = pos + t - 2;
                    if (src >= 0 && src < 3 && pos == arg[f]) {
                        gc[t * 4 + f] += x[src] * dp1;
                        dx[src] += dp1 * cv[t * 4 + f];
                    }
                }
            }
        }
        if (a.dx_all) {
            drow[e] = dx[0];
            drow[EMB + e] = dx[1];
        }
        drow[2 * EMB + e] = dx[2];
    }
#pragma unroll
    for (int k = 0; k < 48; ++k) {
        const float v = wave_sum(gc[k]);
        if (lane == 0) drow[a.cg_col + k] = v;
    }
}

// ------------------------------------------------------------------ AutoInt: multi-head self-attention over the three fields
// (deepctr InteractingLayer, att_embedding_size 8, 4 heads, residual, no scaling -- oracle/fmnets.py).  Token-major compact
// buffers: row 3 b + t = field t of batch row b.  Per layer: P = X W (k_graph_gemm, W = [W_query | W_key | W_value | W_res],
// [d][128]) -> per batch row (one wave): scores Q_h K_h^T [3 x 3] per head, softmax over the fields, A V, + R, relu.
constexpr int ATT_OUT = 32, ATT_DIM = 8, ATT_P = 4 * ATT_OUT;      // 4 heads x 8; 128 projection columns per token
struct AttArgs {
    const float* P;      // [3 rows_pad][128]
    float* A;            // [rows_pad][36]: probabilities [head][field][field]
    float* Y;            // [3 rows_pad][32]
    const float* dY;     // [3 rows_pad][32] (backward)
    float* dP;           // [3 rows_pad][128] (backward)
    float* top; int top_ld;      // forward, last layer: also [rows_pad][96] inside the activation workspace (for the head)
    const float* dtop; int dtop_ld;   // backward, last layer: d Y comes from the gradient workspace
    int rows_pad;
};
__global__ __launch_bounds__(256) void k_graph_att_fwd(const AttArgs a) {
    __shared__ float lds[4][3 * ATT_P + 48];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + w;
    if (r >= a.rows_pad) return;
    float* p = lds[w];
    float* sc = p + 3 * ATT_P;
    for (int i = lane; i < 3 * ATT_P; i += 64) p[i] = a.P[(size_t)r * 3 * ATT_P + i];
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) {
        const int h = lane / 9, t = (lane % 9) / 3, s_ = lane % 3;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < ATT_DIM; ++j) v = fmaf(p[t * ATT_P + h * ATT_DIM + j], p[s_ * ATT_P + ATT_OUT + h * ATT_DIM + j], v);
        sc[lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) {
        const int base = lane - lane % 3;
        const float m = fmaxf(fmaxf(sc[base], sc[base + 1]), sc[base + 2]);
        const float e0 = __expf(sc[base] - m), e1 = __expf(sc[base + 1] - m), e2 = __expf(sc[base + 2] - m);
        const float mine = __expf(sc[lane] - m) / ((e0 + e1) + e2);
        a.A[(size_t)r * 36 + lane] = mine;
        __builtin_amdgcn_wave_barrier();
        sc[lane] = mine;
    } else {
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 3 * ATT_OUT; i += 64) {
        const int t = i / ATT_OUT, c = i % ATT_OUT, h = c / ATT_DIM;
        float o = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) o = fmaf(sc[h * 9 + t * 3 + s_], p[s_ * ATT_P + 2 * ATT_OUT + c], o);
        const float y = fmaxf(o + p[t * ATT_P + 3 * ATT_OUT + c], 0.f);
        a.Y[(size_t)r * 3 * ATT_OUT + i] = y;
        if (a.top) a.top[(size_t)r * a.top_ld + i] = y;
    }
}
__global__ __launch_bounds__(256) void k_graph_att_bwd(const AttArgs a) {
    __shared__ float lds[4][3 * ATT_P + 48 + 48 + 96];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + w;
    if (r >= a.rows_pad) return;
    float* p = lds[w];
    float* pa = p + 3 * ATT_P;       // probabilities
    float* ds = pa + 48;             // d scores
    float* dz = ds + 48;             // d (O + R) = d Y through the relu
    for (int i = lane; i < 3 * ATT_P; i += 64) p[i] = a.P[(size_t)r * 3 * ATT_P + i];
    if (lane < 36) pa[lane] = a.A[(size_t)r * 36 + lane];
    for (int i = lane; i < 3 * ATT_OUT; i += 64) {
        const float dy = a.dtop ? a.dtop[(size_t)r * a.dtop_ld + i] : a.dY[(size_t)r * 3 * ATT_OUT + i];
        dz[i] = a.Y[(size_t)r * 3 * ATT_OUT + i] > 0.f ? dy : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    float da = 0.f;
    if (lane < 36) {
        const int h = lane / 9, t = (lane % 9) / 3, s_ = lane % 3;
#pragma unroll
        for (int j = 0; j < ATT_DIM; ++j) da = fmaf(dz[t * ATT_OUT + h * ATT_DIM + j], p[s_ * ATT_P + 2 * ATT_OUT + h * ATT_DIM + j], da);
        ds[lane] = da * pa[lane];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) {
        const int base = lane - lane % 3;
        const float dot = (ds[base] + ds[base + 1]) + ds[base + 2];
        const float v = pa[lane] * (da - dot);
        __builtin_amdgcn_wave_barrier();
        ds[lane] = v;
    } else {
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    float* dp = a.dP + (size_t)r * 3 * ATT_P;
    for (int i = lane; i < 3 * ATT_OUT; i += 64) {
        const int t = i / ATT_OUT, c = i % ATT_OUT, h = c / ATT_DIM;
        float dq = 0.f, dk = 0.f, dv = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            dq = fmaf(ds[h * 9 + t * 3 + s_], p[s_ * ATT_P + ATT_OUT + c], dq);           // d Q[t] = sum_s dS[t][s] K[s]
            dk = fmaf(ds[h * 9 + s_ * 3 + t], p[s_ * ATT_P + c], dk);                      // d K[t] = sum_s dS[s][t] Q[s]
            dv = fmaf(pa[h * 9 + s_ * 3 + t], dz[s_ * ATT_OUT + c], dv);                   // d V[t] = sum_s A[s][t] dZ[s]
        }
        dp[t * ATT_P + c] = dq;
        dp[t * ATT_P + ATT_OUT + c] = dk;
        dp[t * ATT_P + 2 * ATT_OUT + c] = dv;
        dp[t * ATT_P + 3 * ATT_OUT + c] = dz[i];
    }
}
// out[r][k] = sum_c d[r][c] W[k][c] for a NARROW result (k < n_k <= 32; c < n_c): d X of the 32-wide attention layers
__global__ __launch_bounds__(256) void k_graph_small_nt(const float* d, int n_c, const float* W, int n_k, int rows, float* out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * n_k) return;
    const int r = idx / n_k, k = idx - r * n_k;
    const f32x4* dr = reinterpret_cast<const f32x4*>(d + (size_t)r * n_c);       // n_c is a multiple of 4 (128)
    const f32x4* wr = reinterpret_cast<const f32x4*>(W + (size_t)k * n_c);
    float s = 0.f;
    for (int c4 = 0; c4 < n_c / 4; c4 += 4) {           // 16-B loads, eight in flight; the chain keeps its order
        f32x4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x[u] = dr[c4 + u];
            y[u] = wr[c4 + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) s = fmaf(x[u][q], y[u][q], s);
    }
    out[idx] = s;
}
// d x[:, first .. first + n) (+)= compact [rows][384] columns first .. first + n
__global__ __launch_bounds__(256) void k_graph_add_x(float* dact, int ld, const float* src, int rows, int first, int n) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * n) return;
    const int r = idx / n, c = first + (idx - r * n);
    dact[(size_t)r * ld + c] += src[(size_t)r * XDIM + c];
}

// ------------------------------------------------------------------ gate: softmax(q Wg) and the mixture, one wave per row
struct GateArgs {
    float* act; float* dact; int ld;
    int q_col, n_q;             // gate DNN output
    const float* wg; int n_e;   // [n_q][n_e]
    int e_col[MAX_MIX];         // last-layer output of each mixed expert
    int n_h;                    // expert width
    int g_col, m_col;           // gate probabilities (n_e columns), mixture (n_h columns)
    int rows_pad;
    float gate_scale;           // 1 / keep of the dropout behind every DNN layer (1 in inference)
};
// One workgroup per batch row (4 waves: the gate logits / the d gate sums are dealt over the waves, the expert width over all
// 256 lanes) -- with one wave per row the 1,024 rows of a batch were 1,024 waves on 1,024 SIMDs, each walking its experts'
// reductions one after the other (23 us for 12 experts).  Same summation orders as that form: bit-identical results.
__global__ __launch_bounds__(256) void k_graph_gate_fwd(const GateArgs a) {
    __shared__ float sl[MAX_MIX];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* row = a.act + (size_t)r * a.ld;
    for (int e = w; e < a.n_e; e += 4) {
        float s = 0.f;
        for (int j = lane; j < a.n_q; j += 64) s = fmaf(row[a.q_col + j], a.wg[j * a.n_e + e], s);
        s = wave_sum(s);
        if (lane == 0) sl[e] = s;
    }
    __syncthreads();
    float mx = -3.0e38f;
    for (int e = 0; e < a.n_e; ++e) mx = fmaxf(mx, sl[e]);
    float den = 0.f;
    for (int e = 0; e < a.n_e; ++e) den += __expf(sl[e] - mx);
    if (tid < a.n_e) row[a.g_col + tid] = __expf(sl[tid] - mx) / den;
    for (int cidx = tid; cidx < a.n_h; cidx += 256) {
        float m = 0.f;
        for (int e = 0; e < a.n_e; ++e) m += (__expf(sl[e] - mx) / den) * row[a.e_col[e] + cidx];
        row[a.m_col + cidx] = m;
    }
}
// d mixture -> d expert outputs (times their relu / dropout gate = d z of the experts' last layers), d gate logits
// (kept: dWg = q^T dgl) and d q (times q's gate = d z of the gate DNN's last layer)
__global__ __launch_bounds__(256) void k_graph_gate_bwd(const GateArgs a) {
    __shared__ float sdg[MAX_MIX], sgp[MAX_MIX];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* row = a.act + (size_t)r * a.ld;
    float* drow = a.dact + (size_t)r * a.ld;
    for (int e = w; e < a.n_e; e += 4) {
        float t = 0.f;
        for (int cidx = lane; cidx < a.n_h; cidx += 64) t = fmaf(drow[a.m_col + cidx], row[a.e_col[e] + cidx], t);
        t = wave_sum(t);
        if (lane == 0) {
            sdg[e] = t;
            sgp[e] = row[a.g_col + e];
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int e = 0; e < a.n_e; ++e) s = fmaf(sgp[e], sdg[e], s);
    for (int cidx = tid; cidx < a.n_h; cidx += 256) {
        const float dm = drow[a.m_col + cidx];
        for (int e = 0; e < a.n_e; ++e) {
            const float h = row[a.e_col[e] + cidx];
            drow[a.e_col[e] + cidx] = h > 0.f ? (sgp[e] * dm) * a.gate_scale : 0.f;
        }
    }
    __syncthreads();        // (every read of d mixture is done before g_col / q_col, which may neighbour it, are written)
    if (tid < a.n_e) drow[a.g_col + tid] = sgp[tid] * (sdg[tid] - s);        // d gate logit
    for (int j = tid; j < a.n_q; j += 256) {
        float v = 0.f;
        for (int e = 0; e < a.n_e; ++e) v = fmaf(sgp[e] * (sdg[e] - s), a.wg[j * a.n_e + e], v);
        drow[a.q_col + j] = row[a.q_col + j] > 0.f ? v * a.gate_scale : 0.f;
    }
}

// ------------------------------------------------------------------ head: Dense(1, no bias) + global bias, sigmoid, Keras BCE
struct HeadArgs {
    const float* act; float* dact; int ld;
    int t_col, n_t;
    int n_plain;                // leading columns that no dropout follows (AutoInt: the attention output): gate only
    const float* w; const float* gb;
    const float* y; int rows, rows_pad;
    const float* extra;         // NFM: the linear logit of every position (null otherwise)
    float* dlogit; float* rowloss;
    int train; float gate_scale;
    const float* thresholds; uint32_t* hist; float* pred_out;       // inference
    const float* log_var; const int32_t* domrow;    // weighted loss (training only): d logit scaled by 1 / var^2, var =
                                                    // log_var[domain of the batch's FIRST row] (weighted_loss.py:38-41)
};
__global__ __launch_bounds__(256) void k_graph_head(const HeadArgs a) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= a.rows_pad) return;
    const float* row = a.act + (size_t)r * a.ld;
    float s = 0.f;
    for (int cidx = lane; cidx < a.n_t; cidx += 64) s = fmaf(row[a.t_col + cidx], a.w[cidx], s);
    const float logit = wave_sum(s) + a.gb[0] + (a.extra ? a.extra[r] : 0.f);
    float p;
    if (logit >= 0.f) {
        p = 1.0f / (1.0f + __expf(-logit));
    } else {
        const float ez = __expf(logit);
        p = ez / (1.0f + ez);
    }
    const bool valid = r < a.rows;
    const float y = a.y[r];
    const float lo = 1e-7f, hi = 1.0f - 1e-7f;
    const float pc = fminf(fmaxf(p, lo), hi);
    const float zc = __logf(pc / (1.0f - pc));
    const float loss = fmaxf(zc, 0.f) - zc * y + __logf(1.0f + __expf(-fabsf(zc)));
    if (lane == 0) a.rowloss[r] = valid ? loss : 0.f;
    if (a.train) {
        const float inside = (p >= lo && p <= hi) ? 1.0f : 0.0f;
        float dl = valid ? ((p - y) * inside) / (float)a.rows : 0.f;
        if (a.log_var) {
            const float var = a.log_var[a.domrow[0]];
            dl *= 1.0f / (var * var);
        }
        if (lane == 0) a.dlogit[r] = dl;
        float* drow = a.dact + (size_t)r * a.ld;
        for (int cidx = lane; cidx < a.n_t; cidx += 64)
            drow[a.t_col + cidx] = row[a.t_col + cidx] > 0.f ? (dl * a.w[cidx]) * (cidx < a.n_plain ? 1.0f : a.gate_scale) : 0.f;
    } else if (lane == 0 && valid) {
        int blo = 0, bhi = 500;          // AUC bin = number of thresholds strictly below p (utils/metrics_utils.py:309)
        while (blo < bhi) {
            const int mid = (blo + bhi) >> 1;
            if (a.thresholds[mid] < p) blo = mid + 1; else bhi = mid;
        }
        atomicAdd(a.hist + (y != 0.f ? 501 : 0) + blo, 1u);
        if (a.pred_out) a.pred_out[r] = p;
    }
}

// batch loss = mean of the rows' BCE + l2 (sum of squares of the three tables); one workgroup, fixed order.
// mode 0: out[0] = loss;  mode 1 (evaluation): out[0] += loss
__global__ __launch_bounds__(256) void k_graph_loss(const float* rowloss, int rows, const float* dm, int dm_count, float l2,
                                                    const float* frozen_sumsq, float* out, int mode, const float* lin_d,
                                                    int n_lin_d, float l2_lin, const float* log_var = nullptr,
                                                    const int32_t* domrow = nullptr, float* g_log_var = nullptr,
                                                    int n_domain = 0) {
    __shared__ float red[256];
    float s = 0.f, q = 0.f;
    for (int b = threadIdx.x; b < rows; b += 256) s += rowloss[b];
    for (int e = threadIdx.x; e < dm_count; e += 256) q = fmaf(dm[e], dm[e], q);
    float ql = 0.f;         // NFM: l2_lin * (sum of squares of the three 1-d linear tables)
    if (threadIdx.x == 0 && lin_d) {
        for (int e = 0; e < n_lin_d; ++e) ql = fmaf(lin_d[e], lin_d[e], ql);
        ql = l2_lin * ((frozen_sumsq[2] + frozen_sumsq[3]) + ql);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float tot = red[0];
    __syncthreads();
    red[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float data = tot / (float)rows;
        if (log_var) {      // weighted_loss.py:30-35: mean(BCE / var^2 + log var); d / d var = -2 mean(BCE) / var^3 + 1 / var
            const int d0 = domrow[0];
            const float var = log_var[d0];
            for (int d = 0; d < n_domain; ++d) g_log_var[d] = d == d0 ? -2.0f * data / (var * var * var) + 1.0f / var : 0.f;
            data = (1.0f / (var * var)) * data + logf(var);
        }
        const float loss = data + l2 * ((frozen_sumsq[0] + frozen_sumsq[1]) + red[0]) + ql;
        out[0] = mode ? out[0] + loss : loss;
    }
}
__global__ void k_graph_scale(float* x, float s) { x[0] *= s; }
// ------------------------------------------------------------------ domain table gradient
// g[d][c] = sum over the batch rows of domain d of d x[b][256 + c]  +  2 l2 Dm[d][c]   (rows in batch order)
// grid (8 column blocks, domains): 16 columns x 16 row groups per workgroup, 8 loads in flight, summed through LDS in a
// fixed order; a domain with no row in the batch (all but one of them in a domain step) leaves after one look at the ids.
__device__ __forceinline__ void domain_grad_body(const float* dx, int ld, int x_col, const int32_t* domrow, int rows,
                                                 const float* dm, float two_l2, float* g, const int bx, const int by,
                                                 const GradSink& sk) {
    __shared__ float red[CS_GROUPS][CS_COLS + 1];
    const int d = by, c = threadIdx.x & (CS_COLS - 1), rg = threadIdx.x / CS_COLS;
    const int col = bx * CS_COLS + c;
    int mine = 0;
    for (int b = threadIdx.x; b < rows; b += 256) mine |= domrow[b] == d;
    if (!__syncthreads_or(mine)) {
        if (rg == 0) sink1(sk, g + d * EMB + col, two_l2 * dm[d * EMB + col]);
        return;
    }
    const float* p = dx + x_col + col;
    float s = 0.f;
    int b = rg;
    for (; b + 7 * CS_GROUPS < rows; b += 8 * CS_GROUPS) {
        float v[8];
        int id[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            id[k] = domrow[b + k * CS_GROUPS];
            v[k] = p[(size_t)(b + k * CS_GROUPS) * ld];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += id[k] == d ? v[k] : 0.f;
    }
    for (; b < rows; b += CS_GROUPS) s += domrow[b] == d ? p[(size_t)b * ld] : 0.f;
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CS_GROUPS; ++k) t += red[k][c];
        sink1(sk, g + d * EMB + col, t + two_l2 * dm[d * EMB + col]);
    }
}
__global__ __launch_bounds__(256) void k_graph_domain_grad(const float* dx, int ld, int x_col, const int32_t* domrow, int rows,
                                                           const float* dm, float two_l2, float* g) {
    GradSink none;
    none.p = nullptr;
    domain_grad_body(dx, ld, x_col, domrow, rows, dm, two_l2, g, blockIdx.x, blockIdx.y, none);
}
// ---- the tail of a step's backward pass in ONE launch: the ends of the queued weight gradients (k_graph_wfinish_multi's
// workgroups), the narrow contractions that were queued with them (head / gate kernels, PNN's inner-product rows, the
// attention projections: k_graph_small_tn's workgroups) and the domain table's gradient (k_graph_domain_grad's) -- all of
// them read what the backward pass left in the workspaces and write gradients nothing but the optimiser reads.
constexpr int MAX_TQ = 6;
struct TailJobs {
    int n_tn;
    int tn_first[MAX_TQ + 1];       // first workgroup (behind the weight gradients' ends) of narrow contraction q
    const float* in[MAX_TQ]; const float* d[MAX_TQ]; float* out[MAX_TQ]; float* sum_out[MAX_TQ];
    int in_ld[MAX_TQ], d_ld[MAX_TQ], rows[MAX_TQ], n_j[MAX_TQ], n_e[MAX_TQ];
    int dg_first, dg_blocks;        // the domain table's gradient: 8 column blocks x n_domain (0 blocks: not in this launch)
    const float* dx; int ld, x_col; const int32_t* domrow; int dg_rows; const float* dm; float two_l2; float* g_dm;
    int lin_blocks;                 // the 1-d linear domain table's gradient (k_graph_lin_domain_grad's workgroups: one per domain)
    const float* lin_dlogit; const float* lin_w; float lin_two_l2; float* lin_g;
    GradSink sink;                  // p non-null: every finished gradient element steps its parameter right here
};
__global__ __launch_bounds__(256) void k_graph_tail(const WFinish f, const TailJobs j) {
    const int nf = f.n ? f.first[f.n] : 0;
    int bid = blockIdx.x;
    if (bid < nf) {
        wfinish_multi_body(f, bid, j.sink);
        return;
    }
    bid -= nf;
    if (bid < j.dg_first) {
        int q = 0;
        while (q + 1 < j.n_tn && bid >= j.tn_first[q + 1]) ++q;
        small_tn_body(j.in[q], j.in_ld[q], j.d[q], j.d_ld[q], j.rows[q], j.n_j[q], j.n_e[q], j.out[q], j.sum_out[q],
                      bid - j.tn_first[q], j.tn_first[q + 1] - j.tn_first[q], j.sink);
        return;
    }
    bid -= j.dg_first;
    if (bid < j.dg_blocks) {
        domain_grad_body(j.dx, j.ld, j.x_col, j.domrow, j.dg_rows, j.dm, j.two_l2, j.g_dm, bid % (EMB / CS_COLS), bid / (EMB / CS_COLS),
                         j.sink);
        return;
    }
    bid -= j.dg_blocks;
    if (bid < j.lin_blocks) lin_domain_grad_body(j.lin_dlogit, j.domrow, j.dg_rows, j.lin_w, j.lin_two_l2, j.lin_g, bid, j.sink);
}

// ------------------------------------------------------------------ optimiser on a range of the flat vector
struct AdamArgs {           // up to two ranges of the flat vector in one launch (the shared block and the task's block)
    float *p, *m, *v;       // bases of the flat vector / its slots (m: the accumulator for MAMDR_OPT_ACCUMULATE)
    const float* g;         // gradient base, addressed like p
    int64_t off4[2], n4[2]; // float4 offsets / counts of the ranges
    int optimizer;          // MAMDR_OPT_ADAM / MAMDR_OPT_SGD / MAMDR_OPT_ACCUMULATE (m += g, nothing else)
    float alpha, omb1, omb2, eps;
};
__global__ __launch_bounds__(256) void k_graph_adam(const AdamArgs a) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n4[0]) i += a.off4[0];
    else if (i - a.n4[0] < a.n4[1]) i = i - a.n4[0] + a.off4[1];
    else return;
    f32x4 p = reinterpret_cast<const f32x4*>(a.p)[i];
    const f32x4 g = reinterpret_cast<const f32x4*>(a.g)[i];
    f32x4 m = a.optimizer != MAMDR_OPT_SGD ? reinterpret_cast<const f32x4*>(a.m)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 v = a.optimizer == MAMDR_OPT_ADAM ? reinterpret_cast<const f32x4*>(a.v)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        opt_elem(a.optimizer, g[k], pk, mk, vk, a.alpha, a.omb1, a.omb2, a.eps);
        p[k] = pk; m[k] = mk; v[k] = vk;
    }
    if (a.optimizer != MAMDR_OPT_ACCUMULATE) reinterpret_cast<f32x4*>(a.p)[i] = p;
    if (a.optimizer != MAMDR_OPT_SGD) reinterpret_cast<f32x4*>(a.m)[i] = m;
    if (a.optimizer == MAMDR_OPT_ADAM) reinterpret_cast<f32x4*>(a.v)[i] = v;
}

// ------------------------------------------------------------------ host side: structure of the tower
struct Layer {
    int64_t w_off, b_off;
    int in, out;
    uint32_t id;            // dropout stream id = position among all DNN kernels in flat-vector order
};
struct Dnn {
    std::string name;
    std::vector<Layer> layers;
    int in_dim;
};
struct TensorInfo {
    std::string name;
    int64_t off, rows, cols;
};
struct Task {
    std::vector<int> mix;   // indices into dnns: the experts this task mixes (specific ones first, then the shared ones)
    int gate = -1;          // gate DNN
    int64_t wg_off = 0;
    int tower = -1;
    int64_t head_w = 0, head_gb = 0;
    int64_t blk_off = 0, blk_end = 0;
    // column plan of the activation / gradient workspaces for this task's path
    std::vector<std::vector<int>> col;  // col[dnn index in `path`][layer] -> first column of that layer's output
    std::vector<int> path;              // dnns on the path: mix..., gate, tower
    int g_col = 0, m_col = 0, n_cols = 0;
};
struct SplitData {
    const int32_t *uid = nullptr, *pid = nullptr, *dom = nullptr;
    const float* label = nullptr;
    int64_t n = 0;
    bool bound = false;     // an empty split (n = 0) is bound too
};

}  // namespace

struct mamdr_graph {
    mamdr_graph_config cfg;
    hipStream_t stream = nullptr;
    std::vector<Dnn> dnns;
    std::vector<Task> tasks;
    std::vector<TensorInfo> tensors;
    int64_t n_params = 0, dm_off = 0, shared_end = 0;
    bool gated = false;
    int n_h = 0;                // expert width
    // single-output towers of the deepctr family on the same layers (NFM, PNN): ONE task serves every domain
    bool single = false;
    bool has_lin = false;       // NFM / CCPM / AutoInt: the three 1-d linear tables (deepctr get_linear_logit)
    int f_col = 0;              // interaction features: NFM 128 columns, PNN 3 (+ 1 pad), CCPM 512
    int cg_col = 0;             // CCPM: 64 columns of the gradient workspace for the rows' shares of the 48 conv gradients
    int64_t conv_off = 0;       // CCPM: [w1 6x4 | b1 | w2 4x4 | b2]
    // AutoInt: three attention layers on compact token-major buffers (row 3 b + t = field t of batch row b)
    int64_t att_w[3] = {0, 0, 0};
    int top_col = 0;            // [attention output 96 | last DNN layer] contiguous in the activation workspace (head input)
    float *xt = nullptr, *dxt = nullptr;
    float *attP[3] = {nullptr, nullptr, nullptr}, *attdP[3] = {nullptr, nullptr, nullptr};
    float *attA[3] = {nullptr, nullptr, nullptr}, *attY[3] = {nullptr, nullptr, nullptr}, *attdY[3] = {nullptr, nullptr, nullptr};
    int64_t lin_d_off = 0;      // NFM: 1-d linear table of the domain feature (behind the global bias)
    int64_t lv_off = -1;        // weighted loss: `log_var` [n_domain], the last tensor (-1: plain loss)
    int64_t lin_u_off = 0, lin_i_off = 0;       // ... of the user / item features (trainable tables only)
    float *extra = nullptr, *glin_u = nullptr, *glin_i = nullptr;
    // bound state
    float *params = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    float* accum = nullptr;     // meta-gradient accumulator of MAMDR_OPT_ACCUMULATE steps (mamdr_graph_bind_accumulator)
    const float *user_tab = nullptr, *item_tab = nullptr;
    std::vector<SplitData> data;
    int64_t adam_t = 0;
    float b1p = 1.f, b2p = 1.f;
    uint32_t global_step = 0;
    // workspace
    int rows_pad_max = 0, ld = 0;
    float *act = nullptr, *dact = nullptr, *grad = nullptr, *dlogit = nullptr, *rowloss = nullptr, *y = nullptr;
    float* wpart = nullptr;     // split-K partial products of one weight gradient (launch_wgrad)
    size_t wpart_floats = 0;
    float* dxpart = nullptr;    // the members' products of a group's first-layer d x (dnn_backward_group)
    size_t dxpart_floats = 0;
    bool group_ok = true;       // MAMDR_GRAPH_NO_GROUP=1: one launch per expert and layer (A/B, parity of the grouped launches)
    // the step's weight gradients, queued by the backward pass and run in ONE pair of launches at its end (flush_wgrads);
    // MAMDR_GRAPH_NO_DEFER=1: a pair of launches per layer, where the backward pass meets it (A/B)
    struct WProb { const float* A; int lda; const float* B; int ldb; float* out; int M, N, rows; const float* dz; float* db; };
    std::vector<WProb> wq;
    struct TnProb { const float* in; int in_ld; const float* d; int d_ld; int rows, n_j, n_e; float* out; float* sum_out; };
    std::vector<TnProb> tq;     // narrow contractions (k_graph_small_tn's problems) that ride in the queue's second launch
    bool defer_w = true;
    // the optimiser step inside the tail launch (every gradient of a step's two ranges is finished there): no k_graph_adam.
    // Off for the weighted loss (k_graph_loss writes d / d log_var) and under MAMDR_GRAPH_NO_TAIL_OPT=1 (A/B; same bits)
    bool tail_opt = true;
    GradSink sink;              // this step's sink (p null: gradients are stored, k_graph_adam follows)
    int wq_blocks = 512;        // the queue's launch splits the batch rows until it has about this many workgroups
    int32_t* domrow = nullptr;
    float *thresholds = nullptr, *frozen_sumsq = nullptr, *sumsq_partials = nullptr, *eval_acc = nullptr;
    // trainable tables
    bool tables = false;
    int64_t table_floats = 0;
    int32_t *urow = nullptr, *irow = nullptr, *map_u = nullptr, *map_i = nullptr, *hasdup_u = nullptr, *hasdup_i = nullptr;
    float *gbuf_u = nullptr, *gbuf_i = nullptr;
    float* G(int64_t off) const { return grad + (off - table_floats); }      // gradient of the flat vector's element `off`
};

namespace {

int64_t add_tensor(mamdr_graph* g, const std::string& name, int64_t rows, int64_t cols) {
    const int64_t off = g->n_params;
    g->tensors.push_back(TensorInfo{name, off, rows, cols});
    g->n_params = (off + rows * cols + 3) & ~(int64_t)3;
    return off;
}
int add_dnn(mamdr_graph* g, const std::string& name, int in_dim, const int32_t* hidden, int n_hidden, uint32_t& next_id) {
    Dnn d;
    d.name = name;
    d.in_dim = in_dim;
    int in = in_dim;
    for (int l = 0; l < n_hidden; ++l) {
        Layer L;
        L.in = in;
        L.out = hidden[l];
        L.w_off = add_tensor(g, name + "/W" + std::to_string(l), in, hidden[l]);
        L.b_off = add_tensor(g, name + "/b" + std::to_string(l), 1, hidden[l]);
        L.id = next_id++;
        d.layers.push_back(L);
        in = hidden[l];
    }
    g->dnns.push_back(d);
    return (int)g->dnns.size() - 1;
}

// dW[M x N] = A[rows x M]^T . B[rows x N], the rows split over up to 16 workgroups per tile when the tiles alone leave
// most of the 256 CUs idle (a 384 x 512 kernel is 48 tiles); the partial products meet in k_graph_wfinish, which also
// carries the layer's bias gradient (column sums of `dz` into `db`, skipped when db is null)
struct DomainGradJob { const float* dx; int ld, x_col; const int32_t* domrow; int rows; const float* dm; float two_l2; float* g_dm;
                       int n_domain;
                       const float* lin_dlogit; const float* lin_w; float lin_two_l2; float* lin_g; };     // lin_g null: no linear table
void flush_wgrads(mamdr_graph* g, const DomainGradJob* dg = nullptr) {
    const int n = (int)g->wq.size();
    if (!n && g->tq.empty() && !dg) return;
    int tiles = 0;
    for (const auto& q : g->wq) tiles += (q.M / GT) * (q.N / GT);
    int S = 1;
    while (S < 16 && tiles * S < g->wq_blocks) S *= 2;
    int split[MAX_WQ];
    for (;; S /= 2) {       // per problem: the largest power of two <= S that leaves every split >= 4 k-tiles; all must fit
        size_t need = 0;
        for (int p = 0; p < n; ++p) {
            const auto& q = g->wq[p];
            const int nkt = q.rows / GK;
            int sp = 1;
            while (sp < S && nkt % (2 * sp) == 0 && nkt / (2 * sp) >= 4) sp *= 2;
            split[p] = sp;
            if (sp > 1) need += (size_t)sp * q.M * q.N;
        }
        if (need <= g->wpart_floats || S == 1) break;
    }
    WMulti m;
    WFinish f;
    memset(&m, 0, sizeof(m));
    memset(&f, 0, sizeof(f));
    m.n = f.n = n;
    size_t used = 0;
    int nb = 0, nf = 0;
    for (int p = 0; p < n; ++p) {
        const auto& q = g->wq[p];
        const int sp = split[p];
        float* part = g->wpart + used;
        if (sp > 1) used += (size_t)sp * q.M * q.N;
        m.first[p] = nb;
        m.A[p] = q.A;
        m.lda[p] = q.lda;
        m.B[p] = q.B;
        m.ldb[p] = q.ldb;
        m.C[p] = sp > 1 ? part : q.out;
        m.N[p] = q.N;
        m.tx[p] = q.N / GT;
        m.ty[p] = q.M / GT;
        m.K[p] = q.rows / sp;
        m.mn[p] = q.M * q.N;
        nb += m.tx[p] * m.ty[p] * sp;
        f.first[p] = nf;
        f.part[p] = sp > 1 ? part : q.out;
        f.out[p] = q.out;
        f.dz[p] = q.dz;
        f.db[p] = q.db;
        f.split[p] = sp;
        // (a gradient that the contraction wrote in place still has to pass through the optimiser when the sink steps)
        f.nb_red[p] = (sp > 1 || g->sink.p) ? (q.M * q.N / 4 + 255) / 256 : 0;
        f.N[p] = q.db ? q.N : 0;
        f.rows[p] = q.rows;
        f.ld[p] = g->ld;
        f.mn[p] = q.M * q.N;
        nf += f.nb_red[p] + (q.db ? (q.N + CS_COLS - 1) / CS_COLS : 0);
    }
    m.first[n] = nb;
    f.first[n] = nf;
    if (nb) GLAUNCH(k_graph_wgrad_multi, dim3(nb), dim3(256), 0, g->stream, m);
    TailJobs j;
    memset(&j, 0, sizeof(j));
    int nt = 0;
    j.n_tn = (int)g->tq.size();
    for (int q = 0; q < j.n_tn; ++q) {
        const auto& t = g->tq[q];
        j.tn_first[q] = nt;
        j.in[q] = t.in;
        j.in_ld[q] = t.in_ld;
        j.d[q] = t.d;
        j.d_ld[q] = t.d_ld;
        j.rows[q] = t.rows;
        j.n_j[q] = t.n_j;
        j.n_e[q] = t.n_e;
        j.out[q] = t.out;
        j.sum_out[q] = t.sum_out;
        nt += (t.n_j * t.n_e + CS_COLS - 1) / CS_COLS + (t.sum_out ? 1 : 0);
    }
    j.tn_first[j.n_tn] = nt;
    j.dg_first = nt;
    if (dg) {
        j.dg_blocks = (EMB / CS_COLS) * dg->n_domain;
        j.dx = dg->dx;
        j.ld = dg->ld;
        j.x_col = dg->x_col;
        j.domrow = dg->domrow;
        j.dg_rows = dg->rows;
        j.dm = dg->dm;
        j.two_l2 = dg->two_l2;
        j.g_dm = dg->g_dm;
        if (dg->lin_g) {
            j.lin_blocks = dg->n_domain;
            j.lin_dlogit = dg->lin_dlogit;
            j.lin_w = dg->lin_w;
            j.lin_two_l2 = dg->lin_two_l2;
            j.lin_g = dg->lin_g;
        }
    }
    j.sink = g->sink;
    const int nblk = nf + nt + j.dg_blocks + j.lin_blocks;
    if (nblk) GLAUNCH(k_graph_tail, dim3(nblk), dim3(256), 0, g->stream, f, j);
    g->wq.clear();
    g->tq.clear();
}
// a narrow contraction of the backward pass: queued beside the weight gradients (their outputs are gradients too)
void small_tn(mamdr_graph* g, const float* in, int in_ld, const float* d, int d_ld, int rows, int n_j, int n_e, float* out,
              float* sum_out = nullptr) {
    if (!g->defer_w) {
        launch_small_tn(g->stream, in, in_ld, d, d_ld, rows, n_j, n_e, out, sum_out);
        return;
    }
    if ((int)g->tq.size() == MAX_TQ) {
        g->sink.p = nullptr;
        flush_wgrads(g);
    }
    g->tq.push_back(mamdr_graph::TnProb{in, in_ld, d, d_ld, rows, n_j, n_e, out, sum_out});
}
void queue_wgrad(mamdr_graph* g, const float* A, int lda, const float* B, int ldb, float* out, int M, int N, int rows,
                 const float* dz, float* db) {
    if ((int)g->wq.size() == MAX_WQ) {      // a flush in mid-step must not step parameters the backward pass still reads:
        g->sink.p = nullptr;                // this step's gradients are stored and k_graph_adam runs at its end
        flush_wgrads(g);
    }
    g->wq.push_back(mamdr_graph::WProb{A, lda, B, ldb, out, M, N, rows, dz, db});
}
void launch_wgrad(mamdr_graph* g, GemmArgs a, int M, int N, int rows, const float* dz, float* db) {
    if (g->defer_w) {
        queue_wgrad(g, a.A, a.lda, a.B, a.ldb, a.C, M, N, rows, dz, db);
        return;
    }
    const int tiles = (M / GT) * (N / GT), nkt = rows / GK;
    int split = 1;
    while (split < 16 && tiles * split < 256 && nkt % (2 * split) == 0 && nkt / (2 * split) >= 4 &&
           (size_t)(2 * split) * M * N <= g->wpart_floats)
        split *= 2;
    float* out = a.C;
    a.K = rows / split;
    a.zstride = (size_t)M * N;
    if (split > 1) a.C = g->wpart;
    GLAUNCH(k_graph_gemm<2>, dim3(N / GT, M / GT, split), dim3(256), 0, g->stream, a);
    const int64_t n4 = (int64_t)M * N / 4;
    const int nb_red = split > 1 ? (int)((n4 + 255) / 256) : 0, nb_cs = db ? (N + CS_COLS - 1) / CS_COLS : 0;
    if (nb_red + nb_cs)
        GLAUNCH(k_graph_wfinish, dim3(nb_red + nb_cs), dim3(256), 0, g->stream, g->wpart, split, a.zstride, n4, out,
                           nb_red, dz, g->ld, rows, db, N);
}

// 32 x 32 tiles (gemm_tile32) while the 64 x 64 ones would leave CUs idle or nearly so; MAMDR_GRAPH_TILE32_BELOW=<tiles> (0: never)
// (process-wide like the environment it comes from; re-read at EVERY mamdr_graph_create -- round 6: a value set by one context's
// environment used to stay in force for every later context of the process, and a test that set MAMDR_GRAPH_TILE32_BELOW=0
// changed the rounding of every generic-layer run after it in the same session)
constexpr int TILE32_BELOW_DEFAULT = 512;
std::atomic<int> g_tile32_below{TILE32_BELOW_DEFAULT};
inline bool use_tile32(int M, int N, int n_group = 1) {
    return (M / GT) * (N / GT) * n_group < g_tile32_below.load(std::memory_order_relaxed) && N % T32 == 0;
}
void launch_gemm(int mode, const GemmArgs& a, int M, int N, hipStream_t s) {
    const dim3 grid(N / GT, M / GT), block(256);
    if (mode != 2 && use_tile32(M, N)) {
        const dim3 g32(N / T32, M / T32);
        if (mode == 0) GLAUNCH(k_graph_gemm32<0>, g32, block, 0, s, a);
        else GLAUNCH(k_graph_gemm32<1>, g32, block, 0, s, a);
        return;
    }
    if (mode == 0) GLAUNCH(k_graph_gemm<0>, grid, block, 0, s, a);
    else if (mode == 1) GLAUNCH(k_graph_gemm<1>, grid, block, 0, s, a);
    else GLAUNCH(k_graph_gemm<2>, grid, block, 0, s, a);
}

struct StepCtx {
    int rows, rp;               // rows of the batch, padded to 64
    bool train;
    uint32_t seed, step, drop_thresh;
    float keep_scale;
    bool use_dropout;
};

// forward of one DNN: input columns `in_col` (width in_dim) of the activation workspace
void dnn_forward(mamdr_graph* g, const Dnn& d, const std::vector<int>& cols, int in_col, const StepCtx& sc, int xe_col = -1) {
    int src = in_col;
    for (size_t l = 0; l < d.layers.size(); ++l) {
        const Layer& L = d.layers[l];
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        if (l == 0 && xe_col >= 0) {        // PNN: the three inner products times rows 384..386 of the first kernel
            a.xe = g->act + xe_col;
            a.xe_ld = g->ld;
            a.we = g->params + L.w_off + (size_t)L.in * L.out;
            a.n_xe = 3;
        }
        a.A = g->act + src;
        a.lda = g->ld;
        a.B = g->params + L.w_off;
        a.ldb = L.out;
        a.C = g->act + cols[l];
        a.ldc = g->ld;
        a.K = L.in;
        a.bias = g->params + L.b_off;
        a.relu = 1;
        a.use_dropout = sc.use_dropout ? 1 : 0;
        a.drop_key = dropout_layer_key(sc.seed, sc.step, L.id);
        a.drop_thresh = sc.drop_thresh;
        a.keep_scale = sc.keep_scale;
        a.n_cols = L.out;
        launch_gemm(0, a, sc.rp, L.out, g->stream);
        src = cols[l];
    }
}
// backward of one DNN whose last layer's d z already sits in the gradient workspace.  The input's gradient goes to
// `din_col` of the gradient workspace: times the gate of `in_gate_col` (the producer's relu / dropout) when >= 0;
// accumulated when `din_acc`; only columns [din_first, din_first + din_n) of the input when din_n > 0 (first layers
// on x: the domain columns alone matter while the tables are frozen); skipped when din_col < 0.
void dnn_backward(mamdr_graph* g, const Dnn& d, const std::vector<int>& cols, int in_col, int din_col, int in_gate_col,
                  bool din_acc, int din_first, int din_n, const StepCtx& sc) {
    for (int l = (int)d.layers.size() - 1; l >= 0; --l) {
        const Layer& L = d.layers[l];
        const int src = l == 0 ? in_col : cols[l - 1];
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        a.A = g->act + src;             // dW = in^T dz
        a.lda = g->ld;
        a.B = g->dact + cols[l];
        a.ldb = g->ld;
        a.C = g->G(L.w_off);
        a.ldc = L.out;
        launch_wgrad(g, a, L.in, L.out, sc.rp, g->dact + cols[l], g->G(L.b_off));
        memset(&a, 0, sizeof(a));
        a.A = g->dact + cols[l];        // d in = dz W^T
        a.lda = g->ld;
        a.K = L.out;
        a.gate_scale = sc.keep_scale;
        if (l > 0) {
            a.B = g->params + L.w_off;
            a.ldb = L.out;
            a.C = g->dact + cols[l - 1];
            a.ldc = g->ld;
            a.gate_y = g->act + cols[l - 1];
            a.gate_ld = g->ld;
            launch_gemm(1, a, sc.rp, L.in, g->stream);
        } else if (din_col >= 0) {
            const int first = din_n > 0 ? din_first : 0, n = din_n > 0 ? din_n : L.in;
            a.B = g->params + L.w_off + (size_t)first * L.out;
            a.ldb = L.out;
            a.C = g->dact + din_col + first;
            a.ldc = g->ld;
            if (in_gate_col >= 0) {
                a.gate_y = g->act + in_gate_col + first;
                a.gate_ld = g->ld;
            }
            a.accumulate = din_acc ? 1 : 0;
            launch_gemm(1, a, sc.rp, n, g->stream);
        }
    }
}

// ---- a group of DNNs of ONE shape on ONE input (the experts a task mixes), layer by layer in single launches
bool same_shape(const mamdr_graph* g, const std::vector<int>& ids) {
    if (ids.size() < 2 || ids.size() > (size_t)MAX_GROUP) return false;
    const Dnn& d0 = g->dnns[ids[0]];
    for (int id : ids) {
        const Dnn& d = g->dnns[id];
        if (d.in_dim != d0.in_dim || d.layers.size() != d0.layers.size()) return false;
        for (size_t l = 0; l < d.layers.size(); ++l)
            if (d.layers[l].in != d0.layers[l].in || d.layers[l].out != d0.layers[l].out) return false;
    }
    return true;
}
void dnn_forward_group(mamdr_graph* g, const std::vector<int>& ids, const std::vector<std::vector<int>>& cols, int in_col,
                       const StepCtx& sc) {
    const Dnn& d0 = g->dnns[ids[0]];
    const int n = (int)ids.size();
    for (size_t l = 0; l < d0.layers.size(); ++l) {
        const Layer& L0 = d0.layers[l];
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        GroupTab t;
        memset(&t, 0, sizeof(t));
        t.n = n;
        a.A = g->act;
        a.lda = g->ld;
        a.B = g->params;
        a.ldb = L0.out;
        a.C = g->act;
        a.ldc = g->ld;
        a.K = L0.in;
        a.bias = g->params;
        a.relu = 1;
        a.use_dropout = sc.use_dropout ? 1 : 0;
        a.drop_thresh = sc.drop_thresh;
        a.keep_scale = sc.keep_scale;
        a.n_cols = L0.out;
        for (int e = 0; e < n; ++e) {
            const Layer& L = g->dnns[ids[e]].layers[l];
            t.a_off[e] = l == 0 ? in_col : cols[e][l - 1];
            t.b_off[e] = L.w_off;
            t.c_off[e] = cols[e][l];
            t.bias_off[e] = L.b_off;
            t.drop_key[e] = dropout_layer_key(sc.seed, sc.step, L.id);
        }
        if (use_tile32(sc.rp, L0.out, n))
            GLAUNCH(k_graph_gemm32_group<0>, dim3(L0.out / T32, sc.rp / T32, n), dim3(256), 0, g->stream, a, t);
        else
            GLAUNCH(k_graph_gemm_group<0>, dim3(L0.out / GT, sc.rp / GT, n), dim3(256), 0, g->stream, a, t);
    }
}
// backward of the group (every member's last-layer d z sits in the gradient workspace): weight / bias gradients and the
// inner layers' d inputs in grouped launches; the FIRST layers' d x adds up over the members and stays one launch each
void dnn_backward_group(mamdr_graph* g, const std::vector<int>& ids, const std::vector<std::vector<int>>& cols, int in_col,
                        bool din_acc, int din_first, int din_n, const StepCtx& sc) {
    const Dnn& d0 = g->dnns[ids[0]];
    const int n = (int)ids.size();
    for (int l = (int)d0.layers.size() - 1; l >= 0; --l) {
        const Layer& L0 = d0.layers[l];
        const int M = L0.in, N = L0.out;
        if (g->defer_w) {
            for (int e = 0; e < n; ++e) {
                const Layer& L = g->dnns[ids[e]].layers[l];
                queue_wgrad(g, g->act + (l == 0 ? in_col : cols[e][l - 1]), g->ld, g->dact + cols[e][l], g->ld, g->G(L.w_off), M, N,
                            sc.rp, g->dact + cols[e][l], g->G(L.b_off));
            }
        } else {    // dW_e = in_e^T dz_e, db_e = column sums of dz_e
            const int tiles = (M / GT) * (N / GT), nkt = sc.rp / GK;
            int split = 1;
            while (split < 16 && tiles * n * split < 256 && nkt % (2 * split) == 0 && nkt / (2 * split) >= 4 &&
                   (size_t)(2 * split) * n * M * N <= g->wpart_floats)
                split *= 2;
            GemmArgs a;
            memset(&a, 0, sizeof(a));
            GroupTab t, f;
            memset(&t, 0, sizeof(t));
            memset(&f, 0, sizeof(f));
            t.n = f.n = n;
            t.tiles_y = M / GT;
            a.A = g->act;
            a.lda = g->ld;
            a.B = g->dact;
            a.ldb = g->ld;
            a.C = split > 1 ? g->wpart : g->grad;
            a.ldc = N;
            a.K = sc.rp / split;
            a.zstride = (size_t)M * N;
            for (int e = 0; e < n; ++e) {
                const Layer& L = g->dnns[ids[e]].layers[l];
                t.a_off[e] = l == 0 ? in_col : cols[e][l - 1];
                t.b_off[e] = cols[e][l];
                t.c_off[e] = split > 1 ? (int64_t)e * split * M * N : L.w_off - g->table_floats;
                f.a_off[e] = cols[e][l];
                f.c_off[e] = L.w_off - g->table_floats;
                f.bias_off[e] = L.b_off - g->table_floats;
            }
            GLAUNCH(k_graph_gemm_group<2>, dim3(N / GT, (M / GT) * n, split), dim3(256), 0, g->stream, a, t);
            const int64_t n4 = (int64_t)M * N / 4;
            const int nb_red = split > 1 ? (int)((n4 + 255) / 256) : 0, nb_cs = (N + CS_COLS - 1) / CS_COLS;
            GLAUNCH(k_graph_wfinish_group, dim3(nb_red + nb_cs, n), dim3(256), 0, g->stream, g->wpart, split,
                               a.zstride, n4, g->grad, nb_red, g->dact, g->ld, sc.rp, g->grad, N, f);
        }
        if (l > 0) {    // d in_e = dz_e W_e^T through the producer's relu / dropout gate
            GemmArgs a;
            memset(&a, 0, sizeof(a));
            GroupTab t;
            memset(&t, 0, sizeof(t));
            t.n = n;
            a.A = g->dact;
            a.lda = g->ld;
            a.B = g->params;
            a.ldb = N;
            a.C = g->dact;
            a.ldc = g->ld;
            a.K = N;
            a.gate_scale = sc.keep_scale;
            a.gate_y = g->act;
            a.gate_ld = g->ld;
            for (int e = 0; e < n; ++e) {
                const Layer& L = g->dnns[ids[e]].layers[l];
                t.a_off[e] = cols[e][l];
                t.b_off[e] = L.w_off;
                t.c_off[e] = cols[e][l - 1];
                t.gate_off[e] = cols[e][l - 1];
            }
            if (use_tile32(sc.rp, M, n))
                GLAUNCH(k_graph_gemm32_group<1>, dim3(M / T32, sc.rp / T32, n), dim3(256), 0, g->stream, a, t);
            else
                GLAUNCH(k_graph_gemm_group<1>, dim3(M / GT, sc.rp / GT, n), dim3(256), 0, g->stream, a, t);
        } else {
            // d x = sum_e dz_e W_e[first : first + nn]^T
            const int first = din_n > 0 ? din_first : 0, nn = din_n > 0 ? din_n : M;
            GemmArgs a;
            memset(&a, 0, sizeof(a));
            GroupTab t;
            memset(&t, 0, sizeof(t));
            t.n = n;
            a.A = g->dact;
            a.lda = g->ld;
            a.K = N;
            a.gate_scale = sc.keep_scale;
            a.B = g->params + (size_t)first * N;
            a.ldb = N;
            const size_t stride = (size_t)sc.rp * nn;
            if ((size_t)n * stride <= g->dxpart_floats) {
                // every member's product as a launch-mate of the others (n x the tiles), summed in member order
                a.C = g->dxpart;
                a.ldc = nn;
                for (int e = 0; e < n; ++e) {
                    t.a_off[e] = cols[e][0];
                    t.b_off[e] = g->dnns[ids[e]].layers[0].w_off;
                    t.c_off[e] = (int64_t)e * stride;
                }
                if (use_tile32(sc.rp, nn, n))
                    GLAUNCH(k_graph_gemm32_group<1>, dim3(nn / T32, sc.rp / T32, n), dim3(256), 0, g->stream, a, t);
                else
                    GLAUNCH(k_graph_gemm_group<1>, dim3(nn / GT, sc.rp / GT, n), dim3(256), 0, g->stream, a, t);
                const int64_t tot = (int64_t)sc.rp * (nn / 4);
                GLAUNCH(k_graph_dx_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, g->stream, g->dxpart, n,
                                   stride, sc.rp, nn / 4, g->dact + in_col + first, g->ld, din_acc ? 1 : 0);
            } else {    // one contraction whose reduction index runs through all members
                a.C = g->dact + in_col + first;
                a.ldc = g->ld;
                a.accumulate = din_acc ? 1 : 0;
                for (int e = 0; e < n; ++e) {
                    t.a_off[e] = cols[e][0];
                    t.b_off[e] = g->dnns[ids[e]].layers[0].w_off;
                }
                if (use_tile32(sc.rp, nn))
                    GLAUNCH(k_graph_gemm32_kcat, dim3(nn / T32, sc.rp / T32), dim3(256), 0, g->stream, a, t);
                else
                    GLAUNCH(k_graph_gemm_kcat, dim3(nn / GT, sc.rp / GT), dim3(256), 0, g->stream, a, t);
            }
        }
    }
}

void fill_gate(const mamdr_graph* g, const Task& t, const StepCtx& sc, GateArgs& ga) {
    memset(&ga, 0, sizeof(ga));
    ga.act = g->act;
    ga.dact = g->dact;
    ga.ld = g->ld;
    const size_t gi = t.mix.size();                 // position of the gate DNN in `path`
    const Dnn& gd = g->dnns[t.gate];
    ga.q_col = t.col[gi].back();
    ga.n_q = gd.layers.back().out;
    ga.wg = g->params + t.wg_off;
    ga.n_e = (int)t.mix.size();
    for (size_t e = 0; e < t.mix.size(); ++e) ga.e_col[e] = t.col[e].back();
    ga.n_h = g->n_h;
    ga.g_col = t.g_col;
    ga.m_col = t.m_col;
    ga.rows_pad = sc.rp;
    ga.gate_scale = sc.keep_scale;
}

// forward of task d on the gathered batch; -> column of the tower's output
void fill_feat(const mamdr_graph* g, const Task& t, const StepCtx& sc, FeatArgs& fa) {
    memset(&fa, 0, sizeof(fa));
    fa.act = g->act;
    fa.dact = g->dact;
    fa.ld = g->ld;
    fa.f_col = g->f_col;
    fa.rows_pad = sc.rp;
    fa.kind = g->cfg.kind == MAMDR_GRAPH_NFM ? 0 : (g->cfg.kind == MAMDR_GRAPH_DEEPFM ? 2 : 1);
    fa.extra = g->extra;
    fa.dlogit = g->dlogit;
    fa.dx_all = g->tables ? 1 : 0;
    const Layer& L0 = g->dnns[t.tower].layers[0];
    fa.w_ip = g->params + L0.w_off + (size_t)L0.in * L0.out;
    fa.z_col = t.col[0][0];
    fa.n_out = L0.out;
}

void fill_ccpm(const mamdr_graph* g, const StepCtx& sc, CcpmArgs& ca) {
    memset(&ca, 0, sizeof(ca));
    ca.act = g->act;
    ca.dact = g->dact;
    ca.ld = g->ld;
    ca.f_col = g->f_col;
    ca.cg_col = g->cg_col;
    ca.rows_pad = sc.rp;
    ca.dx_all = g->tables ? 1 : 0;
    ca.conv = g->params + g->conv_off;
}

int task_forward(mamdr_graph* g, const Task& t, const StepCtx& sc) {
    if (g->single && g->cfg.kind == MAMDR_GRAPH_AUTOINT) {
        for (int l = 0; l < 3; ++l) {
            const int d_in = l == 0 ? EMB : ATT_OUT;
            GemmArgs a;
            memset(&a, 0, sizeof(a));
            a.A = l == 0 ? g->xt : g->attY[l - 1];     // token-major [3 rows_pad][d]
            a.lda = d_in;
            a.B = g->params + g->att_w[l];
            a.ldb = ATT_P;
            a.C = g->attP[l];
            a.ldc = ATT_P;
            a.K = d_in;
            a.n_cols = ATT_P;
            launch_gemm(0, a, 3 * sc.rp, ATT_P, g->stream);
            AttArgs aa;
            memset(&aa, 0, sizeof(aa));
            aa.P = g->attP[l];
            aa.A = g->attA[l];
            aa.Y = g->attY[l];
            aa.rows_pad = sc.rp;
            if (l == 2) {
                aa.top = g->act + g->top_col;
                aa.top_ld = g->ld;
            }
            GLAUNCH(k_graph_att_fwd, dim3(sc.rp / 4), dim3(256), 0, g->stream, aa);
        }
        dnn_forward(g, g->dnns[t.tower], t.col[0], 0, sc);
        return g->top_col;
    }
    if (g->single) {
        if (g->cfg.kind == MAMDR_GRAPH_CCPM) {
            CcpmArgs ca;
            fill_ccpm(g, sc, ca);
            GLAUNCH(k_graph_ccpm_fwd, dim3(sc.rp / 4), dim3(256), 0, g->stream, ca);
            dnn_forward(g, g->dnns[t.tower], t.col[0], g->f_col, sc);
            return t.col[0].back();
        }
        const int kind = g->cfg.kind;
        if (kind == MAMDR_GRAPH_MLP || kind == MAMDR_GRAPH_WDL) {       // the plain DNN on x (WDL's linear part rides in `extra`)
            dnn_forward(g, g->dnns[t.tower], t.col[0], 0, sc);
            return t.col[0].back();
        }
        FeatArgs fa;
        fill_feat(g, t, sc, fa);
        GLAUNCH(k_graph_feat_fwd, dim3(sc.rp / 4), dim3(256), 0, g->stream, fa);
        const bool nfm = kind == MAMDR_GRAPH_NFM;
        dnn_forward(g, g->dnns[t.tower], t.col[0], nfm ? g->f_col : 0, sc, kind == MAMDR_GRAPH_PNN ? g->f_col : -1);
        return t.col[0].back();
    }
    if (g->group_ok && same_shape(g, t.mix)) {
        std::vector<std::vector<int>> mc(t.col.begin(), t.col.begin() + t.mix.size());
        dnn_forward_group(g, t.mix, mc, 0, sc);
    } else {
        for (size_t e = 0; e < t.mix.size(); ++e) dnn_forward(g, g->dnns[t.mix[e]], t.col[e], 0, sc);
    }
    int tower_in;
    if (g->gated) {
        const size_t gi = t.mix.size();
        dnn_forward(g, g->dnns[t.gate], t.col[gi], 0, sc);
        GateArgs ga;
        fill_gate(g, t, sc, ga);
        GLAUNCH(k_graph_gate_fwd, dim3(sc.rp), dim3(256), 0, g->stream, ga);
        tower_in = t.m_col;
    } else {
        tower_in = t.col[0].back();
    }
    const size_t ti = t.path.size() - 1;
    dnn_forward(g, g->dnns[t.tower], t.col[ti], tower_in, sc);
    return t.col[ti].back();
}

int check(const mamdr_graph* g) {
    if (!g) return gfail(MAMDR_EINVAL, "null graph context");
    return MAMDR_OK;
}
// trainable tables: the regulariser term of a reported loss needs their current sums of squares
void refresh_sumsq(mamdr_graph* g) {
    launch_sumsq(g->params, (int64_t)g->cfg.n_user * EMB, g->sumsq_partials, g->frozen_sumsq + 0, g->stream);
    launch_sumsq(g->params + (size_t)g->cfg.n_user * EMB, (int64_t)g->cfg.n_item * EMB, g->sumsq_partials, g->frozen_sumsq + 1,
                 g->stream);
    if (g->has_lin) {
        launch_sumsq(g->params + g->lin_u_off, g->cfg.n_user, g->sumsq_partials, g->frozen_sumsq + 2, g->stream);
        launch_sumsq(g->params + g->lin_i_off, g->cfg.n_item, g->sumsq_partials, g->frozen_sumsq + 3, g->stream);
    }
}
int ready(const mamdr_graph* g) {
    if (!g->params) return gfail(MAMDR_ESTATE, "mamdr_graph_bind_state has not been called");
    if (!g->tables && (!g->user_tab || !g->item_tab)) return gfail(MAMDR_ESTATE, "frozen user / item tables are not bound");
    return MAMDR_OK;
}
SplitData* split_of(mamdr_graph* g, int domain, int split) {
    if (domain < 0 || domain >= g->cfg.n_domain || split < 0 || split > 2) return nullptr;
    return &g->data[(size_t)domain * 3 + split];
}
void fill_gather(const mamdr_graph* g, const SplitData& d, const int32_t* perm, int64_t row_base, const StepCtx& sc,
                 GatherArgs& ga) {
    memset(&ga, 0, sizeof(ga));
    ga.user_tab = g->user_tab;
    ga.item_tab = g->item_tab;
    ga.dm = g->params + g->dm_off;
    ga.uid = d.uid;
    ga.pid = d.pid;
    ga.dom = d.dom;
    ga.perm = perm;
    ga.label = d.label;
    ga.row_base = row_base;
    ga.n_rows_split = d.n;
    ga.rows = sc.rows;
    ga.rows_pad = sc.rp;
    ga.n_user = g->cfg.n_user;
    ga.n_item = g->cfg.n_item;
    ga.n_domain = g->cfg.n_domain;
    ga.x = g->act;
    ga.ld = g->ld;
    ga.domrow = g->domrow;
    ga.y = g->y;
    ga.xt = g->xt;
    if (g->extra) {
        ga.extra = g->extra;
        ga.lin_d = g->params + g->lin_d_off;
        if (g->tables) {
            ga.lin_u = g->params + g->lin_u_off;
            ga.lin_i = g->params + g->lin_i_off;
        }
    }
    if (g->tables && sc.train) {
        ga.urow = g->urow;
        ga.irow = g->irow;
        ga.map_u = g->map_u;
        ga.map_i = g->map_i;
    }
}

}  // namespace

extern "C" {

const char* mamdr_graph_last_error(void) { return g_gerr; }

int mamdr_graph_create(const mamdr_graph_config* cfg, void* stream, mamdr_graph** out) {
    if (!cfg || !out) return gfail(MAMDR_EINVAL, "null argument");
    *out = nullptr;
    (void)mamdr::env_warn_unknown();
    if (cfg->abi_version != MAMDR_ABI_VERSION) return gfail(MAMDR_EINVAL, "abi_version %d != %d", cfg->abi_version, MAMDR_ABI_VERSION);
    if (cfg->emb_dim != EMB) return gfail(MAMDR_EINVAL, "emb_dim must be %d", EMB);
    if (cfg->n_user <= 0 || cfg->n_item <= 0 || cfg->n_domain <= 0 || cfg->max_batch <= 0) return gfail(MAMDR_EINVAL, "bad sizes");
    if (cfg->emb_trainable && cfg->max_batch > 16384) return gfail(MAMDR_EINVAL, "trainable tables: max_batch <= 16384");
    if (cfg->kind < MAMDR_GRAPH_SHARED_BOTTOM || cfg->kind > MAMDR_GRAPH_DEEPFM)
        return gfail(MAMDR_EINVAL, "unknown graph tower kind %d", cfg->kind);
    if (!(cfg->dropout >= 0.f && cfg->dropout < 1.f)) return gfail(MAMDR_EINVAL, "dropout rate must be in [0,1)");
    const bool single = cfg->kind >= MAMDR_GRAPH_NFM;
    const bool has_lin = cfg->kind == MAMDR_GRAPH_NFM || cfg->kind == MAMDR_GRAPH_CCPM || cfg->kind == MAMDR_GRAPH_AUTOINT ||
                         cfg->kind == MAMDR_GRAPH_WDL || cfg->kind == MAMDR_GRAPH_DEEPFM;
    const bool gated = cfg->kind == MAMDR_GRAPH_MMOE || cfg->kind == MAMDR_GRAPH_PLE;
    if (cfg->n_expert_hidden < 1 || cfg->n_expert_hidden > 4 || (!single && (cfg->n_tower_hidden < 1 || cfg->n_tower_hidden > 4)) ||
        (gated && (cfg->n_gate_hidden < 1 || cfg->n_gate_hidden > 4)))
        return gfail(MAMDR_EINVAL, "hidden_dim%s need 1..4 layers", single ? "" : (gated ? " / tower_hidden_dim / gate_dnn_hidden_units"
                                                                                           : " / tower_hidden_dim"));
    auto widths_ok = [](const int32_t* h, int n) {
        for (int i = 0; i < n; ++i)
            if (h[i] <= 0 || h[i] % 64) return false;
        return true;
    };
    if (!widths_ok(cfg->expert_hidden, cfg->n_expert_hidden) || (!single && !widths_ok(cfg->tower_hidden, cfg->n_tower_hidden)) ||
        (gated && !widths_ok(cfg->gate_hidden, cfg->n_gate_hidden)))
        return gfail(MAMDR_EINVAL, "layer widths must be multiples of 64 (the reference's configs use 64 ... 512)");
    int n_shared = single ? 0 : 1, n_specific = 0;
    if (cfg->kind == MAMDR_GRAPH_MMOE) n_shared = cfg->num_experts;
    if (cfg->kind == MAMDR_GRAPH_PLE) { n_shared = cfg->shared_expert_num; n_specific = cfg->specific_expert_num; }
    if (!single && (n_shared < 0 || n_specific < 0 || n_shared + n_specific < 1 || n_shared + n_specific > MAX_MIX))
        return gfail(MAMDR_EINVAL, "a task must mix 1..%d experts", MAX_MIX);
    if (!single && cfg->uncertainty_weight)     // uncertainty_weight.py:41-45 wraps model.inputs / outputs[0] of ONE Keras model
        return gfail(MAMDR_ENOTBUILT, "the weighted loss wraps a single-output tower (the multi-task towers are a dict of models)");

    mamdr_graph* g = new (std::nothrow) mamdr_graph();
    if (!g) return gfail(MAMDR_EHIP, "out of host memory");
    g->cfg = *cfg;
    g->stream = (hipStream_t)stream;
    if (const char* ev = getenv("MAMDR_GRAPH_NO_GROUP")) g->group_ok = atoi(ev) == 0;
    if (const char* ev = getenv("MAMDR_GRAPH_NO_DEFER")) g->defer_w = atoi(ev) == 0;
    {
        const char* ev = getenv("MAMDR_GRAPH_TILE32_BELOW");
        g_tile32_below.store(ev ? atoi(ev) : TILE32_BELOW_DEFAULT, std::memory_order_relaxed);
    }
    if (const char* ev = getenv("MAMDR_GRAPH_NO_TAIL_OPT")) g->tail_opt = atoi(ev) == 0;
    g->sink.p = nullptr;
    if (const char* ev = getenv("MAMDR_GRAPH_WQ_BLOCKS")) g->wq_blocks = atoi(ev) > 0 ? atoi(ev) : g->wq_blocks;
    g->gated = gated;
    g->single = single;
    g->has_lin = has_lin;
    g->n_h = cfg->expert_hidden[cfg->n_expert_hidden - 1];
    g->data.resize((size_t)cfg->n_domain * 3);
    // ---- flat vector: the block every task's model trains, then one block per task (oracle/mtl.py Spec.tensors)
    uint32_t next_id = 0;
    g->tables = cfg->emb_trainable != 0;
    if (g->tables) {        // [user table | item table] contiguous at the head (k_emb_sweep walks them as one range)
        add_tensor(g, "user_emb", cfg->n_user, EMB);
        add_tensor(g, "item_emb", cfg->n_item, EMB);
        if (has_lin) {      // their 1-d linear tables train with them (deepctr: same feature column)
            g->lin_u_off = add_tensor(g, "lin_user", cfg->n_user, 1);
            g->lin_i_off = add_tensor(g, "lin_item", cfg->n_item, 1);
        }
        g->table_floats = g->n_params;
    }
    g->dm_off = add_tensor(g, "domain_emb", cfg->n_domain, EMB);
    std::vector<int> shared;
    for (int e = 0; e < n_shared; ++e) {
        const std::string nm = cfg->kind == MAMDR_GRAPH_SHARED_BOTTOM ? "bottom"
                               : (cfg->kind == MAMDR_GRAPH_MMOE ? "expert_" + std::to_string(e) : "shared_expert_" + std::to_string(e));
        shared.push_back(add_dnn(g, nm, XDIM, cfg->expert_hidden, cfg->n_expert_hidden, next_id));
    }
    g->shared_end = g->n_params;
    g->tasks.resize(single ? 1 : cfg->n_domain);
    int max_cols = 0;
    if (single) {
        // oracle/fmnets.py param_names: domain_emb | W0 W1 W2 | b0 b1 b2 | wo | gb | (NFM) lin_domain -- one block, all of it
        // trained by every step.  NFM: DNN on the 128 interaction columns; PNN: on x (+ 3 inner products into W0's last rows)
        Task& t = g->tasks[0];
        Dnn d;
        d.name = "dnn";
        const int nfm = cfg->kind == MAMDR_GRAPH_NFM;
        const bool ccpm = cfg->kind == MAMDR_GRAPH_CCPM;
        if (ccpm) {         // conv1_w [6][4] | conv1_b [4] | conv2_w [4 in][4 out] | conv2_b [4]: 48 contiguous floats
            g->conv_off = add_tensor(g, "conv1_w", 6, 4);
            add_tensor(g, "conv1_b", 1, 4);
            add_tensor(g, "conv2_w", 4, 4);
            add_tensor(g, "conv2_b", 1, 4);
        }
        const bool autoint = cfg->kind == MAMDR_GRAPH_AUTOINT;
        if (autoint) {      // [W_query | W_key | W_value | W_res] per layer: [128][128], then [32][128] twice
            g->att_w[0] = add_tensor(g, "att0_w", EMB, ATT_P);
            g->att_w[1] = add_tensor(g, "att1_w", ATT_OUT, ATT_P);
            g->att_w[2] = add_tensor(g, "att2_w", ATT_OUT, ATT_P);
        }
        d.in_dim = nfm ? EMB : (ccpm ? 4 * EMB : XDIM);
        int in = d.in_dim;
        for (int l = 0; l < cfg->n_expert_hidden; ++l) {
            Layer L;
            L.in = in;
            L.out = cfg->expert_hidden[l];
            L.w_off = add_tensor(g, "W" + std::to_string(l), (l == 0 && cfg->kind == MAMDR_GRAPH_PNN) ? in + 3 : in, L.out);
            L.b_off = 0;
            L.id = (uint32_t)l;
            d.layers.push_back(L);
            in = L.out;
        }
        for (int l = 0; l < cfg->n_expert_hidden; ++l) d.layers[l].b_off = add_tensor(g, "b" + std::to_string(l), 1, d.layers[l].out);
        g->dnns.push_back(d);
        t.tower = 0;
        t.head_w = add_tensor(g, "wo", autoint ? 3 * ATT_OUT + in : in, 1);
        t.head_gb = add_tensor(g, "gb", 1, 1);
        if (has_lin) g->lin_d_off = add_tensor(g, "lin_domain", cfg->n_domain, 1);
        if (cfg->uncertainty_weight) g->lv_off = add_tensor(g, "log_var", cfg->n_domain, 1);
        g->shared_end = g->n_params;
        t.blk_off = t.blk_end = g->n_params;
        int c = XDIM;
        g->f_col = c;
        c += nfm ? EMB : (ccpm ? 4 * EMB : 4);
        if (ccpm) { g->cg_col = c; c += 64; }
        t.path.push_back(0);
        std::vector<int> cols;
        for (size_t l = 0; l < d.layers.size(); ++l) {
            if (autoint && l + 1 == d.layers.size()) {      // head input = [attention output 96 | last DNN layer]
                g->top_col = c;
                c += 3 * ATT_OUT;
            }
            cols.push_back(c);
            c += d.layers[l].out;
        }
        t.col.push_back(cols);
        t.n_cols = c;
        max_cols = c;
    }
    for (int d = 0; d < (single ? 0 : cfg->n_domain); ++d) {
        Task& t = g->tasks[d];
        t.blk_off = g->n_params;
        for (int e = 0; e < n_specific; ++e)
            t.mix.push_back(add_dnn(g, "task_" + std::to_string(d) + "_expert_" + std::to_string(e), XDIM, cfg->expert_hidden,
                                    cfg->n_expert_hidden, next_id));
        for (int s : shared) t.mix.push_back(s);
        if (gated) {
            t.gate = add_dnn(g, "gate_" + std::to_string(d), XDIM, cfg->gate_hidden, cfg->n_gate_hidden, next_id);
            t.wg_off = add_tensor(g, "gate_" + std::to_string(d) + "/Wg", cfg->gate_hidden[cfg->n_gate_hidden - 1], (int64_t)t.mix.size());
        }
        t.tower = add_dnn(g, "tower_" + std::to_string(d), g->n_h, cfg->tower_hidden, cfg->n_tower_hidden, next_id);
        t.head_w = add_tensor(g, "head_" + std::to_string(d) + "/w", cfg->tower_hidden[cfg->n_tower_hidden - 1], 1);
        t.head_gb = add_tensor(g, "head_" + std::to_string(d) + "/gb", 1, 1);
        t.blk_end = g->n_params;
        // column plan: x | every layer output on the path | gate probabilities | mixture
        int c = XDIM;
        t.path = t.mix;
        if (gated) t.path.push_back(t.gate);
        t.path.push_back(t.tower);
        for (int di : t.path) {
            std::vector<int> cols;
            for (const Layer& L : g->dnns[di].layers) { cols.push_back(c); c += L.out; }
            t.col.push_back(cols);
        }
        if (gated) {
            t.g_col = c; c += ((int)t.mix.size() + 3) & ~3;
            t.m_col = c; c += g->n_h;
        }
        t.n_cols = c;
        if (c > max_cols) max_cols = c;
    }
    g->ld = (max_cols + 3) & ~3;
    g->rows_pad_max = (cfg->max_batch + GT - 1) / GT * GT;
    float thr[500];
    thr[0] = (float)(0.0 - 1e-7);
    for (int i = 0; i < 498; ++i) thr[i + 1] = (float)((double)(i + 1) * 1.0 / (double)(500 - 1));
    thr[499] = (float)(1.0 + 1e-7);
    const size_t rp = (size_t)g->rows_pad_max;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void**)&g->act, rp * g->ld * sizeof(float));
    alloc((void**)&g->dact, rp * g->ld * sizeof(float));
    alloc((void**)&g->grad, (size_t)(g->n_params - g->table_floats) * sizeof(float));
    size_t max_w = (size_t)EMB * ATT_P;
    for (const Dnn& d : g->dnns)
        for (const Layer& L : d.layers) max_w = std::max(max_w, (size_t)L.in * L.out);
    g->wpart_floats = 16 * max_w;
    if (g->defer_w) {       // the queue holds every problem's partial products at once: 8 splits of the widest step's kernels
        size_t max_path = 3 * (size_t)EMB * ATT_P;
        for (const Task& t : g->tasks) {
            size_t w = cfg->kind == MAMDR_GRAPH_AUTOINT ? 3 * (size_t)EMB * ATT_P : 0;
            for (int id : t.path)
                for (const Layer& L : g->dnns[id].layers) w += (size_t)L.in * L.out;
            max_path = std::max(max_path, w);
        }
        g->wpart_floats = std::max(g->wpart_floats, 8 * max_path);
    }
    alloc((void**)&g->wpart, g->wpart_floats * sizeof(float));
    size_t max_mix = 0;
    for (const Task& t : g->tasks) max_mix = std::max(max_mix, t.mix.size());
    if (g->gated && g->group_ok && max_mix > 1) {
        g->dxpart_floats = max_mix * rp * (size_t)(g->tables ? XDIM : EMB);
        alloc((void**)&g->dxpart, g->dxpart_floats * sizeof(float));
    }
    if (g->tables) {
        alloc((void**)&g->urow, rp * sizeof(int32_t));
        alloc((void**)&g->irow, rp * sizeof(int32_t));
        alloc((void**)&g->hasdup_u, rp * sizeof(int32_t));
        alloc((void**)&g->hasdup_i, rp * sizeof(int32_t));
        alloc((void**)&g->map_u, (size_t)cfg->n_user * sizeof(int32_t));
        alloc((void**)&g->map_i, (size_t)cfg->n_item * sizeof(int32_t));
        alloc((void**)&g->gbuf_u, rp * EMB * sizeof(float));
        alloc((void**)&g->gbuf_i, rp * EMB * sizeof(float));
    }
    alloc((void**)&g->dlogit, rp * sizeof(float));
    alloc((void**)&g->rowloss, rp * sizeof(float));
    alloc((void**)&g->y, rp * sizeof(float));
    alloc((void**)&g->domrow, rp * sizeof(int32_t));
    alloc((void**)&g->thresholds, sizeof(thr));
    alloc((void**)&g->frozen_sumsq, 4 * sizeof(float));
    alloc((void**)&g->sumsq_partials, 1024 * sizeof(float));
    alloc((void**)&g->eval_acc, 4 * sizeof(float));
    if (cfg->kind == MAMDR_GRAPH_AUTOINT) {
        alloc((void**)&g->xt, rp * XDIM * sizeof(float));
        alloc((void**)&g->dxt, rp * XDIM * sizeof(float));
        for (int l = 0; l < 3; ++l) {
            alloc((void**)&g->attP[l], 3 * rp * ATT_P * sizeof(float));
            alloc((void**)&g->attdP[l], 3 * rp * ATT_P * sizeof(float));
            alloc((void**)&g->attA[l], rp * 36 * sizeof(float));
            alloc((void**)&g->attY[l], 3 * rp * ATT_OUT * sizeof(float));
            alloc((void**)&g->attdY[l], 3 * rp * ATT_OUT * sizeof(float));
        }
    }
    if (has_lin) {
        alloc((void**)&g->extra, rp * sizeof(float));
        if (g->tables) {
            alloc((void**)&g->glin_u, rp * sizeof(float));
            alloc((void**)&g->glin_i, rp * sizeof(float));
        }
    }
    if (e == hipSuccess) e = hipMemsetAsync(g->grad, 0, (size_t)(g->n_params - g->table_floats) * sizeof(float), g->stream);
    if (g->tables && e == hipSuccess) {
        e = hipMemsetAsync(g->hasdup_u, 0, rp * sizeof(int32_t), g->stream);
        if (e == hipSuccess) e = hipMemsetAsync(g->hasdup_i, 0, rp * sizeof(int32_t), g->stream);
        if (e == hipSuccess) e = hipMemsetAsync(g->urow, 0xff, rp * sizeof(int32_t), g->stream);
        if (e == hipSuccess) e = hipMemsetAsync(g->irow, 0xff, rp * sizeof(int32_t), g->stream);
        launch_emb_map_init(g->map_u, cfg->n_user, g->stream);
        launch_emb_map_init(g->map_i, cfg->n_item, g->stream);
    }
    if (e == hipSuccess) e = hipMemsetAsync(g->dact, 0, rp * g->ld * sizeof(float), g->stream);
    if (e == hipSuccess) e = hipMemsetAsync(g->frozen_sumsq, 0, 4 * sizeof(float), g->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(g->thresholds, thr, sizeof(thr), hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) {
        mamdr_graph_destroy(g);
        return gfail(MAMDR_EHIP, "workspace: %s", hipGetErrorString(e));
    }
    *out = g;
    return MAMDR_OK;
}

int mamdr_graph_destroy(mamdr_graph* g) {
    if (!g) return MAMDR_OK;
    (void)hipStreamSynchronize(g->stream);
    void* ptrs[] = {g->act, g->dact, g->grad, g->dlogit, g->rowloss, g->y, g->domrow, g->thresholds, g->frozen_sumsq,
                    g->sumsq_partials, g->eval_acc, g->urow, g->irow, g->map_u, g->map_i, g->hasdup_u, g->hasdup_i,
                    g->gbuf_u, g->gbuf_i, g->extra, g->glin_u, g->glin_i, g->xt, g->dxt, g->wpart, g->dxpart,
                    g->attP[0], g->attP[1], g->attP[2], g->attdP[0], g->attdP[1], g->attdP[2], g->attA[0], g->attA[1], g->attA[2],
                    g->attY[0], g->attY[1], g->attY[2], g->attdY[0], g->attdY[1], g->attdY[2]};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete g;
    return MAMDR_OK;
}

int64_t mamdr_graph_param_count(const mamdr_graph* g) { return g ? g->n_params : 0; }
int32_t mamdr_graph_tensor_count(const mamdr_graph* g) { return g ? (int32_t)g->tensors.size() : 0; }
int mamdr_graph_tensor_info(const mamdr_graph* g, int32_t i, char* name, int32_t name_cap, int64_t* offset, int64_t* rows,
                            int64_t* cols) {
    if (check(g)) return MAMDR_EINVAL;
    if (i < 0 || i >= (int32_t)g->tensors.size() || !name || name_cap < 2 || !offset || !rows || !cols)
        return gfail(MAMDR_EINVAL, "bad tensor query");
    const TensorInfo& t = g->tensors[i];
    snprintf(name, (size_t)name_cap, "%s", t.name.c_str());
    *offset = t.off;
    *rows = t.rows;
    *cols = t.cols;
    return MAMDR_OK;
}
/* the two ranges of the flat vector a step of task `domain` trains */
int mamdr_graph_task_ranges(const mamdr_graph* g, int domain, int64_t* shared_off, int64_t* shared_count, int64_t* task_off,
                            int64_t* task_count) {
    if (check(g)) return MAMDR_EINVAL;
    if (domain < 0 || domain >= g->cfg.n_domain) return gfail(MAMDR_EINVAL, "domain out of range");
    *shared_off = g->dm_off;
    *shared_count = g->shared_end - g->dm_off;
    const Task& t = g->tasks[g->single ? 0 : domain];
    *task_off = t.blk_off;
    *task_count = t.blk_end - t.blk_off;
    return MAMDR_OK;
}

int mamdr_graph_bind_state(mamdr_graph* g, float* d_params, float* d_m, float* d_v) {
    if (check(g)) return MAMDR_EINVAL;
    if (!d_params || !d_m || !d_v) return gfail(MAMDR_EINVAL, "null state pointer");
    if (((uintptr_t)d_params | (uintptr_t)d_m | (uintptr_t)d_v) & 15) return gfail(MAMDR_EINVAL, "state pointers must be 16-byte aligned");
    g->params = d_params;
    g->adam_m = d_m;
    g->adam_v = d_v;
    if (g->tables) {
        g->user_tab = d_params;
        g->item_tab = d_params + (size_t)g->cfg.n_user * EMB;
    }
    return MAMDR_OK;
}
int mamdr_graph_optimizer_reset(mamdr_graph* g) {
    if (check(g)) return MAMDR_EINVAL;
    if (!g->params) return gfail(MAMDR_ESTATE, "state not bound");
    GHIP(hipMemsetAsync(g->adam_m, 0, (size_t)g->n_params * sizeof(float), g->stream));
    GHIP(hipMemsetAsync(g->adam_v, 0, (size_t)g->n_params * sizeof(float), g->stream));
    g->adam_t = 0;
    g->b1p = g->b2p = 1.f;
    return MAMDR_OK;
}
// the counterpart of mamdr_set_counters for the generic-layer engine: TF's running beta powers for `optimizer_steps` steps
// (one optimizer object for all D compiled models: deep_mtl_ctr.py:52-55) and the position of the dropout stream
int mamdr_graph_set_counters(mamdr_graph* g, int64_t optimizer_steps, int64_t dropout_steps) {
    if (check(g)) return MAMDR_EINVAL;
    if (optimizer_steps < 0 || optimizer_steps > (int64_t)0x7ffffff0 || dropout_steps < 0 || dropout_steps > (int64_t)0xffffffffLL)
        return gfail(MAMDR_EINVAL, "mamdr_graph_set_counters(%lld, %lld): out of range", (long long)optimizer_steps, (long long)dropout_steps);
    g->adam_t = optimizer_steps;
    float b1 = 1.f, b2 = 1.f;
    for (int64_t t = 0; t < optimizer_steps && (b1 != 0.f || b2 != 0.f); ++t) {
        b1 *= g->cfg.adam_beta1;
        b2 *= g->cfg.adam_beta2;
    }
    g->b1p = b1;
    g->b2p = b2;
    g->global_step = (uint32_t)dropout_steps;
    return MAMDR_OK;
}
int mamdr_graph_set_adam_eps(mamdr_graph* g, float eps) {
    if (check(g)) return MAMDR_EINVAL;
    if (!(eps > 0.f)) return gfail(MAMDR_EINVAL, "adam epsilon %g", (double)eps);
    g->cfg.adam_eps = eps;
    return MAMDR_OK;
}
int64_t mamdr_graph_launch_count(void) { return (int64_t)g_graph_launches.load(); }
int64_t mamdr_graph_optimizer_steps(const mamdr_graph* g) { return g ? g->adam_t : 0; }
int64_t mamdr_graph_dropout_steps(const mamdr_graph* g) { return g ? (int64_t)g->global_step : 0; }

int mamdr_graph_bind_table(mamdr_graph* g, int seg, const float* d_rows, int64_t n_rows) {
    if (check(g)) return MAMDR_EINVAL;
    if (g->tables) return gfail(MAMDR_ESTATE, "tables are trainable: they live in the flat vector");
    if (!d_rows || ((uintptr_t)d_rows & 15)) return gfail(MAMDR_EINVAL, "table pointer null or not 16-byte aligned");
    if (seg == MAMDR_SEG_USER_EMB) {
        if (n_rows != g->cfg.n_user) return gfail(MAMDR_EINVAL, "user table has %lld rows, config says %d", (long long)n_rows, g->cfg.n_user);
        g->user_tab = d_rows;
        launch_sumsq(d_rows, n_rows * EMB, g->sumsq_partials, g->frozen_sumsq + 0, g->stream);
    } else if (seg == MAMDR_SEG_ITEM_EMB) {
        if (n_rows != g->cfg.n_item) return gfail(MAMDR_EINVAL, "item table has %lld rows, config says %d", (long long)n_rows, g->cfg.n_item);
        g->item_tab = d_rows;
        launch_sumsq(d_rows, n_rows * EMB, g->sumsq_partials, g->frozen_sumsq + 1, g->stream);
    } else {
        return gfail(MAMDR_EINVAL, "segment %d is not a bindable table", seg);
    }
    GHIP(hipGetLastError());
    return MAMDR_OK;
}

int mamdr_graph_bind_domain_data(mamdr_graph* g, int domain, int split, const int32_t* d_uid, const int32_t* d_pid,
                                 const int32_t* d_domain, const float* d_label, int64_t n_rows) {
    if (check(g)) return MAMDR_EINVAL;
    SplitData* d = split_of(g, domain, split);
    if (!d) return gfail(MAMDR_EINVAL, "domain %d / split %d out of range", domain, split);
    if (n_rows < 0 || n_rows > 0x7fffffff) return gfail(MAMDR_EINVAL, "n_rows out of range");
    if (n_rows > 0 && (!d_uid || !d_pid || !d_domain || !d_label)) return gfail(MAMDR_EINVAL, "null column pointer");
    d->bound = true;
    d->uid = d_uid;
    d->pid = d_pid;
    d->dom = d_domain;
    d->label = d_label;
    d->n = n_rows;
    return MAMDR_OK;
}

int mamdr_graph_train_steps(mamdr_graph* g, int domain, const int32_t* d_perm, int64_t first_step, int64_t n_steps,
                            int32_t batch, uint32_t dropout_seed, int32_t optimizer, float lr, float* d_loss_out) {
    return mamdr_graph_train_steps_n(g, domain, d_perm, -1, first_step, n_steps, batch, dropout_seed, optimizer, lr, d_loss_out);
}

int mamdr_graph_train_steps_n(mamdr_graph* g, int domain, const int32_t* d_perm, int64_t pass_rows, int64_t first_step,
                              int64_t n_steps, int32_t batch, uint32_t dropout_seed, int32_t optimizer, float lr,
                              float* d_loss_out) {
    if (check(g)) return MAMDR_EINVAL;
    if (ready(g)) return MAMDR_ESTATE;
    SplitData* d = split_of(g, domain, MAMDR_SPLIT_TRAIN);
    if (!d || !d->bound) return gfail(MAMDR_ESTATE, "train split of domain %d is not bound", domain);
    if (batch <= 0 || batch > g->cfg.max_batch) return gfail(MAMDR_EINVAL, "batch %d outside (0, max_batch=%d]", batch, g->cfg.max_batch);
    if (optimizer != MAMDR_OPT_ADAM && optimizer != MAMDR_OPT_SGD && optimizer != MAMDR_OPT_ACCUMULATE)
        return gfail(MAMDR_EINVAL, "unknown optimizer %d", optimizer);
    if (optimizer == MAMDR_OPT_ACCUMULATE && !g->accum)
        return gfail(MAMDR_ESTATE, "MAMDR_OPT_ACCUMULATE needs mamdr_graph_bind_accumulator first");
    if (first_step < 0 || n_steps < 0) return gfail(MAMDR_EINVAL, "negative step range");
    if (pass_rows < 0) pass_rows = d->n;                 // the whole split
    if (pass_rows > d->n) return gfail(MAMDR_EINVAL, "pass of %lld rows exceeds the %lld rows of domain %d",
                                       (long long)pass_rows, (long long)d->n, domain);
    if (pass_rows < d->n && !d_perm) return gfail(MAMDR_EINVAL, "a pass over part of a split needs its permutation");
    const int64_t pass_steps = (pass_rows + batch - 1) / batch;
    if (first_step + n_steps > pass_steps)
        return gfail(MAMDR_EINVAL, "steps [%lld,%lld) exceed the %lld batches of domain %d", (long long)first_step,
                     (long long)(first_step + n_steps), (long long)pass_steps, domain);
    if (n_steps == 0) return MAMDR_OK;
    const Task& t = g->tasks[g->single ? 0 : domain];
    const float rate = g->cfg.dropout;
    double thr = (double)rate * 4294967296.0;
    const float omb1 = 1.0f - g->cfg.adam_beta1, omb2 = 1.0f - g->cfg.adam_beta2;
    // DIAGNOSTIC (MAMDR_GRAPH_DIAG_REPLAY=1, wrong values, right timing): the second step of a call is captured into a
    // hipGraph and REPLAYED for the full-batch steps that follow -- what a step costs when the host issues one graph launch
    // instead of ~21 kernel launches (launch-rate sensitivity)
    // Compiled in only with -DMAMDR_DIAG (tools/build_variant.sh): a production build that finds the variable set says so
    // once and ignores it -- a replayed step reuses the captured batch, dropout position and Adam alpha.
#ifdef MAMDR_DIAG
    static const bool diag_replay = getenv("MAMDR_GRAPH_DIAG_REPLAY") && atoi(getenv("MAMDR_GRAPH_DIAG_REPLAY")) != 0;
#else
    constexpr bool diag_replay = false;
    static const bool diag_warned = []() {
        if (getenv("MAMDR_GRAPH_DIAG_REPLAY") && atoi(getenv("MAMDR_GRAPH_DIAG_REPLAY")) != 0)
            fprintf(stderr, "mamdr: MAMDR_GRAPH_DIAG_REPLAY is a diagnostic of -DMAMDR_DIAG builds (wrong values by design); ignored\n");
        return true;
    }();
    (void)diag_warned;
#endif
    hipGraph_t dgraph = nullptr;
    hipGraphExec_t dexec = nullptr;
    for (int64_t s = 0; s < n_steps; ++s) {
        const int64_t row_base = (first_step + s) * batch;
        g->wq.clear();          // (a step that failed half-way leaves its queues behind)
        g->tq.clear();
        if (diag_replay && dexec && pass_rows - row_base >= batch) {
            (void)hipGraphLaunch(dexec, g->stream);
            g->global_step += 1;
            continue;
        }
        const bool capturing = diag_replay && !dexec && s == 1 && pass_rows - row_base >= batch && !d_loss_out;
        if (capturing) (void)hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal);
        struct EndCap {
            bool on; hipStream_t st; hipGraph_t* gr; hipGraphExec_t* ex;
            ~EndCap() {
                if (!on) return;
                if (hipStreamEndCapture(st, gr) == hipSuccess && hipGraphInstantiate(ex, *gr, nullptr, nullptr, 0) == hipSuccess)
                    (void)hipGraphLaunch(*ex, st);
            }
        } endcap{capturing, g->stream, &dgraph, &dexec};
        StepCtx sc;
        sc.rows = (int)((pass_rows - row_base) < batch ? (pass_rows - row_base) : batch);
        sc.rp = (sc.rows + GT - 1) / GT * GT;
        sc.train = true;
        sc.seed = dropout_seed;
        sc.step = g->global_step;
        sc.drop_thresh = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)(int64_t)thr;
        sc.keep_scale = (float)(1.0 / (1.0 - (double)rate));
        sc.use_dropout = rate > 0.f && optimizer != MAMDR_OPT_ACCUMULATE;     // the meta pass runs in learning phase 0 (maml.py:107-109)
        if (!sc.use_dropout) sc.keep_scale = 1.0f;
        float alpha = lr;
        if (optimizer == MAMDR_OPT_ADAM) {      // ONE optimizer object for all domain models: its beta powers advance every step
            g->adam_t += 1;
            g->b1p *= g->cfg.adam_beta1;
            g->b2p *= g->cfg.adam_beta2;
            alpha = lr * sqrtf(1.0f - g->b2p) / (1.0f - g->b1p);
        }
        const bool fuse_opt = g->defer_w && g->tail_opt && g->lv_off < 0;
        g->sink.p = nullptr;
        if (fuse_opt) {
            g->sink.g_base = g->grad;
            g->sink.p = g->params + g->table_floats;
            g->sink.m = (optimizer == MAMDR_OPT_ACCUMULATE ? g->accum : g->adam_m) + g->table_floats;
            g->sink.v = g->adam_v + g->table_floats;
            g->sink.optimizer = optimizer;
            g->sink.alpha = alpha;
            g->sink.omb1 = omb1;
            g->sink.omb2 = omb2;
            g->sink.eps = g->cfg.adam_eps;
        }
        GatherArgs ga;
        fill_gather(g, *d, d_perm, row_base, sc, ga);
        GLAUNCH(k_graph_gather, dim3(sc.rp / 4), dim3(256), 0, g->stream, ga);
        const int t_col = task_forward(g, t, sc);
        const Dnn& tower = g->dnns[t.tower];
        HeadArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.act = g->act;
        ha.dact = g->dact;
        ha.ld = g->ld;
        ha.t_col = t_col;
        ha.n_t = tower.layers.back().out;
        if (g->cfg.kind == MAMDR_GRAPH_AUTOINT) {       // head on [attention output | last DNN layer]
            ha.n_t += 3 * ATT_OUT;
            ha.n_plain = 3 * ATT_OUT;
        }
        ha.w = g->params + t.head_w;
        ha.gb = g->params + t.head_gb;
        ha.y = g->y;
        ha.rows = sc.rows;
        ha.rows_pad = sc.rp;
        ha.dlogit = g->dlogit;
        ha.rowloss = g->rowloss;
        ha.extra = g->extra;
        ha.train = 1;
        ha.gate_scale = sc.keep_scale;
        const bool weighted = g->lv_off >= 0;
        if (weighted) {
            ha.log_var = g->params + g->lv_off;
            ha.domrow = g->domrow;
        }
        GLAUNCH(k_graph_head, dim3(sc.rp / 4), dim3(256), 0, g->stream, ha);
        if (d_loss_out && g->tables) refresh_sumsq(g);
        if (d_loss_out || weighted)     // the weighted loss takes d / d log_var from the batch's mean BCE: this launch, always
            GLAUNCH(k_graph_loss, dim3(1), dim3(256), 0, g->stream, g->rowloss, sc.rows, g->params + g->dm_off,
                               g->cfg.n_domain * EMB, g->cfg.l2_emb, g->frozen_sumsq, d_loss_out ? d_loss_out + s : g->eval_acc + 1, 0,
                               g->extra ? g->params + g->lin_d_off : nullptr, g->cfg.n_domain, g->cfg.l2_linear,
                               weighted ? g->params + g->lv_off : nullptr, g->domrow, weighted ? g->G(g->lv_off) : nullptr,
                               g->cfg.n_domain);
        // ---- backward
        // head: dw = t^T dlogit, dgb = sum dlogit
        small_tn(g, g->act + t_col, g->ld, g->dlogit, 1,
                           sc.rp, ha.n_t, 1, g->G(t.head_w), g->G(t.head_gb));
        const size_t ti = t.path.size() - 1;
        // the x columns of the gradient workspace collect d x from every first layer (the domain columns alone while the
        // tables are frozen): the first writer overwrites, the others add
        bool dx_started = false;
        const int dx_first = g->tables ? 0 : 2 * EMB, dx_n = g->tables ? 0 : EMB;
        if (g->single && g->cfg.kind == MAMDR_GRAPH_AUTOINT) {
            // DNN on x as any first layer; then the attention stack from the head's d [attention output] down to d x
            dnn_backward(g, tower, t.col[0], 0, 0, -1, false, dx_first, dx_n, sc);
            for (int l = 2; l >= 0; --l) {
                const int d_in = l == 0 ? EMB : ATT_OUT;
                AttArgs aa;
                memset(&aa, 0, sizeof(aa));
                aa.P = g->attP[l];
                aa.A = g->attA[l];
                aa.Y = g->attY[l];
                aa.dY = g->attdY[l];
                aa.dP = g->attdP[l];
                aa.rows_pad = sc.rp;
                if (l == 2) {
                    aa.dtop = g->dact + g->top_col;
                    aa.dtop_ld = g->ld;
                }
                GLAUNCH(k_graph_att_bwd, dim3(sc.rp / 4), dim3(256), 0, g->stream, aa);
                const float* xin = l == 0 ? g->xt : g->attY[l - 1];
                if (l == 0) {
                    GemmArgs a;
                    memset(&a, 0, sizeof(a));
                    a.A = xin;                      // dW = X^T dP over the 3 B token rows
                    a.lda = d_in;
                    a.B = g->attdP[l];
                    a.ldb = ATT_P;
                    a.C = g->G(g->att_w[l]);
                    a.ldc = ATT_P;
                    launch_wgrad(g, a, d_in, ATT_P, 3 * sc.rp, nullptr, nullptr);
                    memset(&a, 0, sizeof(a));
                    a.A = g->attdP[l];              // d X = dP W^T
                    a.lda = ATT_P;
                    a.B = g->params + g->att_w[l];
                    a.ldb = ATT_P;
                    a.C = g->dxt;
                    a.ldc = d_in;
                    a.K = ATT_P;
                    launch_gemm(1, a, 3 * sc.rp, d_in, g->stream);
                    const int first = g->tables ? 0 : 2 * EMB, n = g->tables ? XDIM : EMB;
                    GLAUNCH(k_graph_add_x, dim3((sc.rp * n + 255) / 256), dim3(256), 0, g->stream, g->dact, g->ld, g->dxt,
                                       sc.rp, first, n);
                } else {
                    small_tn(g, xin, d_in,
                                       g->attdP[l], ATT_P, 3 * sc.rp, d_in, ATT_P, g->G(g->att_w[l]));
                    if (use_tile32(3 * sc.rp, d_in)) {      // d Y[l-1] = d P . W^T, [3 B x 128] x [32 x 128]^T
                        GemmArgs a;
                        memset(&a, 0, sizeof(a));
                        a.A = g->attdP[l];
                        a.lda = ATT_P;
                        a.B = g->params + g->att_w[l];
                        a.ldb = ATT_P;
                        a.C = g->attdY[l - 1];
                        a.ldc = d_in;
                        a.K = ATT_P;
                        launch_gemm(1, a, 3 * sc.rp, d_in, g->stream);
                    } else {
                        GLAUNCH(k_graph_small_nt, dim3((3 * sc.rp * d_in + 255) / 256), dim3(256), 0, g->stream, g->attdP[l],
                                ATT_P, g->params + g->att_w[l], d_in, 3 * sc.rp, g->attdY[l - 1]);
                    }
                }
            }
            if (!g->defer_w) launch_lin_domain_grad(g->stream, g->dlogit,
                               g->domrow, sc.rows, g->params + g->lin_d_off, 2.0f * g->cfg.l2_linear, g->cfg.n_domain,
                               g->G(g->lin_d_off));
        } else if (g->single && g->cfg.kind == MAMDR_GRAPH_CCPM) {
            // d features from the first layer; the convolutions' backward per row; their 48 gradients summed over the batch
            dnn_backward(g, tower, t.col[0], g->f_col, g->f_col, -1, false, 0, 0, sc);
            CcpmArgs ca;
            fill_ccpm(g, sc, ca);
            GLAUNCH(k_graph_ccpm_bwd, dim3(sc.rp / 4), dim3(256), 0, g->stream, ca);
            if (g->defer_w)     // the 48 column sums as a queue entry without a contraction (its end = the bias-gradient workgroups)
                queue_wgrad(g, nullptr, 0, nullptr, 0, nullptr, 0, 48, sc.rp, g->dact + g->cg_col, g->G(g->conv_off));
            else
                launch_colsum(g->stream, g->dact + g->cg_col, g->ld, sc.rp, g->G(g->conv_off), 48);
            if (!g->defer_w) launch_lin_domain_grad(g->stream, g->dlogit,
                               g->domrow, sc.rows, g->params + g->lin_d_off, 2.0f * g->cfg.l2_linear, g->cfg.n_domain,
                               g->G(g->lin_d_off));
        } else if (g->single && (g->cfg.kind == MAMDR_GRAPH_MLP || g->cfg.kind == MAMDR_GRAPH_WDL ||
                                 g->cfg.kind == MAMDR_GRAPH_DEEPFM)) {
            // the DNN on x as any first layer (deepctr.py:26-32,36-38 with any hidden_dim); DeepFM: + the FM term's d x
            dnn_backward(g, tower, t.col[0], 0, 0, -1, false, dx_first, dx_n, sc);
            if (g->cfg.kind == MAMDR_GRAPH_DEEPFM) {
                FeatArgs fa;
                fill_feat(g, t, sc, fa);
                GLAUNCH(k_graph_feat_bwd, dim3(sc.rp / 4), dim3(256), 0, g->stream, fa);
            }
            if (g->has_lin && !g->defer_w)
                launch_lin_domain_grad(g->stream, g->dlogit, g->domrow, sc.rows, g->params + g->lin_d_off, 2.0f * g->cfg.l2_linear,
                                       g->cfg.n_domain, g->G(g->lin_d_off));
        } else if (g->single) {
            const bool nfm = g->cfg.kind == MAMDR_GRAPH_NFM;
            FeatArgs fa;
            fill_feat(g, t, sc, fa);
            if (nfm) {
                // d(interaction columns) from the first layer, then d x = df (sum of the other two fields)
                dnn_backward(g, tower, t.col[0], g->f_col, g->f_col, -1, false, 0, 0, sc);
            } else {
                // rows 0..383 of the first kernel as any first layer on x; rows 384..386 against the inner products
                dnn_backward(g, tower, t.col[0], 0, 0, -1, false, dx_first, dx_n, sc);
                const Layer& L0 = tower.layers[0];
                small_tn(g, g->act + g->f_col, g->ld,
                                   g->dact + t.col[0][0], g->ld, sc.rp, 3, L0.out, g->G(L0.w_off + (int64_t)L0.in * L0.out));
            }
            GLAUNCH(k_graph_feat_bwd, dim3(sc.rp / 4), dim3(256), 0, g->stream, fa);
            if (nfm)
                if (!g->defer_w) launch_lin_domain_grad(g->stream, g->dlogit,
                                   g->domrow, sc.rows, g->params + g->lin_d_off, 2.0f * g->cfg.l2_linear, g->cfg.n_domain,
                                   g->G(g->lin_d_off));
        } else if (g->gated) {
            dnn_backward(g, tower, t.col[ti], t.m_col, t.m_col, -1, false, 0, 0, sc);
            GateArgs gta;
            fill_gate(g, t, sc, gta);
            GLAUNCH(k_graph_gate_bwd, dim3(sc.rp), dim3(256), 0, g->stream, gta);
            const size_t gi = t.mix.size();
            const Dnn& gd = g->dnns[t.gate];
            small_tn(g, g->act + gta.q_col,
                               g->ld, g->dact + t.g_col, g->ld, sc.rp, gta.n_q, gta.n_e, g->G(t.wg_off));
            dnn_backward(g, gd, t.col[gi], 0, 0, -1, dx_started, dx_first, dx_n, sc);
            dx_started = true;
            if (g->group_ok && same_shape(g, t.mix)) {
                std::vector<std::vector<int>> mc(t.col.begin(), t.col.begin() + t.mix.size());
                dnn_backward_group(g, t.mix, mc, 0, dx_started, dx_first, dx_n, sc);
                dx_started = true;
            } else {
                for (size_t e = 0; e < t.mix.size(); ++e) {
                    dnn_backward(g, g->dnns[t.mix[e]], t.col[e], 0, 0, -1, dx_started, dx_first, dx_n, sc);
                    dx_started = true;
                }
            }
        } else {
            // the tower's input IS the bottom's output: its gradient passes through the bottom's last relu / dropout gate
            dnn_backward(g, tower, t.col[ti], t.col[0].back(), t.col[0].back(), t.col[0].back(), false, 0, 0, sc);
            dnn_backward(g, g->dnns[t.mix[0]], t.col[0], 0, 0, -1, false, dx_first, dx_n, sc);
        }
        if (g->defer_w) {       // every layer's dW / db of this step, the narrow contractions, the domain table's gradient:
            const bool lin = g->single && g->has_lin;
            const DomainGradJob dg{g->dact, g->ld, 2 * EMB, g->domrow, sc.rows, g->params + g->dm_off, 2.0f * g->cfg.l2_emb,
                                   g->G(g->dm_off), g->cfg.n_domain, g->dlogit, lin ? g->params + g->lin_d_off : nullptr,
                                   2.0f * g->cfg.l2_linear, lin ? g->G(g->lin_d_off) : nullptr};      // one pair of launches
            flush_wgrads(g, &dg);
        } else {
            GLAUNCH(k_graph_domain_grad, dim3(EMB / CS_COLS, g->cfg.n_domain), dim3(256), 0, g->stream, g->dact, g->ld, 2 * EMB,
                    g->domrow, sc.rows, g->params + g->dm_off, 2.0f * g->cfg.l2_emb, g->G(g->dm_off));
        }
        if (g->tables) {
            // TF1's dense step over both tables: g = 2 l2 p + scatter-add of d x[:, user | item columns]
            EmbStepArgs ea;
            memset(&ea, 0, sizeof(ea));
            float* const slot_m = optimizer == MAMDR_OPT_ACCUMULATE ? g->accum : g->adam_m;
            ea.p = g->params;
            ea.m = slot_m;
            ea.v = g->adam_v;
            ea.dxe = g->dact;
            ea.dx_ld = g->ld;
            ea.dlogit = g->dlogit;
            ea.rows = sc.rows;
            ea.opt.optimizer = optimizer;
            ea.opt.alpha = alpha;
            ea.opt.omb1 = omb1;
            ea.opt.omb2 = omb2;
            ea.opt.eps = g->cfg.adam_eps;
            ea.opt.two_l2 = 2.0f * g->cfg.l2_emb;
            ea.t[0].n_rows = g->cfg.n_user;
            ea.t[0].brow = g->urow;
            ea.t[0].map = g->map_u;
            ea.t[0].gbuf = g->gbuf_u;
            ea.t[0].hasdup = g->hasdup_u;
            ea.t[0].dx_off = 0;
            ea.t[1].n_rows = g->cfg.n_item;
            ea.t[1].brow = g->irow;
            ea.t[1].map = g->map_i;
            ea.t[1].gbuf = g->gbuf_i;
            ea.t[1].hasdup = g->hasdup_i;
            ea.t[1].dx_off = EMB;
            if (g->has_lin) {       // their 1-d linear tables: scatter-add of d loss / d logit, same rule
                ea.two_l2_lin = 2.0f * g->cfg.l2_linear;
                ea.t[0].lin_p = g->params + g->lin_u_off;
                ea.t[0].lin_m = slot_m + g->lin_u_off;
                ea.t[0].lin_v = g->adam_v + g->lin_u_off;
                ea.t[0].glin = g->glin_u;
                ea.t[1].lin_p = g->params + g->lin_i_off;
                ea.t[1].lin_m = slot_m + g->lin_i_off;
                ea.t[1].lin_v = g->adam_v + g->lin_i_off;
                ea.t[1].glin = g->glin_i;
            }
            launch_emb_reduce(ea, g->stream);
            launch_emb_sweep(ea, g->stream);
            if (g->has_lin) launch_lin_sweep(ea, g->stream);      // (reads the row maps, then resets them)
        }
        // ---- optimiser on the two ranges this task's model trains (one launch; none when the tail launch stepped them)
        if (!fuse_opt || !g->sink.p) {      // (sink dropped: a queue overflowed in mid-step)
            const int64_t off[2] = {g->dm_off, t.blk_off}, cnt[2] = {g->shared_end - g->dm_off, t.blk_end - t.blk_off};
            AdamArgs aa;
            aa.p = g->params;
            aa.m = optimizer == MAMDR_OPT_ACCUMULATE ? g->accum : g->adam_m;
            aa.v = g->adam_v;
            aa.g = g->grad - g->table_floats;       // G(off) = grad + off - table_floats
            for (int k = 0; k < 2; ++k) {
                aa.off4[k] = off[k] / 4;
                aa.n4[k] = cnt[k] / 4;
            }
            aa.optimizer = optimizer;
            aa.alpha = alpha;
            aa.omb1 = omb1;
            aa.omb2 = omb2;
            aa.eps = g->cfg.adam_eps;
            const int64_t n4 = aa.n4[0] + aa.n4[1];
            if (n4 > 0) GLAUNCH(k_graph_adam, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, g->stream, aa);
        }
        g->global_step += 1;
    }
    if (dexec) (void)hipGraphExecDestroy(dexec);
    if (dgraph) (void)hipGraphDestroy(dgraph);
    GHIP(hipGetLastError());
    return MAMDR_OK;
}

int mamdr_graph_bind_accumulator(mamdr_graph* g, float* d_acc) {
    if (check(g)) return MAMDR_EINVAL;
    if (d_acc && ((uintptr_t)d_acc & 15)) return gfail(MAMDR_EINVAL, "accumulator is not 16-byte aligned");
    g->accum = d_acc;
    return MAMDR_OK;
}

int mamdr_graph_eval_domain(mamdr_graph* g, int domain, int split, int32_t batch, float* d_loss_out, uint32_t* d_hist,
                            float* d_pred_out) {
    if (check(g)) return MAMDR_EINVAL;
    if (ready(g)) return MAMDR_ESTATE;
    SplitData* d = split_of(g, domain, split);
    if (!d || !d->bound) return gfail(MAMDR_ESTATE, "split %d of domain %d is not bound", split, domain);
    if (batch <= 0 || batch > g->cfg.max_batch) return gfail(MAMDR_EINVAL, "batch %d outside (0, max_batch=%d]", batch, g->cfg.max_batch);
    if (!d_loss_out || !d_hist) return gfail(MAMDR_EINVAL, "null output pointer");
    if (d->n <= 0) return gfail(MAMDR_EINVAL, "empty split");
    const Task& t = g->tasks[g->single ? 0 : domain];
    GHIP(hipMemsetAsync(d_hist, 0, 2 * 501 * sizeof(uint32_t), g->stream));
    GHIP(hipMemsetAsync(g->eval_acc, 0, sizeof(float), g->stream));
    if (g->tables) refresh_sumsq(g);
    const int64_t n_batches = (d->n + batch - 1) / batch;
    for (int64_t b = 0; b < n_batches; ++b) {
        StepCtx sc;
        memset(&sc, 0, sizeof(sc));
        const int64_t row_base = b * batch;
        sc.rows = (int)((d->n - row_base) < batch ? (d->n - row_base) : batch);
        sc.rp = (sc.rows + GT - 1) / GT * GT;
        sc.keep_scale = 1.0f;
        GatherArgs ga;
        fill_gather(g, *d, nullptr, row_base, sc, ga);
        GLAUNCH(k_graph_gather, dim3(sc.rp / 4), dim3(256), 0, g->stream, ga);
        const int t_col = task_forward(g, t, sc);
        HeadArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.act = g->act;
        ha.dact = g->dact;
        ha.ld = g->ld;
        ha.t_col = t_col;
        ha.n_t = g->dnns[t.tower].layers.back().out + (g->cfg.kind == MAMDR_GRAPH_AUTOINT ? 3 * ATT_OUT : 0);
        ha.w = g->params + t.head_w;
        ha.gb = g->params + t.head_gb;
        ha.y = g->y;
        ha.rows = sc.rows;
        ha.rows_pad = sc.rp;
        ha.dlogit = g->dlogit;
        ha.rowloss = g->rowloss;
        ha.extra = g->extra;
        ha.gate_scale = 1.0f;
        ha.thresholds = g->thresholds;
        ha.hist = d_hist;
        ha.pred_out = d_pred_out ? d_pred_out + row_base : nullptr;
        GLAUNCH(k_graph_head, dim3(sc.rp / 4), dim3(256), 0, g->stream, ha);
        GLAUNCH(k_graph_loss, dim3(1), dim3(256), 0, g->stream, g->rowloss, sc.rows, g->params + g->dm_off,
                           g->cfg.n_domain * EMB, g->cfg.l2_emb, g->frozen_sumsq, g->eval_acc, 1,
                           g->extra ? g->params + g->lin_d_off : nullptr, g->cfg.n_domain, g->cfg.l2_linear);
    }
    GLAUNCH(k_graph_scale, dim3(1), dim3(1), 0, g->stream, g->eval_acc, 1.0f / (float)n_batches);
    GHIP(hipMemcpyAsync(d_loss_out, g->eval_acc, sizeof(float), hipMemcpyDeviceToDevice, g->stream));
    GHIP(hipGetLastError());
    return MAMDR_OK;
}

}  // extern "C"
