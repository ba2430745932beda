// Internal launch interface between the C ABI (mamdr_api.hip) and the kernel
// translation units.  Not part of the public boundary (include/mamdr_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "mamdr_device.h"

namespace mamdr {

// optimiser hyper-parameters of one step (shared by the dense and the table paths)
struct OptArgsLite {
    int optimizer;             // 0 adam, 1 sgd
    float alpha;               // adam: lr*sqrt(1-b2^t)/(1-b1^t); sgd: lr
    float omb1, omb2, eps;
    float two_l2;
};
constexpr int32_t EMB_UNTOUCHED = 0x7fffffff;

// Profiling (mamdr_api.hip: Prof): while a timed launch is being issued, g_prof_stop is the event the launch
// carries as its OWN stop event (hipExtLaunchKernelGGL: recorded by the kernel's completion, no marker packet
// between the kernels).  A kernel's time is then its stop event minus the stop event of the kernel before it.
extern thread_local hipEvent_t g_prof_stop;
int env_warn_unknown();          // mamdr_api.hip: MAMDR_* names of the environment missing from env_registry.h, reported once
#define MAMDR_LAUNCH(kernel, grid, block, lds, stream, ...)                                               \
    do {                                                                                                  \
        if (::mamdr::g_prof_stop) {                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ::mamdr::g_prof_stop, 0, __VA_ARGS__); \
            ::mamdr::g_prof_stop = nullptr;                                                               \
        } else {                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                            \
        }                                                                                                 \
    } while (0)

// ---- domain table under k_wgrad_adam.  Its gradient is a sum over the S workgroups' partials, i.e. a device-wide
// dependency inside the step.  Instead of a kernel of its own per step (4 us of launch + drain for 1,280 elements)
// the step stays PENDING: the next step's tower kernel applies it on the fly -- every workgroup recomputes the
// row(s) its samples use from (p, m, v before the step, the partials), workgroup d < n_domain also writes row d
// back -- and k_dm_finish materialises the last step of a mamdr_train_steps call.  All three sites run dm_step4:
// explicit roundings, the same bits wherever it is instantiated.
constexpr int DM_PARTS = 32;       // partial gradients per step = S workgroups of k_wgrad_adam
constexpr int DM_PAIRS = DM_PARTS / 2;
struct DmStep {
    const float* snap;         // [3][n_domain][EMB]: p, m, v of the domain table BEFORE the pending step; null = none pending
    const float* pdm;          // [n_part][n_domain][EMB] partial gradients of the pending step
    int n_part, n_domain;
    int optimizer;             // of the pending step: 0 adam, 1 sgd, 2 accumulate
    float alpha, omb1, omb2, eps, two_l2;
};
// Canonical arithmetic (every site, bit for bit): pair sums s_j = pdm[2 j] + pdm[2 j + 1] (j < 16), group sums
// G_q = ((s_4q + s_4q+1) + s_4q+2) + s_4q+3 (q < 4), g = ((G_0 + G_1) + G_2) + G_3, g += 2 l2 p, optimiser step with
// separately rounded operations.  (A tree, so that four lanes can each sum a quarter of the partials.)
__device__ __forceinline__ f32x4 dm_pair(const DmStep& q, int d, int c4, int j) {
#pragma clang fp contract(off)
    const size_t row = (size_t)d * EMB + 4 * c4, plane = (size_t)q.n_domain * EMB;
    const f32x4 a = *reinterpret_cast<const f32x4*>(q.pdm + (size_t)(2 * j) * plane + row);
    const f32x4 b = *reinterpret_cast<const f32x4*>(q.pdm + (size_t)(2 * j + 1) * plane + row);
    return a + b;
}
// (p, m, v of the row BEFORE the step: separate, so that a caller can request them ahead of a barrier)
__device__ __forceinline__ void dm_load4(const DmStep& q, int d, int c4, f32x4& p, f32x4& m, f32x4& v) {
    const size_t row = (size_t)d * EMB + 4 * c4, plane = (size_t)q.n_domain * EMB;
    p = *reinterpret_cast<const f32x4*>(q.snap + row);
    m = *reinterpret_cast<const f32x4*>(q.snap + plane + row);
    v = *reinterpret_cast<const f32x4*>(q.snap + 2 * plane + row);
}
__device__ __forceinline__ void dm_apply4(const DmStep& q, f32x4 g, f32x4& p, f32x4& m, f32x4& v) {
#pragma clang fp contract(off)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gk = g[k] + q.two_l2 * p[k];
        if (q.optimizer == 0) {
            m[k] = m[k] + (gk - m[k]) * q.omb1;
            v[k] = v[k] + (gk * gk - v[k]) * q.omb2;
            p[k] = p[k] - (m[k] * q.alpha) / (sqrtf(v[k]) + q.eps);
        } else if (q.optimizer == 1) {
            p[k] = p[k] - gk * q.alpha;
        } else {
            m[k] = m[k] + gk;
        }
    }
}
// one lane does it all (the writer workgroups and k_dm_finish: off the critical path): four batches of loads
__device__ __forceinline__ void dm_step4(const DmStep& q, int d, int c4, f32x4& p, f32x4& m, f32x4& v) {
#pragma clang fp contract(off)
    static_assert(DM_PAIRS == 16, "four groups of four pair sums");
    dm_load4(q, d, c4, p, m, v);
    f32x4 G[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        f32x4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = dm_pair(q, d, c4, 4 * gq + k);
        G[gq] = ((t[0] + t[1]) + t[2]) + t[3];
    }
    const f32x4 g = ((G[0] + G[1]) + G[2]) + G[3];
    dm_apply4(q, g, p, m, v);
}

// One launch of the fused tower kernel: gather -> MLP forward -> BCE -> (train:
// backward activation chain) over a contiguous range of positions of one split.
struct TowerArgs {
    // tables and dense parameters
    const float* user_tab;
    const float* item_tab;
    const float* dense;        // dense block base (domain table first)
    DenseLayout L;
    int n_user, n_item, n_domain;
    // bound split columns
    const int32_t* uid;
    const int32_t* pid;
    const int32_t* dom;
    const float* label;
    const int32_t* perm;       // nullable
    int64_t row_base;          // position (in perm order) of row 0 of this launch
    int64_t n_rows_split;      // rows in the bound split (for clamping)
    int rows;                  // rows covered by this launch (train: rows of the batch)
    int batch;                 // eval: batch size (loss is averaged per batch); train: == rows
    // dropout stream
    uint32_t seed, step, drop_thresh;
    float keep_scale;
    int use_dropout;
    // train outputs (workspace)
    float* acts;               // [rows_pad][ACT_LD]
    float* dz;                 // [rows_pad][DZ_LD]
    float* dlogit;             // [rows_pad]
    int32_t* domrow;           // [rows_pad]
    // trainable user / item tables only (null otherwise)
    float* dxe;                // [rows_pad][dx_ld]  d loss / d [user | item (| domain)] embedding row
    int dx_ld;                 // 256, or 384 for the Star tower
    const float* pn_aff;       // Star: [scale 384 | shift 384 | mean 384 | inv 384 | ...] of PartitionedNorm, null otherwise
    float* pn_part;            // Star training: [tiles][2][384] per-tile sums of PartitionedNorm's backward (s1 = sum dxn,
                               // s2 = sum dxn * xhat), written by the tower's tail -- k_star_pnb_partial's work without its launch
    int32_t* urow;             // [rows_pad] user row of each batch position (-1 = padding)
    int32_t* irow;             // [rows_pad]
    int32_t* map_u;            // [n_user] / [n_item]: atomicMin of the batch position touching the row (null: frozen tables)
    int32_t* map_i;
    float* loss_part;          // [tiles] sum of per-row BCE of the tile
    // DeepFM: logit += FM second-order term + linear tables
    int deepfm;                // FM instances: 1 DeepFM, 2 WDL (linear tables only), 3 PNN (inner products of the field pairs)
    float* ipbuf;              // PNN: [rows_pad][4] inner products <u,i> <u,d> <i,d> of the batch's rows (train)
    const float* lin_user;     // [n_user] / [n_item] 1-d tables; null = frozen at their zero initialisation
    const float* lin_item;
    float* fmq;                // train: [rows_pad][EMB] dlogit * (user + item embedding), for the domain-table gradient
    // uncertainty weighting (model_zoo/uncertainty_weight/weighted_loss.py:30-43): the BCE term of the loss is
    // divided by var^2, var = dense[uw_off] (the batch's domain); -1 = off
    int uw_off;
    const float* wT;           // k_tower4 only: transposed W1 / W2 copies
    // k_wgrad_adam path: the tower also snapshots W0[256:384, :] and the domain table (it only READS the weights;
    // the fused kernel steps them while other workgroups still need the pre-update values); null otherwise
    float* w0dom_snap;
    // domain table (see DmStep): the pending step to apply on the fly, the live rows workgroup d < n_domain writes
    // back, and the snapshot [3][n_domain][EMB] of (p, m, v) as this step's forward pass saw them
    // pre-gathered pass (k_pass_prep; frozen tables on the k_wgrad_adam path): row i of this launch reads its
    // [user | item] embedding rows, domain and label at xpre[i], pdom[i], plabel[i] -- no chain of dependent loads
    const float* xpre;         // [rows][2 EMB], null = gather through perm / uid / pid
    const int32_t* pdom;
    const float* plabel;
    int no_w1l;                // k_tower4: keep streaming W1 / W1^T (MAMDR_T4_NO_W1L=1, diagnostic)
    int w2_direct;             // k_tower4<.., W1L, PRE>: the backward pass reads W2 itself instead of the copy W2T (the first
                               // step of a call whose copies are stale)
    DmStep dms;
    int dm_hint;               // the domain the caller expects every row of the batch to carry (the pass's domain)
    float* dm_live_p;
    float* dm_live_m;
    float* dm_live_v;
    float* dm_snap_out;
    // eval outputs
    const float* thresholds;   // 500 fp32 AUC thresholds
    uint32_t* hist;            // [2][501]
    float* pred_out;           // nullable, [n_rows] in position order
#ifdef MAMDR_STAMPS
    unsigned long long* stamps; // diagnostic build only: [tiles][16] s_memtime stamps
#endif
};

// weight-gradient GEMMs (K = batch rows) + bias / output-layer / domain-table sums
struct TileDesc {
    int a_kind, a_off;         // 0: acts column block, 1: ones (row 0), 2: one-hot(domain) rows a_off.., 3: ipbuf columns (PNN)
    int b_kind, b_off;         // 0: dz column block, 1: dlogit (col 0), 2: fmq column block
    int dst_off, dst_ld;       // destination in the dense-block gradient slab
    int m_valid, n_valid;      // valid rows / cols of the 32x32 tile
    int big;                   // 1: full 64x64 tile of acts^T dz (LDS-staged path), 0: 32x32 tile with synthesised operands
};

struct WgradArgs {
    const float* acts;
    const float* dz;
    const float* dlogit;
    const int32_t* domrow;
    const float* fmq;          // DeepFM / PNN only
    const float* ipbuf;        // PNN only: [rows_pad][4]
    const TileDesc* tiles;
    int n_tiles;
    int rows_pad;              // batch rows rounded up to TILE_ROWS
    int n_groups;              // K-split groups (one gradient slab each)
    int rows_per_group;        // multiple of 8
    float* slabs;              // [n_groups][slab_ld]
    int slab_ld;
    // loss of the step (optional)
    const float* loss_part;
    int n_loss_tiles;
    int rows;                  // actual batch rows
    const float* dense;        // for the domain-table regulariser
    int dm_count;
    float l2_emb;
    const float* frozen_sumsq; // [4] sums of squares: user, item table; DeepFM linear user, item table
    int ld_off, ld_count;      // DeepFM: linear domain table inside the dense block (regulariser l2_lin)
    float l2_lin;
    int lv_off, lv_count, uw_d; // uncertainty weighting: per-domain scalars in the dense block, the batch's domain
    float* loss_out;           // nullable: 1 float
    const float* w0dom;        // W0[256:384, :] (live weights)
    float* w0dom_copy;         // its pre-update snapshot, read by k_update
    float* dm_copy;            // pre-update snapshot of the domain table (k_update: dW0[256:384] by linearity), nullable
#ifdef MAMDR_STAMPS
    unsigned long long* stamps; // diagnostic build only
#endif
};

struct UpdateArgs {
    float* p;                  // dense block of params
    float* m;
    float* v;
    const float* slabs;
    int n_groups;
    int slab_ld;
    int count4;                // float4 elements
    int dm_count;              // elements [0, dm_count): domain table, gradient = S . W0dom^T + 2*l2*p
    int s_off;                 // offset of S = onehot(domain)^T dz1 ([n_domain][256]) inside a slab
    const float* w0dom_copy;
    const float* dm_copy;      // not null: dW0[256:384, :] = Dm^T . S (the slabs carry no tiles for those rows)
    int n_domain;
    int s2_off;                // DeepFM: offset of S2 = onehot(domain)^T fmq ([n_domain][EMB]) inside a slab, 0 = none
    int ld_off, ld_count;      // DeepFM: linear domain table (gradient += 2 l2_lin p)
    float two_l2_lin;
    float two_l2;
    int optimizer;             // 0 adam, 1 sgd
    float alpha;               // adam: lr*sqrt(1-b2^t)/(1-b1^t); sgd: lr
    float omb1, omb2, eps;
    float* wT;                 // nullable: transposed W1 / W2 copies to keep current
    int w1_off, w2_off;        // offsets of W1 / W2 in the dense block
    int w0_off, w0t;           // W0 offset; w0t: also keep W0T (trainable tables)
    int no_sdm;                // NFM: rows 256..383 of W0 do not meet the domain row -- no S . W0dom^T term in the table's gradient
#ifdef MAMDR_STAMPS
    unsigned long long* stamps; // diagnostic build only: [workgroups][4] s_memtime stamps (entry, operands summed, exit)
#endif
};

// pre-update snapshot of W0[256:384, :] for k_wgrad_adam, by the LAST wave of every tower workgroup (it is not on
// the row bookkeeping's critical path): float4 i for i in this workgroup's share
__device__ __forceinline__ void tower_snapshots(const TowerArgs& a, int n_threads, int n_tiles) {
    if (a.w0dom_snap == nullptr) return;
    const int st = (int)threadIdx.x - (n_threads - 64);
    if (st < 0) return;
    const int total = EMB * H1 / 4;
    const int per = (total + n_tiles - 1) / n_tiles;
    const int i0 = (int)blockIdx.x * per, i1 = min(i0 + per, total);
    const f32x4* w0dom = reinterpret_cast<const f32x4*>(a.dense + a.L.w0 + 2 * EMB * H1);
    for (int i = i0 + st; i < i1; i += 64) reinterpret_cast<f32x4*>(a.w0dom_snap)[i] = w0dom[i];
}
// ---- the tile workgroups' domain-table duty (DmStep).  Everything is requested at kernel start, beside the weight
// prefetch, from the domain the caller expects (dm_hint: per-domain datasets carry one domain per batch), so the
// round trip to the partials hides behind the row bookkeeping's own dependent loads:
//   reader  all 512 lanes fetch one pair sum each of row dm_hint (16 pairs x 32 float4) -> LDS -> behind a barrier
//           32 lanes add them in order and step the row: the tile's x[256:384] (if the batch's rows do carry
//           dm_hint; otherwise every lane of a domain segment runs dm_step4 alone)
//   writer  workgroup d < n_domain does the same for row d with lanes 32..63 and writes it back: live p / m / v and
//           the snapshot [3][n_domain][EMB] this step's k_wgrad_adam and the next step's towers read
__device__ __forceinline__ void dm_apply1(const DmStep& q, float g, float& p, float& m, float& v) {
#pragma clang fp contract(off)
    const float gk = g + q.two_l2 * p;
    if (q.optimizer == 0) {
        m = m + (gk - m) * q.omb1;
        v = v + (gk * gk - v) * q.omb2;
        p = p - (m * q.alpha) / (sqrtf(v) + q.eps);
    } else if (q.optimizer == 1) {
        p = p - gk * q.alpha;
    } else {
        m = m + gk;
    }
}
// ---- the 4-row tower's form, without LDS and without a barrier: wave w owns the 16 domain columns its second
// layer-0 segment contracts; lane (l = lane & 15, q = lane >> 4) loads partials 8 q .. 8 q + 7 of column 16 w + l,
// sums its group G_q, and four cross-lane reads complete the tree.  Requested behind the row bookkeeping, consumed
// between layer 0's segments.
struct DmWave {
    float r[8], w[8];          // partials of the batch's domain row / of the row this workgroup writes back
    float rp, rm, rv, wp, wm, wv;
};
// addresses first (pure arithmetic, from the domain the caller expects: done while the row bookkeeping's loads are in
// flight), loads second (behind the bookkeeping barrier: a dozen instructions)
struct DmWaveAddr {
    const float* rb;           // partial 8 q of column c of row dm_hint; + k * plane for the others
    const float* wb;           // ... of row `tile`
    const float* rs;           // snapshot p of (dm_hint, c); + plane: m, + 2 plane: v
    const float* ws;           // snapshot (or live) p of (tile, c)
};
__device__ __forceinline__ void dm_wave_addr(const TowerArgs& a, int tile, DmWaveAddr& q) {
    const int lane = (int)threadIdx.x & 63, l = lane & 15, gq = lane >> 4, c = 16 * ((int)threadIdx.x >> 6) + l;
    const size_t plane = (size_t)a.n_domain * EMB;
    const size_t er = (size_t)a.dm_hint * EMB + c, ew = (size_t)min(tile, a.n_domain - 1) * EMB + c;
    q.rb = a.dms.pdm + (size_t)(8 * gq) * plane + er;
    q.wb = a.dms.pdm + (size_t)(8 * gq) * plane + ew;
    q.rs = a.dms.snap + er;
    q.ws = (a.dms.snap ? a.dms.snap : a.dm_live_p) + ew;
}
__device__ __forceinline__ void dm_wave_begin(const TowerArgs& a, int tile, bool do_read, const DmWaveAddr& q, DmWave& t) {
    const bool pend = a.dms.snap != nullptr, wr = tile < a.n_domain;      // uniform
    const size_t plane = (size_t)a.n_domain * EMB;
#pragma unroll
    for (int k = 0; k < 8; ++k) t.r[k] = t.w[k] = 0.f;
    t.rp = t.rm = t.rv = t.wp = t.wm = t.wv = 0.f;
    if (pend && do_read) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t.r[k] = q.rb[(size_t)k * plane];
        t.rp = q.rs[0];
        t.rm = q.rs[plane];
        t.rv = q.rs[2 * plane];
    }
    if (wr) {
        if (pend) {
#pragma unroll
            for (int k = 0; k < 8; ++k) t.w[k] = q.wb[(size_t)k * plane];
            t.wp = q.ws[0];
            t.wm = q.ws[plane];
            t.wv = q.ws[2 * plane];
        } else {
            const size_t off = (size_t)(q.ws - a.dm_live_p);
            t.wp = q.ws[0];
            t.wm = a.dm_live_m[off];
            t.wv = a.dm_live_v[off];
        }
    }
}
__device__ __forceinline__ float dm_wave_sum(const float (&x)[8]) {
#pragma clang fp contract(off)
    const int l = (int)threadIdx.x & 15;
    const float G = (((x[0] + x[1]) + (x[2] + x[3])) + (x[4] + x[5])) + (x[6] + x[7]);
    const float G0 = __shfl(G, l), G1 = __shfl(G, l + 16), G2 = __shfl(G, l + 32), G3 = __shfl(G, l + 48);
    return ((G0 + G1) + G2) + G3;
}
// one element: the 16 staged pair sums of column c in order, then the step (the scalar form of dm_step4: same bits)
__device__ __forceinline__ void dm_elem_finish(const DmStep& q, int c, const float* parts, float& p, float& m, float& v) {
#pragma clang fp contract(off)
    float t[DM_PAIRS];
#pragma unroll
    for (int j = 0; j < DM_PAIRS; ++j) t[j] = parts[j * EMB + c];
    float G[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) G[gq] = ((t[4 * gq] + t[4 * gq + 1]) + t[4 * gq + 2]) + t[4 * gq + 3];
    dm_apply1(q, ((G[0] + G[1]) + G[2]) + G[3], p, m, v);
}
// per-lane state of the tile workgroups' duty.  Only waves 4..7 take part: wave 0 runs the row bookkeeping's chain of
// dependent loads, and loads retire in order -- anything requested ahead of that chain is waited for with it.
//   waves 4..7 (256 lanes)  two pair sums each of the hinted row (and of the row to write back): lane L' = lane - 256
//                            owns float4 c4 = L' & 31 of pairs g and g + 8, g = L' >> 5
//   waves 4, 5 (reader)     lane L' < 128: (p, m, v) of column L' of the hinted row
//   waves 6, 7 (writer)     lane L' - 128: the same for the row this workgroup writes back
struct DmTile {
    f32x4 pr[2][2], pw[2][2];  // the two partials of each pair, as loaded: added only when staged (an add right behind
                               // the loads would make the wave wait out their round trip before its next barrier)
    float p, m, v;
};
__device__ __forceinline__ void dm_pair_load(const DmStep& q, int d, int c4, int j, f32x4 (&t)[2]) {
    const size_t row = (size_t)d * EMB + 4 * c4, plane = (size_t)q.n_domain * EMB;
    t[0] = *reinterpret_cast<const f32x4*>(q.pdm + (size_t)(2 * j) * plane + row);
    t[1] = *reinterpret_cast<const f32x4*>(q.pdm + (size_t)(2 * j + 1) * plane + row);
}
__device__ __forceinline__ f32x4 dm_pair_add(const f32x4 (&t)[2]) {
#pragma clang fp contract(off)
    return t[0] + t[1];
}
__device__ __forceinline__ void dm_tile_begin(const TowerArgs& a, int tile, int d_read, DmTile& t) {
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // scalar branches below: no divergence
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    t.pr[0][0] = t.pr[0][1] = t.pr[1][0] = t.pr[1][1] = zero;
    t.pw[0][0] = t.pw[0][1] = t.pw[1][0] = t.pw[1][1] = zero;
    t.p = t.m = t.v = 0.f;
    if (wv < 4) return;
    const bool pend = a.dms.snap != nullptr, wr = tile < a.n_domain;      // both uniform
    const int dw = min(tile, a.n_domain - 1);
    const int lp = (int)threadIdx.x - 256, grp = lp >> 5, c4 = lp & 31;
    if (pend) {
        dm_pair_load(a.dms, d_read, c4, grp, t.pr[0]);
        dm_pair_load(a.dms, d_read, c4, grp + 8, t.pr[1]);
        if (wr) {
            dm_pair_load(a.dms, dw, c4, grp, t.pw[0]);
            dm_pair_load(a.dms, dw, c4, grp + 8, t.pw[1]);
        }
    }
    const size_t plane = (size_t)a.n_domain * EMB;
    const int c = lp & (EMB - 1);
    if (wv < 6) {
        if (pend) {
            const size_t e = (size_t)d_read * EMB + c;
            t.p = a.dms.snap[e];
            t.m = a.dms.snap[plane + e];
            t.v = a.dms.snap[2 * plane + e];
        }
    } else if (wr) {
        const size_t e = (size_t)dw * EMB + c;
        if (pend) {
            t.p = a.dms.snap[e];
            t.m = a.dms.snap[plane + e];
            t.v = a.dms.snap[2 * plane + e];
        } else {
            t.p = a.dm_live_p[e];
            t.m = a.dm_live_m[e];
            t.v = a.dm_live_v[e];
        }
    }
}
// pair sums -> LDS parts[2][16][128] floats (16 KB)
__device__ __forceinline__ void dm_tile_stage(const TowerArgs& a, int tile, const DmTile& t, float* parts) {
    if (!a.dms.snap || threadIdx.x < 256) return;
    const int lp = (int)threadIdx.x - 256;
    *reinterpret_cast<f32x4*>(parts + lp * 4) = dm_pair_add(t.pr[0]);
    *reinterpret_cast<f32x4*>(parts + 1024 + lp * 4) = dm_pair_add(t.pr[1]);
    if (tile < a.n_domain) {
        *reinterpret_cast<f32x4*>(parts + 2048 + lp * 4) = dm_pair_add(t.pw[0]);
        *reinterpret_cast<f32x4*>(parts + 3072 + lp * 4) = dm_pair_add(t.pw[1]);
    }
}
// behind the barrier that follows dm_tile_stage: the hinted row's element c after the pending step
__device__ __forceinline__ float dm_tile_reader(const TowerArgs& a, int c, const float* parts, const DmTile& t) {
    float p = t.p, m = t.m, v = t.v;
    dm_elem_finish(a.dms, c, parts, p, m, v);
    return p;
}
__device__ __forceinline__ void dm_store_elem(const TowerArgs& a, size_t e, bool stepped, float p, float m, float v) {
    const size_t plane = (size_t)a.n_domain * EMB;
    if (stepped) {
        if (a.dms.optimizer != 2) a.dm_live_p[e] = p;
        if (a.dms.optimizer != 1) a.dm_live_m[e] = m;
        if (a.dms.optimizer == 0) a.dm_live_v[e] = v;
    }
    a.dm_snap_out[e] = p;
    a.dm_snap_out[plane + e] = m;
    a.dm_snap_out[2 * plane + e] = v;
}
// writer lanes (waves 6, 7: c = lane - 384), behind the same barrier
__device__ __forceinline__ void dm_tile_writer(const TowerArgs& a, int tile, int n_tiles, int c, const float* parts,
                                               const DmTile& t) {
    const bool pend = a.dms.snap != nullptr;
    if (tile < a.n_domain) {
        float p = t.p, m = t.m, v = t.v;
        if (pend) dm_elem_finish(a.dms, c, parts + 2048, p, m, v);
        dm_store_elem(a, (size_t)tile * EMB + c, pend, p, m, v);
    }
    // grids smaller than the domain count (tiny batches): the remaining rows, one lane chain per float4
    if (c < EMB / 4) {
        for (int d = tile + n_tiles; d < a.n_domain; d += n_tiles) {
            f32x4 p, m, v;
            const size_t row = (size_t)d * EMB + 4 * c, plane = (size_t)a.n_domain * EMB;
            if (pend) {
                dm_step4(a.dms, d, c, p, m, v);
                if (a.dms.optimizer != 2) *reinterpret_cast<f32x4*>(a.dm_live_p + row) = p;
                if (a.dms.optimizer != 1) *reinterpret_cast<f32x4*>(a.dm_live_m + row) = m;
                if (a.dms.optimizer == 0) *reinterpret_cast<f32x4*>(a.dm_live_v + row) = v;
            } else {
                p = *reinterpret_cast<const f32x4*>(a.dm_live_p + row);
                m = *reinterpret_cast<const f32x4*>(a.dm_live_m + row);
                v = *reinterpret_cast<const f32x4*>(a.dm_live_v + row);
            }
            *reinterpret_cast<f32x4*>(a.dm_snap_out + row) = p;
            *reinterpret_cast<f32x4*>(a.dm_snap_out + plane + row) = m;
            *reinterpret_cast<f32x4*>(a.dm_snap_out + 2 * plane + row) = v;
        }
    }
}

// k_wgrad_adam (fused_kernels.hip): weight gradients + optimiser step of the dense block in one launch
struct FusedArgs {
    const float* acts;         // [rows_pad][ACT_LD]
    const float* dz;           // [rows_pad][DZ_LD]
    const float* dlogit;       // [rows_pad]
    const int32_t* domrow;     // [rows_pad]
    const float* xa;           // A operand of dW0[0:256, :]: rows of [user | item] embeddings, leading dimension xa_ld
    int xa_ld;
    int rows_pad, rows;
    float* p;                  // dense block of params / slots (m = accumulator for optimizer 2)
    float* m;
    float* v;
    DenseLayout L;
    int n_domain;
    const float* w0dom_snap;   // pre-update W0[256:384, :] and domain table (tower_snapshots)
    const float* dm_snap;
    float* pdm;                // [8][n_domain][EMB] partial domain-table gradients
    float* wT;                 // nullable: k_tower4's transposed W1 / W2 copies
    int optimizer;             // 0 adam, 1 sgd, 2 accumulate
    float alpha, omb1, omb2, eps, two_l2;
    const float* loss_part;    // loss of the step (optional)
    int n_loss_tiles;
    const float* frozen_sumsq;
    float l2_emb;
    float* loss_out;
    // rider workgroups (round 5): the NEXT tower launch's pre-gathered rows, touched ahead of time from the XCD whose
    // workgroup will read them (workgroup b of a 1-d grid runs on XCD b mod 8; what a kernel leaves in an XCD's L2
    // survives the kernel boundary: 240 cycles instead of 1,400 cold -- see GatherPf).  The tower's prologue waits
    // 5.3 K cycles for its first data (profiles/r05_stamps_tower4_taobao10_bs1024.txt); pf_tiles = 0: no rider
    const float* pf_x;         // next step's rows in the pass buffer [pf_rows][2 EMB]
    const int32_t* pf_dom;
    const float* pf_lab;
    int pf_tiles;              // four-row tiles of the next step
    float* pf_sink;
#ifdef MAMDR_STAMPS
    unsigned long long* stamps; // diagnostic build only: [workgroups][8] s_memtime stamps of wave 0
#endif
};
void launch_wgrad_adam(const FusedArgs& a, hipStream_t s);
// positions [pos0, pos0 + n) of a pass resolved once per mamdr_train_steps call: src = perm[pos] (or pos), clamped;
// xpre[i] = [user row | item row] of src, pdom[i] / plabel[i] its domain (clamped) and label
struct PassPrepArgs {
    const float* user_tab;
    const float* item_tab;
    const int32_t* uid;
    const int32_t* pid;
    const int32_t* dom;
    const float* label;
    const int32_t* perm;       // nullable
    int64_t pos0, n, n_rows_split;
    int n_user, n_item, n_domain;
    int pad_dom;               // domain written for the padding rows (the pass's domain)
    float* xpre;
    int32_t* pdom;
    float* plabel;
    // k_transpose_w's work riding in the same launch (both open a mamdr_train_steps call): workgroups
    // [n_prep_wgs, n_prep_wgs + TRANSPOSE_WGS) when tw_wT is set
    const float* tw_dense;
    DenseLayout tw_L;
    float* tw_wT;
    int n_prep_wgs;
};
// several passes in ONE launch (mamdr_pregather_passes): pass k owns workgroups [wg_end[k - 1], wg_end[k]) and the
// rows [out_off, out_off + n + 16) of the pass buffer
constexpr int PREP_MAX_PASSES = 16;
struct PassPrepMultiArgs {
    const float* user_tab;
    const float* item_tab;
    int n_user, n_item, n_domain, n_pass;
    float* xpre;
    int32_t* pdom;
    float* plabel;
    int wg_end[PREP_MAX_PASSES];
    struct Pass {
        const int32_t* uid;
        const int32_t* pid;
        const int32_t* dom;
        const float* label;
        const int32_t* perm;   // nullable
        int64_t n, n_rows_split, out_off;
        int pad_dom;
    } p[PREP_MAX_PASSES];
};
void launch_pass_prep_multi(const PassPrepMultiArgs& a, hipStream_t s);
// ---- W1 [256][128] as an LDS image (both towers, when a workgroup has its CU to itself).  LDS-DMA
// (global_load_lds_dwordx4: 16 B per lane, two rows per wave instruction, no registers) writes a lane-linear image, so
// the swizzle sits on the SOURCE address: the 16-B chunk q of row r lives at chunk position q ^ (r & 31) -- row reads
// (layer 1's B operand) and column-chunk reads (its backward contraction) are both bank-conflict free.
// The requests are inline asm, i.e. NOT in the compiler's vmcnt bookkeeping: a counted LDS-DMA makes hipcc wait
// vmcnt(0) at the next use of any load.  Uncounted, they only make the counted waits behind them conservative (vmcnt
// retires in order); w1_image_landed() is the wait, a workgroup barrier must follow before another wave reads.
template <int ROWS>
__device__ __forceinline__ void w1_image_request(const float* __restrict__ W1, float* w1s, const int row0) {
    const int lane = threadIdx.x & 63;
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(w1s + row0 * H2));
#pragma unroll
    for (int kk = 0; kk < ROWS; kk += 2) {
        const int k = row0 + kk + (lane >> 5);
        const float* src = W1 + k * H2 + 4 * ((lane & 31) ^ (k & 31));
        const uint32_t dst = base + kk * H2 * 4;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
}
__device__ __forceinline__ void w1_image_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// float offset of W1[row][4 chunk + c] in the image
__device__ __forceinline__ int w1_image_at(int row, int chunk) { return row * H2 + 4 * (chunk ^ (row & 31)); }

constexpr int TRANSPOSE_WGS = (WT_FLOATS + 255) / 256;
// W1T / W2T / W0T element e from the live weights (see k_transpose_w)
__device__ __forceinline__ void transpose_w_elem(const float* dense, const DenseLayout& L, float* wT, int e) {
    if (e < H1 * H2) {
        const int r = e / H2, c = e - r * H2;               // W1[r][c], r < 256, c < 128
        wT[W1T_OFF + c * H1 + r] = dense[L.w1 + e];
    } else if (e < H1 * H2 + H2 * H3) {
        const int f = e - H1 * H2;
        const int r = f / H3, c = f - r * H3;               // W2[r][c], r < 128, c < 64
        wT[W2T_OFF + c * H2 + r] = dense[L.w2 + f];
    } else if (e < WT_FLOATS) {
        const int f = e - (H1 * H2 + H2 * H3);
        const int r = f / H1, c = f - r * H1;               // W0[r][c], r < 256 (user | item rows), c < 256
        wT[W0T_OFF + c * (2 * EMB) + r] = dense[L.w0 + f];
    }
}
void launch_pass_prep(const PassPrepArgs& a, hipStream_t s);
// materialise a pending domain-table step (end of a mamdr_train_steps call): live p / m / v := dm_step4
void launch_dm_finish(const DmStep& q, float* live_p, float* live_m, float* live_v, hipStream_t s);

void launch_tower_train(const TowerArgs& a, hipStream_t s);
void launch_tower_eval(const TowerArgs& a, hipStream_t s);
// -> 0, or one of T4_E_* (a state the caller's own predicate should have excluded / a refused LDS limit): the C ABI reports
// them as MAMDR_ESTATE / MAMDR_EHIP through mamdr_last_error -- the library never aborts
constexpr int T4_E_W2D_LDS = 1;        // k_tower4<.., W2D> was refused its LDS limit
constexpr int T4_E_W2D_STATE = 2;      // w2_direct asked of a launch that does not take the W1-image instance
int launch_tower4_train(const TowerArgs& a, hipStream_t s);
// the W1-image instance of the pre-gathered tower can run (its LDS limit was granted): grids of up to one tile per CU
bool tower4_w1l_ready();
bool tower4_takes_w1l(int64_t rows, int no_w1l);      // launch_tower4_train's choice of the W1-image instance for a batch
void launch_transpose_w(const float* dense, const DenseLayout& L, float* wT, hipStream_t s);
struct EvalFinishArgs {
    const float* loss_part;
    int64_t n_rows;
    int batch;
    const float* dense;
    int dm_count;
    float l2_emb;
    const float* frozen_sumsq; // [4]
    int ld_off, ld_count;
    float l2_lin;
    float* loss_out;
};
void launch_eval_finish(const EvalFinishArgs& a, hipStream_t s);
struct GatherPf;
void launch_wgrad(const WgradArgs& a, hipStream_t s, const GatherPf* pf = nullptr);
bool launch_wgrad_pairs(const WgradArgs& a, hipStream_t s, const GatherPf* pf);      // false: the 96 KB of LDS were refused
// the NEXT step's gather, touched ahead of time by rider workgroups in k_update's launch.  What one kernel leaves in an
// XCD's L2 survives the kernel boundary (tools/probes/l2_survive_probe.hip: 240 cycles for a line the same workgroup id
// touched in the kernel before, 1,400 cold, 600 when another XCD touched it), and workgroup b of any 1-d grid runs on XCD
// b mod 8: rider b' = tile (mod 8) walks tile's dependent chain perm -> uid / pid / domain / label -> table rows, so the
// tower's three dependent misses per step become three L2 hits.
struct GatherPf {
    const int32_t *perm, *uid, *pid, *dom;
    const float* label;
    const float *user_tab, *item_tab;
    int64_t row_base, n_rows_split;
    int rows, n_user, n_item, n_tiles;     // n_tiles = 0: no rider
    float* sink;
};
void launch_update(const UpdateArgs& a, hipStream_t s, const GatherPf* pf = nullptr);
void launch_gather(const TowerArgs& a, float* out, hipStream_t s);
void launch_sumsq(const float* x, int64_t n, float* partials /*>=1024 floats*/, float* out, hipStream_t s);

size_t tower_lds_bytes();

// emb_kernels.hip: trainable embedding tables (TF1 dense Adam over every row)
struct EmbTable {
    int64_t n_rows;
    const int32_t* brow;       // [rows] table row of each batch position (-1 = padding)
    int32_t* map;              // [n_rows] first batch position touching the row, EMB_UNTOUCHED otherwise (set by k_tower)
    float* gbuf;               // [rows][EMB] summed row gradients, indexed by representative position
    int32_t* hasdup;           // [rows] 1 if the representative's row occurs again in the batch (self-resetting)
    int32_t* last;             // [n_rows] Adam step up to which the row's (p, m, v) are current (lazy mode), or null
    int dx_off;                // 0 = user slice of dxe, EMB = item slice
    // DeepFM 1-d linear table of the same feature (null otherwise): gradient = scatter-add of dlogit
    float* lin_p;
    float* lin_m;
    float* lin_v;
    float* glin;               // [rows] summed dlogit, indexed by representative position
};
struct EmbStepArgs {
    float* p;                  // [user table | item table], contiguous at the head of the flat vector
    float* m;
    float* v;
    EmbTable t[2];             // user, item
    const float* dxe;          // [rows][dx_ld]
    int dx_ld;                 // 256, or 384 for the Star tower
    const float* dlogit;       // [rows]
    int rows;                  // batch rows
    float two_l2_lin;
    OptArgsLite opt;
    // lazy dense-Adam bookkeeping (emb_kernels.hip): alpha of every Adam step in a ring, current step
    const float* alpha_log;
    int log_mask;
    int t_now;
    int flags_done;            // duplicate flags were set by k_emb_catchup (lazy mode): skip k_emb_flag
    int apply_now;             // lazy mode: k_emb_reduce applies Adam step t_now to the rows it reduces
    // Star tower: dxe holds d loss / d NORMALISED input; PartitionedNorm's backward through the batch statistics,
    // dx = coef (( dxn - s1 / B) - xhat s2 / B), xhat = (x - mean) inv, is applied to every gathered gradient row on the
    // fly (x = the table row itself) -- k_star_pnb_apply's arithmetic without its pass over the batch.  null = dxe is d x
    const float* pn_sums;      // [2][384] column sums s1, s2 over the batch (k_star_pnb_final); non-null = the switch
    const float* pn_means;     // [2][384] s1 / B, s2 / B (k_star_pnb_final forms the quotients once: the same IEEE division
                               // every consumer did per element)
    const float* pn;           // PartitionedNorm workspace [scale | shift | mean | inv | coef | ...] x 384
};
struct EmbRowsArgs {           // k_emb_rows: row ids + representatives of the batch BEFORE the tower runs
    const int32_t* uid;
    const int32_t* pid;
    const int32_t* perm;
    int64_t row_base, n_rows_split;
    int rows, rows_pad, n_user, n_item;
    int32_t* urow;
    int32_t* irow;
    int32_t* map_u;
    int32_t* map_i;
    float* alpha_log;
    int log_idx;
    float alpha;
};
void launch_emb_rows(const EmbRowsArgs& a, hipStream_t s);
// k_wgrad + k_emb_reduce (+ the NEXT step's k_emb_rows, into the alternate row / map buffers) in one launch;
// k_update (+ DeepFM's k_lin_sweep) (+ the NEXT step's k_emb_catchup) in one launch
struct StarPnBwdArgs;
void launch_wgrad_reduce(const WgradArgs& a, const EmbStepArgs& e, const EmbRowsArgs* next_rows, const StarPnBwdArgs* star_dm,
                         hipStream_t s);     // star_dm: also the Star tower's domain-row column sums (k_star_dm_final)
void launch_update_lin(const UpdateArgs& a, const EmbStepArgs& e, bool lin, const EmbStepArgs* next_catchup, hipStream_t s);
void launch_emb_catchup(const EmbStepArgs& a, hipStream_t s);     // rows of the batch -> current at t_now - 1
void launch_emb_flush(const EmbStepArgs& a, hipStream_t s);       // every row -> current at t_now
void launch_emb_reduce(const EmbStepArgs& a, hipStream_t s);
void launch_emb_sweep(const EmbStepArgs& a, hipStream_t s);
void launch_lin_sweep(const EmbStepArgs& a, hipStream_t s);
void launch_emb_map_init(int32_t* map, int64_t n, hipStream_t s);

// star_kernels.hip: Star tower = PartitionedNorm + StarFCN (model_zoo/Star).  The step reuses
// k_tower / k_wgrad on an "effective" dense block (kernel_shared * kernel_specific[d], ...) built per step.
struct StarPrepArgs {
    const float* blk;          // Star block of the live weights
    StarLayout SL;
    DenseLayout L;             // layout of the effective block
    int n_domain, d;
    float* eff;                // effective dense block [L.alloc]
    float* pn;                 // [PN_WS_FLOATS]
    const float* part;         // training: [chunks][2][384] DOUBLES: sum x / sum x^2 of the raw input columns per chunk
    int n_chunks, rows;
    float* aux;                // moving statistics (read at eval, updated in training)
    StarAuxLayout AL;
    int train;
    int skip_eff;              // the effective block is current already (written by the previous step's k_star_update)
};
struct StarPnBwdArgs {
    const float* user_tab;
    const float* item_tab;
    const float* dm_row;       // domain table row d
    const int32_t* urow;       // [rows] (-1 = padding)
    const int32_t* irow;
    int rows, n_chunks;
    float* dxe;                // in: d loss / d normalised input [rows][384]; out: d loss / d raw input
    const float* pn;
    float* part;               // [chunks][2][384]
    float* sums;               // [2][384] s1 = sum dxn, s2 = sum dxn * xhat
    float* means;              // [2][384] s1 / B, s2 / B
    float* dmpart;             // [chunks][EMB] column sums of dx[:, 256:384]
    float* dmsum;              // [EMB] their total
    int fused;                 // 1: k_star_pnb_final is the only launch (it also finishes dmsum; the table rows get their
                               // d x inside k_emb_reduce); 0: k_star_pnb_apply rewrites dxe, k_star_dm_final sums dmpart;
                               // 2 (the default inside a call, round 6): k_star_pnb_final is the only launch, the table
                               // rows get their d x inside k_emb_reduce, the domain columns' partials are formed by
                               // star_pnb_dom_body in k_wgrad_reduce and summed + stepped by star_dm_step_body in
                               // k_star_update_catchup -- k_star_pnb_apply's arithmetic and order, without its launch
};
struct StarUpdateArgs {
    float* p;                  // Star block of weights / Adam m / Adam v (or accumulator)
    float* m;
    float* v;
    StarLayout SL;
    DenseLayout L;
    int n_domain, d;
    const float* slabs;        // k_wgrad output on the effective layout
    int n_groups, slab_ld;
    const float* sums;         // PartitionedNorm [2][384]
    const float* dmsum;        // [EMB] gradient of the domain-table row d
    const float* xdom;         // [EMB] the batch's normalised domain row (k_star_prep)
    OptArgsLite opt;
    // lazy replay of the other domains' slices (Adam only): the launch covers slice d alone and logs the step's alpha
    int only_live;
    float* alpha_log;          // [log_mask + 1] alphas of this call's steps, slot = step index inside the call
    int log_idx;
    int dm_elsewhere;          // 1: the live domain row is stepped by star_dm_step_body in the same launch
    float* eff_out;            // nullable: the NEXT step's effective dense block (same domain) -- K_l = Ws_l * Wd_l[d],
                               // b_l = bs_l + bd_l[d], the output unit, row d of the domain table -- from the values
                               // just stepped (k_star_prep then skips that part)
};
// k_star_catchup: every slice but d_live takes the n_steps zero-gradient Adam steps it skipped (same arithmetic, same
// order: bit-identical to the per-step sweep)
struct StarCatchArgs {
    float* p;
    float* m;
    float* v;
    StarLayout SL;
    int n_domain, d_live;
    const float* alpha_log;
    int first_idx, n_steps, log_mask;
    float omb1, omb2, eps;
};
void launch_star_catchup(const StarCatchArgs& a, hipStream_t s);
void launch_star_stats(const TowerArgs& a, float* part, float* step_counter, hipStream_t s);
void launch_star_prep(const StarPrepArgs& a, hipStream_t s);
void launch_star_pn_bwd(const StarPnBwdArgs& a, bool dm_final, hipStream_t s, bool partial_done = false);
void launch_star_update(const StarUpdateArgs& a, hipStream_t s);
// k_star_update + the NEXT step's k_emb_catchup in one launch (lazy table Adam)
void launch_star_update_catchup(const StarUpdateArgs& a, const EmbStepArgs& next_catchup, const StarPnBwdArgs* dm, hipStream_t s);

// outer_kernels.hip (compiled with -ffp-contract=off)
void launch_interp(float* dst, const float* a, const float* b, float scale, int64_t n, hipStream_t s);
void launch_moving_average(float* unbiased, float* biased, const float* value, float decay, float denom, int64_t n,
                           hipStream_t s);
void launch_merge(float* dst, const float* t, const float* p, int mode, int64_t n, hipStream_t s);
void launch_dr_advance(float* phi, float* w, float* merged, const float* theta, float gamma, int mode, int assign, int64_t n,
                       hipStream_t s);
void launch_dr_advance_dm(float* phi, float* w, float* merged, const float* theta, float gamma, int mode, int assign, int64_t n,
                          const DmStep& q, float* live_m, float* live_v, int64_t dm_off4, int dm_n4, hipStream_t s);
void launch_sub(float* dst, const float* a, const float* b, int64_t n, hipStream_t s);
void launch_accumulate(float* acc, const float* a, const float* b, const float* shared, float divisor, int64_t n,
                       hipStream_t s);
void launch_apply_accumulated(float* dst, float* acc, float divisor, float scale, int64_t n, hipStream_t s);
void launch_adam_apply(float* p, float* m, float* v, const float* g, float gscale, float alpha, float omb1, float omb2,
                       float eps, int64_t n, hipStream_t s);
// PCGrad projection (model_zoo/pcgrad.py:152-160) over tensors described as (offset, rows, cols) slices
constexpr int PCG_MAX_SEG = 24;
struct PcgArgs {
    float* fin;                // running gradient (= current gradient), updated in place
    float* aux;                // auxiliary gradient, projected in place
    int n_seg;
    int64_t off[PCG_MAX_SEG];
    int64_t row_start[PCG_MAX_SEG + 1];   // prefix sums of the row counts
    int cols[PCG_MAX_SEG];
};
void launch_pcgrad(const PcgArgs& a, hipStream_t s);

}  // namespace mamdr
