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
#define MAMDR_LAUNCH(kernel, grid, block, lds, stream, ...)                                               \
    do {                                                                                                  \
        if (::mamdr::g_prof_stop) {                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ::mamdr::g_prof_stop, 0, __VA_ARGS__); \
            ::mamdr::g_prof_stop = nullptr;                                                               \
        } else {                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                            \
        }                                                                                                 \
    } while (0)

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
    const float* pn_aff;       // Star: [scale 384 | shift 384] of PartitionedNorm, null otherwise
    int32_t* urow;             // [rows_pad] user row of each batch position (-1 = padding)
    int32_t* irow;             // [rows_pad]
    int32_t* map_u;            // [n_user] / [n_item]: atomicMin of the batch position touching the row (null: frozen tables)
    int32_t* map_i;
    float* loss_part;          // [tiles] sum of per-row BCE of the tile
    // DeepFM: logit += FM second-order term + linear tables
    int deepfm;
    const float* lin_user;     // [n_user] / [n_item] 1-d tables; null = frozen at their zero initialisation
    const float* lin_item;
    float* fmq;                // train: [rows_pad][EMB] dlogit * (user + item embedding), for the domain-table gradient
    // uncertainty weighting (model_zoo/uncertainty_weight/weighted_loss.py:30-43): the BCE term of the loss is
    // divided by var^2, var = dense[uw_off] (the batch's domain); -1 = off
    int uw_off;
    const float* wT;           // k_tower4 only: transposed W1 / W2 copies
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
    int a_kind, a_off;         // 0: acts column block, 1: ones (row 0), 2: one-hot(domain) rows a_off..
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
    const float* fmq;          // DeepFM only
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
};

void launch_tower_train(const TowerArgs& a, hipStream_t s);
void launch_tower_eval(const TowerArgs& a, hipStream_t s);
void launch_tower4_train(const TowerArgs& a, hipStream_t s);
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
void launch_wgrad(const WgradArgs& a, hipStream_t s);
void launch_update(const UpdateArgs& a, hipStream_t s);
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
    const float* part;         // training: [chunks][2][384] chunk mean / M2 of the raw input columns
    int n_chunks, rows;
    float* aux;                // moving statistics (read at eval, updated in training)
    StarAuxLayout AL;
    int train;
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
    float* dmpart;             // [chunks][EMB] column sums of dx[:, 256:384]
    float* dmsum;              // [EMB] their total
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
};
void launch_star_stats(const TowerArgs& a, float* part, float* step_counter, hipStream_t s);
void launch_star_prep(const StarPrepArgs& a, hipStream_t s);
void launch_star_pn_bwd(const StarPnBwdArgs& a, bool dm_final, hipStream_t s);
void launch_star_update(const StarUpdateArgs& a, hipStream_t s);
// k_star_update + the NEXT step's k_emb_catchup in one launch (lazy table Adam)
void launch_star_update_catchup(const StarUpdateArgs& a, const EmbStepArgs& next_catchup, hipStream_t s);

// outer_kernels.hip (compiled with -ffp-contract=off)
void launch_interp(float* dst, const float* a, const float* b, float scale, int64_t n, hipStream_t s);
void launch_merge(float* dst, const float* t, const float* p, int mode, int64_t n, hipStream_t s);
void launch_dr_advance(float* phi, float* w, float* merged, const float* theta, float gamma, int mode, int assign, int64_t n,
                       hipStream_t s);
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
