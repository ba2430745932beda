// C ABI of libmamdr_hip.so (declared in include/mamdr_hip.h).
//
// Host-side state is tiny: bound pointers, the Adam step count with its fp32
// running beta powers (TF keeps them as beta1_power / beta2_power variables), the
// global inner-step counter that indexes the dropout stream, and a private
// workspace sized for max_batch.  Everything numeric runs in the kernels of
// step_kernels.hip / outer_kernels.hip on the context's stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mamdr_hip.h"
#include "mamdr_kernels.h"
#include "env_registry.h"

using namespace mamdr;

namespace mamdr {
thread_local hipEvent_t g_prof_stop = nullptr;
}

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(MAMDR_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct SplitData {
    const int32_t* uid = nullptr;
    const int32_t* pid = nullptr;
    const int32_t* dom = nullptr;
    const float* label = nullptr;
    int64_t n = 0;
    bool bound = false;     // an EMPTY split (n = 0, null columns) is bound too: a pass over it has no steps
};

struct EventPair {
    hipEvent_t a, b;
    bool own_a = true;         // false: `a` is the previous kernel's stop event (owned by that pair)
};

}  // namespace

struct mamdr_ctx {
    mamdr_config cfg;
    hipStream_t stream = nullptr;
    DenseLayout L;
    int64_t table_floats = 0;   // trainable user+item floats in front of the dense block
    bool deepfm = false;
    bool nfm = false;           // linear tables + DNN over the bi-interaction (MAMDR_TOWER_NFM): FM instances, mode 4
    bool pnn = false;           // the mlp tower + three inner-product inputs (MAMDR_TOWER_PNN): FM instances of the towers, mode 3
    float* ipbuf = nullptr;     // PNN: [rows_pad][4] the batch's inner products (A operand of dW0x's tiles)
    bool star = false;
    StarLayout SL;
    StarAuxLayout AL;
    int64_t n_meta = 0;
    float* aux = nullptr;           // bound PartitionedNorm state (Star)
    float* eff = nullptr;           // Star: effective dense block of the step's domain
    float* pn = nullptr;            // Star: [PN_WS_FLOATS]
    float* star_part = nullptr;     // Star: [chunks][2][384] partials (forward statistics as doubles, then backward sums as floats)
    float* star_sums = nullptr;     // Star: [2][384] PN sums + [128] domain-row gradient
    float* star_dmpart = nullptr;   // Star: [chunks][EMB]
    int64_t lin_user_off = 0;   // DeepFM + trainable tables: 1-d linear tables behind the embedding tables
    int64_t lin_item_off = 0;
    int64_t n_params = 0;       // floats of the flat vector (incl. padding)
    // bound state
    float* params = nullptr;
    float* adam_m = nullptr;
    float* adam_v = nullptr;
    float* accum = nullptr;         // meta-gradient accumulator (MAMDR_OPT_ACCUMULATE)
    const float* user_tab = nullptr;
    const float* item_tab = nullptr;
    std::vector<SplitData> data;   // [domain*3 + split]
    // optimiser / stream counters (host side)
    int64_t adam_t = 0;
    float b1p = 1.0f, b2p = 1.0f;
    uint32_t global_step = 0;
    // workspace
    int rows_pad_max = 0;
    float* acts = nullptr;
    float* dz = nullptr;
    float* dlogit = nullptr;
    float* w0dom_copy = nullptr;
    float* dm_copy = nullptr;       // pre-update snapshot of the domain table (dW0[256:384] by linearity)
    bool lin_w0dom = false;         // k_wgrad carries no tiles for W0[256:384]: k_update rebuilds that gradient from S
    float* wT = nullptr;            // transposed W1 / W2 (k_tower4)
    // mlp tower with frozen tables: weight gradients + optimiser step in one launch (k_wgrad_adam) + k_dm_finish
    // instead of k_wgrad -> slabs -> k_update (MAMDR_FUSED=0 keeps the slab path)
    bool fused = false;
    float* star_alpha = nullptr;    // Star tower: alphas of the current call's steps (lazy replay of the other domains' slices)
    int star_dense_slices = 0;      // MAMDR_STAR_DENSE_SLICES=1: every slice swept every step (diagnostic; same bits)
    bool star_pn_in_tower = true;   // PartitionedNorm backward: per-tile sums in the tower's tail (MAMDR_STAR_PNB_KERNEL=1:
                                    // k_star_pnb_partial as a launch of its own; same bits)
    // MAMDR_STAR_PNB_FUSED=1 (measured, not adopted -- DESIGN.md section 8): no k_star_pnb_apply either; the table rows
    // get PartitionedNorm's backward inside k_emb_reduce and the domain row's gradient comes in closed form from
    // k_star_pnb_final.  Saves the 6.3 us launch, costs 2.3 us in k_wgrad_reduce -- and the closed form is EXACTLY zero
    // where the per-row sum leaves rounding residue, which Adam turns into a random walk of the domain row (in the
    // reference too): the moving statistics then lag differently and validation AUC moves by ~1e-3.
    bool star_pn_fused = false;
    // k_star_pnb_apply's work spread over the launches around it inside a call (StarPnBwdArgs::fused == 2; same bits).
    // MAMDR_STAR_PNB_APPLY=1 keeps the launch (diagnostic).
    bool star_pn_no_apply = true;
    int t4_no_w1l = 0;              // MAMDR_T4_NO_W1L=1: k_tower4 without the W1 image in LDS (diagnostic)
    int fused_max_batch = 1024;     // batches up to this size take the fused path (MAMDR_FUSED=2: every batch size):
                                    // 4 rows x the CU count, set at mamdr_create
    int tower4_max_rows = 2048;     // steps of up to this many (padded) rows run k_tower4, see mamdr_create
    float* pdm = nullptr;           // [32][n_domain][EMB] partial domain-table gradients
    // the domain table's step stays pending until the next tower kernel applies it (DmStep, mamdr_kernels.h):
    // two snapshots [3][n_domain][EMB] of (p, m, v) alternate between steps
    // the rows of a call pre-gathered once (k_pass_prep): [cap][2 EMB] + domain / label per position, grown on demand
    float* xpre = nullptr;
    int32_t* pdom = nullptr;
    float* plabel = nullptr;
    int64_t pre_cap = 0;
    // passes gathered ahead of their calls (mamdr_pregather_passes): entry k's rows sit at [off, off + n + 16) of xpre
    struct PgEntry { int domain; const int32_t* perm; int64_t n, off; int batch; };
    std::vector<PgEntry> pg;
    size_t pg_pos = 0;
    int64_t pg_hits = 0;            // calls served from an entry (mamdr_pregather_hits)
    int64_t pg_launches = 0;        // hints that launched k_pass_prep_multi (mamdr_pregather_launches)
    bool gather_pf = true;          // MAMDR_NO_GATHER_PF=1: no riders in k_update's launch touching the next step's gather
    bool wgrad_pairs = false;          // MAMDR_WGRAD_PAIRS=1: k_wgrad8 (eight waves, one slab per pair of row groups) where a step has > 8 groups
    bool gather_pf_in_wgrad = true;    // the riders of the next step's gather sit in k_wgrad's launch (round 5: k_update, bound by what it pulls over
                                       // the fabric, loses 0.42 us without them, k_wgrad gains 0.13; profiles/r05_ab_riders_place.txt);
                                       // MAMDR_GATHER_PF_IN=update: in k_update's launch as in rounds 3 - 4
    bool fused_pf = false;          // MAMDR_FUSED_PF=1: riders in k_wgrad_adam's launch touch the next tower launch's pre-gathered rows (round 5:
                                    // measured and left off -- the tower gains 0.08 us, k_wgrad_adam's second round of blocks costs 0.8;
                                    // profiles/r05_ab_fused_pf.txt)
    bool use_pre = true;            // MAMDR_NO_PREGATHER=1: the towers gather through perm / uid / pid every step
    float* dmsnap[2] = {nullptr, nullptr};
    int dm_cur = 0;
    // ... ACROSS calls too (round 4): an Adam call leaves its last step pending; the first tower of the next fused Adam
    // call applies it, anything else that reads or replaces the live state materialises it first (finish_dm, from
    // sync_tables -- the contract of mamdr_sync_tables).  MAMDR_DM_EACH=1 / MAMDR_DM_CALL=1: after every step / call.
    DmStep dm_pending{};
    bool dm_finish_call = false;
    // the transposed copies in wT hold the live W1 / W2 (/ W0[0:256]): true after a call whose steps kept them current,
    // false once the live state may have been replaced from outside (sync_tables) or stepped without them
    bool wT_valid = false;
    bool w2_direct_ok = true;       // MAMDR_NO_W2_DIRECT=1: always build the copies at the start of a call (k_transpose_w)
    bool dm_finish_each = false;    // MAMDR_DM_EACH=1: materialise after every step (k_dm_finish per step; A/B measurements)
    int tower_tile = 0;             // 0 auto, 4 / 16 forced (env MAMDR_TOWER_TILE)
    // trainable user / item tables
    float* dxe = nullptr;
    int32_t* urow = nullptr;
    int32_t* irow = nullptr;
    int32_t* map_u = nullptr;
    int32_t* map_i = nullptr;
    float* gbuf_u = nullptr;
    float* gbuf_i = nullptr;
    int32_t* hasdup_u = nullptr;
    int32_t* hasdup_i = nullptr;
    // lazy dense Adam over the trainable tables (emb_kernels.hip); MAMDR_DENSE_ADAM=1 keeps the per-step sweep
    bool lazy = false;
    bool tables_dirty = false;      // some rows lag behind adam_t
    int32_t* last_u = nullptr;      // [n_user] / [n_item] Adam step each row is current at
    int32_t* last_i = nullptr;
    float* alpha_log = nullptr;     // ring of the per-step alpha
    int log_cap = 1 << 16;
    // Adam steps between forced flushes.  Every missed step is replayed exactly once either way; the flush
    // replays at full occupancy, the per-row catch-up before a gather is a serial chain per row, so short gaps
    // win until the flush's own table traffic shows (Amazon-6, 10 % rows: 7.7 K domain-steps/s without a period,
    // 11.6 K at 16, 12.1 K at 32, 12.0 K at 64, 10.9 K at 256).  MAMDR_LAZY_FLUSH_EVERY overrides.
    int flush_every = 32;
    int64_t flush_t = 0;            // adam_t of the last flush
    int64_t n_flush = 0;            // k_emb_flush launches so far / those forced by the flush period (mamdr_table_flushes)
    int64_t n_flush_forced = 0;
    float* fmq = nullptr;           // DeepFM: [rows_pad][EMB]
    float* glin_u = nullptr;        // DeepFM + trainable tables: [rows_pad]
    float* glin_i = nullptr;
    int32_t* domrow = nullptr;
    float* loss_part = nullptr;     // train: per tile of a batch
    float* eval_part = nullptr;     // eval: per tile of a split (grown on bind)
    int64_t eval_part_cap = 0;
    float* slabs = nullptr;
    int max_groups = 16;
    int rpg_override = 0;       // MAMDR_RPG: rows per K-split group of k_wgrad (diagnostic)
    bool tail_fuse = true;      // MAMDR_NO_TAILFUSE=1: k_emb_reduce / k_lin_sweep as launches of their own
    // the other half of the row / map double buffer: the NEXT step's k_emb_rows rides in this step's last launch
    int32_t* urow_alt = nullptr;
    int32_t* irow_alt = nullptr;
    int32_t* map_u_alt = nullptr;
    int32_t* map_i_alt = nullptr;
    bool rows_ready = false;    // the current buffers already hold the rows of the step about to run
    bool catchup_ready = false; // ... and those rows were already brought up to the previous step
    int slab_ld = 0;            // dense block + S region ([n_domain][256]) (+ DeepFM S2 region [n_domain][128])
    int s2_off = 0;
    TileDesc* tiles = nullptr;
    int n_tiles = 0;
    float* thresholds = nullptr;
    float* frozen_sumsq = nullptr;  // [4] user, item table; DeepFM linear user, item table
    float* sumsq_partials = nullptr;
#ifdef MAMDR_STAMPS
    unsigned long long* stamps = nullptr;
#endif
    // profiling
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;    // recycled profiling events
    void ev_pool_push(hipEvent_t e) { ev_pool.push_back(e); }
    // a kernel's time = its stop event minus the stop event of the kernel launched right before it on the stream
    // (start markers of their own, attached or recorded, run ahead of the previous kernel's completion when the
    // host is ahead, or add a packet between the kernels); chain_ok: prev_b is that immediately preceding event
    hipEvent_t prev_b = nullptr;
    bool chain_ok = false;
    std::vector<EventPair> ev[MAMDR_KERNEL_COUNT];
};

namespace {

std::vector<TileDesc> build_tiles(const DenseLayout& L, int n_domain, bool deepfm, int s2_off, bool lin_w0dom, bool star,
                                  bool pnn = false, bool nfm = false) {
    std::vector<TileDesc> t;
    struct G { int a_off, M, b_off, N, dst; };
    // dW0 = x^T dz1, dW1 = h1^T dz2, dW2 = h2^T dz3
    const G gemms[3] = {{0, XDIM, 0, H1, L.w0}, {XDIM, H1, H1, H2, L.w1}, {XDIM + H1, H2, H1 + H2, H3, L.w2}};
    // (64x64 tiles first: the kernel stages their operands through LDS)
    // (lin_w0dom: rows 256..383 of x are per-domain constants, their part of dW0 follows from S in k_update)
    // (NFM: rows 0..255 of W0 meet the raw user / item rows of the tile but are no parameters -- they stay zero: no tiles)
    for (const G& g : gemms)
        for (int m0 = (nfm && g.dst == L.w0) ? 2 * EMB : 0; m0 < ((lin_w0dom && g.dst == L.w0) ? 2 * EMB : g.M); m0 += 64)
            for (int n0 = 0; n0 < g.N; n0 += 64)
                t.push_back(TileDesc{0, g.a_off + m0, 0, g.b_off + n0, g.dst + m0 * g.N + n0, g.N, 64, 64, 1});
    // biases = column sums of dz (A = ones in row 0)
    const int boff[3] = {L.b0, L.b1, L.b2}, bn[3] = {H1, H2, H3}, zoff[3] = {0, H1, H1 + H2};
    for (int l = 0; l < 3; ++l)
        for (int n0 = 0; n0 < bn[l]; n0 += 32) t.push_back(TileDesc{1, 0, 0, zoff[l] + n0, boff[l] + n0, 0, 1, 32});
    // output unit: dwo = h3^T dlogit, dgb = sum dlogit
    for (int m0 = 0; m0 < H3; m0 += 32) t.push_back(TileDesc{0, XDIM + H1 + H2 + m0, 1, 0, L.wo + m0, 1, 32, 1});
    t.push_back(TileDesc{1, 0, 1, 0, L.gb, 0, 1, 1});
    // domain table, by linearity: S = onehot(domain)^T dz1 ([n_domain][256], behind the dense block in
    // the slab); k_update turns it into dDm = S . W0[256:384,:]^T
    // (not for the Star tower: its domain-row gradient comes through PartitionedNorm's backward)
    for (int m0 = 0; m0 < (star ? 0 : n_domain); m0 += 32)
        for (int n0 = 0; n0 < H1; n0 += 32) {
            const int mv = n_domain - m0 < 32 ? n_domain - m0 : 32;
            t.push_back(TileDesc{2, m0, 0, n0, L.alloc + m0 * H1 + n0, H1, mv, 32});
        }
    if (deepfm || pnn)
        for (int m0 = 0; m0 < n_domain; m0 += 32) {
            const int mv = n_domain - m0 < 32 ? n_domain - m0 : 32;
            // per-row part of the domain-table gradient: S2 = onehot(domain)^T fmq (DeepFM: dlogit * (u + i); PNN: the
            // inner products' chain rule, dip_ud * u + dip_id * i)
            for (int n0 = 0; n0 < EMB; n0 += 32) t.push_back(TileDesc{2, m0, 2, n0, s2_off + m0 * EMB + n0, EMB, mv, 32});
            // linear domain table: onehot(domain)^T dlogit
            if (deepfm) t.push_back(TileDesc{2, m0, 1, 0, L.ld + m0, 1, mv, 1});
        }
    // PNN: the three extra rows of the first kernel, dW0x = ip^T dz1 (A = the batch's inner products, ipbuf [B][4])
    if (pnn)
        for (int n0 = 0; n0 < H1; n0 += 32) t.push_back(TileDesc{3, 0, 0, n0, L.wx + n0, H1, 3, 32});
    return t;
}

int check_ctx(const mamdr_ctx* c) {
    if (!c) return fail(MAMDR_EINVAL, "null context");
    return MAMDR_OK;
}

SplitData* split_of(mamdr_ctx* c, int domain, int split) {
    if (domain < 0 || domain >= c->cfg.n_domain || split < 0 || split > 2) return nullptr;
    return &c->data[(size_t)domain * 3 + split];
}

// per-kernel device time from stop events chained along the stream (see mamdr_ctx::prev_b)
struct Prof {
    mamdr_ctx* c;
    int k;
    EventPair e{nullptr, nullptr, true};
    Prof(mamdr_ctx* c_, int k_, bool = false) : c(c_), k(k_) {
        if (!c->profile || c->ev[k].size() >= 200000) return;
        auto take = [&]() {
            hipEvent_t ev = nullptr;
            if (!c->ev_pool.empty()) {          // (pool refilled by mamdr_profile_reset: no event creation per launch)
                ev = c->ev_pool.back();
                c->ev_pool.pop_back();
            } else {
                (void)hipEventCreate(&ev);
            }
            return ev;
        };
        e.b = take();
        if (c->chain_ok && c->prev_b) {
            e.a = c->prev_b;
            e.own_a = false;
        } else {                                // nothing timed right before: an explicit start marker
            e.a = take();
            (void)hipEventRecord(e.a, c->stream);
        }
        g_prof_stop = e.b;                      // the launch issued inside this scope carries it (MAMDR_LAUNCH)
    }
    ~Prof() {
        if (!e.b) return;
        if (g_prof_stop) {                      // no launch took it (should not happen): record it the plain way
            g_prof_stop = nullptr;
            (void)hipEventRecord(e.b, c->stream);
        }
        c->ev[k].push_back(e);
        c->prev_b = e.b;
        c->chain_ok = true;
    }
};
// a launch that is not timed went out: the next timed kernel needs a start marker of its own
static inline void prof_break(mamdr_ctx* c) { c->chain_ok = false; }

void fill_tower_common(const mamdr_ctx* c, const SplitData& d, TowerArgs& a) {
    memset(&a, 0, sizeof(a));
    a.user_tab = c->cfg.emb_trainable ? c->params : c->user_tab;
    a.item_tab = c->cfg.emb_trainable ? c->params + (size_t)c->cfg.n_user * EMB : c->item_tab;
    a.dense = c->params + c->table_floats;
    a.L = c->L;
    a.n_user = c->cfg.n_user;
    a.n_item = c->cfg.n_item;
    a.n_domain = c->cfg.n_domain;
    a.uid = d.uid;
    a.pid = d.pid;
    a.dom = d.dom;
    a.label = d.label;
    a.n_rows_split = d.n;
    a.thresholds = c->thresholds;
    a.deepfm = c->nfm ? 4 : (c->deepfm ? (c->cfg.tower == MAMDR_TOWER_WDL ? 2 : 1) : (c->pnn ? 3 : 0));
    a.ipbuf = c->ipbuf;
    a.uw_off = -1;
    if (c->deepfm && c->cfg.emb_trainable) {
        a.lin_user = c->params + c->lin_user_off;
        a.lin_item = c->params + c->lin_item_off;
    }
}

int ready(const mamdr_ctx* c) {
    if (!c->params) return fail(MAMDR_ESTATE, "mamdr_bind_state has not been called");
    if (c->star && !c->aux) return fail(MAMDR_ESTATE, "Star tower: mamdr_bind_aux has not been called");
    if (!c->cfg.emb_trainable && (!c->user_tab || !c->item_tab))
        return fail(MAMDR_ESTATE, "frozen user/item tables are not bound (mamdr_bind_table)");
    return MAMDR_OK;
}

}  // namespace


// ---- trainable user / item tables: shared by the mlp / deepfm and the Star step
static void fill_emb_args(const mamdr_ctx* c, int32_t optimizer, float alpha, float omb1, float omb2, float two_l2,
                          int rows, int dx_ld, EmbStepArgs& ea) {
    memset(&ea, 0, sizeof(ea));
    float* slot_m = optimizer == MAMDR_OPT_ACCUMULATE ? c->accum : c->adam_m;
    ea.p = c->params;
    ea.m = slot_m;
    ea.v = c->adam_v;
    ea.dxe = c->dxe;
    ea.dx_ld = dx_ld;
    ea.dlogit = c->dlogit;
    ea.rows = rows;
    ea.two_l2_lin = 2.0f * c->cfg.l2_linear;
    ea.opt.optimizer = optimizer;
    ea.opt.alpha = alpha;
    ea.opt.omb1 = omb1;
    ea.opt.omb2 = omb2;
    ea.opt.eps = c->cfg.adam_eps;
    ea.opt.two_l2 = two_l2;
    ea.alpha_log = c->alpha_log;
    ea.log_mask = c->log_cap - 1;
    ea.t_now = (int)c->adam_t;
    if (c->star && c->star_pn_fused) {         // PartitionedNorm's backward rides in k_emb_reduce (EmbStepArgs::pn_sums)
        ea.pn_sums = c->star_sums;
        ea.pn_means = c->star_sums + 2 * XDIM + EMB;
        ea.pn = c->pn;
    }
    EmbTable& tu = ea.t[0];
    EmbTable& ti = ea.t[1];
    tu.n_rows = c->cfg.n_user;
    tu.brow = c->urow;
    tu.map = c->map_u;
    tu.gbuf = c->gbuf_u;
    tu.hasdup = c->hasdup_u;
    tu.last = c->last_u;
    tu.dx_off = 0;
    ti.n_rows = c->cfg.n_item;
    ti.brow = c->irow;
    ti.map = c->map_i;
    ti.gbuf = c->gbuf_i;
    ti.hasdup = c->hasdup_i;
    ti.last = c->last_i;
    ti.dx_off = EMB;
    if (c->deepfm) {
        tu.lin_p = c->params + c->lin_user_off;
        tu.lin_m = slot_m + c->lin_user_off;
        tu.lin_v = c->adam_v + c->lin_user_off;
        tu.glin = c->glin_u;
        ti.lin_p = c->params + c->lin_item_off;
        ti.lin_m = slot_m + c->lin_item_off;
        ti.lin_v = c->adam_v + c->lin_item_off;
        ti.glin = c->glin_i;
    }
}

constexpr int STAR_ALPHA_CAP = 1 << 12;      // steps between two replays of the lagging Star slices (power of two)
static float table_two_l2(const mamdr_ctx* c) { return c->star ? 0.f : 2.0f * c->cfg.l2_emb; }

// materialise a domain-table step the k_wgrad_adam path left pending
static void finish_dm(mamdr_ctx* c) {
    if (!c->dm_pending.snap) return;
    float* const m = (c->dm_pending.optimizer == MAMDR_OPT_ACCUMULATE ? c->accum : c->adam_m) + c->table_floats + c->L.dm;
    {
        Prof p(c, MAMDR_KERNEL_UPDATE);
        launch_dm_finish(c->dm_pending, c->params + c->table_floats + c->L.dm, m, c->adam_v + c->table_floats + c->L.dm,
                         c->stream);
    }
    c->dm_pending.snap = nullptr;
}

// the live state current and about to be read or replaced from outside: the pending domain-table step applied, every
// table row at adam_t (no-op when nothing lags); the transposed weight copies can no longer be trusted
static void sync_tables(mamdr_ctx* c) {
    finish_dm(c);
    c->wT_valid = false;
    if (!c->tables_dirty) return;
    EmbStepArgs ea;
    fill_emb_args(c, MAMDR_OPT_ADAM, 0.f, 1.0f - c->cfg.adam_beta1, 1.0f - c->cfg.adam_beta2, table_two_l2(c), 0,
                  c->star ? XDIM : 2 * EMB, ea);
    {
        Prof p(c, MAMDR_KERNEL_FLUSH);
        launch_emb_flush(ea, c->stream);
    }
    c->tables_dirty = false;
    c->flush_t = c->adam_t;
    c->n_flush += 1;
}

// k_emb_rows arguments of the batch at row_base for Adam step `t` (alt: into the other half of the double buffer)
static void fill_rows_args(const mamdr_ctx* c, const SplitData& d, const int32_t* d_perm, int64_t row_base, int rows,
                           int rows_pad, float alpha, int64_t t, bool alt, EmbRowsArgs& ra) {
    memset(&ra, 0, sizeof(ra));
    ra.uid = d.uid;
    ra.pid = d.pid;
    ra.perm = d_perm;
    ra.row_base = row_base;
    ra.n_rows_split = d.n;
    ra.rows = rows;
    ra.rows_pad = rows_pad;
    ra.n_user = c->cfg.n_user;
    ra.n_item = c->cfg.n_item;
    ra.urow = alt ? c->urow_alt : c->urow;
    ra.irow = alt ? c->irow_alt : c->irow;
    ra.map_u = alt ? c->map_u_alt : c->map_u;
    ra.map_i = alt ? c->map_i_alt : c->map_i;
    ra.alpha_log = c->alpha_log;
    ra.log_idx = (int)(t & (c->log_cap - 1));
    ra.alpha = alpha;
}

// lazy mode, before the tower of Adam step adam_t (already incremented): row ids + representatives of the
// batch, alpha of this step into the ring, rows of the batch brought up to adam_t - 1
static void emb_pre_step(mamdr_ctx* c, const SplitData& d, const int32_t* d_perm, int64_t row_base, int rows,
                         int rows_pad, float alpha, float omb1, float omb2) {
    // the ring must not wrap over a lagging row's range; and a bounded gap keeps the per-row serial replay of
    // k_emb_catchup short (the flush replays the same steps at full occupancy)
    if (c->adam_t - c->flush_t >= c->log_cap - 2 || c->adam_t - c->flush_t > c->flush_every) {
        c->adam_t -= 1;
        if (c->tables_dirty) c->n_flush_forced += 1;
        sync_tables(c);
        c->adam_t += 1;
    }
    if (!c->rows_ready) {
        EmbRowsArgs ra;
        fill_rows_args(c, d, d_perm, row_base, rows, rows_pad, alpha, c->adam_t, false, ra);
        Prof p(c, MAMDR_KERNEL_AUX);
        launch_emb_rows(ra, c->stream);
    }
    c->rows_ready = false;
    if (!c->catchup_ready) {
        EmbStepArgs ea;
        fill_emb_args(c, MAMDR_OPT_ADAM, alpha, omb1, omb2, table_two_l2(c), rows, c->star ? XDIM : 2 * EMB, ea);
        Prof p(c, MAMDR_KERNEL_AUX);
        launch_emb_catchup(ea, c->stream);
    }
    c->catchup_ready = false;
    c->tables_dirty = true;
}

// after the tower / wgrad: scatter-add of the row gradients and the optimiser on the tables
static void emb_post_step(mamdr_ctx* c, int32_t optimizer, float alpha, float omb1, float omb2, int rows) {
    EmbStepArgs ea;
    fill_emb_args(c, optimizer, alpha, omb1, omb2, table_two_l2(c), rows, c->star ? XDIM : 2 * EMB, ea);
    if (c->lazy && optimizer == MAMDR_OPT_ADAM) {
        // duplicates were flagged by the catch-up kernel; the reducing workgroup applies the step itself
        ea.flags_done = 1;
        ea.apply_now = 1;
        {
            Prof p(c, MAMDR_KERNEL_EMB_SWEEP);
            launch_emb_reduce(ea, c->stream);
        }
        if (c->deepfm) {
            Prof p(c, MAMDR_KERNEL_AUX);
            launch_lin_sweep(ea, c->stream);     // reads the row maps, then releases them
        }
        return;
    }
    launch_emb_reduce(ea, c->stream);
    prof_break(c);
    {
        Prof p(c, MAMDR_KERNEL_EMB_SWEEP);
        launch_emb_sweep(ea, c->stream);
    }
    if (c->deepfm) {
        Prof p(c, MAMDR_KERNEL_AUX);
        launch_lin_sweep(ea, c->stream);
    }
}

// ---- Star tower: one training step on `rows` rows of domain `domain` (star.py:70-97; kernels in star_kernels.hip)
// next_rows (nullable): the NEXT step's k_emb_rows arguments (alternate buffers), riding in this step's last launch
// lazy_idx >= 0: only slice `domain` of the per-domain tensors is stepped (the others are replayed by the caller,
// k_star_catchup) and the step's alpha is logged at that slot
static int star_train_step(mamdr_ctx* c, const SplitData& d, int domain, const int32_t* d_perm, int64_t row_base, int rows,
                           int32_t optimizer, float alpha, float omb1, float omb2, float* loss_out,
                           const EmbRowsArgs* next_rows, const EmbStepArgs* next_catchup, int lazy_idx, bool eff_current,
                           bool eff_for_next) {
    const int rows_pad = (rows + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
    const int chunks = (rows + STAR_CHUNK - 1) / STAR_CHUNK;
    float* blk = c->params + c->table_floats;
    TowerArgs ta;
    fill_tower_common(c, d, ta);
    ta.perm = d_perm;
    ta.row_base = row_base;
    ta.rows = rows;
    ta.batch = rows;
    if (c->cfg.emb_trainable && c->lazy && optimizer == MAMDR_OPT_ADAM)
        emb_pre_step(c, d, d_perm, row_base, rows, rows_pad, alpha, omb1, omb2);
    StarPrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.blk = blk;
    pa.SL = c->SL;
    pa.L = c->L;
    pa.n_domain = c->cfg.n_domain;
    pa.d = domain;
    pa.eff = c->eff;
    pa.pn = c->pn;
    pa.part = c->star_part;
    pa.n_chunks = chunks;
    pa.rows = rows;
    pa.aux = c->aux;
    pa.AL = c->AL;
    pa.train = 1;
    pa.skip_eff = eff_current ? 1 : 0;      // (the previous step of this call wrote it: k_star_update, eff_out)
    {
        Prof p(c, MAMDR_KERNEL_AUX);            // k_star_stats + k_star_prep as one timed group
        // forward statistics read the raw rows (domain table straight from the flat vector: SL.dm == L.dm == 0)
        launch_star_stats(ta, c->star_part, c->aux + c->AL.steps + domain, c->stream);
        launch_star_prep(pa, c->stream);
    }

    ta.dense = c->eff;
    ta.pn_aff = c->pn;
    ta.pn_part = c->star_pn_in_tower ? c->star_part : nullptr;
    ta.use_dropout = 0;
    ta.keep_scale = 1.0f;
    ta.acts = c->acts;
    ta.dz = c->dz;
    ta.dlogit = c->dlogit;
    ta.domrow = c->domrow;
    ta.dxe = c->dxe;
    ta.dx_ld = XDIM;
    ta.urow = c->urow;
    ta.irow = c->irow;
    ta.map_u = c->map_u;          // null with frozen tables
    ta.map_i = c->map_i;
    ta.loss_part = c->loss_part;
#ifdef MAMDR_STAMPS
    ta.stamps = c->stamps;
#endif
    {
        Prof p(c, MAMDR_KERNEL_FWD_BWD);
        launch_tower_train(ta, c->stream);
    }
    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.acts = c->acts;
    wa.dz = c->dz;
    wa.dlogit = c->dlogit;
    wa.domrow = c->domrow;
    wa.tiles = c->tiles;
    wa.n_tiles = c->n_tiles;
    wa.rows_pad = rows_pad;
    int rpg = rows_pad <= 512 ? 256 : (rows_pad <= 1024 ? 128 : (rows_pad <= 4096 ? 256 : 512));
    if (c->rpg_override > 0) rpg = c->rpg_override;        // MAMDR_RPG (diagnostic)
    int groups = (rows_pad + rpg - 1) / rpg;
    if (groups > c->max_groups) {
        rpg = ((rows_pad + c->max_groups - 1) / c->max_groups + 7) / 8 * 8;
        groups = (rows_pad + rpg - 1) / rpg;
    }
    wa.n_groups = groups;
    wa.rows_per_group = rpg;
    wa.slabs = c->slabs;
    wa.slab_ld = c->slab_ld;
    wa.w0dom = c->eff + c->L.w0 + (size_t)(2 * EMB) * H1;
    wa.w0dom_copy = c->w0dom_copy;
    wa.loss_part = c->loss_part;
    wa.n_loss_tiles = rows_pad / TILE_ROWS;
    wa.rows = rows;
    wa.dense = c->eff;
    wa.dm_count = 0;              // no regularisers in this tower: loss = mean BCE
    wa.l2_emb = 0.f;
    wa.frozen_sumsq = c->frozen_sumsq;
    wa.loss_out = loss_out;
    StarPnBwdArgs ba;
    memset(&ba, 0, sizeof(ba));
    ba.user_tab = ta.user_tab;
    ba.item_tab = ta.item_tab;
    ba.dm_row = blk + c->SL.dm + (size_t)domain * EMB;
    ba.urow = c->urow;
    ba.irow = c->irow;
    ba.rows = rows;
    ba.n_chunks = chunks;
    ba.dxe = c->dxe;
    ba.pn = c->pn;
    ba.part = c->star_part;
    ba.sums = c->star_sums;
    ba.dmpart = c->star_dmpart;
    ba.dmsum = c->star_sums + 2 * XDIM;
    ba.means = c->star_sums + 2 * XDIM + EMB;
    ba.fused = c->star_pn_fused ? 1 : 0;
    // lazy table Adam with fused tails: PartitionedNorm's backward first (it only needs the tower's outputs), then
    // [k_wgrad + k_emb_reduce(t) + k_emb_rows(t+1)], then [k_star_update + k_emb_catchup(t+1)]
    const bool tail = c->tail_fuse && c->cfg.emb_trainable && c->lazy && optimizer == MAMDR_OPT_ADAM && !c->profile &&
                      !loss_out;
    // ... and (round 6) without k_star_pnb_apply when a catch-up launch follows: the table rows take PartitionedNorm's
    // backward inside k_emb_reduce, the domain columns' partial sums ride in k_wgrad_reduce and are finished and stepped
    // in k_star_update_catchup (StarPnBwdArgs::fused == 2; the same roundings in the same order as the launch it replaces)
    const bool no_apply = tail && next_catchup && !ba.fused && c->star_pn_no_apply;
    if (no_apply) ba.fused = 2;
    if (tail) {
        {
            Prof p(c, MAMDR_KERNEL_AUX);
            launch_star_pn_bwd(ba, false, c->stream, c->star_pn_in_tower);   // (its last kernel, the domain-row column sums, rides below)
        }
        EmbStepArgs tea;
        fill_emb_args(c, optimizer, alpha, omb1, omb2, table_two_l2(c), rows, XDIM, tea);
        tea.flags_done = 1;
        tea.apply_now = 1;
        if (no_apply) {
            tea.pn_sums = c->star_sums;
            tea.pn_means = c->star_sums + 2 * XDIM + EMB;
            tea.pn = c->pn;
        }
        Prof p(c, MAMDR_KERNEL_WGRAD);
        launch_wgrad_reduce(wa, tea, next_rows, ba.fused == 1 ? nullptr : &ba, c->stream);
    } else {
        {
            Prof p(c, MAMDR_KERNEL_WGRAD);
            launch_wgrad(wa, c->stream);
        }
        Prof p(c, MAMDR_KERNEL_AUX);                  // PartitionedNorm's backward: 4 launches as one timed group
        launch_star_pn_bwd(ba, true, c->stream, c->star_pn_in_tower);
    }

    float* slot_m = optimizer == MAMDR_OPT_ACCUMULATE ? c->accum : c->adam_m;
    StarUpdateArgs ua;
    memset(&ua, 0, sizeof(ua));
    ua.p = blk;
    ua.m = slot_m + c->table_floats;
    ua.v = c->adam_v + c->table_floats;
    ua.SL = c->SL;
    ua.L = c->L;
    ua.n_domain = c->cfg.n_domain;
    ua.d = domain;
    ua.slabs = c->slabs;
    ua.n_groups = groups;
    ua.slab_ld = c->slab_ld;
    ua.sums = c->star_sums;
    ua.dmsum = c->star_sums + 2 * XDIM;
    ua.xdom = c->lin_w0dom ? c->pn + PN_XDOM_OFF : nullptr;
    ua.opt.optimizer = optimizer;
    ua.opt.alpha = alpha;
    ua.opt.omb1 = omb1;
    ua.opt.omb2 = omb2;
    ua.opt.eps = c->cfg.adam_eps;
    ua.opt.two_l2 = 0.f;
    if (lazy_idx >= 0) {
        ua.only_live = 1;
        ua.alpha_log = c->star_alpha;
        ua.log_idx = lazy_idx;
        if (eff_for_next) ua.eff_out = c->eff;
    }
    {
        Prof p(c, MAMDR_KERNEL_UPDATE);
        ua.dm_elsewhere = no_apply ? 1 : 0;
        if (tail && next_catchup) launch_star_update_catchup(ua, *next_catchup, no_apply ? &ba : nullptr, c->stream);
        else launch_star_update(ua, c->stream);
    }
    if (tail && next_rows) {
        std::swap(c->urow, c->urow_alt);
        std::swap(c->irow, c->irow_alt);
        std::swap(c->map_u, c->map_u_alt);
        std::swap(c->map_i, c->map_i_alt);
        c->rows_ready = true;
        c->catchup_ready = next_catchup != nullptr;
        c->tables_dirty = true;
    }
    if (c->cfg.emb_trainable && !tail) emb_post_step(c, optimizer, alpha, omb1, omb2, rows);
    return MAMDR_OK;
}

extern char** environ;
namespace mamdr {
// once per process (thread-safe static initialiser): a MAMDR_* name in the environment that nobody reads is reported
int env_warn_unknown() {
    static const int unknown = []() {
        int n = 0;
        for (char** e = environ; e && *e; ++e) {
            if (strncmp(*e, "MAMDR_", 6) != 0) continue;
            const char* eq = strchr(*e, '=');
            const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
            bool known = false;
            for (int i = 0; i < kNumEnvSwitches && !known; ++i) {
                const char* k = kEnvSwitches[i].name;
                const size_t kl = strlen(k);
                if (kl && k[kl - 1] == '*') known = len >= kl - 1 && strncmp(*e, k, kl - 1) == 0;
                else known = len == kl && strncmp(*e, k, kl) == 0;
            }
            if (!known) {
                fprintf(stderr, "mamdr: environment variable %.*s is not a switch this build reads (mamdr_env_switches() lists them)\n",
                        (int)len, *e);
                n += 1;
            }
        }
        return n;
    }();
    return unknown;
}
}  // namespace mamdr

extern "C" {

const char* mamdr_last_error(void) { return g_err; }
int mamdr_abi_version(void) { return MAMDR_ABI_VERSION; }

// ---- environment switches: one table (env_registry.h), handed out and checked against the process environment
const char* mamdr_env_switches(void) {
    static const std::string table = []() {
        std::string t;
        for (int i = 0; i < kNumEnvSwitches; ++i)
            t += std::string(kEnvSwitches[i].name) + "\t" + kEnvSwitches[i].reader + "\t" + kEnvSwitches[i].effect + "\n";
        return t;
    }();
    return table.c_str();
}
int mamdr_env_unknown(void) { return mamdr::env_warn_unknown(); }


int mamdr_create(const mamdr_config* cfg, void* stream, mamdr_ctx** out) {
    (void)mamdr::env_warn_unknown();
    if (!cfg || !out) return fail(MAMDR_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->abi_version != MAMDR_ABI_VERSION)
        return fail(MAMDR_EINVAL, "abi_version %d != %d", cfg->abi_version, MAMDR_ABI_VERSION);
    if (cfg->tower != MAMDR_TOWER_MLP && cfg->tower != MAMDR_TOWER_DEEPFM && cfg->tower != MAMDR_TOWER_STAR &&
        cfg->tower != MAMDR_TOWER_WDL && cfg->tower != MAMDR_TOWER_PNN && cfg->tower != MAMDR_TOWER_NFM)
        return fail(MAMDR_EINVAL, "unknown tower kind %d", cfg->tower);
    if (cfg->emb_dim != EMB || cfg->hidden[0] != H1 || cfg->hidden[1] != H2 || cfg->hidden[2] != H3)
        return fail(MAMDR_EINVAL, "kernels are specialised for emb_dim 128 and hidden (256,128,64); got %d (%d,%d,%d)",
                    cfg->emb_dim, cfg->hidden[0], cfg->hidden[1], cfg->hidden[2]);
    if (cfg->n_user <= 0 || cfg->n_item <= 0 || cfg->n_domain <= 0)
        return fail(MAMDR_EINVAL, "n_user/n_item/n_domain must be positive");
    if (cfg->max_batch <= 0 || cfg->max_batch % TILE_ROWS != 0)
        return fail(MAMDR_EINVAL, "max_batch must be a positive multiple of %d", TILE_ROWS);
    if (cfg->max_batch > 16384) return fail(MAMDR_EINVAL, "max_batch %d exceeds 16384", cfg->max_batch);
    if (!(cfg->dropout >= 0.f && cfg->dropout < 1.f)) return fail(MAMDR_EINVAL, "dropout rate must be in [0,1)");
    if (cfg->uncertainty_weight && cfg->tower == MAMDR_TOWER_STAR)
        return fail(MAMDR_ENOTBUILT, "uncertainty weighting is built for the deepctr towers of the step kernels (mlp / deepfm / wdl / pnn / nfm)");
    if ((cfg->tower == MAMDR_TOWER_PNN || cfg->tower == MAMDR_TOWER_NFM) && cfg->max_batch > 2048)
        return fail(MAMDR_ENOTBUILT, "the pnn / nfm towers' training step is built on the four-row tower: batches of up to 2,048 rows, "
                                     "not %d (the generic-layer engine, mamdr_graph_*, takes any batch size)", cfg->max_batch);

    mamdr_ctx* c = new (std::nothrow) mamdr_ctx();
    if (!c) return fail(MAMDR_EINVAL, "out of host memory");
    c->cfg = *cfg;
    c->stream = (hipStream_t)stream;
    c->nfm = cfg->tower == MAMDR_TOWER_NFM;
    c->deepfm = cfg->tower == MAMDR_TOWER_DEEPFM || cfg->tower == MAMDR_TOWER_WDL || c->nfm;   // linear tables (+ FM term)
    c->pnn = cfg->tower == MAMDR_TOWER_PNN;
    c->L = DenseLayout::make(cfg->n_domain, c->deepfm, cfg->uncertainty_weight != 0, c->pnn);
    c->table_floats = cfg->emb_trainable ? ((int64_t)cfg->n_user + cfg->n_item) * EMB : 0;
    if (c->deepfm && cfg->emb_trainable) {
        // each 1-d table padded to 4 floats so that the dense block stays 16-B aligned
        c->lin_user_off = c->table_floats;
        c->lin_item_off = c->lin_user_off + (((int64_t)cfg->n_user + 3) & ~(int64_t)3);
        c->table_floats = c->lin_item_off + (((int64_t)cfg->n_item + 3) & ~(int64_t)3);
    }
    c->star = cfg->tower == MAMDR_TOWER_STAR;
    c->SL = StarLayout::make(cfg->n_domain);
    c->AL = StarAuxLayout::make(cfg->n_domain);
    c->n_params = c->table_floats + (c->star ? c->SL.alloc : c->L.alloc);
    c->n_meta = c->star ? c->table_floats + c->SL.n_meta : c->n_params;
    c->data.resize((size_t)cfg->n_domain * 3);
    c->rows_pad_max = cfg->max_batch;
    if (const char* tt = getenv("MAMDR_TOWER_TILE")) c->tower_tile = atoi(tt);
    c->slab_ld = c->L.alloc + cfg->n_domain * H1;
    if (c->deepfm || c->pnn) {      // per-row terms of the domain-table gradient: S2 = onehot(domain)^T fmq
        c->s2_off = c->slab_ld;
        c->slab_ld += cfg->n_domain * EMB;
    }

    const size_t rp = (size_t)c->rows_pad_max;
    // dW0[256:384] without tiles: Dm^T . S in k_update, or (Star: one normalised domain row per batch) the
    // rank-1 form in k_star_update
    // (NFM: rows 256..383 of the input tile carry the bi-interaction, not the domain row: plain tiles)
    c->lin_w0dom = (c->star || cfg->n_domain <= 64) && !getenv("MAMDR_NO_W0LIN") && !c->nfm;
    std::vector<TileDesc> tiles = build_tiles(c->L, cfg->n_domain, c->deepfm, c->s2_off, c->lin_w0dom, c->star, c->pnn, c->nfm);
    c->n_tiles = (int)tiles.size();
    float thr[500];
    thr[0] = (float)(0.0 - 1e-7);
    for (int i = 0; i < 498; ++i) thr[i + 1] = (float)((double)(i + 1) * 1.0 / (double)(500 - 1));
    thr[499] = (float)(1.0 + 1e-7);

#define ALLOC(ptr, bytes)                                                        \
    do {                                                                         \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                      \
        if (e_ != hipSuccess) {                                                  \
            mamdr_destroy(c);                                                    \
            return fail(MAMDR_EHIP, "hipMalloc(%zu): %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                                        \
    } while (0)
    ALLOC(c->acts, rp * ACT_LD * sizeof(float));
    ALLOC(c->dz, rp * DZ_LD * sizeof(float));
    ALLOC(c->dlogit, rp * sizeof(float));
    ALLOC(c->w0dom_copy, (size_t)EMB * H1 * sizeof(float));
    ALLOC(c->dm_copy, (size_t)cfg->n_domain * EMB * sizeof(float));
    if (c->star) {
        const size_t chunks = (rp + STAR_CHUNK - 1) / STAR_CHUNK;
        ALLOC(c->eff, (size_t)c->L.alloc * sizeof(float));
        ALLOC(c->pn, (size_t)PN_WS_FLOATS * sizeof(float));
        ALLOC(c->star_alpha, (size_t)STAR_ALPHA_CAP * sizeof(float));
        if (const char* sd = getenv("MAMDR_STAR_DENSE_SLICES")) c->star_dense_slices = atoi(sd) != 0;
        ALLOC(c->star_part, chunks * 2 * XDIM * sizeof(double));    // forward: double sums; backward: float sums
        ALLOC(c->star_sums, (4 * XDIM + EMB) * sizeof(float));      // sums | domain-row gradient | s1 / B, s2 / B
        ALLOC(c->star_dmpart, chunks * EMB * sizeof(float));
    }
    if (cfg->emb_trainable || c->star) {
        ALLOC(c->dxe, rp * (c->star ? XDIM : 2 * EMB) * sizeof(float));
        ALLOC(c->urow, rp * sizeof(int32_t));
        ALLOC(c->irow, rp * sizeof(int32_t));
    }
    if (cfg->emb_trainable) {
        ALLOC(c->map_u, (size_t)cfg->n_user * sizeof(int32_t));
        ALLOC(c->map_i, (size_t)cfg->n_item * sizeof(int32_t));
        ALLOC(c->urow_alt, rp * sizeof(int32_t));
        ALLOC(c->irow_alt, rp * sizeof(int32_t));
        ALLOC(c->map_u_alt, (size_t)cfg->n_user * sizeof(int32_t));
        ALLOC(c->map_i_alt, (size_t)cfg->n_item * sizeof(int32_t));
        ALLOC(c->gbuf_u, rp * EMB * sizeof(float));
        ALLOC(c->gbuf_i, rp * EMB * sizeof(float));
        ALLOC(c->hasdup_u, rp * sizeof(int32_t));
        ALLOC(c->hasdup_i, rp * sizeof(int32_t));
        const char* dense_env = getenv("MAMDR_DENSE_ADAM");
        c->lazy = !(dense_env && atoi(dense_env) != 0);
        if (const char* fe = getenv("MAMDR_LAZY_FLUSH_EVERY")) c->flush_every = atoi(fe) > 0 ? atoi(fe) : c->flush_every;
        if (const char* cap_env = getenv("MAMDR_LAZY_LOG_CAP")) {      // tests: force the alpha ring to wrap
            const int cap = atoi(cap_env);
            if (cap >= 4 && (cap & (cap - 1)) == 0) c->log_cap = cap;
        }
        ALLOC(c->last_u, (size_t)cfg->n_user * sizeof(int32_t));
        ALLOC(c->last_i, (size_t)cfg->n_item * sizeof(int32_t));
        ALLOC(c->alpha_log, (size_t)c->log_cap * sizeof(float));
        if (c->deepfm) {
            ALLOC(c->glin_u, rp * sizeof(float));
            ALLOC(c->glin_i, rp * sizeof(float));
        }
    }
    if (c->deepfm || c->pnn) ALLOC(c->fmq, rp * EMB * sizeof(float));
    if (c->pnn) ALLOC(c->ipbuf, rp * 4 * sizeof(float));
    ALLOC(c->domrow, rp * sizeof(int32_t));
    ALLOC(c->loss_part, (rp / 4) * sizeof(float));
    ALLOC(c->wT, (size_t)WT_FLOATS * sizeof(float));
    {
        if (const char* nw = getenv("MAMDR_T4_NO_W1L")) c->t4_no_w1l = atoi(nw) != 0;
        // Which tower for how many rows (frozen-table mlp, measured at 2,048 rows of Taobao-10, us / step): k_tower4 +
        // k_wgrad_adam 40.4 (512 four-row tiles: two rounds of workgroups, no W1 image), k_tower + k_wgrad_adam 39.0,
        // k_tower + k_wgrad + k_update 38.6 (128 sixteen-row tiles, the lean instance: 21.4 us against k_tower4's 26.9).
        // So the four-row tower and the fused path serve what fits ONE round of workgroups (4 rows x CUs = 1,024 rows);
        // with trainable tables or the DeepFM terms the four-row tower stays ahead up to 2,048 rows (Amazon-6 at
        // 2,048: 29.3 vs 34.2 us).
        {
            int dev = 0, n_cu = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
                n_cu <= 0)
                n_cu = 256;
            const int one_round = std::min(2048, std::max(256, 4 * n_cu));
            c->fused_max_batch = one_round;
            if (cfg->tower == MAMDR_TOWER_MLP && !cfg->emb_trainable) c->tower4_max_rows = one_round;
        }
        const char* fe = getenv("MAMDR_FUSED");
        c->fused = cfg->tower == MAMDR_TOWER_MLP && !cfg->emb_trainable && !cfg->uncertainty_weight && c->lin_w0dom &&
                   cfg->n_domain <= 64 && !(fe && atoi(fe) == 0);
        if (c->fused && fe && atoi(fe) == 2) c->fused_max_batch = 1 << 30;
        if (c->fused) {
            ALLOC(c->pdm, (size_t)DM_PARTS * cfg->n_domain * EMB * sizeof(float));
            ALLOC(c->dmsnap[0], (size_t)3 * cfg->n_domain * EMB * sizeof(float));
            ALLOC(c->dmsnap[1], (size_t)3 * cfg->n_domain * EMB * sizeof(float));
            if (const char* de = getenv("MAMDR_DM_EACH")) c->dm_finish_each = atoi(de) != 0;
            if (const char* de = getenv("MAMDR_DM_CALL")) c->dm_finish_call = atoi(de) != 0;
            if (const char* de = getenv("MAMDR_NO_W2_DIRECT")) c->w2_direct_ok = atoi(de) == 0;
            if (const char* pe = getenv("MAMDR_NO_PREGATHER")) c->use_pre = atoi(pe) == 0;
        }
    }
    if (const char* ev = getenv("MAMDR_NO_TAILFUSE")) c->tail_fuse = atoi(ev) == 0;
    if (const char* ev = getenv("MAMDR_STAR_PNB_KERNEL")) c->star_pn_in_tower = atoi(ev) == 0;
    if (const char* ev = getenv("MAMDR_STAR_PNB_FUSED")) c->star_pn_fused = atoi(ev) != 0 && c->star_pn_in_tower;
    if (const char* ev = getenv("MAMDR_STAR_PNB_APPLY")) c->star_pn_no_apply = atoi(ev) == 0;
    if (const char* ev = getenv("MAMDR_MAX_GROUPS")) c->max_groups = atoi(ev) > 0 ? atoi(ev) : c->max_groups;   // diagnostic
    if (const char* ev = getenv("MAMDR_RPG")) c->rpg_override = atoi(ev) / 8 * 8;
    if (const char* ev = getenv("MAMDR_NO_GATHER_PF")) c->gather_pf = atoi(ev) == 0;
    if (const char* ev = getenv("MAMDR_FUSED_PF")) c->fused_pf = atoi(ev) != 0;
    if (const char* ev = getenv("MAMDR_WGRAD_PAIRS")) c->wgrad_pairs = atoi(ev) != 0;
    if (const char* ev = getenv("MAMDR_GATHER_PF_IN")) c->gather_pf_in_wgrad = strcmp(ev, "update") != 0;
    ALLOC(c->slabs, (size_t)c->max_groups * c->slab_ld * sizeof(float));
    ALLOC(c->tiles, tiles.size() * sizeof(TileDesc));
    ALLOC(c->thresholds, sizeof(thr));
    ALLOC(c->frozen_sumsq, 4 * sizeof(float));
    ALLOC(c->sumsq_partials, 1024 * sizeof(float));
#undef ALLOC
    hipError_t e = hipMemsetAsync(c->slabs, 0, (size_t)c->max_groups * c->slab_ld * sizeof(float), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->frozen_sumsq, 0, 4 * sizeof(float), c->stream);
    if (const char* ev = getenv("MAMDR_RPG")) c->rpg_override = atoi(ev) / 8 * 8;
    if (cfg->emb_trainable) {
        if (e == hipSuccess) e = hipMemsetAsync(c->hasdup_u, 0, rp * sizeof(int32_t), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->hasdup_i, 0, rp * sizeof(int32_t), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->last_u, 0, (size_t)cfg->n_user * sizeof(int32_t), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->last_i, 0, (size_t)cfg->n_item * sizeof(int32_t), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->alpha_log, 0, (size_t)c->log_cap * sizeof(float), c->stream);
        launch_emb_map_init(c->map_u, cfg->n_user, c->stream);
        launch_emb_map_init(c->map_i, cfg->n_item, c->stream);
        launch_emb_map_init(c->map_u_alt, cfg->n_user, c->stream);
        launch_emb_map_init(c->map_i_alt, cfg->n_item, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c->tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->thresholds, thr, sizeof(thr), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // host staging buffers go out of scope
    if (e != hipSuccess) {
        mamdr_destroy(c);
        return fail(MAMDR_EHIP, "workspace initialisation: %s", hipGetErrorString(e));
    }
    *out = c;
    return MAMDR_OK;
}

int mamdr_destroy(mamdr_ctx* c) {
    if (!c) return MAMDR_OK;
    for (int k = 0; k < MAMDR_KERNEL_COUNT; ++k)
        for (EventPair& p : c->ev[k]) {
            if (p.own_a) (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    void* ptrs[] = {c->urow_alt, c->irow_alt, c->map_u_alt, c->map_i_alt, c->acts, c->dz, c->dlogit, c->w0dom_copy, c->dm_copy, c->wT, c->dxe, c->urow, c->irow, c->map_u, c->map_i, c->gbuf_u, c->gbuf_i, c->hasdup_u, c->hasdup_i, c->last_u, c->last_i, c->alpha_log, c->star_alpha, c->pdm, c->dmsnap[0], c->dmsnap[1], c->xpre, c->pdom, c->plabel, c->fmq, c->ipbuf, c->glin_u, c->glin_i, c->eff, c->pn, c->star_part, c->star_sums, c->star_dmpart, c->domrow, c->loss_part, c->eval_part, c->slabs,
                    c->tiles, c->thresholds, c->frozen_sumsq, c->sumsq_partials};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete c;
    return MAMDR_OK;
}

int64_t mamdr_param_count(const mamdr_ctx* c) { return c ? c->n_params : 0; }
int64_t mamdr_meta_count(const mamdr_ctx* c) { return c ? c->n_meta : 0; }
int64_t mamdr_aux_count(const mamdr_ctx* c) { return (c && c->star) ? c->AL.count : 0; }

int mamdr_bind_aux(mamdr_ctx* c, float* d_aux) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (!c->star) return fail(MAMDR_ESTATE, "this tower has no auxiliary state");
    if (!d_aux || ((uintptr_t)d_aux & 15)) return fail(MAMDR_EINVAL, "aux pointer null or not 16-byte aligned");
    c->aux = d_aux;
    return MAMDR_OK;
}

int mamdr_param_segment(const mamdr_ctx* c, int seg, int64_t* offset, int64_t* count) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (!offset || !count) return fail(MAMDR_EINVAL, "null argument");
    const DenseLayout& L = c->L;
    const int64_t base = c->table_floats;
    int64_t off = 0, cnt = 0;
    if (c->star) {
        const StarLayout& S = c->SL;
        const int64_t D = c->cfg.n_domain;
        const bool tr = c->cfg.emb_trainable != 0;
        if (seg >= MAMDR_SEG_STAR_WS0 && seg <= MAMDR_SEG_STAR_WS2) {
            off = base + S.ws[seg - MAMDR_SEG_STAR_WS0]; cnt = StarLayout::ksize(seg - MAMDR_SEG_STAR_WS0);
        } else if (seg >= MAMDR_SEG_STAR_BS0 && seg <= MAMDR_SEG_STAR_BS2) {
            off = base + S.bs[seg - MAMDR_SEG_STAR_BS0]; cnt = StarLayout::bsize(seg - MAMDR_SEG_STAR_BS0);
        } else if (seg >= MAMDR_SEG_STAR_WD0 && seg <= MAMDR_SEG_STAR_WD2) {
            off = base + S.wd[seg - MAMDR_SEG_STAR_WD0]; cnt = D * StarLayout::ksize(seg - MAMDR_SEG_STAR_WD0);
        } else if (seg >= MAMDR_SEG_STAR_BD0 && seg <= MAMDR_SEG_STAR_BD2) {
            off = base + S.bd[seg - MAMDR_SEG_STAR_BD0]; cnt = D * StarLayout::bsize(seg - MAMDR_SEG_STAR_BD0);
        } else {
            switch (seg) {
                case MAMDR_SEG_USER_EMB: off = 0; cnt = tr ? (int64_t)c->cfg.n_user * EMB : 0; break;
                case MAMDR_SEG_ITEM_EMB: off = tr ? (int64_t)c->cfg.n_user * EMB : 0; cnt = tr ? (int64_t)c->cfg.n_item * EMB : 0; break;
                case MAMDR_SEG_DOMAIN_EMB: off = base + S.dm; cnt = D * EMB; break;
                case MAMDR_SEG_PN_GAMMA_SHARED: off = base + S.pgs; cnt = XDIM; break;
                case MAMDR_SEG_PN_BETA_SHARED: off = base + S.pbs; cnt = XDIM; break;
                case MAMDR_SEG_PN_GAMMA_SPEC: off = base + S.pgd; cnt = D * XDIM; break;
                case MAMDR_SEG_PN_BETA_SPEC: off = base + S.pbd; cnt = D * XDIM; break;
                case MAMDR_SEG_WO: off = base + S.wo; cnt = H3; break;
                case MAMDR_SEG_GB: off = base + S.gb; cnt = 1; break;
                default:
                    if (seg < 0 || seg >= MAMDR_SEG_COUNT) return fail(MAMDR_EINVAL, "unknown segment %d", seg);
                    off = 0; cnt = 0;      // a segment of another tower
            }
        }
        *offset = off;
        *count = cnt;
        return MAMDR_OK;
    }
    if (seg >= MAMDR_SEG_STAR_WS0 && seg <= MAMDR_SEG_STAR_BD2) {  // Star segments are absent from this tower
        *offset = 0;
        *count = 0;
        return MAMDR_OK;
    }
    switch (seg) {
        case MAMDR_SEG_USER_EMB: off = 0; cnt = c->cfg.emb_trainable ? (int64_t)c->cfg.n_user * EMB : 0; break;
        case MAMDR_SEG_ITEM_EMB:
            off = c->cfg.emb_trainable ? (int64_t)c->cfg.n_user * EMB : 0;
            cnt = c->cfg.emb_trainable ? (int64_t)c->cfg.n_item * EMB : 0;
            break;
        case MAMDR_SEG_DOMAIN_EMB: off = base + L.dm; cnt = (int64_t)c->cfg.n_domain * EMB; break;
        case MAMDR_SEG_W0:
            off = base + L.w0 + (c->nfm ? 2 * EMB * H1 : 0);
            cnt = c->nfm ? EMB * H1 : XDIM * H1;
            break;
        case MAMDR_SEG_W1: off = base + L.w1; cnt = H1 * H2; break;
        case MAMDR_SEG_W2: off = base + L.w2; cnt = H2 * H3; break;
        case MAMDR_SEG_B0: off = base + L.b0; cnt = H1; break;
        case MAMDR_SEG_B1: off = base + L.b1; cnt = H2; break;
        case MAMDR_SEG_B2: off = base + L.b2; cnt = H3; break;
        case MAMDR_SEG_WO: off = base + L.wo; cnt = H3; break;
        case MAMDR_SEG_GB: off = base + L.gb; cnt = 1; break;
        case MAMDR_SEG_LIN_USER:
            off = c->lin_user_off;
            cnt = (c->deepfm && c->cfg.emb_trainable) ? c->cfg.n_user : 0;
            break;
        case MAMDR_SEG_LIN_ITEM:
            off = c->lin_item_off;
            cnt = (c->deepfm && c->cfg.emb_trainable) ? c->cfg.n_item : 0;
            break;
        case MAMDR_SEG_LIN_DOMAIN: off = base + L.ld; cnt = L.ld_count; break;
        case MAMDR_SEG_LOG_VAR: off = base + L.lv; cnt = L.lv_count; break;
        case MAMDR_SEG_W0X: off = base + L.wx; cnt = L.wx_count; break;
        default: return fail(MAMDR_EINVAL, "unknown segment %d", seg);
    }
    *offset = off;
    *count = cnt;
    return MAMDR_OK;
}

int mamdr_bind_state(mamdr_ctx* c, float* d_params, float* d_adam_m, float* d_adam_v) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (!d_params || !d_adam_m || !d_adam_v) return fail(MAMDR_EINVAL, "null state pointer");
    if (((uintptr_t)d_params | (uintptr_t)d_adam_m | (uintptr_t)d_adam_v) & 15)
        return fail(MAMDR_EINVAL, "state pointers must be 16-byte aligned");
    if (c->params) sync_tables(c);
    c->params = d_params;
    c->adam_m = d_adam_m;
    c->adam_v = d_adam_v;
    return MAMDR_OK;
}

int mamdr_optimizer_reset(mamdr_ctx* c) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (!c->adam_m) return fail(MAMDR_ESTATE, "mamdr_bind_state has not been called");
    sync_tables(c);             // pending moves of the lagging rows belong to the old optimiser state
    if (c->last_u) {
        HIP_TRY(hipMemsetAsync(c->last_u, 0, (size_t)c->cfg.n_user * sizeof(int32_t), c->stream));
        HIP_TRY(hipMemsetAsync(c->last_i, 0, (size_t)c->cfg.n_item * sizeof(int32_t), c->stream));
    }
    c->flush_t = 0;
    HIP_TRY(hipMemsetAsync(c->adam_m, 0, (size_t)c->n_params * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->adam_v, 0, (size_t)c->n_params * sizeof(float), c->stream));
    c->adam_t = 0;
    c->b1p = 1.0f;
    c->b2p = 1.0f;
    return MAMDR_OK;
}

int64_t mamdr_optimizer_steps(const mamdr_ctx* c) { return c ? c->adam_t : 0; }

// Restore the two host-side counters of a run (a checkpoint's `beta1_power` / `beta2_power` variables and the position of
// the dropout stream): the live state is brought up to date first (pending domain-table step, lagging table rows), then
// every table row counts as current AT the new step count -- the caller supplies weights and slots that belong to it.
int mamdr_set_counters(mamdr_ctx* c, int64_t optimizer_steps, int64_t dropout_steps) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (optimizer_steps < 0 || optimizer_steps > (int64_t)0x7ffffff0 || dropout_steps < 0 || dropout_steps > (int64_t)0xffffffffLL)
        return fail(MAMDR_EINVAL, "mamdr_set_counters(%lld, %lld): out of range", (long long)optimizer_steps, (long long)dropout_steps);
    if (!c->adam_m) return fail(MAMDR_ESTATE, "mamdr_bind_state has not been called");
    sync_tables(c);
    if (c->last_u) {
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->last_u), (int)optimizer_steps, (size_t)c->cfg.n_user, c->stream));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->last_i), (int)optimizer_steps, (size_t)c->cfg.n_item, c->stream));
    }
    c->flush_t = optimizer_steps;
    c->adam_t = optimizer_steps;
    float b1 = 1.0f, b2 = 1.0f;                 // TF's running products, one fp32 rounding per step (as the step loop forms them)
    for (int64_t t = 0; t < optimizer_steps && (b1 != 0.f || b2 != 0.f); ++t) {      // (both products end at 0: ~1e5 steps)
        b1 = b1 * c->cfg.adam_beta1;
        b2 = b2 * c->cfg.adam_beta2;
    }
    c->b1p = b1;
    c->b2p = b2;
    c->global_step = (uint32_t)dropout_steps;
    return MAMDR_OK;
}
int64_t mamdr_table_flushes(const mamdr_ctx* c, int32_t forced_only) {
    return !c ? 0 : forced_only ? c->n_flush_forced : c->n_flush;
}

int mamdr_sync_tables(mamdr_ctx* c) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    sync_tables(c);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}

int mamdr_bind_accumulator(mamdr_ctx* c, float* d_acc) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (!d_acc || ((uintptr_t)d_acc & 15)) return fail(MAMDR_EINVAL, "accumulator pointer null or not 16-byte aligned");
    c->accum = d_acc;
    return MAMDR_OK;
}

int mamdr_bind_table(mamdr_ctx* c, int seg, const float* d_rows, int64_t n_rows) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (c->cfg.emb_trainable) return fail(MAMDR_ESTATE, "tables are trainable: they live in the flat vector");
    if (!d_rows || ((uintptr_t)d_rows & 15)) return fail(MAMDR_EINVAL, "table pointer null or not 16-byte aligned");
    c->pg.clear();              // rows gathered ahead of their calls came from the old table
    if (seg == MAMDR_SEG_USER_EMB) {
        if (n_rows != c->cfg.n_user) return fail(MAMDR_EINVAL, "user table has %lld rows, config says %d", (long long)n_rows, c->cfg.n_user);
        c->user_tab = d_rows;
        launch_sumsq(d_rows, n_rows * EMB, c->sumsq_partials, c->frozen_sumsq + 0, c->stream);
    } else if (seg == MAMDR_SEG_ITEM_EMB) {
        if (n_rows != c->cfg.n_item) return fail(MAMDR_EINVAL, "item table has %lld rows, config says %d", (long long)n_rows, c->cfg.n_item);
        c->item_tab = d_rows;
        launch_sumsq(d_rows, n_rows * EMB, c->sumsq_partials, c->frozen_sumsq + 1, c->stream);
    } else {
        return fail(MAMDR_EINVAL, "segment %d is not a bindable table", seg);
    }
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}

int mamdr_bind_domain_data(mamdr_ctx* c, int domain, int split, const int32_t* d_uid, const int32_t* d_pid,
                           const int32_t* d_domain, const float* d_label, int64_t n_rows) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    SplitData* d = split_of(c, domain, split);
    if (!d) return fail(MAMDR_EINVAL, "domain %d / split %d out of range", domain, split);
    if (n_rows < 0 || n_rows > 0x7fffffff) return fail(MAMDR_EINVAL, "n_rows out of range");
    if (n_rows > 0 && (!d_uid || !d_pid || !d_domain || !d_label)) return fail(MAMDR_EINVAL, "null column pointer");
    d->bound = true;
    d->uid = d_uid;
    d->pid = d_pid;
    d->dom = d_domain;
    d->label = d_label;
    d->n = n_rows;
    c->pg.clear();              // (a pass gathered ahead of its call may have come from the old columns)
    const int64_t tiles = (n_rows + TILE_ROWS - 1) / TILE_ROWS;
    if (tiles > c->eval_part_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->eval_part) HIP_TRY(hipFree(c->eval_part));
        c->eval_part = nullptr;
        HIP_TRY(hipMalloc((void**)&c->eval_part, (size_t)tiles * sizeof(float)));
        c->eval_part_cap = tiles;
    }
    return MAMDR_OK;
}

// the pass buffer holds at least `rows` positions (contents are lost when it grows)
static int grow_pass_buffer(mamdr_ctx* c, int64_t rows) {
    if (rows <= c->pre_cap) return MAMDR_OK;
    const int64_t cap = rows + rows / 4 + 1024;
    c->pg.clear();
    if (c->xpre) { (void)hipFree(c->xpre); (void)hipFree(c->pdom); (void)hipFree(c->plabel); }
    c->xpre = nullptr; c->pdom = nullptr; c->plabel = nullptr; c->pre_cap = 0;
    HIP_TRY(hipMalloc((void**)&c->xpre, (size_t)cap * 2 * EMB * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->pdom, (size_t)cap * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&c->plabel, (size_t)cap * sizeof(float)));
    c->pre_cap = cap;
    return MAMDR_OK;
}

int64_t mamdr_pregather_hits(const mamdr_ctx* c) { return c ? c->pg_hits : 0; }
int64_t mamdr_pregather_launches(const mamdr_ctx* c) { return c ? c->pg_launches : 0; }

int mamdr_pregather_passes(mamdr_ctx* c, int32_t n_passes, const int32_t* h_domains, const int32_t* const* h_d_perms,
                           const int64_t* h_pass_rows, int32_t batch) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (ready(c)) return MAMDR_ESTATE;
    if (n_passes < 0 || (n_passes > 0 && !h_domains)) return fail(MAMDR_EINVAL, "pregather: bad pass list");
    c->pg.clear();
    c->pg_pos = 0;
    // a hint: it only has an effect where a call would gather its pass itself (frozen tables, k_wgrad_adam path)
    if (n_passes == 0 || batch <= 0 || batch > c->cfg.max_batch || !c->use_pre ||
        !(c->fused && (batch + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS <= c->fused_max_batch))
        return MAMDR_OK;
    if (n_passes > PREP_MAX_PASSES) n_passes = PREP_MAX_PASSES;      // the later ones gather themselves
    PassPrepMultiArgs a;
    memset(&a, 0, sizeof(a));
    int64_t off = 0;
    int wgs = 0;
    for (int k = 0; k < n_passes; ++k) {
        SplitData* d = split_of(c, h_domains[k], MAMDR_SPLIT_TRAIN);
        if (!d || !d->bound) return fail(MAMDR_ESTATE, "pregather: train split of domain %d is not bound", h_domains[k]);
        int64_t n = h_pass_rows ? h_pass_rows[k] : -1;
        if (n < 0) n = d->n;
        if (n > d->n) return fail(MAMDR_EINVAL, "pregather: pass of %lld rows exceeds the %lld rows of domain %d", (long long)n,
                                  (long long)d->n, h_domains[k]);
        PassPrepMultiArgs::Pass& p = a.p[k];
        p.uid = d->uid;
        p.pid = d->pid;
        p.dom = d->dom;
        p.label = d->label;
        p.perm = h_d_perms ? h_d_perms[k] : nullptr;
        p.n = n;
        p.n_rows_split = d->n;
        p.out_off = off;
        p.pad_dom = h_domains[k];
        const int64_t w = n > 0 ? (n + 16 + 3) / 4 : 0;      // (an empty pass has no steps: nothing to gather)
        wgs += (int)w;
        a.wg_end[k] = wgs;
        c->pg.push_back(mamdr_ctx::PgEntry{h_domains[k], p.perm, n, off, batch});
        off += 4 * w;
    }
    a.n_pass = n_passes;
    if (wgs == 0) return MAMDR_OK;
    {
        std::vector<mamdr_ctx::PgEntry> keep = c->pg;      // (growing the buffer drops the entries of the OLD buffer)
        if (int rc = grow_pass_buffer(c, off)) return rc;
        c->pg = keep;
    }
    a.user_tab = c->user_tab;
    a.item_tab = c->item_tab;
    a.n_user = c->cfg.n_user;
    a.n_item = c->cfg.n_item;
    a.n_domain = c->cfg.n_domain;
    a.xpre = c->xpre;
    a.pdom = c->pdom;
    a.plabel = c->plabel;
    prof_break(c);
    {
        Prof p(c, MAMDR_KERNEL_AUX);
        launch_pass_prep_multi(a, c->stream);
        c->pg_launches++;
    }
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}

int mamdr_train_steps(mamdr_ctx* c, int domain, const int32_t* d_perm, int64_t first_step, int64_t n_steps,
                      int32_t batch, uint32_t dropout_seed, int32_t optimizer, float lr, float* d_loss_out) {
    return mamdr_train_steps_n(c, domain, d_perm, -1, first_step, n_steps, batch, dropout_seed, optimizer, lr, d_loss_out);
}

int mamdr_train_steps_n(mamdr_ctx* c, int domain, const int32_t* d_perm, int64_t pass_rows, int64_t first_step,
                        int64_t n_steps, int32_t batch, uint32_t dropout_seed, int32_t optimizer, float lr,
                        float* d_loss_out) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (ready(c)) return MAMDR_ESTATE;
    SplitData* d = split_of(c, domain, MAMDR_SPLIT_TRAIN);
    if (!d || !d->bound) return fail(MAMDR_ESTATE, "train split of domain %d is not bound", domain);
    if (batch <= 0 || batch > c->cfg.max_batch) return fail(MAMDR_EINVAL, "batch %d outside (0, max_batch=%d]", batch, c->cfg.max_batch);
    if (optimizer != MAMDR_OPT_ADAM && optimizer != MAMDR_OPT_SGD && optimizer != MAMDR_OPT_ACCUMULATE)
        return fail(MAMDR_EINVAL, "unknown optimizer %d", optimizer);
    if (optimizer == MAMDR_OPT_ACCUMULATE && !c->accum)
        return fail(MAMDR_ESTATE, "MAMDR_OPT_ACCUMULATE needs mamdr_bind_accumulator first");
    if (first_step < 0 || n_steps < 0) return fail(MAMDR_EINVAL, "negative step range");
    if (pass_rows < 0) pass_rows = d->n;                 // the whole split
    if (pass_rows > d->n) return fail(MAMDR_EINVAL, "pass of %lld rows exceeds the %lld rows of domain %d",
                                      (long long)pass_rows, (long long)d->n, domain);
    const int64_t pass_steps = (pass_rows + batch - 1) / batch;
    if (first_step + n_steps > pass_steps)
        return fail(MAMDR_EINVAL, "steps [%lld,%lld) exceed the %lld batches of domain %d", (long long)first_step,
                    (long long)(first_step + n_steps), (long long)pass_steps, domain);
    if (n_steps == 0) return MAMDR_OK;      // an empty pass (empty domain, meta_train_step window of nothing): no launch, no state change

    const float rate = c->cfg.dropout;
    double thr = (double)rate * 4294967296.0;
    const uint32_t drop_thresh = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)(int64_t)thr;
    const float keep_scale = (float)(1.0 / (1.0 - (double)rate));
    const float omb1 = 1.0f - c->cfg.adam_beta1, omb2 = 1.0f - c->cfg.adam_beta2;
    // SGD / accumulate steps update the tables densely: bring every lagging row up to date first
    if (optimizer != MAMDR_OPT_ADAM) sync_tables(c);

    // small batches run the 4-row-tile tower (all CUs busy); it needs transposed W1 / W2 copies:
    // refreshed here because the caller may have assigned new weights, kept current by k_update
    const bool may_use4 = !c->star && c->tower_tile != 16;
    // ... only when a step of THIS call is small enough for that tower (the rows of a pass's steps never grow: the
    // last one is the smallest) -- a 4,096-row call over a domain without a short last batch needs no copies at all
    bool need_wT = false;
    if (may_use4 && n_steps > 0) {
        const int64_t last_base = (first_step + n_steps - 1) * (int64_t)batch;
        const int64_t last_rows = std::min<int64_t>(batch, pass_rows - last_base);
        const int64_t last_pad = (last_rows + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
        need_wT = c->tower_tile == 4 || last_pad <= c->tower4_max_rows;
    }

    c->rows_ready = false;
    c->catchup_ready = false;
    prof_break(c);
    // one path per call (the pending domain-table step lives across the steps of a call): k_wgrad_adam for batches up
    // to fused_max_batch rows (measured: 27.3 vs 29.5 us / step at 1,024 rows, a tie at 4,096)
    const bool fused = c->fused && (batch + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS <= c->fused_max_batch;
    // k_wgrad_adam path: the domain table's step of the previous step -- of this call or, between two Adam calls, of
    // the previous call (c->dm_pending); every other kind of call starts from the materialised table
    DmStep& dm_pending = c->dm_pending;
    if (!(fused && optimizer == MAMDR_OPT_ADAM)) finish_dm(c);
    // ... and on that path the rows of the whole call are resolved and gathered once (frozen tables; 4-row tower)
    const int64_t pre_pos0 = first_step * batch;
    const int64_t pre_n = std::min<int64_t>((first_step + n_steps) * (int64_t)batch, pass_rows) - pre_pos0;
    const bool pre = fused && c->use_pre && n_steps > 0 && pre_n > 0;
    // the transposed copies: built here unless the previous call's steps left them current (no sync_tables since).  On
    // the k_wgrad_adam path with the W1 image (k_tower4<.., W1L, PRE>) only W2T is read, and only the call's FIRST tower
    // can find it stale -- k_wgrad_adam rewrites every copy with the step -- so that tower reads W2 itself (w2_direct:
    // 32 B runs of 128 rows, four loads per lane) and nothing is transposed at all
    bool build_wT = need_wT && !c->wT_valid;
    bool w2_direct = false;
    if (build_wT && pre && c->w2_direct_ok && optimizer != MAMDR_OPT_ACCUMULATE && !c->t4_no_w1l && c->tower_tile != 16 &&
        tower4_w1l_ready() &&
        tower4_takes_w1l(std::min<int64_t>(batch, pass_rows - first_step * (int64_t)batch), c->t4_no_w1l)) {
        build_wT = false;
        w2_direct = true;
    }
    // ... unless mamdr_pregather_passes gathered this pass ahead of the call: entries are consumed in order (an entry
    // stays current while calls keep working on its pass); a call that matches none drops them all
    int64_t pre_base = 0;              // position of row pre_pos0 inside the pass buffer
    bool pre_cached = false;
    // ... and a call whose FIRST step runs the 16-row tower (which reads no copy) needs none built either: k_update
    // rewrites the copy of every element it steps, so they are current from the call's second step on -- before the
    // short last batch that runs the four-row tower (the rows of a pass's steps never grow)
    if (build_wT && optimizer != MAMDR_OPT_ACCUMULATE && !fused && !c->star) {
        const int64_t first_rows = std::min<int64_t>(batch, pass_rows - first_step * (int64_t)batch);
        const int64_t first_pad = (first_rows + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
        if (!(c->tower_tile == 4 || first_pad <= c->tower4_max_rows) && n_steps > 1) build_wT = false;
    }
    if (pre) {
        for (size_t k = c->pg_pos; k < c->pg.size(); ++k) {
            const mamdr_ctx::PgEntry& e = c->pg[k];
            if (e.domain == domain && e.perm == d_perm && e.n == pass_rows && e.batch == batch) {
                c->pg_pos = k;
                pre_base = e.off + pre_pos0;
                pre_cached = true;
                c->pg_hits += 1;
                break;
            }
        }
        if (!pre_cached) c->pg.clear();
    }
    if (pre && pre_cached) {
        if (build_wT) launch_transpose_w(c->params + c->table_floats, c->L, c->wT, c->stream);
    } else if (pre) {
        if (int rc = grow_pass_buffer(c, pre_n + 16)) return rc;
        PassPrepArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.user_tab = c->user_tab;
        pa.item_tab = c->item_tab;
        pa.uid = d->uid;
        pa.pid = d->pid;
        pa.dom = d->dom;
        pa.label = d->label;
        pa.perm = d_perm;
        pa.pos0 = pre_pos0;
        pa.n = pre_n;
        pa.n_rows_split = d->n;
        pa.n_user = c->cfg.n_user;
        pa.n_item = c->cfg.n_item;
        pa.n_domain = c->cfg.n_domain;
        pa.pad_dom = domain;
        pa.xpre = c->xpre;
        pa.pdom = c->pdom;
        pa.plabel = c->plabel;
        if (build_wT) {                // ... in k_pass_prep's launch (one launch less per call)
            pa.tw_dense = c->params + c->table_floats;
            pa.tw_L = c->L;
            pa.tw_wT = c->wT;
        }
        Prof p(c, MAMDR_KERNEL_AUX);
        launch_pass_prep(pa, c->stream);
    } else if (build_wT) {
        launch_transpose_w(c->params + c->table_floats, c->L, c->wT, c->stream);
    }
    // (what the call's steps leave behind: k_wgrad_adam / k_update keep the copies of what they step current when the
    // call uses them; steps without them make them stale; accumulate steps change no weight)
    if (optimizer != MAMDR_OPT_ACCUMULATE) c->wT_valid = need_wT;
    else if (build_wT) c->wT_valid = true;
    // Star tower: a batch carries one domain, so D - 1 of the D slices of every per-domain tensor see a zero gradient
    // and only decay -- TF1's dense Adam still moves them every step (star_kernels.hip).  Inside a call those steps are
    // postponed: k_star_update covers the live slice only and logs the step's alpha, k_star_catchup replays the
    // skipped steps when the call ends (the same arithmetic in the same order: bit-identical; one sweep of the 13
    // slices per call instead of one per step).  Calls of a single step gain nothing and sweep as before.
    const bool star_lazy = c->star && optimizer == MAMDR_OPT_ADAM && n_steps >= 2 && !d_loss_out && !c->star_dense_slices &&
                           c->cfg.n_domain > 1;
    int64_t star_lag = 0;
    for (int64_t s = 0; s < n_steps; ++s) {
        const int64_t row_base = (first_step + s) * batch;
        const int rows = (int)((pass_rows - row_base) < batch ? (pass_rows - row_base) : batch);
        const int rows_pad = (rows + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
        // the reported loss carries l2 * sum(table^2) over EVERY row: bring lagging rows up to date first
        // (only callers that ask for the per-step loss pay this flush; the meta loops do not)
        if (d_loss_out) sync_tables(c);
        float step_alpha = lr;
        if (optimizer == MAMDR_OPT_ADAM) {
            c->adam_t += 1;
            c->b1p = c->b1p * c->cfg.adam_beta1;
            c->b2p = c->b2p * c->cfg.adam_beta2;
            step_alpha = lr * sqrtf(1.0f - c->b2p) / (1.0f - c->b1p);
        }
        if (c->star) {
            // the next step of this call (lazy Adam, launches fused): its rows are resolved in this step's last launch
            EmbRowsArgs nr;
            const bool pre = s + 1 < n_steps && c->tail_fuse && c->cfg.emb_trainable && c->lazy &&
                             optimizer == MAMDR_OPT_ADAM && !c->profile && !d_loss_out;
            if (pre) {
                const int64_t nb = (first_step + s + 1) * batch;
                const int nrows = (int)((pass_rows - nb) < batch ? (pass_rows - nb) : batch);
                const float b1n = c->b1p * c->cfg.adam_beta1, b2n = c->b2p * c->cfg.adam_beta2;
                fill_rows_args(c, *d, d_perm, nb, nrows, (nrows + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS,
                               lr * sqrtf(1.0f - b2n) / (1.0f - b1n), c->adam_t + 1, true, nr);
            }
            EmbStepArgs nea;
            if (pre) {
                fill_emb_args(c, MAMDR_OPT_ADAM, nr.alpha, omb1, omb2, table_two_l2(c), nr.rows, XDIM, nea);
                nea.t_now = (int)c->adam_t + 1;
                nea.t[0].brow = c->urow_alt;
                nea.t[0].map = c->map_u_alt;
                nea.t[1].brow = c->irow_alt;
                nea.t[1].map = c->map_i_alt;
            }
            const int rc = star_train_step(c, *d, domain, d_perm, row_base, rows, optimizer, step_alpha, omb1, omb2,
                                           d_loss_out ? d_loss_out + s : nullptr, pre ? &nr : nullptr, pre ? &nea : nullptr,
                                           star_lazy ? (int)(star_lag & (STAR_ALPHA_CAP - 1)) : -1, star_lazy && s > 0,
                                           star_lazy && s + 1 < n_steps);
            if (rc) return rc;
            c->global_step += 1;
            if (star_lazy) {
                star_lag += 1;
                if (star_lag == STAR_ALPHA_CAP || s + 1 == n_steps) {       // the other slices catch up: log full / call over
                    StarCatchArgs ca;
                    memset(&ca, 0, sizeof(ca));
                    ca.p = c->params + c->table_floats;
                    ca.m = c->adam_m + c->table_floats;
                    ca.v = c->adam_v + c->table_floats;
                    ca.SL = c->SL;
                    ca.n_domain = c->cfg.n_domain;
                    ca.d_live = domain;
                    ca.alpha_log = c->star_alpha;
                    ca.first_idx = 0;
                    ca.n_steps = (int)star_lag;
                    ca.log_mask = STAR_ALPHA_CAP - 1;
                    ca.omb1 = omb1;
                    ca.omb2 = omb2;
                    ca.eps = c->cfg.adam_eps;
                    {
                        Prof p(c, MAMDR_KERNEL_AUX);
                        launch_star_catchup(ca, c->stream);
                    }
                    star_lag = 0;
                }
            }
            continue;
        }

        if (c->cfg.emb_trainable && c->lazy && optimizer == MAMDR_OPT_ADAM)
            emb_pre_step(c, *d, d_perm, row_base, rows, rows_pad, step_alpha, omb1, omb2);
        TowerArgs ta;
        fill_tower_common(c, *d, ta);
        ta.perm = d_perm;
        ta.row_base = row_base;
        ta.rows = rows;
        ta.batch = rows;
        ta.seed = dropout_seed;
        ta.step = c->global_step;
        ta.drop_thresh = drop_thresh;
        ta.keep_scale = keep_scale;
        // the meta pass runs with the Keras learning phase at its default (0): dropout off
        ta.use_dropout = (rate > 0.f && optimizer != MAMDR_OPT_ACCUMULATE) ? 1 : 0;
        ta.acts = c->acts;
        ta.dz = c->dz;
        ta.dlogit = c->dlogit;
        ta.domrow = c->domrow;
        ta.dxe = c->dxe;
        ta.dx_ld = 2 * EMB;
        ta.urow = c->urow;
        ta.irow = c->irow;
        ta.map_u = c->map_u;
        ta.map_i = c->map_i;
        ta.loss_part = c->loss_part;
        ta.fmq = c->fmq;
        if (c->L.lv_count > 0) ta.uw_off = c->L.lv + domain;
#ifdef MAMDR_STAMPS
        ta.stamps = c->stamps ? c->stamps + (c->global_step & 1) * 16384 : nullptr;      // two steps side by side
#endif
        ta.wT = c->wT;
        ta.no_w1l = c->t4_no_w1l;
        float* const dense_m = (optimizer == MAMDR_OPT_ACCUMULATE ? c->accum : c->adam_m) + c->table_floats;
        const bool use4 = may_use4 && (c->tower_tile == 4 || rows_pad <= c->tower4_max_rows);
        if ((c->pnn || c->nfm) && !use4)
            return fail(MAMDR_ENOTBUILT, "pnn / nfm tower: a training step of %d rows needs the four-row tower (MAMDR_TOWER_TILE=16?)", rows);
        if (fused) {
            ta.w0dom_snap = c->w0dom_copy;
            c->dm_cur ^= 1;
            ta.dms = dm_pending;                       // the previous step of this call (snap == null: none)
            ta.dm_hint = domain;
            if (pre) {
                ta.xpre = c->xpre + (size_t)(pre_base + row_base - pre_pos0) * 2 * EMB;
                ta.pdom = c->pdom + (pre_base + row_base - pre_pos0);
                ta.plabel = c->plabel + (pre_base + row_base - pre_pos0);
            }
            ta.dm_live_p = c->params + c->table_floats + c->L.dm;
            ta.dm_live_m = dense_m + c->L.dm;
            ta.dm_live_v = c->adam_v + c->table_floats + c->L.dm;
            ta.dm_snap_out = c->dmsnap[c->dm_cur];
            ta.w2_direct = (w2_direct && s == 0) ? 1 : 0;
        }
        {
            Prof p(c, MAMDR_KERNEL_FWD_BWD);
#ifdef MAMDR_TOWER_TWICE
            // diagnostic build (tools/stamp_tower.py with MAMDR_DIAG_FLAGS=-DMAMDR_TOWER_TWICE): the same tower launch twice in a
            // row -- idempotent (the pending domain-table step is formed from its snapshot, every output is overwritten) -- so that
            // the stamps of the SECOND launch show the kernel with its own code and data still where the first left them
            if (use4) (void)launch_tower4_train(ta, c->stream);
#endif
            const int t4e = use4 ? launch_tower4_train(ta, c->stream) : (launch_tower_train(ta, c->stream), 0);
            if (t4e == T4_E_W2D_LDS) return fail(MAMDR_EHIP, "k_tower4<W2D> was refused its LDS limit (hipFuncSetAttribute)");
            if (t4e) return fail(MAMDR_ESTATE, "w2_direct without the W1-image instance of k_tower4 (step %lld of the call)", (long long)s);
        }
        if (fused) {
            FusedArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.acts = c->acts;
            fa.dz = c->dz;
            fa.dlogit = c->dlogit;
            fa.domrow = c->domrow;
            fa.xa = ta.xpre ? ta.xpre : c->acts;
            fa.xa_ld = ta.xpre ? 2 * EMB : ACT_LD;
            fa.rows_pad = rows_pad;
            fa.rows = rows;
            fa.p = c->params + c->table_floats;
            fa.m = dense_m;
            fa.v = c->adam_v + c->table_floats;
            fa.L = c->L;
            fa.n_domain = c->cfg.n_domain;
            fa.w0dom_snap = c->w0dom_copy;
            fa.dm_snap = c->dmsnap[c->dm_cur];         // p plane: the domain table as this step's forward pass saw it
            fa.pdm = c->pdm;
            fa.wT = (need_wT && optimizer != MAMDR_OPT_ACCUMULATE) ? c->wT : nullptr;
            fa.optimizer = optimizer;
            fa.alpha = step_alpha;
            fa.omb1 = omb1;
            fa.omb2 = omb2;
            fa.eps = c->cfg.adam_eps;
            fa.two_l2 = 2.0f * c->cfg.l2_emb;
            fa.loss_part = c->loss_part;
            fa.n_loss_tiles = use4 ? rows_pad / 4 : rows_pad / TILE_ROWS;
            fa.frozen_sumsq = c->frozen_sumsq;
            fa.l2_emb = c->cfg.l2_emb;
            fa.loss_out = d_loss_out ? d_loss_out + s : nullptr;
#ifdef MAMDR_STAMPS
            fa.stamps = c->stamps ? c->stamps + 65536 + (c->global_step & 1) * 4096 : nullptr;
#endif
            // prefetch riders: the next tower launch's pre-gathered rows -- the next step of this call, or (last step) the
            // first step of the pass the pregather hint says the next call will run
            if (pre && c->fused_pf && ta.xpre) {
                const int64_t here = pre_base + row_base - pre_pos0;         // this step's first row in the pass buffer
                int64_t next_off = -1, next_rows = 0;
                if (s + 1 < n_steps) {
                    next_off = here + batch;
                    next_rows = std::min<int64_t>(batch, pass_rows - (row_base + batch));
                } else if (pre_cached && c->pg_pos + 1 < c->pg.size() && c->pg[c->pg_pos + 1].batch == batch) {
                    const mamdr_ctx::PgEntry& e = c->pg[c->pg_pos + 1];
                    next_off = e.off;
                    next_rows = std::min<int64_t>(batch, e.n);
                }
                if (next_off >= 0 && next_rows > 0) {
                    fa.pf_x = c->xpre + (size_t)next_off * 2 * EMB;
                    fa.pf_dom = c->pdom + next_off;
                    fa.pf_lab = c->plabel + next_off;
                    fa.pf_tiles = (int)((next_rows + 3) / 4);
                    fa.pf_sink = c->loss_part;
                }
            }
            {
                Prof p(c, MAMDR_KERNEL_WGRAD);
                launch_wgrad_adam(fa, c->stream);
            }
            // the domain table's step stays pending: the next step's tower kernel applies it, the last one of the
            // call is materialised below
            dm_pending.snap = c->dmsnap[c->dm_cur];
            dm_pending.pdm = c->pdm;
            dm_pending.n_part = DM_PARTS;
            dm_pending.n_domain = c->cfg.n_domain;
            dm_pending.optimizer = optimizer;
            dm_pending.alpha = step_alpha;
            dm_pending.omb1 = omb1;
            dm_pending.omb2 = omb2;
            dm_pending.eps = c->cfg.adam_eps;
            dm_pending.two_l2 = 2.0f * c->cfg.l2_emb;
            // (an Adam call leaves its last step pending for the next call's first tower / the next sync_tables)
            if (c->dm_finish_each || (s + 1 == n_steps && (optimizer != MAMDR_OPT_ADAM || c->dm_finish_call))) finish_dm(c);
            c->global_step += 1;
            continue;
        }

        if (c->cfg.emb_trainable && d_loss_out) {
            prof_break(c);
            // the regulariser of the reported loss needs the current tables' sums of squares
            launch_sumsq(c->params, (int64_t)c->cfg.n_user * EMB, c->sumsq_partials, c->frozen_sumsq + 0, c->stream);
            launch_sumsq(c->params + (size_t)c->cfg.n_user * EMB, (int64_t)c->cfg.n_item * EMB, c->sumsq_partials,
                         c->frozen_sumsq + 1, c->stream);
            if (c->deepfm) {
                launch_sumsq(c->params + c->lin_user_off, c->cfg.n_user, c->sumsq_partials, c->frozen_sumsq + 2, c->stream);
                launch_sumsq(c->params + c->lin_item_off, c->cfg.n_item, c->sumsq_partials, c->frozen_sumsq + 3, c->stream);
            }
        }
        WgradArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.acts = c->acts;
        wa.dz = c->dz;
        wa.dlogit = c->dlogit;
        wa.domrow = c->domrow;
        wa.fmq = c->fmq;
        wa.ipbuf = c->ipbuf;
        wa.ld_off = c->L.ld;
        wa.ld_count = c->L.ld_count;
        wa.l2_lin = c->cfg.l2_linear;
        wa.lv_off = c->L.lv;
        wa.lv_count = c->L.lv_count;
        wa.uw_d = domain;
        wa.tiles = c->tiles;
        wa.n_tiles = c->n_tiles;
        wa.rows_pad = rows_pad;
        // rows per K-split group (measured: 1024 rows, 8 groups of 128: 31.0 us/step vs 32.0 with 4 of 256;
        // batches of <= 512 rows keep 256-row groups: one or two slabs)
        int rpg = rows_pad <= 512 ? 256 : (rows_pad <= 1024 ? 128 : (rows_pad <= 4096 ? 256 : 512));
        if (c->rpg_override > 0) rpg = c->rpg_override;        // MAMDR_RPG (diagnostic)
        int groups = (rows_pad + rpg - 1) / rpg;
        if (groups > c->max_groups) {
            rpg = ((rows_pad + c->max_groups - 1) / c->max_groups + 7) / 8 * 8;
            groups = (rows_pad + rpg - 1) / rpg;
        }
        wa.n_groups = groups;
        wa.rows_per_group = rpg;
        wa.slabs = c->slabs;
        wa.slab_ld = c->slab_ld;
        wa.w0dom = c->params + c->table_floats + c->L.w0 + (size_t)(2 * EMB) * H1;
        wa.w0dom_copy = c->w0dom_copy;
        wa.dm_copy = c->lin_w0dom ? c->dm_copy : nullptr;
        wa.loss_part = c->loss_part;
        wa.n_loss_tiles = use4 ? rows_pad / 4 : rows_pad / TILE_ROWS;
        wa.rows = rows;
        wa.dense = c->params + c->table_floats;
        wa.dm_count = c->cfg.n_domain * EMB;
        wa.l2_emb = c->cfg.l2_emb;
        wa.frozen_sumsq = c->frozen_sumsq;
        wa.loss_out = d_loss_out ? d_loss_out + s : nullptr;
#ifdef MAMDR_STAMPS
        wa.stamps = c->stamps ? c->stamps + 65536 : nullptr;
#endif
        // lazy table Adam: k_emb_reduce (and DeepFM's k_lin_sweep) only need the tower's outputs and write state
        // no dense kernel touches -> they ride in k_wgrad's / k_update's launches (profiling runs keep them apart
        // for per-kernel times; a reported loss reads the tables between the two and keeps them apart too)
        const bool tail = c->tail_fuse && c->cfg.emb_trainable && c->lazy && optimizer == MAMDR_OPT_ADAM && !d_loss_out &&
                          !c->profile;
        EmbStepArgs tea, nea;
        EmbRowsArgs nr;
        // the next step of this call is known: its row ids / maps are resolved in this step's k_wgrad launch (into
        // the alternate buffers) and its catch-up runs in this step's k_update launch
        const bool pre = tail && s + 1 < n_steps;
        if (tail) {
            fill_emb_args(c, optimizer, step_alpha, omb1, omb2, table_two_l2(c), rows, 2 * EMB, tea);
            tea.flags_done = 1;
            tea.apply_now = 1;
        }
        if (pre) {
            const int64_t nb = (first_step + s + 1) * batch;
            const int nrows = (int)((pass_rows - nb) < batch ? (pass_rows - nb) : batch);
            const float b1n = c->b1p * c->cfg.adam_beta1, b2n = c->b2p * c->cfg.adam_beta2;
            const float alpha_n = lr * sqrtf(1.0f - b2n) / (1.0f - b1n);
            fill_rows_args(c, *d, d_perm, nb, nrows, (nrows + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS, alpha_n, c->adam_t + 1,
                           true, nr);
            fill_emb_args(c, MAMDR_OPT_ADAM, alpha_n, omb1, omb2, table_two_l2(c), nrows, 2 * EMB, nea);
            nea.t_now = (int)c->adam_t + 1;
            nea.t[0].brow = c->urow_alt;
            nea.t[0].map = c->map_u_alt;
            nea.t[1].brow = c->irow_alt;
            nea.t[1].map = c->map_i_alt;
        }
        // frozen tables, another step of this call follows on the 16-row tower: its gather is touched by riders (GatherPf,
        // mamdr_kernels.h) -- in k_wgrad's launch (default since round 5), or (MAMDR_GATHER_PF_IN=update) in k_update's
        GatherPf pf;
        memset(&pf, 0, sizeof(pf));
        if (!tail && c->gather_pf && !c->cfg.emb_trainable && !c->star && s + 1 < n_steps) {
            const int64_t nb = row_base + batch;
            const int nrows = (int)std::min<int64_t>(batch, pass_rows - nb);
            const int npad = (nrows + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
            const bool next4 = may_use4 && (c->tower_tile == 4 || npad <= c->tower4_max_rows);
            if (nrows > 0 && !next4) {
                pf.perm = d_perm;
                pf.uid = d->uid;
                pf.pid = d->pid;
                pf.dom = d->dom;
                pf.label = d->label;
                pf.user_tab = c->user_tab;
                pf.item_tab = c->item_tab;
                pf.row_base = nb;
                pf.n_rows_split = d->n;
                pf.rows = nrows;
                pf.n_user = c->cfg.n_user;
                pf.n_item = c->cfg.n_item;
                pf.n_tiles = npad / TILE_ROWS;
                pf.sink = c->loss_part;
            }
        }
        bool paired = false;        // k_wgrad8: one slab per PAIR of row groups (MAMDR_WGRAD_PAIRS; steps of more than 8 groups)
        {
            Prof p(c, MAMDR_KERNEL_WGRAD);
            if (tail) launch_wgrad_reduce(wa, tea, pre ? &nr : nullptr, nullptr, c->stream);
            else {
                if (c->wgrad_pairs && !c->star && !c->cfg.emb_trainable && groups > 8)
                    paired = launch_wgrad_pairs(wa, c->stream, c->gather_pf_in_wgrad ? &pf : nullptr);
                if (!paired) launch_wgrad(wa, c->stream, c->gather_pf_in_wgrad ? &pf : nullptr);
            }
        }

        UpdateArgs ua;
        memset(&ua, 0, sizeof(ua));
        ua.p = c->params + c->table_floats;
        ua.m = (optimizer == MAMDR_OPT_ACCUMULATE ? c->accum : c->adam_m) + c->table_floats;
        ua.v = c->adam_v + c->table_floats;
        ua.slabs = c->slabs;
        ua.n_groups = paired ? (groups + 1) / 2 : groups;
#ifdef MAMDR_STAMPS
        ua.stamps = c->stamps ? c->stamps + 65536 + 8192 : nullptr;
#endif
        ua.slab_ld = c->slab_ld;
        ua.s_off = c->L.alloc;
        ua.w0dom_copy = c->w0dom_copy;
        ua.dm_copy = c->lin_w0dom ? c->dm_copy : nullptr;
        ua.n_domain = c->cfg.n_domain;
        ua.count4 = c->L.alloc / 4;
        ua.dm_count = c->cfg.n_domain * EMB;
        ua.two_l2 = 2.0f * c->cfg.l2_emb;
        ua.s2_off = c->s2_off;
        ua.ld_off = c->L.ld;
        ua.ld_count = c->L.ld_count;
        ua.two_l2_lin = 2.0f * c->cfg.l2_linear;
        ua.optimizer = optimizer;
        ua.alpha = step_alpha;
        ua.omb1 = omb1;
        ua.omb2 = omb2;
        ua.eps = c->cfg.adam_eps;
        ua.wT = (need_wT && optimizer != MAMDR_OPT_ACCUMULATE) ? c->wT : nullptr;
        ua.w1_off = c->L.w1;
        ua.w2_off = c->L.w2;
        ua.w0_off = c->L.w0;
        ua.w0t = (c->cfg.emb_trainable && !c->nfm) ? 1 : 0;
        ua.no_sdm = c->nfm ? 1 : 0;
        {
            Prof p(c, MAMDR_KERNEL_UPDATE);
            if (tail) {
                launch_update_lin(ua, tea, c->deepfm, pre ? &nea : nullptr, c->stream);
                if (pre) {
                    std::swap(c->urow, c->urow_alt);
                    std::swap(c->irow, c->irow_alt);
                    std::swap(c->map_u, c->map_u_alt);
                    std::swap(c->map_i, c->map_i_alt);
                    c->rows_ready = true;
                    c->catchup_ready = true;
                    c->tables_dirty = true;
                }
            } else {
                // frozen tables, another step of this call follows on the 16-row tower: its gather is touched by riders
                // (GatherPf, mamdr_kernels.h)
                launch_update(ua, c->stream, c->gather_pf_in_wgrad ? nullptr : &pf);
            }
        }
        if (c->cfg.emb_trainable && !tail) emb_post_step(c, optimizer, ua.alpha, omb1, omb2, rows);
        c->global_step += 1;
    }
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}

int mamdr_eval_domain(mamdr_ctx* c, int domain, int split, int32_t batch, float* d_loss_out, uint32_t* d_hist,
                      float* d_pred_out) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (ready(c)) return MAMDR_ESTATE;
    SplitData* d = split_of(c, domain, split);
    if (!d || !d->bound) return fail(MAMDR_ESTATE, "split %d of domain %d is not bound", split, domain);
    if (!d_loss_out || !d_hist) return fail(MAMDR_EINVAL, "null output pointer");
    sync_tables(c);
    if (batch <= 0 || batch % TILE_ROWS != 0) return fail(MAMDR_EINVAL, "eval batch must be a positive multiple of %d", TILE_ROWS);
    if (d->n <= 0) return fail(MAMDR_EINVAL, "split %d of domain %d is empty", split, domain);
    HIP_TRY(hipMemsetAsync(d_hist, 0, 2 * 501 * sizeof(uint32_t), c->stream));
    TowerArgs ta;
    fill_tower_common(c, *d, ta);
    ta.perm = nullptr;
    ta.row_base = 0;
    ta.rows = (int)d->n;
    ta.batch = batch;
    ta.loss_part = c->eval_part;
    ta.hist = d_hist;
    ta.pred_out = d_pred_out;
    if (c->star) {
        // inference: domain `domain`'s moving statistics and merged kernels (partitioned_norm.py:143-165)
        StarPrepArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.blk = c->params + c->table_floats;
        pa.SL = c->SL;
        pa.L = c->L;
        pa.n_domain = c->cfg.n_domain;
        pa.d = domain;
        pa.eff = c->eff;
        pa.pn = c->pn;
        pa.aux = c->aux;
        pa.AL = c->AL;
        pa.train = 0;
        launch_star_prep(pa, c->stream);
        ta.dense = c->eff;
        ta.pn_aff = c->pn;
    }
    {
        Prof p(c, MAMDR_KERNEL_EVAL);
        launch_tower_eval(ta, c->stream);
    }
    if (c->cfg.emb_trainable) {
        launch_sumsq(c->params, (int64_t)c->cfg.n_user * EMB, c->sumsq_partials, c->frozen_sumsq + 0, c->stream);
        launch_sumsq(c->params + (size_t)c->cfg.n_user * EMB, (int64_t)c->cfg.n_item * EMB, c->sumsq_partials,
                     c->frozen_sumsq + 1, c->stream);
        if (c->deepfm) {
            launch_sumsq(c->params + c->lin_user_off, c->cfg.n_user, c->sumsq_partials, c->frozen_sumsq + 2, c->stream);
            launch_sumsq(c->params + c->lin_item_off, c->cfg.n_item, c->sumsq_partials, c->frozen_sumsq + 3, c->stream);
        }
    }
    EvalFinishArgs fa;
    fa.loss_part = c->eval_part;
    fa.n_rows = d->n;
    fa.batch = batch;
    fa.dense = c->params + c->table_floats;
    fa.dm_count = c->star ? 0 : c->cfg.n_domain * EMB;
    fa.l2_emb = c->star ? 0.f : c->cfg.l2_emb;
    fa.frozen_sumsq = c->frozen_sumsq;
    fa.ld_off = c->L.ld;
    fa.ld_count = c->L.ld_count;
    fa.l2_lin = c->cfg.l2_linear;
    fa.loss_out = d_loss_out;
    launch_eval_finish(fa, c->stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}

int mamdr_gather_rows(mamdr_ctx* c, int domain, int split, const int32_t* d_perm, int64_t first_row,
                      int64_t n_rows, float* d_out) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (ready(c)) return MAMDR_ESTATE;
    SplitData* d = split_of(c, domain, split);
    if (!d || !d->bound) return fail(MAMDR_ESTATE, "split %d of domain %d is not bound", split, domain);
    if (!d_out) return fail(MAMDR_EINVAL, "null output pointer");
    if (first_row < 0 || n_rows < 0 || first_row + n_rows > d->n) return fail(MAMDR_EINVAL, "row range outside the split");
    if (n_rows == 0) return MAMDR_OK;
    sync_tables(c);
    TowerArgs ta;
    fill_tower_common(c, *d, ta);
    ta.perm = d_perm;
    ta.row_base = first_row;
    ta.rows = (int)n_rows;
    {
        Prof p(c, MAMDR_KERNEL_GATHER);
        launch_gather(ta, d_out, c->stream);
    }
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}

// ---- outer updates
static int check_vec(const void* p, const char* name) {
    if (!p) return fail(MAMDR_EINVAL, "%s is null", name);
    if ((uintptr_t)p & 15) return fail(MAMDR_EINVAL, "%s is not 16-byte aligned", name);
    return MAMDR_OK;
}
// (an empty vector -- n = 0 -- may be a null pointer: nothing is read or written)
#define CHECK_VEC(p) do { if (n != 0 && check_vec((p), #p)) return MAMDR_EINVAL; } while (0)

int mamdr_interp(float* d_dst, const float* d_a, const float* d_b, float scale, int64_t n, void* stream) {
    CHECK_VEC(d_dst); CHECK_VEC(d_a); CHECK_VEC(d_b);
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    launch_interp(d_dst, d_a, d_b, scale, n, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_moving_average(float* d_unbiased, float* d_biased, const float* d_value, float decay, float denom, int64_t n,
                         void* stream) {
    CHECK_VEC(d_unbiased); CHECK_VEC(d_biased); CHECK_VEC(d_value);
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    if (!(denom > 0.f)) return fail(MAMDR_EINVAL, "moving average: debias denominator %g (local step < 1?)", (double)denom);
    launch_moving_average(d_unbiased, d_biased, d_value, decay, denom, n, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_merge(float* d_dst, const float* d_theta, const float* d_phi, int32_t mode, int64_t n, void* stream) {
    CHECK_VEC(d_dst); CHECK_VEC(d_theta); CHECK_VEC(d_phi);
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    if (mode != MAMDR_MERGE_PLUS && mode != MAMDR_MERGE_TIMES) return fail(MAMDR_EINVAL, "unknown merge mode %d", mode);
    launch_merge(d_dst, d_theta, d_phi, mode, n, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_dr_advance(float* d_phi, float* d_w, float* d_merged, const float* d_theta, float gamma, int32_t mode,
                     int32_t assign_model, int64_t n, void* stream) {
    CHECK_VEC(d_phi); CHECK_VEC(d_w); CHECK_VEC(d_merged); CHECK_VEC(d_theta);
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    if (mode != MAMDR_MERGE_PLUS && mode != MAMDR_MERGE_TIMES) return fail(MAMDR_EINVAL, "unknown merge mode %d", mode);
    launch_dr_advance(d_phi, d_w, d_merged, d_theta, gamma, mode == MAMDR_MERGE_PLUS ? 0 : 1, assign_model != 0, n,
                      (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}

int mamdr_dr_advance_live(mamdr_ctx* c, float* d_phi, float* d_merged, const float* d_theta, float gamma, int32_t mode,
                          int32_t assign_model, int64_t meta_off, int64_t n) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (ready(c)) return MAMDR_ESTATE;
    CHECK_VEC(d_phi); CHECK_VEC(d_merged); CHECK_VEC(d_theta);
    if (n < 0 || meta_off < 0 || (meta_off & 3) || meta_off + n > c->n_params)
        return fail(MAMDR_EINVAL, "range [%lld, %lld) outside the %lld live parameters (or not 16-byte aligned)",
                    (long long)meta_off, (long long)(meta_off + n), (long long)c->n_params);
    if (n == 0) return MAMDR_OK;
    if (mode != MAMDR_MERGE_PLUS && mode != MAMDR_MERGE_TIMES) return fail(MAMDR_EINVAL, "unknown merge mode %d", mode);
    float* const w = c->params + meta_off;
    const int64_t dm0 = c->table_floats + c->L.dm, dmn = (int64_t)c->cfg.n_domain * EMB;
    if (c->dm_pending.snap && c->dm_pending.optimizer == MAMDR_OPT_ADAM && dm0 >= meta_off && dm0 + dmn <= meta_off + n &&
        !c->tables_dirty && !c->dm_finish_call) {
        // the pending domain-table step is materialised by the lanes that own its elements (no k_dm_finish launch)
        prof_break(c);
        launch_dr_advance_dm(d_phi, w, d_merged, d_theta, gamma, mode == MAMDR_MERGE_PLUS ? 0 : 1, assign_model != 0, n,
                             c->dm_pending, c->adam_m + dm0, c->adam_v + dm0, (dm0 - meta_off) >> 2, (int)(dmn >> 2), c->stream);
        c->dm_pending.snap = nullptr;
        c->wT_valid = false;
    } else {
        sync_tables(c);
        prof_break(c);
        launch_dr_advance(d_phi, w, d_merged, d_theta, gamma, mode == MAMDR_MERGE_PLUS ? 0 : 1, assign_model != 0, n, c->stream);
    }
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_sub(float* d_dst, const float* d_a, const float* d_b, int64_t n, void* stream) {
    CHECK_VEC(d_dst); CHECK_VEC(d_a); CHECK_VEC(d_b);
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    launch_sub(d_dst, d_a, d_b, n, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_accumulate(float* d_acc, const float* d_a, const float* d_b, const float* d_shared, float divisor,
                     int64_t n, void* stream) {
    CHECK_VEC(d_acc); CHECK_VEC(d_a); CHECK_VEC(d_b);
    if (d_shared && ((uintptr_t)d_shared & 15)) return fail(MAMDR_EINVAL, "d_shared is not 16-byte aligned");
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    if (divisor == 0.f) return fail(MAMDR_EINVAL, "divisor must be non-zero");
    launch_accumulate(d_acc, d_a, d_b, d_shared, divisor, n, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_apply_accumulated(float* d_dst, float* d_acc, float divisor, float scale, int64_t n, void* stream) {
    CHECK_VEC(d_dst); CHECK_VEC(d_acc);
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    launch_apply_accumulated(d_dst, d_acc, divisor, scale, n, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_adam_apply(float* d_p, float* d_m, float* d_v, const float* d_g, float grad_scale, float lr, float beta1,
                     float beta2, float eps, float beta1_power, float beta2_power, int64_t n, void* stream) {
    CHECK_VEC(d_p); CHECK_VEC(d_m); CHECK_VEC(d_v); CHECK_VEC(d_g);
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    const float alpha = lr * sqrtf(1.0f - beta2_power) / (1.0f - beta1_power);
    launch_adam_apply(d_p, d_m, d_v, d_g, grad_scale, alpha, 1.0f - beta1, 1.0f - beta2, eps, n, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_pcgrad_project(float* d_final, float* d_aux, const int64_t* h_offsets, const int64_t* h_rows,
                         const int32_t* h_cols, int32_t n_seg, void* stream) {
    if (!d_final || !d_aux || !h_offsets || !h_rows || !h_cols) return fail(MAMDR_EINVAL, "null pointer");
    if (n_seg < 0 || n_seg > PCG_MAX_SEG) return fail(MAMDR_EINVAL, "n_seg %d outside [0, %d]", n_seg, PCG_MAX_SEG);
    PcgArgs a;
    memset(&a, 0, sizeof(a));
    a.fin = d_final;
    a.aux = d_aux;
    a.n_seg = n_seg;
    for (int i = 0; i < n_seg; ++i) {
        if (h_offsets[i] < 0 || h_rows[i] < 0 || h_cols[i] <= 0 || h_cols[i] > 4096)
            return fail(MAMDR_EINVAL, "tensor %d: bad offset / rows / cols", i);
        a.off[i] = h_offsets[i];
        a.cols[i] = h_cols[i];
        a.row_start[i + 1] = a.row_start[i] + h_rows[i];
    }
    launch_pcgrad(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MAMDR_OK;
}
int mamdr_copy(float* d_dst, const float* d_src, int64_t n, void* stream) {
    if (!d_dst || !d_src) return fail(MAMDR_EINVAL, "null pointer");
    if (n < 0) return fail(MAMDR_EINVAL, "negative length");
    if (n == 0) return MAMDR_OK;
    if (n == 0) return MAMDR_OK;
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MAMDR_OK;
}

// ---- host helper: tf.data shuffle-buffer order (restated in oracle/rng.py)
int mamdr_shuffle_perm(int64_t n, int64_t buffer_size, uint64_t seed, int32_t* h_out) {
    if (n < 0 || n > 0x7fffffff) return fail(MAMDR_EINVAL, "n out of range");
    if (n == 0) return MAMDR_OK;
    if (!h_out) return fail(MAMDR_EINVAL, "null output");
    if (buffer_size < 1) buffer_size = 1;
    int64_t filled = n < buffer_size ? n : buffer_size;
    std::vector<int32_t> buf((size_t)filled);
    for (int64_t i = 0; i < filled; ++i) buf[(size_t)i] = (int32_t)i;
    int64_t next = filled;
    uint64_t state = seed;
    for (int64_t i = 0; i < n; ++i) {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const uint64_t j = ((z >> 32) * (uint64_t)filled) >> 32;
        h_out[i] = buf[(size_t)j];
        if (next < n) {
            buf[(size_t)j] = (int32_t)next++;
        } else {
            buf[(size_t)j] = buf[(size_t)filled - 1];
            --filled;
        }
    }
    return MAMDR_OK;
}

int mamdr_shuffle_perms(int32_t n_passes, const int64_t* h_n, int64_t buffer_size, const uint64_t* h_seeds,
                        int32_t* h_out) {
    if (n_passes < 0 || (n_passes > 0 && (!h_n || !h_seeds))) return fail(MAMDR_EINVAL, "bad pass list");
    int64_t off = 0;
    for (int32_t k = 0; k < n_passes; ++k) {
        const int rc = mamdr_shuffle_perm(h_n[k], buffer_size, h_seeds[k], h_out ? h_out + off : nullptr);
        if (rc) return rc;
        off += h_n[k];
    }
    return MAMDR_OK;
}

#ifdef MAMDR_STAMPS
// diagnostic build only (tools/stamp_tower.py)
int mamdr_debug_set_stamps(mamdr_ctx* c, unsigned long long* d_stamps) {
    c->stamps = d_stamps;
    return MAMDR_OK;
}
#endif

// ---- profiling
int64_t mamdr_dropout_steps(const mamdr_ctx* c) { return c ? (int64_t)c->global_step : 0; }

int mamdr_step_path(const mamdr_ctx* c, int32_t batch) {
    if (!c || !c->fused) return 0;
    return (batch + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS <= c->fused_max_batch ? 1 : 0;
}

int mamdr_set_tower_tile(mamdr_ctx* c, int32_t rows) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (rows != 0 && rows != 4 && rows != 16) return fail(MAMDR_EINVAL, "tower tile of %d rows (0 = automatic, 4, 16)", rows);
    if (rows == 16 && (c->pnn || c->nfm)) return fail(MAMDR_EINVAL, "the pnn / nfm towers exist as four-row tiles only");
    if (rows != c->tower_tile) {
        c->tower_tile = rows;
        c->pg.clear();          // (passes gathered ahead were laid out for the step path of the old choice)
    }
    return MAMDR_OK;
}
int mamdr_tower_tile(const mamdr_ctx* c, int32_t batch) {
    if (!c || batch <= 0) return MAMDR_EINVAL;
    const int64_t pad = ((int64_t)batch + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
    const bool may_use4 = !c->star && c->tower_tile != 16;
    return may_use4 && (c->tower_tile == 4 || pad <= c->tower4_max_rows) ? 4 : 16;
}
int mamdr_profile_enable(mamdr_ctx* c, int32_t enable) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    c->profile = enable != 0;
    return MAMDR_OK;
}
int mamdr_profile_reset(mamdr_ctx* c) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < MAMDR_KERNEL_COUNT; ++k) {
        for (EventPair& p : c->ev[k]) {        // kept for the next profiled run
            if (p.own_a) c->ev_pool_push(p.a);
            c->ev_pool_push(p.b);
        }
        c->ev[k].clear();
    }
    c->prev_b = nullptr;
    c->chain_ok = false;
    return MAMDR_OK;
}
int mamdr_profile_read(mamdr_ctx* c, int32_t kernel, double* total_ms, int64_t* launches) {
    if (check_ctx(c)) return MAMDR_EINVAL;
    if (kernel < 0 || kernel >= MAMDR_KERNEL_COUNT || !total_ms || !launches) return fail(MAMDR_EINVAL, "bad argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    double sum = 0.0;
    for (EventPair& p : c->ev[kernel]) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)c->ev[kernel].size();
    return MAMDR_OK;
}

}  // extern "C"
