// Star tower: PartitionedNorm + StarFCN around the shared step kernels.
//
// Per training step on domain d (model_zoo/Star/star.py:70-97, partitioned_norm.py:102-203,
// star_fcn.py:105-139):
//   k_star_stats    per-chunk sum x / sum x^2 (double) of the 384 raw input columns of the batch (gathered rows)
//   k_star_prep     first 24 blocks: add the chunks -> batch mean / population variance, update domain d's
//                   zero-debiased moving statistics, emits the PartitionedNorm affine
//                   (scale = gamma_s * gamma_d[d] * rsqrt(var + eps), shift = beta_s + beta_d[d] - mean * scale);
//                   other blocks: effective dense block  K_l = W_shared_l * W_specific_l[d],
//                   b_l = b_shared_l + b_specific_l[d]  in the layout k_tower / k_wgrad read
//   k_tower<train, DXW=384>, k_wgrad      (step_kernels.hip) on the effective block
//   k_star_pnb_*    PartitionedNorm backward through the batch statistics: column sums
//                   s1 = sum dxn, s2 = sum dxn * xhat, then dx = gamma * inv * (dxn - s1/B - xhat * s2/B)
//   (k_emb_reduce / k_emb_sweep with trainable tables)
//   k_star_update   chain rule onto shared / specific tensors + TF1 Adam over EVERY slice (the
//                   specific tensors of the other domains get zero gradient but still decay and move,
//                   as tf.train.AdamOptimizer's sparse rule does).  Inside a mamdr_train_steps call the
//                   other domains' slices are not swept step by step: the launch covers the live slice,
//                   logs the step's alpha, and k_star_catchup replays the skipped zero-gradient steps at
//                   the end of the call (adam_zero_step: the same roundings either way, bit-identical);
//                   it also writes the next step's effective block (eff_out) so that k_star_prep only
//                   merges the batch statistics
// Every reduction runs in a fixed order (no float atomics).
#include <cstring>

#include "emb_bodies.h"
#include "star_bodies.h"

namespace mamdr {

namespace {
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One Adam step of an element whose gradient is exactly zero (the Star slices of the domains a batch does not carry):
// explicit roundings, shared by the per-step sweep and by k_star_catchup's replay -- the same bits either way.
// (square root and reciprocal on the hardware units, v_sqrt_f32 / v_rcp_f32, 1 ulp, as emb_bodies.h's adam_elem: the
// replay of a long call is bound by exactly this sequence -- with the correctly rounded library forms it cost 2.6 us per
// replayed step on Amazon-13 -- and both sites share it)
__device__ __forceinline__ void adam_zero_step(float& p, float& m, float& v, float alpha, float omb1, float omb2, float eps) {
#pragma clang fp contract(off)
    m = m + (0.f - m) * omb1;
    v = v + (0.f - v) * omb2;
    p = p - (m * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
}
// (SGD and accumulate steps leave a zero-gradient element alone)
__device__ __forceinline__ void opt_zero(const OptArgsLite& o, float* p, float* m, float* v, size_t i) {
    if (o.optimizer != 0) return;
    float pp = p[i], mm = m[i], vv = v[i];
    adam_zero_step(pp, mm, vv, o.alpha, o.omb1, o.omb2, o.eps);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
}
// (returns the parameter as the step leaves it)
__device__ __forceinline__ float opt_apply(const OptArgsLite& o, float g, float* p, float* m, float* v, size_t i) {
    float pn = p[i];
    if (o.optimizer == 0) {
        float mm = m[i], vv = v[i];
        mm = mm + (g - mm) * o.omb1;
        vv = vv + (g * g - vv) * o.omb2;
        m[i] = mm;
        v[i] = vv;
        pn = pn - (mm * o.alpha) / (sqrtf(vv) + o.eps);
        p[i] = pn;
    } else if (o.optimizer == 1) {
        pn = pn - g * o.alpha;
        p[i] = pn;
    } else {
        m[i] = m[i] + g;          // accumulate only: m is the meta-gradient accumulator
    }
    return pn;
}
}  // namespace

// ------------------------------------------------------------------ forward statistics
// grid = chunks of STAR_CHUNK batch rows, block = 384 threads (one per input column; a row of the
// batch is three coalesced 512-B table rows)
__global__ __launch_bounds__(XDIM) void k_star_stats(const TowerArgs a, float* part, float* step_counter) {
    __shared__ int rowi[3 * STAR_CHUNK];
    const int c = threadIdx.x, ch = blockIdx.x;
    const int r0 = ch * STAR_CHUNK;
    const int nb = min(STAR_CHUNK, a.rows - r0);
    if (c < STAR_CHUNK) {
        int64_t src = 0;
        if (c < nb) {
            const int64_t pos = a.row_base + r0 + c;
            src = a.perm ? (int64_t)a.perm[pos] : pos;
            if (src < 0) src = 0;
            if (src >= a.n_rows_split) src = a.n_rows_split - 1;
        }
        rowi[c] = clampi(a.uid[src], 0, a.n_user - 1);
        rowi[STAR_CHUNK + c] = clampi(a.pid[src], 0, a.n_item - 1);
        rowi[2 * STAR_CHUNK + c] = clampi(a.dom[src], 0, a.n_domain - 1);
    }
    __syncthreads();
    const int seg = c >> 7, k = c & (EMB - 1);
    const float* base = seg == 0 ? a.user_tab : (seg == 1 ? a.item_tab : a.dense + a.L.dm);
    float x[STAR_CHUNK];
#pragma unroll
    for (int r = 0; r < STAR_CHUNK; ++r) x[r] = base[(size_t)rowi[seg * STAR_CHUNK + r] * EMB + k];
    // sum x and sum x^2 of the chunk's rows in double (round 3): the batch moments k_star_prep forms from them are the
    // correctly rounded ones up to ~1e-16 -- what nn.moments' float64-accurate reference value is -- instead of fp32
    // chunk statistics merged in fp32 (1e-7 apart: enough to send two fp32 runs of this tower down different paths)
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < STAR_CHUNK; ++r) {
        const double xv = (r < nb) ? (double)x[r] : 0.0;
        s1 += xv;
        s2 += xv * xv;
    }
    double* dpart = reinterpret_cast<double*>(part);
    dpart[(size_t)ch * 2 * XDIM + c] = s1;
    dpart[(size_t)ch * 2 * XDIM + XDIM + c] = s2;
    // local step of domain d's moving averages: bumped here so that every block of k_star_prep reads
    // the same, final value
    if (ch == 0 && c == 0) *step_counter += 1.0f;
}
void launch_star_stats(const TowerArgs& a, float* part, float* step_counter, hipStream_t s) {
    const int chunks = (a.rows + STAR_CHUNK - 1) / STAR_CHUNK;
    hipLaunchKernelGGL(k_star_stats, dim3(chunks), dim3(XDIM), 0, s, a, part, step_counter);
}

// ------------------------------------------------------------------ per-step preparation
// (round 3: 16 columns x 32 lanes, 24 blocks -- with 32 x 16 on 12 blocks a thread walked 32 of the 512 chunks of an
// 8,192-row batch in four dependent rounds of loads and the launch took 10.4 us; the lane count is part of the
// summation order, the same for every batch size)
constexpr int PN_COLS = 16;                    // columns per finalize block
constexpr int PN_LANES = 32;                   // chunk lanes per column (PN_COLS * PN_LANES = 512 threads)
constexpr int PN_BLOCKS = XDIM / PN_COLS;      // 24
constexpr int DMF_COLS = 32;                   // k_star_dm_final: columns per block (x STAR_DM_LANES = 512 threads)

__global__ __launch_bounds__(512) void k_star_prep(const StarPrepArgs a) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < PN_BLOCKS) {
        // thread (column cl, lane j) owns chunks j, j + PN_LANES, ...; the lanes of a column are combined in lane
        // order -- a fixed order for every batch size
        __shared__ double sh_s1[PN_LANES][PN_COLS], sh_s2[PN_LANES][PN_COLS];
        const int cl = tid & (PN_COLS - 1), j = tid / PN_COLS;
        const int c = blockIdx.x * PN_COLS + cl;
        float mean = 0.f, var = 1.f;
        // everything lane 0 needs besides the chunks is requested first: the kernel is a chain of cold round trips
        // (the chunk partials were written by the previous kernel on other XCDs), not arithmetic
        const size_t o = (size_t)a.d * XDIM + c;
        float t_step = 0.f, bm = 0.f, bv = 0.f, mov_m = 0.f, mov_v = 1.f;
        float gam_s = 0.f, gam_d = 0.f, bet_s = 0.f, bet_d = 0.f, raw_dm = 0.f;
        if (j == 0) {
            gam_s = a.blk[a.SL.pgs + c];
            gam_d = a.blk[a.SL.pgd + a.d * XDIM + c];
            bet_s = a.blk[a.SL.pbs + c];
            bet_d = a.blk[a.SL.pbd + a.d * XDIM + c];
            if (c >= 2 * EMB) raw_dm = a.blk[a.SL.dm + (size_t)a.d * EMB + (c - 2 * EMB)];
            if (a.train) {
                t_step = a.aux[a.AL.steps + a.d];
                bm = a.aux[a.AL.biased_mean + o];
                bv = a.aux[a.AL.biased_var + o];
            } else {
                mov_m = a.aux[a.AL.mov_mean + o];
                mov_v = a.aux[a.AL.mov_var + o];
            }
        }
        if (a.train) {
            // Round 3: the chunks carry sum x and sum x^2 in DOUBLE (k_star_stats) and are simply added -- no chain of
            // Chan merges (two divisions each, 16 per thread and then 31 more on lane 0: the launch took 9.6 us for
            // 24 blocks), and batch moments accurate to ~1e-16 before their one rounding to fp32.
            const double* dpart = reinterpret_cast<const double*>(a.part);
            double q1[16], q2[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {       // a thread's whole share at 8,192 rows (512 chunks / 32 lanes) in flight
                const int ch = min(j + u * PN_LANES, a.n_chunks - 1);
                q1[u] = dpart[(size_t)ch * 2 * XDIM + c];
                q2[u] = dpart[(size_t)ch * 2 * XDIM + XDIM + c];
            }
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (j + u * PN_LANES < a.n_chunks) {
                    t1 += q1[u];
                    t2 += q2[u];
                }
            }
            for (int ch = j + 16 * PN_LANES; ch < a.n_chunks; ch += PN_LANES) {       // (batches beyond 8,192 rows)
                t1 += dpart[(size_t)ch * 2 * XDIM + c];
                t2 += dpart[(size_t)ch * 2 * XDIM + XDIM + c];
            }
            sh_s1[j][cl] = t1;
            sh_s2[j][cl] = t2;
        }
        __syncthreads();
        if (j != 0) return;
        if (a.train) {
            double S1 = 0.0, S2 = 0.0;
            for (int q = 0; q < PN_LANES; ++q) {        // lane order: fixed for every batch size
                S1 += sh_s1[q][cl];
                S2 += sh_s2[q][cl];
            }
            const double B = (double)a.rows;
            const double mu = S1 / B;
            mean = (float)mu;
            // population variance about the fp32 mean the tower normalises with (nn.moments: mean((x - mean)^2)):
            // E[x^2] - 2 mean E[x] + mean^2, in double
            const double mf = (double)mean;
            const double vd = S2 / B - 2.0 * mf * mu + mf * mf;
            var = (float)(vd > 0.0 ? vd : 0.0);
            // assign_moving_average(zero_debias=True): biased += (value - biased) * (1 - momentum);
            // moving = biased / (1 - momentum^step); the step was bumped by k_star_stats
            const float factor = 1.0f - powf(PN_MOMENTUM, t_step);
            const float omm = 1.0f - PN_MOMENTUM;
            bm += (mean - bm) * omm;
            bv += (var - bv) * omm;
            a.aux[a.AL.biased_mean + o] = bm;
            a.aux[a.AL.biased_var + o] = bv;
            a.aux[a.AL.mov_mean + o] = bm / factor;
            a.aux[a.AL.mov_var + o] = bv / factor;
        } else {
            mean = mov_m;
            var = mov_v;
        }
        const float inv = 1.0f / sqrtf(var + PN_EPS);
        const float gamma = gam_s * gam_d;
        const float beta = bet_s + bet_d;
        const float scale = __fmul_rn(inv, gamma);
        a.pn[c] = scale;
        a.pn[XDIM + c] = __fsub_rn(beta, __fmul_rn(mean, scale));
        a.pn[2 * XDIM + c] = mean;
        a.pn[3 * XDIM + c] = inv;
        a.pn[4 * XDIM + c] = __fmul_rn(gamma, inv);
        // the normalised domain row, rounded exactly as the tower's gather rounds it: every sample of the batch
        // carries it in x[256:384], so dW0[256:384, :] = xdom (x) column sums of dz1 (k_star_update)
        if (c >= 2 * EMB) {
            const float raw = raw_dm;
            a.pn[PN_XDOM_OFF + (c - 2 * EMB)] = __fadd_rn(__fmul_rn(raw, scale), __fsub_rn(beta, __fmul_rn(mean, scale)));
        }
        return;
    }
    // effective dense block, one element per thread (not launched with skip_eff)
    const int e = ((int)blockIdx.x - PN_BLOCKS) * 512 + tid;
    const DenseLayout& L = a.L;
    if (e >= L.count) return;
    float v;
    if (e < L.w0) {
        v = a.blk[a.SL.dm + e];
    } else if (e < L.b0) {
        const int l = e < L.w1 ? 0 : (e < L.w2 ? 1 : 2);
        const int i = e - (l == 0 ? L.w0 : (l == 1 ? L.w1 : L.w2));
        v = a.blk[a.SL.ws[l] + i] * a.blk[a.SL.wd[l] + (size_t)a.d * StarLayout::ksize(l) + i];
    } else if (e < L.wo) {
        const int l = e < L.b1 ? 0 : (e < L.b2 ? 1 : 2);
        const int i = e - (l == 0 ? L.b0 : (l == 1 ? L.b1 : L.b2));
        v = a.blk[a.SL.bs[l] + i] + a.blk[a.SL.bd[l] + a.d * StarLayout::bsize(l) + i];
    } else if (e < L.gb) {
        v = a.blk[a.SL.wo + (e - L.wo)];
    } else {
        v = a.blk[a.SL.gb];
    }
    a.eff[e] = v;
}
void launch_star_prep(const StarPrepArgs& a, hipStream_t s) {
    MAMDR_LAUNCH(k_star_prep, dim3(PN_BLOCKS + (a.skip_eff ? 0 : (a.L.count + 511) / 512)), dim3(512), 0, s, a);
}

// ------------------------------------------------------------------ PartitionedNorm backward
// A chunk = STAR_CHUNK batch rows x 384 columns, one thread per column.  The row indices of the chunk go
// through LDS, then all 16 gradient values and all 16 raw inputs of the thread's column are loaded before the
// first use: the sums below run in row order, but on data that arrived in one round of loads (the rolled
// version paid one dependent HBM round trip per row: 11.6 / 9.5 us per launch at 8192 rows).
struct PnChunk {
    float g[STAR_CHUNK], xh[STAR_CHUNK];
};
__device__ __forceinline__ void star_load_chunk(const StarPnBwdArgs& a, int r0, int nb, int c, int* rowi, PnChunk& k) {
    if (c < 2 * STAR_CHUNK) {
        const int r = c & (STAR_CHUNK - 1);
        const int b = r0 + min(r, nb - 1);
        rowi[c] = c < STAR_CHUNK ? a.urow[b] : a.irow[b];
    }
    __syncthreads();
    const int seg = c >> 7, kk = c & (EMB - 1);
    const float mean = a.pn[2 * XDIM + c], inv = a.pn[3 * XDIM + c];
    float x[STAR_CHUNK];
#pragma unroll
    for (int r = 0; r < STAR_CHUNK; ++r) {
        const int b = r0 + min(r, nb - 1);
        k.g[r] = a.dxe[(size_t)b * XDIM + c];
        const float* row = seg == 0 ? a.user_tab + (size_t)rowi[r] * EMB
                                    : (seg == 1 ? a.item_tab + (size_t)rowi[STAR_CHUNK + r] * EMB : a.dm_row);
        x[r] = row[kk];
    }
#pragma unroll
    for (int r = 0; r < STAR_CHUNK; ++r) k.xh[r] = (x[r] - mean) * inv;
}

__global__ __launch_bounds__(XDIM) void k_star_pnb_partial(const StarPnBwdArgs a) {
    __shared__ int rowi[2 * STAR_CHUNK];
    const int c = threadIdx.x, ch = blockIdx.x;
    const int r0 = ch * STAR_CHUNK;
    const int nb = min(STAR_CHUNK, a.rows - r0);
    PnChunk k;
    star_load_chunk(a, r0, nb, c, rowi, k);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < STAR_CHUNK; ++r) {
        if (r < nb) {
            s1 += k.g[r];
            s2 += k.g[r] * k.xh[r];
        }
    }
    a.part[(size_t)ch * 2 * XDIM + c] = s1;
    a.part[(size_t)ch * 2 * XDIM + XDIM + c] = s2;
}
__global__ __launch_bounds__(512) void k_star_pnb_final(const StarPnBwdArgs a) {
    __shared__ float sh1[PN_LANES][PN_COLS], sh2[PN_LANES][PN_COLS];
    const int cl = threadIdx.x & (PN_COLS - 1), j = threadIdx.x / PN_COLS;
    const int c = blockIdx.x * PN_COLS + cl;
    float s1 = 0.f, s2 = 0.f;
    for (int ch0 = j; ch0 < a.n_chunks; ch0 += 16 * PN_LANES) {     // 16 chunks' partials in flight, summed in order
        float t1[16], t2[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int ch = min(ch0 + u * PN_LANES, a.n_chunks - 1);
            t1[u] = a.part[(size_t)ch * 2 * XDIM + c];
            t2[u] = a.part[(size_t)ch * 2 * XDIM + XDIM + c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (ch0 + u * PN_LANES < a.n_chunks) {
                s1 += t1[u];
                s2 += t2[u];
            }
        }
    }
    sh1[j][cl] = s1;
    sh2[j][cl] = s2;
    __syncthreads();
    if (j != 0) return;
    for (int q = 1; q < PN_LANES; ++q) {
        s1 += sh1[q][cl];
        s2 += sh2[q][cl];
    }
    a.sums[c] = s1;
    a.sums[XDIM + c] = s2;
    {       // the quotients every consumer of d x needs (k_star_pnb_apply forms the same two per thread)
        const float B = (float)a.rows;
        a.means[c] = s1 / B;
        a.means[XDIM + c] = s2 / B;
    }
    // fused form (a.fused): nobody walks the batch again for d x.  The table rows get PartitionedNorm's backward
    // inside k_emb_reduce (EmbStepArgs::pn_sums); the domain row's gradient is the column sum of
    // dx = coef ((dxn - s1 / B) - xhat s2 / B) over the batch, and every sample of the batch carries the SAME domain row
    // (one xhat), so the sum is available here in closed form.  In exact arithmetic it vanishes (the normalised input of
    // a constant column is 0): what is left is rounding residue, as on the per-row path and as in the reference.
    if (a.fused == 1 && c >= 2 * EMB) {
        const float B = (float)a.rows;
        const float m1 = s1 / B, m2 = s2 / B;
        const float xh = (a.dm_row[c - 2 * EMB] - a.pn[2 * XDIM + c]) * a.pn[3 * XDIM + c];
        a.dmsum[c - 2 * EMB] = a.pn[4 * XDIM + c] * ((s1 - B * m1) - xh * (B * m2));
    }
}
__global__ __launch_bounds__(XDIM) void k_star_pnb_apply(const StarPnBwdArgs a) {
    __shared__ int rowi[2 * STAR_CHUNK];
    const int c = threadIdx.x, ch = blockIdx.x;
    const int r0 = ch * STAR_CHUNK;
    const int nb = min(STAR_CHUNK, a.rows - r0);
    const float B = (float)a.rows;
    const float m1 = a.sums[c] / B, m2 = a.sums[XDIM + c] / B;
    const float coef = a.pn[4 * XDIM + c];
    PnChunk k;
    star_load_chunk(a, r0, nb, c, rowi, k);
    float colsum = 0.f;
#pragma unroll
    for (int r = 0; r < STAR_CHUNK; ++r) {
        if (r < nb) {
            const float t = pn_bwd_t(k.g[r], m1, k.xh[r], m2);      // (explicit roundings: star_bodies.h)
            a.dxe[(size_t)(r0 + r) * XDIM + c] = nc_mul(coef, t);
            colsum = __builtin_fmaf(coef, t, colsum);
        }
    }
    if (c >= 2 * EMB) a.dmpart[(size_t)ch * EMB + (c - 2 * EMB)] = colsum;
}
// column sums of dx[:, 256:384]: star_bodies.h (the body also rides in k_wgrad_reduce)
__global__ __launch_bounds__(512) void k_star_dm_final(const StarPnBwdArgs a) {
    __shared__ float sh[STAR_DM_LANES * DMF_COLS];
    star_dm_final_body<DMF_COLS>(a, (int)blockIdx.x, sh);
}
// dm_final = false: the caller lets the domain-row column sums ride in its next launch (k_wgrad_reduce)
// partial_done: the tower's tail already wrote the per-tile sums (TowerArgs::pn_part)
void launch_star_pn_bwd(const StarPnBwdArgs& a, bool dm_final, hipStream_t s, bool partial_done) {
    if (!partial_done) hipLaunchKernelGGL(k_star_pnb_partial, dim3(a.n_chunks), dim3(XDIM), 0, s, a);
    if (a.fused) {      // the column sums are all that is left of PartitionedNorm's backward as a launch
        MAMDR_LAUNCH(k_star_pnb_final, dim3(PN_BLOCKS), dim3(512), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(k_star_pnb_final, dim3(PN_BLOCKS), dim3(512), 0, s, a);
    // (the group's LAST launch carries a profiling scope's stop event: MAMDR_KERNEL_AUX times the group)
    if (dm_final) {
        hipLaunchKernelGGL(k_star_pnb_apply, dim3(a.n_chunks), dim3(XDIM), 0, s, a);
        MAMDR_LAUNCH(k_star_dm_final, dim3(EMB / DMF_COLS), dim3(512), 0, s, a);
    } else {
        MAMDR_LAUNCH(k_star_pnb_apply, dim3(a.n_chunks), dim3(XDIM), 0, s, a);
    }
}

// ------------------------------------------------------------------ chain rule + optimiser
__device__ __forceinline__ float slab_sum(const StarUpdateArgs& u, int off) {
    float g = u.slabs[off];
    for (int s0 = 1; s0 < u.n_groups; s0 += 8) {           // eight slabs in flight, summed in slab order
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = u.slabs[(size_t)min(s0 + k, u.n_groups - 1) * u.slab_ld + off];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + k < u.n_groups) g += t[k];
    }
    return g;
}

// grid.y = domain slice dd.  The thread of the LIVE slice (dd == d) also owns the shared element, so the
// products use the pre-update values of both factors; the other slices only decay (zero gradient).
// (a body: the kernel also hosts k_emb_reduce's workgroups in k_star_update_reduce; (bx, dd) = workgroup coordinates)
__device__ __forceinline__ void star_update_body(const StarUpdateArgs& u, const int bx, const int dd) {
    const int K0 = XDIM * H1, K1 = H1 * H2, K2 = H2 * H3;
    const int n_kernel = K0 + K1 + K2, n_bias = H1 + H2 + H3;
    int e = bx * 256 + threadIdx.x;
    const int d = u.d;
    const bool live = dd == d;
    if (e < n_kernel) {
        const int l = e < K0 ? 0 : (e < K0 + K1 ? 1 : 2);
        const int i = e - (l == 0 ? 0 : (l == 1 ? K0 : K0 + K1));
        const size_t wi = (size_t)u.SL.wd[l] + (size_t)dd * StarLayout::ksize(l) + i;
        if (!live) {
            opt_zero(u.opt, u.p, u.m, u.v, wi);
            return;
        }
        float gK;
        if (u.xdom && l == 0 && i >= 2 * EMB * H1) {
            // rows 256..383 of x are the same normalised domain row for every sample of the batch: the tile-free
            // rank-1 form  dK0[256 + r][c] = xdom[r] * sum_b dz1[b][c]  (the bias gradient of column c)
            const int r = i / H1 - 2 * EMB, c = i - (i / H1) * H1;
            gK = u.xdom[r] * slab_sum(u, u.L.b0 + c);
        } else {
            gK = slab_sum(u, (l == 0 ? u.L.w0 : (l == 1 ? u.L.w1 : u.L.w2)) + i);
        }
        const size_t si = (size_t)u.SL.ws[l] + i;
        const float ws = u.p[si], wd = u.p[wi];
        const float wdn = opt_apply(u.opt, gK * ws, u.p, u.m, u.v, wi);
        const float wsn = opt_apply(u.opt, gK * wd, u.p, u.m, u.v, si);
        if (u.eff_out) u.eff_out[(l == 0 ? u.L.w0 : (l == 1 ? u.L.w1 : u.L.w2)) + i] = wsn * wdn;
        return;
    }
    e -= n_kernel;
    if (e < n_bias) {
        const int l = e < H1 ? 0 : (e < H1 + H2 ? 1 : 2);
        const int i = e - (l == 0 ? 0 : (l == 1 ? H1 : H1 + H2));
        const size_t bi = (size_t)u.SL.bd[l] + (size_t)dd * StarLayout::bsize(l) + i;
        if (!live) {
            opt_zero(u.opt, u.p, u.m, u.v, bi);
            return;
        }
        const float gb = slab_sum(u, (l == 0 ? u.L.b0 : (l == 1 ? u.L.b1 : u.L.b2)) + i);
        const float bdn = opt_apply(u.opt, gb, u.p, u.m, u.v, bi);
        const float bsn = opt_apply(u.opt, gb, u.p, u.m, u.v, (size_t)u.SL.bs[l] + i);
        if (u.eff_out) u.eff_out[(l == 0 ? u.L.b0 : (l == 1 ? u.L.b1 : u.L.b2)) + i] = bsn + bdn;
        return;
    }
    e -= n_bias;
    if (e < XDIM) {               // PartitionedNorm gamma / beta
        const size_t gi = (size_t)u.SL.pgd + dd * XDIM + e, bi = (size_t)u.SL.pbd + dd * XDIM + e;
        if (!live) {
            opt_zero(u.opt, u.p, u.m, u.v, gi);
            opt_zero(u.opt, u.p, u.m, u.v, bi);
            return;
        }
        const float s1 = u.sums[e], s2 = u.sums[XDIM + e];
        const float gs = u.p[u.SL.pgs + e], gd = u.p[gi];
        opt_apply(u.opt, s2 * gs, u.p, u.m, u.v, gi);
        opt_apply(u.opt, s1, u.p, u.m, u.v, bi);
        opt_apply(u.opt, s2 * gd, u.p, u.m, u.v, (size_t)u.SL.pgs + e);
        opt_apply(u.opt, s1, u.p, u.m, u.v, (size_t)u.SL.pbs + e);
        return;
    }
    e -= XDIM;
    if (e < H3 + 1) {             // output unit (owned by the live slice's threads)
        if (!live) return;
        const float g = slab_sum(u, e < H3 ? u.L.wo + e : u.L.gb);
        const float pn = opt_apply(u.opt, g, u.p, u.m, u.v, (size_t)(e < H3 ? u.SL.wo + e : u.SL.gb));
        if (u.eff_out) u.eff_out[e < H3 ? u.L.wo + e : u.L.gb] = pn;
        return;
    }
    e -= H3 + 1;
    if (e < EMB) {                // domain table row dd: only row d is touched (PN's rounding residue)
        if (live) {
            if (u.dm_elsewhere) return;          // (star_dm_step_body, in the same launch)
            const float pn = opt_apply(u.opt, u.dmsum[e], u.p, u.m, u.v, (size_t)u.SL.dm + (size_t)dd * EMB + e);
            if (u.eff_out) u.eff_out[u.L.dm + dd * EMB + e] = pn;
        }
        else opt_zero(u.opt, u.p, u.m, u.v, (size_t)u.SL.dm + (size_t)dd * EMB + e);
    }
}
// The live domain row's step where its gradient is finished (StarUpdateArgs::dm_elsewhere): the column sums of
// d x[:, 256:384] from the per-chunk partials (star_dm_final_body: the same order as k_star_dm_final), then the
// optimiser step of those 16 elements by the threads that hold the sums.  EMB / 16 workgroups of 256 threads.
__device__ __forceinline__ void star_dm_step_body(const StarUpdateArgs& u, const StarPnBwdArgs& sd, int bx, float* sh) {
    star_dm_final_body<16>(sd, bx, sh);
    if (threadIdx.x >= 16) return;
    const int e = bx * 16 + (int)threadIdx.x;
    const float g = sd.dmsum[e];              // (this thread's own store)
    const float pn = opt_apply(u.opt, g, u.p, u.m, u.v, (size_t)u.SL.dm + (size_t)u.d * EMB + e);
    if (u.eff_out) u.eff_out[u.L.dm + u.d * EMB + e] = pn;
}
constexpr int STAR_UPDATE_N = XDIM * H1 + H1 * H2 + H2 * H3 + (H1 + H2 + H3) + XDIM + (H3 + 1) + EMB;
constexpr int STAR_UPDATE_BX = (STAR_UPDATE_N + 255) / 256;
// (only_live: the grid covers slice d alone -- the other slices are replayed by k_star_catchup -- and the step's
// alpha goes into the call's log)
__device__ __forceinline__ void star_log_alpha(const StarUpdateArgs& u, int bx) {
    if (u.alpha_log && bx == 0 && threadIdx.x == 0) u.alpha_log[u.log_idx] = u.opt.alpha;
}
__global__ __launch_bounds__(256) void k_star_update(const StarUpdateArgs u) {
    star_log_alpha(u, (int)blockIdx.x);
    star_update_body(u, (int)blockIdx.x, u.only_live ? u.d : (int)blockIdx.y);
}
// the chain rule + optimiser on the Star block and the NEXT step's k_emb_catchup touch disjoint state: one launch,
// the catch-up workgroups first (their chains are the longer ones).  (This step's k_emb_reduce and the next step's
// k_emb_rows ride in the k_wgrad launch before it, k_wgrad_reduce.)
// (n_dmf workgroups between the two: the live domain row, star_dm_step_body -- the step's PartitionedNorm backward left
// per-chunk partials of its gradient in the launch before this one)
__global__ __launch_bounds__(256) void k_star_update_catchup(const StarUpdateArgs u, const EmbStepArgs nc, const int n_cu,
                                                             const StarPnBwdArgs sd, const int n_dmf) {
    __shared__ float sh[STAR_DM_LANES * 16];
    const int bid = (int)blockIdx.x;
    if (bid < 2 * n_cu) {
        emb_catchup_body(nc, bid % n_cu, bid / n_cu);
        return;
    }
    if (bid < 2 * n_cu + n_dmf) {
        star_dm_step_body(u, sd, bid - 2 * n_cu, sh);
        return;
    }
    const int idx = bid - 2 * n_cu - n_dmf;
    star_log_alpha(u, idx);
    star_update_body(u, idx % STAR_UPDATE_BX, u.only_live ? u.d : idx / STAR_UPDATE_BX);
}

// ---- the skipped zero-gradient steps of every slice but d_live, element by element (the same element map as
// star_update_body's other-slice branches): grid (STAR_UPDATE_BX, n_domain)
__global__ __launch_bounds__(256) void k_star_catchup(const StarCatchArgs a) {
    const int dd = (int)blockIdx.y;
    if (dd == a.d_live) return;
    const int K0 = XDIM * H1, K1 = H1 * H2, K2 = H2 * H3;
    const int n_kernel = K0 + K1 + K2, n_bias = H1 + H2 + H3;
    int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    size_t idx[2];
    int n_idx = 0;
    if (e < n_kernel) {
        const int l = e < K0 ? 0 : (e < K0 + K1 ? 1 : 2);
        const int i = e - (l == 0 ? 0 : (l == 1 ? K0 : K0 + K1));
        idx[n_idx++] = (size_t)a.SL.wd[l] + (size_t)dd * StarLayout::ksize(l) + i;
    } else if ((e -= n_kernel) < n_bias) {
        const int l = e < H1 ? 0 : (e < H1 + H2 ? 1 : 2);
        const int i = e - (l == 0 ? 0 : (l == 1 ? H1 : H1 + H2));
        idx[n_idx++] = (size_t)a.SL.bd[l] + (size_t)dd * StarLayout::bsize(l) + i;
    } else if ((e -= n_bias) < XDIM) {
        idx[n_idx++] = (size_t)a.SL.pgd + dd * XDIM + e;
        idx[n_idx++] = (size_t)a.SL.pbd + dd * XDIM + e;
    } else if ((e -= XDIM) < H3 + 1) {
        return;                   // output unit: shared, stepped with the live slice
    } else if ((e -= H3 + 1) < EMB) {
        idx[n_idx++] = (size_t)a.SL.dm + (size_t)dd * EMB + e;
    } else {
        return;
    }
    for (int k = 0; k < n_idx; ++k) {
        float p = a.p[idx[k]], m = a.m[idx[k]], v = a.v[idx[k]];
        for (int t = 0; t < a.n_steps; ++t)
            adam_zero_step(p, m, v, a.alpha_log[(a.first_idx + t) & a.log_mask], a.omb1, a.omb2, a.eps);
        a.p[idx[k]] = p;
        a.m[idx[k]] = m;
        a.v[idx[k]] = v;
    }
}
void launch_star_update(const StarUpdateArgs& a, hipStream_t s) {
    MAMDR_LAUNCH(k_star_update, dim3(STAR_UPDATE_BX, a.only_live ? 1 : a.n_domain), dim3(256), 0, s, a);
}
void launch_star_update_catchup(const StarUpdateArgs& a, const EmbStepArgs& nc, const StarPnBwdArgs* dm, hipStream_t s) {
    const int n_cu = (nc.rows + 7) / 8;
    StarPnBwdArgs sd;
    memset(&sd, 0, sizeof(sd));
    if (dm) sd = *dm;
    const int n_dmf = dm ? EMB / 16 : 0;
    MAMDR_LAUNCH(k_star_update_catchup, dim3(2 * n_cu + n_dmf + STAR_UPDATE_BX * (a.only_live ? 1 : a.n_domain)), dim3(256), 0, s,
                 a, nc, n_cu, sd, n_dmf);
}
void launch_star_catchup(const StarCatchArgs& a, hipStream_t s) {
    if (a.n_steps <= 0 || a.n_domain <= 1) return;
    MAMDR_LAUNCH(k_star_catchup, dim3(STAR_UPDATE_BX, a.n_domain), dim3(256), 0, s, a);
}

}  // namespace mamdr
