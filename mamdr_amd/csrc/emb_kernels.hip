// Trainable user / item embedding tables (train.emb_trainable, model_zoo/DeepCTR/deepctr.py:95-99).
//
// TF1 semantics: the table gradient is the scatter-add of the batch's row
// gradients PLUS the dense regulariser term 2*l2*W on every row, and tf.train.AdamOptimizer
// updates every element every step.  Per step, for both tables at once:
//   (k_tower)     map[row] = min batch position touching the row          (integer atomicMin: exact)
//   k_emb_flag    positions that are not their row's representative flag the representative
//   k_emb_reduce  a flagged representative gathers the positions of its row in ascending order
//                 (parallel compare + prefix sum) and sums their gradients in that order
//                 -> bitwise reproducible, no float atomics
//   k_emb_sweep   HBM-bound pass over both tables: g = 2 l2 p (+ gbuf[map[row]]), Adam / SGD /
//                 accumulate; resets map[row] on the way
//   k_lin_sweep   DeepFM only: the 1-d linear tables of the same two features, same rule
#include "mamdr_kernels.h"

namespace mamdr {

__global__ __launch_bounds__(256) void k_emb_fill(int32_t* map, int64_t n, int32_t value) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) map[i] = value;
}
void launch_emb_map_init(int32_t* map, int64_t n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_emb_fill, dim3(blocks), dim3(256), 0, s, map, n, EMB_UNTOUCHED);
}

// TF1 ApplyAdam on one element with an explicit rounding sequence: the dense sweep, the lazy catch-up,
// the touched-row update and the flush all go through it, so a row that is advanced lazily ends up with
// exactly the bits the per-step dense sweep would have produced.
__device__ __forceinline__ void adam_elem(float g, float& p, float& m, float& v, float alpha, float omb1, float omb2,
                                          float eps) {
    m = __fmaf_rn(__fsub_rn(g, m), omb1, m);
    v = __fmaf_rn(__fsub_rn(__fmul_rn(g, g), v), omb2, v);
    // sqrt and reciprocal on the hardware units (v_sqrt_f32 / v_rcp_f32, 1 ulp): the replay of long gaps is
    // bound by exactly this sequence, and both paths share it, so they still agree bit for bit
    p = __fsub_rn(p, __fmul_rn(__fmul_rn(m, alpha), __builtin_amdgcn_rcpf(__fadd_rn(__builtin_amdgcn_sqrtf(v), eps))));
}
// lazy mode: the representative workgroup of k_emb_reduce applies Adam step t_now to its row right away
// (element t of row r; the row was brought to t_now - 1 by k_emb_catchup before the gather)
__device__ __forceinline__ void emb_apply_row(const EmbStepArgs& a, const EmbTable& T, bool second, int r, int t,
                                              float gsum) {
    const size_t e = ((size_t)(second ? a.t[0].n_rows : 0) + r) * EMB + t;
    float p = a.p[e], m = a.m[e], v = a.v[e];
    adam_elem(__fadd_rn(__fmul_rn(a.opt.two_l2, p), gsum), p, m, v, a.opt.alpha, a.opt.omb1, a.opt.omb2, a.opt.eps);
    a.p[e] = p;
    a.m[e] = m;
    a.v[e] = v;
    if (t == 0) {
        T.last[r] = a.t_now;
        if (!T.lin_p) T.map[r] = EMB_UNTOUCHED;      // DeepFM: k_lin_sweep still needs the map and resets it
    }
}

__global__ __launch_bounds__(256) void k_emb_flag(const EmbStepArgs a) {
    const EmbTable& T = a.t[blockIdx.y];
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.rows) return;
    const int r = T.brow[b];
    if (r < 0) return;
    const int rep = T.map[r];
    if (rep != b) T.hasdup[rep] = 1;           // same value from every writer
}

// One 128-thread workgroup per (batch position, table); only representatives do work, and only those
// whose row occurs more than once in the batch scan the batch (O(B/128) per such row).
// Thread t checks the contiguous positions [t*per, (t+1)*per); an exclusive prefix sum of the match
// counts (wave shuffles + one LDS hand-over between the two waves) places the matches in ascending
// order in `list`, then thread c sums column c of the listed positions in that order.
__global__ __launch_bounds__(EMB) void k_emb_reduce(const EmbStepArgs a) {
    extern __shared__ int32_t list[];          // [rows] worst case: every position hits the same row
    __shared__ int wave0_total, n_list;
    const EmbTable& T = a.t[blockIdx.y];
    const int b = blockIdx.x, t = threadIdx.x;
    const int r = T.brow[b];
    if (r < 0 || T.map[r] != b) return;        // uniform over the workgroup
    if (!T.hasdup[b]) {                        // the common case: no other position shares the row
        const float g1 = a.dxe[(size_t)b * a.dx_ld + T.dx_off + t];
        if (T.lin_p && t == 0) T.glin[b] = a.dlogit[b];
        if (a.apply_now) emb_apply_row(a, T, blockIdx.y != 0, r, t, g1);
        else T.gbuf[(size_t)b * EMB + t] = g1;
        return;
    }
    const int per = (a.rows + EMB - 1) / EMB;
    const int p0 = max(t * per, b + 1), p1 = min((t + 1) * per, a.rows);   // b is the minimum position
    int cnt = 0;
    for (int i = p0; i < p1; ++i) cnt += (T.brow[i] == r) ? 1 : 0;
    __syncthreads();                           // every thread has read the flag
    if (t == 0) T.hasdup[b] = 0;               // reset for the next step
    int incl = cnt;
    const int lane = t & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (t == 63) wave0_total = incl;
    __syncthreads();
    int off = incl - cnt + (t >= 64 ? wave0_total : 0);
    for (int i = p0; i < p1; ++i)
        if (T.brow[i] == r) list[off++] = i;
    if (t == EMB - 1) n_list = off;            // the last thread's end offset is the total
    __syncthreads();
    const int n = n_list;
    // ascending-position sum; the loads of 8 listed rows are issued together, the adds stay in order
    // (a popular item can occur hundreds of times in a batch of 8192: the chain of dependent loads was
    // the critical path of the kernel)
    float acc = a.dxe[(size_t)b * a.dx_ld + T.dx_off + t];
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = a.dxe[(size_t)list[k + u] * a.dx_ld + T.dx_off + t];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v8[u];
    }
    for (; k < n; ++k) acc += a.dxe[(size_t)list[k] * a.dx_ld + T.dx_off + t];
    if (T.lin_p && t == 0) {                   // DeepFM: the 1-d linear table's row gradient = sum of dlogit
        float accl = a.dlogit[b];
        for (int k = 0; k < n; ++k) accl += a.dlogit[list[k]];
        T.glin[b] = accl;
    }
    if (a.apply_now) emb_apply_row(a, T, blockIdx.y != 0, r, t, acc);
    else T.gbuf[(size_t)b * EMB + t] = acc;
}

__device__ __forceinline__ void opt_step(const OptArgsLite& o, float g, float& p, float& m, float& v) {
    if (o.optimizer == 0) {
        adam_elem(g, p, m, v, o.alpha, o.omb1, o.omb2, o.eps);
    } else {
        p = p - g * o.alpha;
    }
}

// 16 B per lane over [user table | item table] (contiguous in the flat vector); a wave covers two
// 512-B rows, so map[] is read once per half wave.  Without the DeepFM pass, the lane that owns a
// row's first float4 resets the row's map entry (the other 31 lanes of the row read it in the same
// instruction, no other thread ever does).  OPT: 0 Adam, 1 SGD, 2 accumulate (a.m = accumulator);
// all streaming loads of an element are issued before the (rare) dependent gbuf fetch.
template <int OPT>
__global__ __launch_bounds__(256) void k_emb_sweep(const EmbStepArgs a) {
    const int64_t n0 = a.t[0].n_rows;
    const int64_t n4 = (n0 + a.t[1].n_rows) * (EMB / 4);
    const bool reset = a.t[0].lin_p == nullptr;
    for (int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x; e4 < n4; e4 += (int64_t)gridDim.x * 256) {
        const int64_t row = e4 >> 5;
        const int c4 = (int)(e4 & 31);
        // (explicit selects: indexing a.t[] with a per-lane value would spill the struct to scratch)
        const bool second = row >= n0;
        int32_t* map = second ? a.t[1].map : a.t[0].map;
        const float* gbuf = second ? a.t[1].gbuf : a.t[0].gbuf;
        const int64_t lrow = second ? row - n0 : row;
        const int rep = map[lrow];
        if (OPT == 0 && c4 == 0) {               // lazy bookkeeping: the row is current at this step
            int32_t* last = second ? a.t[1].last : a.t[0].last;
            if (last) last[lrow] = a.t_now;
        }
        // streaming access: every element is touched exactly once per step, so the loads / stores are
        // marked non-temporal (measured 364 -> 346 us per Amazon-6 sweep)
#define SW_LD(ptr) __builtin_nontemporal_load(ptr)
#define SW_ST(val, ptr) __builtin_nontemporal_store(val, ptr)
        f32x4 p = SW_LD(reinterpret_cast<const f32x4*>(a.p) + e4);
        f32x4 m = (f32x4){0.f, 0.f, 0.f, 0.f}, v = m;
        if (OPT != 1) m = SW_LD(reinterpret_cast<const f32x4*>(a.m) + e4);
        if (OPT == 0) v = SW_LD(reinterpret_cast<const f32x4*>(a.v) + e4);
        f32x4 g;       // explicit roundings: the lazy path (k_emb_catchup / emb_apply_row / k_emb_flush) must reproduce these bits
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = __fmul_rn(a.opt.two_l2, p[k]);
        if (rep != EMB_UNTOUCHED) {
            const f32x4 gb = reinterpret_cast<const f32x4*>(gbuf + (size_t)rep * EMB)[c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = __fadd_rn(g[k], gb[k]);
            if (reset && c4 == 0) map[lrow] = EMB_UNTOUCHED;
        }
        if (OPT == 2) {
            SW_ST(m + g, reinterpret_cast<f32x4*>(a.m) + e4);
            continue;
        }
        if (OPT == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                adam_elem(g[k], pk, mk, vk, a.opt.alpha, a.opt.omb1, a.opt.omb2, a.opt.eps);
                p[k] = pk; m[k] = mk; v[k] = vk;
            }
            SW_ST(m, reinterpret_cast<f32x4*>(a.m) + e4);
            SW_ST(v, reinterpret_cast<f32x4*>(a.v) + e4);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = p[k] - g[k] * a.opt.alpha;
        }
        SW_ST(p, reinterpret_cast<f32x4*>(a.p) + e4);
    }
}

// DeepFM: 1-d linear tables of the two features (one scalar per table row) + the map reset
__global__ __launch_bounds__(256) void k_lin_sweep(const EmbStepArgs a) {
    const int64_t n0 = a.t[0].n_rows, n_all = n0 + a.t[1].n_rows;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n_all; row += (int64_t)gridDim.x * 256) {
        const bool second = row >= n0;
        int32_t* map = second ? a.t[1].map : a.t[0].map;
        const float* glin = second ? a.t[1].glin : a.t[0].glin;
        float* lin_p = second ? a.t[1].lin_p : a.t[0].lin_p;
        float* lin_m = second ? a.t[1].lin_m : a.t[0].lin_m;
        float* lin_v = second ? a.t[1].lin_v : a.t[0].lin_v;
        const int64_t lrow = second ? row - n0 : row;
        float p = lin_p[lrow];
        float g = a.two_l2_lin * p;
        const int rep = map[lrow];
        if (rep != EMB_UNTOUCHED) {
            g += glin[rep];
            map[lrow] = EMB_UNTOUCHED;
        }
        if (a.opt.optimizer == 2) {
            lin_m[lrow] += g;
            continue;
        }
        if (a.opt.optimizer == 0) {
            float m = lin_m[lrow], v = lin_v[lrow];
            opt_step(a.opt, g, p, m, v);
            lin_m[lrow] = m;
            lin_v[lrow] = v;
        } else {
            p = p - g * a.opt.alpha;
        }
        lin_p[lrow] = p;
    }
}

// ------------------------------------------------------------------ lazy dense Adam
// TF1's Adam moves EVERY table row every step (regulariser gradient 2 l2 p, decaying moments), which the
// dense sweep above pays for with 24 B/element of HBM traffic per step.  The same per-element recurrence
// can be run late: a row that the batch does not touch is left alone, and when it is needed (touched by a
// batch, or the weights are read / replaced: mamdr_sync_tables) its missed steps last[row]+1 .. t are replayed
// in registers with the logged per-step alpha.  Same arithmetic, same order per element -> same bits; the
// HBM traffic of a step drops from the whole table to the rows of the batch.
__global__ __launch_bounds__(256) void k_emb_rows(const EmbRowsArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b == 0) a.alpha_log[a.log_idx] = a.alpha;
    if (b >= a.rows_pad) return;
    if (b >= a.rows) {
        a.urow[b] = -1;
        a.irow[b] = -1;
        return;
    }
    const int64_t pos = a.row_base + b;
    int64_t src = a.perm ? (int64_t)a.perm[pos] : pos;
    if (src < 0) src = 0;
    if (src >= a.n_rows_split) src = a.n_rows_split - 1;
    int u = a.uid[src], i = a.pid[src];
    u = u < 0 ? 0 : (u > a.n_user - 1 ? a.n_user - 1 : u);
    i = i < 0 ? 0 : (i > a.n_item - 1 ? a.n_item - 1 : i);
    a.urow[b] = u;
    a.irow[b] = i;
    atomicMin(a.map_u + u, b);
    atomicMin(a.map_i + i, b);
}
void launch_emb_rows(const EmbRowsArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_emb_rows, dim3((a.rows_pad + 255) / 256), dim3(256), 0, s, a);
}

// 8 batch positions per workgroup, 32 lanes x float4 per 512-B row.  A representative replays its row's
// missed steps up to t_now - 1 (before the gather reads the row); every other position flags its
// representative as "row occurs more than once" for k_emb_reduce.
__global__ __launch_bounds__(256) void k_emb_catchup(const EmbStepArgs a) {
    const int b = blockIdx.x * 8 + (threadIdx.x >> 5), c4 = threadIdx.x & 31;
    if (b >= a.rows) return;
    const bool second = blockIdx.y != 0;
    const EmbTable& T = a.t[blockIdx.y];
    const int r = T.brow[b];
    if (r < 0) return;
    const int rep = T.map[r];
    if (rep != b) {
        if (c4 == 0) T.hasdup[rep] = 1;        // same value from every writer
        return;
    }
    const int last = T.last[r];
    const int t_prev = a.t_now - 1;
    if (last >= t_prev) return;
    const size_t e4 = ((size_t)(second ? a.t[0].n_rows : 0) + r) * (EMB / 4) + c4;
    f32x4 p = reinterpret_cast<const f32x4*>(a.p)[e4];
    f32x4 m = reinterpret_cast<const f32x4*>(a.m)[e4];
    f32x4 v = reinterpret_cast<const f32x4*>(a.v)[e4];
    for (int t = last + 1; t <= t_prev; ++t) {
        const float alpha = a.alpha_log[t & a.log_mask];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = m[k], vk = v[k];
            adam_elem(__fmul_rn(a.opt.two_l2, pk), pk, mk, vk, alpha, a.opt.omb1, a.opt.omb2, a.opt.eps);
            p[k] = pk; m[k] = mk; v[k] = vk;
        }
    }
    reinterpret_cast<f32x4*>(a.p)[e4] = p;
    reinterpret_cast<f32x4*>(a.m)[e4] = m;
    reinterpret_cast<f32x4*>(a.v)[e4] = v;
    if (c4 == 0) T.last[r] = t_prev;           // (the 32 lanes of the row read last[] in one instruction above)
}
void launch_emb_catchup(const EmbStepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_emb_catchup, dim3((a.rows + 7) / 8, 2), dim3(256), 0, s, a);
}

// every row of both tables -> current at t_now (one float4 per thread; rows already current cost one
// 4-byte read per half wave)
__global__ __launch_bounds__(256) void k_emb_flush(const EmbStepArgs a) {
    const int64_t n0 = a.t[0].n_rows;
    const int64_t n4 = (n0 + a.t[1].n_rows) * (EMB / 4);
    const int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e4 >= n4) return;
    const int64_t row = e4 >> 5;
    const bool second = row >= n0;
    int32_t* lastp = (second ? a.t[1].last : a.t[0].last) + (second ? row - n0 : row);
    const int last = *lastp;
    if (last >= a.t_now) return;
    f32x4 p = reinterpret_cast<const f32x4*>(a.p)[e4];
    f32x4 m = reinterpret_cast<const f32x4*>(a.m)[e4];
    f32x4 v = reinterpret_cast<const f32x4*>(a.v)[e4];
    for (int t = last + 1; t <= a.t_now; ++t) {
        const float alpha = a.alpha_log[t & a.log_mask];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = m[k], vk = v[k];
            adam_elem(__fmul_rn(a.opt.two_l2, pk), pk, mk, vk, alpha, a.opt.omb1, a.opt.omb2, a.opt.eps);
            p[k] = pk; m[k] = mk; v[k] = vk;
        }
    }
    reinterpret_cast<f32x4*>(a.p)[e4] = p;
    reinterpret_cast<f32x4*>(a.m)[e4] = m;
    reinterpret_cast<f32x4*>(a.v)[e4] = v;
    if ((e4 & 31) == 0) *lastp = a.t_now;
}
void launch_emb_flush(const EmbStepArgs& a, hipStream_t s) {
    const int64_t n4 = (a.t[0].n_rows + a.t[1].n_rows) * (EMB / 4);
    hipLaunchKernelGGL(k_emb_flush, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a);
}

void launch_emb_reduce(const EmbStepArgs& a, hipStream_t s) {
    if (!a.flags_done) hipLaunchKernelGGL(k_emb_flag, dim3((a.rows + 255) / 256, 2), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_emb_reduce, dim3(a.rows, 2), dim3(EMB), (size_t)a.rows * sizeof(int32_t), s, a);
}
void launch_emb_sweep(const EmbStepArgs& a, hipStream_t s) {
    const int64_t n_all = a.t[0].n_rows + a.t[1].n_rows;
    const int64_t n4 = n_all * (EMB / 4);
    int64_t blocks = (n4 + 255) / 256;
    // one float4 per thread: measured on Amazon-6 (79 M elements) 349 us uncapped vs 415 us with a
    // 4096-workgroup grid-stride loop; the cap only guards the 32-bit grid dimension
    if (blocks > 0x7fffffff) blocks = 0x7fffffff;
    if (a.opt.optimizer == 0) hipLaunchKernelGGL(k_emb_sweep<0>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else if (a.opt.optimizer == 1) hipLaunchKernelGGL(k_emb_sweep<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_emb_sweep<2>, dim3((unsigned)blocks), dim3(256), 0, s, a);
}
void launch_lin_sweep(const EmbStepArgs& a, hipStream_t s) {
    const int64_t n_all = a.t[0].n_rows + a.t[1].n_rows;
    int64_t blocks = (n_all + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_lin_sweep, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

}  // namespace mamdr
