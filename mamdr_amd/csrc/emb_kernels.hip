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
#include "emb_bodies.h"

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

__global__ __launch_bounds__(256) void k_emb_flag(const EmbStepArgs a) {
    const EmbTable& T = a.t[blockIdx.y];
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.rows) return;
    const int r = T.brow[b];
    if (r < 0) return;
    const int rep = T.map[r];
    if (rep != b) T.hasdup[rep] = 1;           // same value from every writer
}

__global__ __launch_bounds__(256) void k_emb_reduce(const EmbStepArgs a) {
    __shared__ uint16_t list_all[4][RED_CAP];
    emb_reduce_body(a, (int)blockIdx.x, (int)blockIdx.y, list_all);
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
__global__ __launch_bounds__(256) void k_lin_sweep(const EmbStepArgs a) { lin_sweep_body(a, (int)blockIdx.x, (int)gridDim.x); }

// ------------------------------------------------------------------ lazy dense Adam
// TF1's Adam moves EVERY table row every step (regulariser gradient 2 l2 p, decaying moments), which the
// dense sweep above pays for with 24 B/element of HBM traffic per step.  The same per-element recurrence
// can be run late: a row that the batch does not touch is left alone, and when it is needed (touched by a
// batch, or the weights are read / replaced: mamdr_sync_tables) its missed steps last[row]+1 .. t are replayed
// in registers with the logged per-step alpha.  Same arithmetic, same order per element -> same bits; the
// HBM traffic of a step drops from the whole table to the rows of the batch.
__global__ __launch_bounds__(256) void k_emb_rows(const EmbRowsArgs a) { emb_rows_body(a, (int)blockIdx.x); }
void launch_emb_rows(const EmbRowsArgs& a, hipStream_t s) {
    MAMDR_LAUNCH(k_emb_rows, dim3((a.rows_pad + 255) / 256), dim3(256), 0, s, a);
}

// 8 batch positions per workgroup, 32 lanes x float4 per 512-B row.  A representative replays its row's
// missed steps up to t_now - 1 (before the gather reads the row); every other position flags its
// representative as "row occurs more than once" for k_emb_reduce.
__global__ __launch_bounds__(256) void k_emb_catchup(const EmbStepArgs a) { emb_catchup_body(a, (int)blockIdx.x, (int)blockIdx.y); }
void launch_emb_catchup(const EmbStepArgs& a, hipStream_t s) {
    MAMDR_LAUNCH(k_emb_catchup, dim3((a.rows + 7) / 8, 2), dim3(256), 0, s, a);
}

// every row of both tables -> current at t_now (one float4 per thread; rows already current cost one
// 4-byte read per half wave)
// Zero-gradient replay (Star): the alphas of the last 256 steps are staged in LDS once per workgroup (the loop read one
// from global memory per step and element group: a dependent VMEM round trip inside an ALU-bound loop).
constexpr int FLUSH_ALPHAS = 256;
template <bool ZERO_G>
__device__ __forceinline__ void flush_replay(const EmbStepArgs& a, const float* alphas, int last, f32x4& p, f32x4& m, f32x4& v) {
    for (int t = last + 1; t <= a.t_now; ++t) {
        const int back = a.t_now - t;
        // same-box A/B (profiles/r04n_flush_alpha_ab.txt): the LDS table makes the zero-gradient loop (8 operations per
        // element-step) 5.3 % faster and the regularised one (13 operations) 4.6 % slower -> only the former takes it
        const float alpha = (ZERO_G && back < FLUSH_ALPHAS) ? alphas[FLUSH_ALPHAS - 1 - back] : a.alpha_log[t & a.log_mask];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = m[k], vk = v[k];
            if (ZERO_G) adam_elem_zero(pk, mk, vk, alpha, a.opt.omb1, a.opt.omb2, a.opt.eps);
            else adam_elem(__fmul_rn(a.opt.two_l2, pk), pk, mk, vk, alpha, a.opt.omb1, a.opt.omb2, a.opt.eps);
            p[k] = pk; m[k] = mk; v[k] = vk;
        }
    }
}
__global__ __launch_bounds__(256) void k_emb_flush(const EmbStepArgs a) {
    __shared__ float alphas[FLUSH_ALPHAS];
    if (a.opt.two_l2 == 0.f) {
        alphas[threadIdx.x] = a.alpha_log[(a.t_now - (FLUSH_ALPHAS - 1) + (int)threadIdx.x) & a.log_mask];
        __syncthreads();
    }
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
    if (a.opt.two_l2 == 0.f) flush_replay<true>(a, alphas, last, p, m, v);      // Star tower: zero-gradient steps
    else flush_replay<false>(a, alphas, last, p, m, v);
    reinterpret_cast<f32x4*>(a.p)[e4] = p;
    reinterpret_cast<f32x4*>(a.m)[e4] = m;
    reinterpret_cast<f32x4*>(a.v)[e4] = v;
    if ((e4 & 31) == 0) *lastp = a.t_now;
}
void launch_emb_flush(const EmbStepArgs& a, hipStream_t s) {
    const int64_t n4 = (a.t[0].n_rows + a.t[1].n_rows) * (EMB / 4);
    MAMDR_LAUNCH(k_emb_flush, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a);
}

void launch_emb_reduce(const EmbStepArgs& a, hipStream_t s) {
    if (!a.flags_done) hipLaunchKernelGGL(k_emb_flag, dim3((a.rows + 255) / 256, 2), dim3(256), 0, s, a);
    MAMDR_LAUNCH(k_emb_reduce, dim3((a.rows + 7) / 8, 2), dim3(256), 0, s, a);
}
void launch_emb_sweep(const EmbStepArgs& a, hipStream_t s) {
    const int64_t n_all = a.t[0].n_rows + a.t[1].n_rows;
    const int64_t n4 = n_all * (EMB / 4);
    int64_t blocks = (n4 + 255) / 256;
    // one float4 per thread: measured on Amazon-6 (79 M elements) 349 us uncapped vs 415 us with a
    // 4096-workgroup grid-stride loop; the cap only guards the 32-bit grid dimension
    if (blocks > 0x7fffffff) blocks = 0x7fffffff;
    if (a.opt.optimizer == 0) MAMDR_LAUNCH(k_emb_sweep<0>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else if (a.opt.optimizer == 1) MAMDR_LAUNCH(k_emb_sweep<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else MAMDR_LAUNCH(k_emb_sweep<2>, dim3((unsigned)blocks), dim3(256), 0, s, a);
}
void launch_lin_sweep(const EmbStepArgs& a, hipStream_t s) {
    const int64_t n_all = a.t[0].n_rows + a.t[1].n_rows;
    int64_t blocks = (n_all + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    MAMDR_LAUNCH(k_lin_sweep, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

}  // namespace mamdr
