// Trainable user / item embedding tables (train.emb_trainable, model_zoo/DeepCTR/deepctr.py:95-99).
//
// TF1 semantics: the table gradient is the scatter-add of the batch's row
// gradients PLUS the dense regulariser term 2*l2*W on every row, and tf.train.AdamOptimizer
// updates every element every step.  Four kernels per table and step:
//   k_emb_mark    map[row] = min batch position touching the row          (integer atomics: exact)
//   k_emb_reduce  the representative position sums the gradients of all positions of its row in
//                 ascending order -> bitwise reproducible, no float atomics
//   k_emb_sweep   HBM-bound pass over the whole table: g = 2 l2 p (+ gbuf[map[row]]), Adam / SGD
//   k_emb_unmark  map[row] = untouched
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

__global__ __launch_bounds__(256) void k_emb_mark(const EmbStepArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.rows) return;
    const int r = a.brow[b];
    if (r >= 0) atomicMin(a.map + r, b);
}

// one 128-thread workgroup per batch position; only representatives do work
__global__ __launch_bounds__(EMB) void k_emb_reduce(const EmbStepArgs a) {
    extern __shared__ int32_t rows_lds[];
    const int b = blockIdx.x, c = threadIdx.x;
    const int r = a.brow[b];
    if (r < 0 || a.map[r] != b) return;       // uniform over the workgroup
    for (int i = c; i < a.rows; i += EMB) rows_lds[i] = a.brow[i];
    __syncthreads();
    float acc = 0.f;
    for (int i = b; i < a.rows; ++i)           // positions before b cannot share the row (b is the minimum)
        if (rows_lds[i] == r) acc += a.dxe[(size_t)i * (2 * EMB) + a.dx_off + c];
    a.gbuf[(size_t)b * EMB + c] = acc;
    if (a.lin_p && c == 0) {                   // DeepFM: the 1-d linear table's row gradient = sum of dlogit
        float accl = 0.f;
        for (int i = b; i < a.rows; ++i)
            if (rows_lds[i] == r) accl += a.dlogit[i];
        a.glin[b] = accl;
    }
}

__device__ __forceinline__ void opt_step(const OptArgsLite& o, float g, float& p, float& m, float& v) {
    if (o.optimizer == 0) {
        m = m + (g - m) * o.omb1;
        v = v + (g * g - v) * o.omb2;
        p = p - (m * o.alpha) / (sqrtf(v) + o.eps);
    } else {
        p = p - g * o.alpha;
    }
}

// 16 B per lane; a wave covers two 512-B rows, so map[] is read once per half wave
__global__ __launch_bounds__(256) void k_emb_sweep(const EmbStepArgs a) {
    const int64_t n4 = a.n_rows * (EMB / 4);
    for (int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x; e4 < n4; e4 += (int64_t)gridDim.x * 256) {
        const int64_t row = e4 >> 5;
        const int c4 = (int)(e4 & 31);
        f32x4 p = reinterpret_cast<const f32x4*>(a.p)[e4];
        f32x4 g = a.opt.two_l2 * p;
        const int rep = a.map[row];
        if (rep != EMB_UNTOUCHED) g += reinterpret_cast<const f32x4*>(a.gbuf + (size_t)rep * EMB)[c4];
        if (a.opt.optimizer == 2) {          // accumulate only (MAML meta pass): a.m is the accumulator
            f32x4 acc = reinterpret_cast<const f32x4*>(a.m)[e4];
            acc += g;
            reinterpret_cast<f32x4*>(a.m)[e4] = acc;
            continue;
        }
        if (a.opt.optimizer == 0) {
            f32x4 m = reinterpret_cast<const f32x4*>(a.m)[e4];
            f32x4 v = reinterpret_cast<const f32x4*>(a.v)[e4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                opt_step(a.opt, g[k], pk, mk, vk);
                p[k] = pk; m[k] = mk; v[k] = vk;
            }
            reinterpret_cast<f32x4*>(a.m)[e4] = m;
            reinterpret_cast<f32x4*>(a.v)[e4] = v;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = p[k] - g[k] * a.opt.alpha;
        }
        reinterpret_cast<f32x4*>(a.p)[e4] = p;
    }
    if (a.lin_p == nullptr) return;
    // DeepFM 1-d linear table of the same feature: one scalar per table row, same update rule
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < a.n_rows; row += (int64_t)gridDim.x * 256) {
        float p = a.lin_p[row];
        float g = a.two_l2_lin * p;
        const int rep = a.map[row];
        if (rep != EMB_UNTOUCHED) g += a.glin[rep];
        if (a.opt.optimizer == 2) {
            a.lin_m[row] += g;
            continue;
        }
        if (a.opt.optimizer == 0) {
            float m = a.lin_m[row], v = a.lin_v[row];
            opt_step(a.opt, g, p, m, v);
            a.lin_m[row] = m;
            a.lin_v[row] = v;
        } else {
            p = p - g * a.opt.alpha;
        }
        a.lin_p[row] = p;
    }
}

__global__ __launch_bounds__(256) void k_emb_unmark(const EmbStepArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.rows) return;
    const int r = a.brow[b];
    if (r >= 0) a.map[r] = EMB_UNTOUCHED;
}

void launch_emb_scatter(const EmbStepArgs& a, hipStream_t s) {
    const int pb = (a.rows + 255) / 256;
    hipLaunchKernelGGL(k_emb_mark, dim3(pb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_emb_reduce, dim3(a.rows), dim3(EMB), (size_t)a.rows * sizeof(int32_t), s, a);
}
void launch_emb_sweep(const EmbStepArgs& a, hipStream_t s) {
    const int64_t n4 = a.n_rows * (EMB / 4);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;       // grid-stride beyond 16 workgroups per CU
    hipLaunchKernelGGL(k_emb_sweep, dim3((unsigned)blocks), dim3(256), 0, s, a);
}
void launch_emb_unmark(const EmbStepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_emb_unmark, dim3((a.rows + 255) / 256), dim3(256), 0, s, a);
}

}  // namespace mamdr
