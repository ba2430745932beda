// Device bodies of the table-update kernels that can also ride in another kernel's launch (horizontal fusion:
// k_wgrad_reduce, k_update_lin in step_kernels.hip; k_star_update_reduce in star_kernels.hip).  Every function
// that rounds carries `#pragma clang fp contract(off)`: emb_kernels.hip is compiled with -ffp-contract=off and
// the lazy table Adam must produce the same bits wherever a body is instantiated.
#pragma once
#include "mamdr_kernels.h"

namespace mamdr {

// separately rounded multiply / add / subtract in ANY translation unit: HIP's __fmul_rn & co are plain operators
// inside header functions, which a TU compiled with -ffp-contract=fast marks contractable -- and the backend then
// fuses them into fmas after inlining.  These carry the pragma themselves.
__device__ __forceinline__ float nc_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float nc_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float nc_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// TF1 ApplyAdam on one element with an explicit rounding sequence: the dense sweep, the lazy catch-up,
// the touched-row update and the flush all go through it, so a row that is advanced lazily ends up with
// exactly the bits the per-step dense sweep would have produced.
__device__ __forceinline__ void adam_elem(float g, float& p, float& m, float& v, float alpha, float omb1, float omb2,
                                          float eps) {
#pragma clang fp contract(off)
    m = __builtin_fmaf(nc_sub(g, m), omb1, m);
    v = __builtin_fmaf(nc_sub(nc_mul(g, g), v), omb2, v);
    // sqrt and reciprocal on the hardware units (v_sqrt_f32 / v_rcp_f32, 1 ulp): the replay of long gaps is
    // bound by exactly this sequence, and both paths share it, so they still agree bit for bit
    p = nc_sub(p, nc_mul(nc_mul(m, alpha), __builtin_amdgcn_rcpf(nc_add(__builtin_amdgcn_sqrtf(v), eps))));
}
// the same step for a gradient that is exactly zero (a row no batch touches in a tower WITHOUT the table regulariser:
// the Star tower, two_l2 = 0).  adam_elem(0 * p, ...) evaluates g - m = -m and g g - v = -v exactly, so dropping those
// operations leaves every bit in place (signed zeros included: the fma adds m / v back) -- 8 instead of 13 operations in
// the replay loops, which are bound by exactly this sequence.
__device__ __forceinline__ void adam_elem_zero(float& p, float& m, float& v, float alpha, float omb1, float omb2, float eps) {
#pragma clang fp contract(off)
    m = __builtin_fmaf(-m, omb1, m);
    v = __builtin_fmaf(-v, omb2, v);
    p = nc_sub(p, nc_mul(nc_mul(m, alpha), __builtin_amdgcn_rcpf(nc_add(__builtin_amdgcn_sqrtf(v), eps))));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));

// lazy mode: the representative wave of k_emb_reduce applies Adam step t_now to its row right away
// (columns 2*lane, 2*lane+1 of row r; the row was brought to t_now - 1 by k_emb_catchup before the gather)
__device__ __forceinline__ void emb_apply_row(const EmbStepArgs& a, const EmbTable& T, bool second, int r, int lane,
                                              f32x2 gsum) {
#pragma clang fp contract(off)
    const size_t e = ((size_t)(second ? a.t[0].n_rows : 0) + r) * EMB + 2 * lane;
    f32x2 p = *reinterpret_cast<const f32x2*>(a.p + e);
    f32x2 m = *reinterpret_cast<const f32x2*>(a.m + e);
    f32x2 v = *reinterpret_cast<const f32x2*>(a.v + e);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        adam_elem(nc_add(nc_mul(a.opt.two_l2, pk), gsum[k]), pk, mk, vk, a.opt.alpha, a.opt.omb1, a.opt.omb2,
                  a.opt.eps);
        p[k] = pk; m[k] = mk; v[k] = vk;
    }
    *reinterpret_cast<f32x2*>(a.p + e) = p;
    *reinterpret_cast<f32x2*>(a.m + e) = m;
    *reinterpret_cast<f32x2*>(a.v + e) = v;
    if (lane == 0) {
        T.last[r] = a.t_now;
        if (!T.lin_p) T.map[r] = EMB_UNTOUCHED;      // DeepFM: k_lin_sweep still needs the map and resets it
    }
}

// the same step on columns 4*c4 .. 4*c4+3 (the no-duplicate path: half a wave per row)
__device__ __forceinline__ void emb_apply_row4(const EmbStepArgs& a, const EmbTable& T, bool second, int r, int c4,
                                               f32x4 gsum) {
#pragma clang fp contract(off)
    const size_t e = ((size_t)(second ? a.t[0].n_rows : 0) + r) * EMB + 4 * c4;
    f32x4 p = *reinterpret_cast<const f32x4*>(a.p + e);
    f32x4 m = *reinterpret_cast<const f32x4*>(a.m + e);
    f32x4 v = *reinterpret_cast<const f32x4*>(a.v + e);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        adam_elem(nc_add(nc_mul(a.opt.two_l2, pk), gsum[k]), pk, mk, vk, a.opt.alpha, a.opt.omb1, a.opt.omb2,
                  a.opt.eps);
        p[k] = pk; m[k] = mk; v[k] = vk;
    }
    *reinterpret_cast<f32x4*>(a.p + e) = p;
    *reinterpret_cast<f32x4*>(a.m + e) = m;
    *reinterpret_cast<f32x4*>(a.v + e) = v;
    if (c4 == 0) {
        T.last[r] = a.t_now;
        if (!T.lin_p) T.map[r] = EMB_UNTOUCHED;
    }
}

// Star: one element of PartitionedNorm's backward (see EmbStepArgs::pn_sums); c = column of the 384-wide input
__device__ __forceinline__ float pn_fix1(const EmbStepArgs& a, int c, float g, float x) {
#pragma clang fp contract(off)
    const float m1 = a.pn_means[c], m2 = a.pn_means[XDIM + c];
    const float xh = nc_mul(nc_sub(x, a.pn[2 * XDIM + c]), a.pn[3 * XDIM + c]);
    // (t in ONE fused multiply-add: k_star_pnb_apply's rounding, star_bodies.h pn_bwd_t)
    return nc_mul(a.pn[4 * XDIM + c], __builtin_fmaf(-m2, xh, nc_sub(g, m1)));
}

__device__ __forceinline__ void opt_step(const OptArgsLite& o, float g, float& p, float& m, float& v) {
#pragma clang fp contract(off)
    if (o.optimizer == 0) {
        adam_elem(g, p, m, v, o.alpha, o.omb1, o.omb2, o.eps);
    } else {
        p = p - g * o.alpha;
    }
}

// 8 batch positions per 256-thread workgroup (one grid row per table); only representatives (the minimum
// position of their row) do work.
// Phase A, the common case -- no other position shares the row: half a wave per position, 32 lanes x float4 =
// the 512-B row, gradient row + p / m / v in one round of loads, Adam, store (the layout of k_emb_catchup).
// Phase B -- rows that occur more than once (flagged by k_emb_catchup / k_emb_flag; Amazon-13, 8192 rows:
// ~900 of 16 K positions, lists of 1-3): the representative's wave looks at the whole batch, 512 positions per
// step (two 16-B loads of brow per lane, four steps in flight; brow is 32 KB and cache-resident).  Steps
// without a match -- all but a few -- cost a compare and one ballot.  In a step with matches the matching
// lanes append their positions to the wave's LDS list at (count + matches in lower lanes + own earlier
// matches): ascending positions without a prefix sum.  The listed positions' gradient rows are then added IN
// THAT ORDER (bitwise reproducible, no float atomics), 8 row loads in flight, lane l owning columns 2l, 2l+1;
// the list is drained whenever the next step might overflow it.
// (History: one 128-thread workgroup per position with a worst-case 32 KB list: 48 us of a 300 us Amazon-13
// step, occupancy-bound on the trivial workgroups; 16-bit lists and two scanning waves: 30 us; one wave per
// two positions scanning 64 positions per step, from global memory or from an LDS copy: 48-56 us -- the
// serial chain of 128 dependent steps per flagged row was the whole kernel.)
constexpr int RED_STEP = 512;                  // positions per scan step (8 per lane)
constexpr int RED_CAP = 2 * RED_STEP;          // list entries per wave
// (bx, table) = the workgroup's coordinates; list_all = [4 waves][RED_CAP] uint16 of LDS (positions fit 16 bits:
// max_batch <= 16384).  A body, so that the kernel can also ride in another kernel's launch (k_wgrad_reduce).
__device__ __forceinline__ void emb_reduce_body(const EmbStepArgs& a, int bx, int table, uint16_t (*list_all)[RED_CAP]) {
#pragma clang fp contract(off)
    const EmbTable& T = a.t[table];
    const bool second = table != 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = lane >> 5, c4 = lane & 31;
    const int b = bx * 8 + w * 2 + h;  // this half-wave's position
    int r = -1;
    bool dupf = false;
    if (b < a.rows) {
        const int rb = T.brow[b];
        if (rb >= 0 && T.map[rb] == b) {
            r = rb;
            dupf = T.hasdup[b] != 0;
        }
    }
    // ---- phase A
    if (r >= 0 && !dupf) {
        f32x4 g4 = *reinterpret_cast<const f32x4*>(a.dxe + (size_t)b * a.dx_ld + T.dx_off + 4 * c4);
        if (a.pn_sums) {
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(a.p + ((size_t)(second ? a.t[0].n_rows : 0) + r) * EMB + 4 * c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) g4[k] = pn_fix1(a, T.dx_off + 4 * c4 + k, g4[k], x4[k]);
        }
        if (T.lin_p && c4 == 0) T.glin[b] = a.dlogit[b];
        if (a.apply_now) emb_apply_row4(a, T, second, r, c4, g4);
        else *reinterpret_cast<f32x4*>(T.gbuf + (size_t)b * EMB + 4 * c4) = g4;
    }
    const unsigned long long dmask = __ballot(dupf);       // bits 0..31: position h = 0, bits 32..63: h = 1
    if (dmask == 0ull) return;                              // wave-uniform
    // ---- phase B
    uint16_t* L = list_all[w];
    const int n_steps = (a.rows + RED_STEP - 1) / RED_STEP;
    const int rows8 = (a.rows + 7) / 8;                     // brow is allocated (and -1 padded) to a multiple of 16
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        if (((dmask >> (32 * s)) & 1ull) == 0ull) continue;            // wave-uniform
        const int bs = bx * 8 + w * 2 + s;
        const int rs = __shfl(r, 32 * s);
        const float* gcol = a.dxe + T.dx_off + 2 * lane;
        // Star: every position's gradient goes through PartitionedNorm's backward before it is added (the row's own
        // values: the same for all positions of the row)
        f32x2 xrow = (f32x2){0.f, 0.f};
        if (a.pn_sums) xrow = *reinterpret_cast<const f32x2*>(a.p + ((size_t)(second ? a.t[0].n_rows : 0) + rs) * EMB + 2 * lane);
        auto fix = [&](f32x2 gq) {
            if (a.pn_sums) {
                gq[0] = pn_fix1(a, T.dx_off + 2 * lane, gq[0], xrow[0]);
                gq[1] = pn_fix1(a, T.dx_off + 2 * lane + 1, gq[1], xrow[1]);
            }
            return gq;
        };
        f32x2 acc = fix(*reinterpret_cast<const f32x2*>(gcol + (size_t)bs * a.dx_ld));
        float accl = T.lin_p ? a.dlogit[bs] : 0.f;         // DeepFM: the 1-d linear table's row gradient
        int cnt = 0;                           // wave-uniform
        auto drain = [&]() {
            __builtin_amdgcn_wave_barrier();
            int k = 0;
            for (; k + 8 <= cnt; k += 8) {
                f32x2 v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v8[u] = *reinterpret_cast<const f32x2*>(gcol + (size_t)L[k + u] * a.dx_ld);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += fix(v8[u]);
            }
            for (; k < cnt; ++k) acc += fix(*reinterpret_cast<const f32x2*>(gcol + (size_t)L[k] * a.dx_ld));
            if (T.lin_p)
                for (int q = 0; q < cnt; ++q) accl += a.dlogit[L[q]];
            __builtin_amdgcn_wave_barrier();
            cnt = 0;
        };
        // lane's eight positions of step `st` (brow is -1 beyond the batch: never a match)
        auto load8 = [&](int st, int4& lo, int4& hi) {
            const int g8 = min(st * 64 + lane, rows8 - 1);  // (clamped lanes re-read the last group: masked in step)
            lo = reinterpret_cast<const int4*>(T.brow)[2 * g8];
            hi = reinterpret_cast<const int4*>(T.brow)[2 * g8 + 1];
        };
        auto step = [&](int st, const int4 lo, const int4 hi) {
            const int base = st * RED_STEP + lane * 8;
            unsigned m8 = (lo.x == rs ? 1u : 0u) | (lo.y == rs ? 2u : 0u) | (lo.z == rs ? 4u : 0u) | (lo.w == rs ? 8u : 0u) |
                          (hi.x == rs ? 16u : 0u) | (hi.y == rs ? 32u : 0u) | (hi.z == rs ? 64u : 0u) | (hi.w == rs ? 128u : 0u);
            // positions <= bs (only bs itself can carry the id) and the -1 padding are not matches
            if (base <= bs) m8 &= base + 7 <= bs ? 0u : ~((2u << (bs - base)) - 1u);
            if (st * 64 + lane >= rows8) m8 = 0u;
            if (__ballot(m8 != 0) == 0ull) return;          // the usual case
            if (cnt + RED_STEP > RED_CAP) drain();
            int lower = 0, total = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned long long mj = __ballot((m8 >> j) & 1u);
                lower += __popcll(mj & below);
                total += __popcll(mj);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if ((m8 >> j) & 1u) L[cnt + lower + __popc(m8 & ((1u << j) - 1u))] = (uint16_t)(base + j);
            cnt += total;
        };
        // bs is the row's minimum position: steps before its step hold no match
        int st = bs / RED_STEP;
        for (; st + 4 <= n_steps; st += 4) {
            int4 lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8(st + u, lo[u], hi[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) step(st + u, lo[u], hi[u]);
        }
        for (; st < n_steps; ++st) {
            int4 lo, hi;
            load8(st, lo, hi);
            step(st, lo, hi);
        }
        drain();
        if (lane == 0) {
            T.hasdup[bs] = 0;                  // reset for the next step (read above by every lane that needs it)
            if (T.lin_p) T.glin[bs] = accl;
        }
        if (a.apply_now) emb_apply_row(a, T, second, rs, lane, acc);
        else *reinterpret_cast<f32x2*>(T.gbuf + (size_t)bs * EMB + 2 * lane) = acc;
    }
}

// DeepFM: 1-d linear tables of the two features (one scalar per table row) + the map reset
// (bx of nb workgroups; a body for the same reason as emb_reduce_body: k_update_lin)
__device__ __forceinline__ void lin_sweep_body(const EmbStepArgs& a, int bx, int nb) {
#pragma clang fp contract(off)
    const int64_t n0 = a.t[0].n_rows, n_all = n0 + a.t[1].n_rows;
    for (int64_t row = (int64_t)bx * 256 + threadIdx.x; row < n_all; row += (int64_t)nb * 256) {
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


// Row ids + representatives of a batch (see k_emb_rows in emb_kernels.hip).  A body: the NEXT step's rows can be
// resolved in the current step's last launch (k_update_lin), into the other half of a double buffer.
__device__ __forceinline__ void emb_rows_body(const EmbRowsArgs& a, int bx) {
    const int b = bx * 256 + threadIdx.x;
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

// 8 batch positions per workgroup, 32 lanes x float4 per 512-B row (see k_emb_catchup in emb_kernels.hip)
// (a body: the NEXT step's catch-up can ride in the current step's k_update launch)
__device__ __forceinline__ void emb_catchup_body(const EmbStepArgs& a, int bx, int table) {
#pragma clang fp contract(off)
    const int b = bx * 8 + (threadIdx.x >> 5), c4 = threadIdx.x & 31;
    if (b >= a.rows) return;
    const bool second = table != 0;
    const EmbTable& T = a.t[table];
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
    if (a.opt.two_l2 == 0.f) {                 // no regulariser: the missed steps have a zero gradient (same bits, fewer operations)
        for (int t = last + 1; t <= t_prev; ++t) {
            const float alpha = a.alpha_log[t & a.log_mask];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                adam_elem_zero(pk, mk, vk, alpha, a.opt.omb1, a.opt.omb2, a.opt.eps);
                p[k] = pk; m[k] = mk; v[k] = vk;
            }
        }
    } else {
        for (int t = last + 1; t <= t_prev; ++t) {
            const float alpha = a.alpha_log[t & a.log_mask];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                adam_elem(nc_mul(a.opt.two_l2, pk), pk, mk, vk, alpha, a.opt.omb1, a.opt.omb2, a.opt.eps);
                p[k] = pk; m[k] = mk; v[k] = vk;
            }
        }
    }
    reinterpret_cast<f32x4*>(a.p)[e4] = p;
    reinterpret_cast<f32x4*>(a.m)[e4] = m;
    reinterpret_cast<f32x4*>(a.v)[e4] = v;
    if (c4 == 0) T.last[r] = t_prev;           // (the 32 lanes of the row read last[] in one instruction above)
}

}  // namespace mamdr
