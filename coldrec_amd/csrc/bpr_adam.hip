// BPR-MF training step for gfx950: gather + BPR/L2 forward + backward scatter-add, dense Adam.
//
// Replaces, per optimiser step of the reference trainers (model/MF.py:19-27,
// model/LightGCN.py:21-28):
//   gather      rec_user_emb[user_idx], rec_item_emb[pos_idx], rec_item_emb[neg_idx]   (aten::index)
//   bpr_loss    util/utils.py:25-29      mean(-log(1e-5 + sigmoid(u.p - u.n)))
//   l2_reg_loss util/utils.py:44-48      reg * (|U_B|_F + |P_B|_F + |N_B|_F) / B
//   backward    autograd: index_put_(accumulate=True) into dense table gradients
//   Adam        torch.optim.Adam defaults (torch/optim/adam.py _single_tensor_adam): EVERY element
//               of both tables moves every step
//
// All of it is HBM/L2-bound row traffic: rows are read with 16 B per lane (one row of d=128 is
// one 512-B transaction of a half-wave), reductions are wave shuffles, and the three Frobenius
// norms -- a grid-wide dependency of the backward pass -- are carried as per-block partials that
// every backward block re-reduces in the same fixed order (deterministic, no atomics, no memset).
// Row gradients are accumulated with hardware fp32 atomics (global_atomic_add_f32).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "crh_common.h"

namespace {

constexpr int BPR_THREADS = 256;
constexpr int BPR_MAX_BLOCKS = 1024;
// entries of one gradient row above which a whole block sums it: a lane group walks a row's list sequentially (one
// round trip per 4 entries), so the longest "light" row sets the kernel's critical path -- 8 keeps it at two trips
#ifndef BPR_HEAVY_N
#define BPR_HEAVY_N 8
#endif
constexpr int BPR_HEAVY = BPR_HEAVY_N;

struct BprArgs {
    const float* tu;   // user-side table      (rows x d)
    const float* tp;   // positive-item table
    const float* tn;   // negative-item table (== tp for MF / LightGCN)
    const int32_t* iu; // row ids per triple, NULL = identity (already gathered tensors)
    const int32_t* ip;
    const int32_t* in_;
    int64_t B;
    int d;
    float reg;
    float* gu;         // dense gradient tables (accumulated into), may be NULL for forward only
    float* gp;
    float* gn;
    float* xbuf;       // [B] pos-neg score difference
    float* partials;   // [nblocks][4]: sum u^2, sum p^2, sum n^2, sum loss
    float* loss_out;   // [2]: bpr, l2   (may be NULL)
    int nblocks_fwd;
    const int32_t* plan;   // reverse index of the batch (crh_bpr_plan_build_host) or NULL
    const float* totals;   // [4] batch sums (u^2, p^2, n^2, loss) supplied by the caller -- the data-parallel
                           // split, where they were all-reduced over the ranks -- or NULL: reduce `partials`
    float* sums_out;       // [4] forward-only entry point: where bpr_sums_kernel leaves the batch sums
    int64_t B_global;      // batch size the means / norms refer to (== B unless data-parallel)
    int own_mod, own_rem;  // deterministic backward over the plan's row slots w with w % own_mod == own_rem only (1, 0 = all):
                           // the row-ownership split of the data-parallel touched-rows step (crh_bpr_bwd_owned_f32)
};

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    // deterministic: wave butterfly then fixed-order sum of the 4 wave results
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <int G>
__global__ __launch_bounds__(BPR_THREADS) void bpr_fwd_kernel(BprArgs a) {
    __shared__ float red[4];
    const int lig = threadIdx.x % G;                 // lane in group
    const int64_t gid = (int64_t)blockIdx.x * (BPR_THREADS / G) + threadIdx.x / G;
    const int64_t gstride = (int64_t)gridDim.x * (BPR_THREADS / G);
    const int nvec = a.d >> 2;
    float su = 0.f, sp = 0.f, sn = 0.f, sl = 0.f;
    for (int64_t b = gid; b < a.B; b += gstride) {
        const int64_t ru = a.iu ? a.iu[b] : b, rp = a.ip ? a.ip[b] : b, rn = a.in_ ? a.in_[b] : b;
        const f32x4* pu = reinterpret_cast<const f32x4*>(a.tu + ru * a.d);
        const f32x4* pp = reinterpret_cast<const f32x4*>(a.tp + rp * a.d);
        const f32x4* pn = reinterpret_cast<const f32x4*>(a.tn + rn * a.d);
        float dp = 0.f, dn = 0.f;
        for (int c = lig; c < nvec; c += G) {
            const f32x4 u = pu[c], p = pp[c], n = pn[c];
            dp += u.x * p.x + u.y * p.y + u.z * p.z + u.w * p.w;
            dn += u.x * n.x + u.y * n.y + u.z * n.z + u.w * n.w;
            su += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
            sp += p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w;
            sn += n.x * n.x + n.y * n.y + n.z * n.z + n.w * n.w;
        }
        dp = group_sum<G>(dp);
        dn = group_sum<G>(dn);
        if (lig == 0) {
            const float x = dp - dn;                          // pos_score - neg_score
            a.xbuf[b] = x;
            const float sig = 1.0f / (1.0f + expf(-x));
            sl += -logf(1e-5f + sig);                       // 10e-6 literal of the reference
        }
    }
    su = block_sum(su, red);
    sp = block_sum(sp, red);
    sn = block_sum(sn, red);
    sl = block_sum(sl, red);
    if (threadIdx.x == 0) {
        float* o = a.partials + (size_t)blockIdx.x * 4;
        o[0] = su; o[1] = sp; o[2] = sn; o[3] = sl;
    }
}

// Batch sums for the backward pass: every block reduces the forward partials in the same order ->
// identical, deterministic totals; in the data-parallel split the caller supplies them (all-reduced).
__device__ __forceinline__ void batch_totals(const BprArgs& a, float* tot, float* red) {
    if (a.totals) {
        if (threadIdx.x < 4) tot[threadIdx.x] = a.totals[threadIdx.x];
    } else {
        // the four sums in ONE pass over the partials (16-byte loads) and one reduction: component by component the
        // arithmetic of block_sum over a thread-strided sum, i.e. the same bits as four separate passes -- which were four
        // dependent memory round trips at the head of every backward launch
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int i = threadIdx.x; i < a.nblocks_fwd; i += BPR_THREADS) {
            const f32x4 x = reinterpret_cast<const f32x4*>(a.partials)[i];
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s.x += __shfl_xor(s.x, off); s.y += __shfl_xor(s.y, off);
            s.z += __shfl_xor(s.z, off); s.w += __shfl_xor(s.w, off);
        }
        __shared__ f32x4 red4[4];
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const f32x4 t0 = red4[0], t1 = red4[1], t2 = red4[2], t3 = red4[3];
            tot[0] = (t0.x + t1.x) + (t2.x + t3.x);
            tot[1] = (t0.y + t1.y) + (t2.y + t3.y);
            tot[2] = (t0.z + t1.z) + (t2.z + t3.z);
            tot[3] = (t0.w + t1.w) + (t2.w + t3.w);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(BPR_THREADS) void bpr_sums_kernel(BprArgs a) {
    __shared__ float red[4];
    __shared__ float tot[4];
    batch_totals(a, tot, red);
    if (threadIdx.x < 4) a.sums_out[threadIdx.x] = tot[threadIdx.x];
}

template <int G>
__global__ __launch_bounds__(BPR_THREADS) void bpr_bwd_kernel(BprArgs a) {
    __shared__ float red[4];
    __shared__ float tot[4];
    batch_totals(a, tot, red);
    const float invB = 1.0f / (float)a.B_global;
    const float nu = sqrtf(tot[0]), np_ = sqrtf(tot[1]), nn = sqrtf(tot[2]);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.loss_out) {
        a.loss_out[0] = tot[3] * invB;
        a.loss_out[1] = a.reg * (nu * invB + np_ * invB + nn * invB);
    }
    if (!a.gu) return;
    // d(reg*|X|_F/B)/dX = reg * X / (B*|X|_F)   (0 at X = 0, as autograd returns)
    const float cu = nu > 0.f ? a.reg * invB / nu : 0.f;
    const float cp = np_ > 0.f ? a.reg * invB / np_ : 0.f;
    const float cn = nn > 0.f ? a.reg * invB / nn : 0.f;

    const int lig = threadIdx.x % G;
    const int64_t gid = (int64_t)blockIdx.x * (BPR_THREADS / G) + threadIdx.x / G;
    const int64_t gstride = (int64_t)gridDim.x * (BPR_THREADS / G);
    const int nvec = a.d >> 2;
    for (int64_t b = gid; b < a.B; b += gstride) {
        const int64_t ru = a.iu ? a.iu[b] : b, rp = a.ip ? a.ip[b] : b, rn = a.in_ ? a.in_[b] : b;
        const float x = a.xbuf[b];
        const float sig = 1.0f / (1.0f + expf(-x));
        const float g = -invB * sig * (1.0f - sig) / (1e-5f + sig);
        const f32x4* pu = reinterpret_cast<const f32x4*>(a.tu + ru * a.d);
        const f32x4* pp = reinterpret_cast<const f32x4*>(a.tp + rp * a.d);
        const f32x4* pn = reinterpret_cast<const f32x4*>(a.tn + rn * a.d);
        float* gu = a.gu + ru * a.d;
        float* gp = a.gp + rp * a.d;
        float* gn = a.gn + rn * a.d;
        for (int c = lig; c < nvec; c += G) {
            const f32x4 u = pu[c], p = pp[c], n = pn[c];
            const int o = c * 4;
            unsafeAtomicAdd(gu + o + 0, g * (p.x - n.x) + cu * u.x);
            unsafeAtomicAdd(gu + o + 1, g * (p.y - n.y) + cu * u.y);
            unsafeAtomicAdd(gu + o + 2, g * (p.z - n.z) + cu * u.z);
            unsafeAtomicAdd(gu + o + 3, g * (p.w - n.w) + cu * u.w);
            unsafeAtomicAdd(gp + o + 0, g * u.x + cp * p.x);
            unsafeAtomicAdd(gp + o + 1, g * u.y + cp * p.y);
            unsafeAtomicAdd(gp + o + 2, g * u.z + cp * p.z);
            unsafeAtomicAdd(gp + o + 3, g * u.w + cp * p.w);
            unsafeAtomicAdd(gn + o + 0, -g * u.x + cn * n.x);
            unsafeAtomicAdd(gn + o + 1, -g * u.y + cn * n.y);
            unsafeAtomicAdd(gn + o + 2, -g * u.z + cn * n.z);
            unsafeAtomicAdd(gn + o + 3, -g * u.w + cn * n.w);
        }
    }
}

// ---------------------------------------------------------------- deterministic backward (no atomics)
// Reverse index of one batch ("plan", int32, built on the host next to the sampler):
//   [0] nu  [1] ni  [2] L = layout batch size (>= the batch; one stride for all batches of an epoch)
//   [3 .. 3+L)  user rows touched (ascending), then (L+1) offsets into the user list, L triple ids
//   grouped by user row; then 2L item rows, (2L+1) offsets, 2L entries  b | (role << 30)
//   (role 0 = positive, 1 = negative); then [n_heavy, heavy row slots...] (see plan_heavy_off).
// One lane group owns one touched row and sums its contributions in list order, then STORES the
// row of the dense gradient table (which is zero everywhere else): no atomics, bit-reproducible.
// The plan ends with the list of HEAVY rows (more than BPR_HEAVY entries): [n_heavy, slot, slot, ...] where a slot
// is the row's position w in the concatenated (user rows, item rows) list; at most 3L/(BPR_HEAVY+1) of them.
__host__ __device__ inline int64_t plan_heavy_cap(int64_t L) { return 3 * L / BPR_HEAVY + 2; }
__host__ __device__ inline int64_t plan_heavy_off(int64_t L) { return 3 + (3 * L + 1) + (6 * L + 1); }
__host__ __device__ inline int64_t plan_ints(int64_t L) { return plan_heavy_off(L) + 1 + plan_heavy_cap(L); }

struct PlanView {
    int n_u, n_i, n_heavy;
    const int32_t *urow, *uptr, *ulist, *irow, *iptr, *ilist, *heavy;
};

__device__ __forceinline__ PlanView plan_view(const int32_t* pl) {
    PlanView v;
    v.n_u = pl[0];
    v.n_i = pl[1];
    const int64_t L = pl[2];
    v.urow = pl + 3;
    v.uptr = v.urow + L;
    v.ulist = v.uptr + (L + 1);
    v.irow = v.ulist + L;
    v.iptr = v.irow + 2 * L;
    v.ilist = v.iptr + (2 * L + 1);
    v.n_heavy = pl[plan_heavy_off(L)];
    v.heavy = pl + plan_heavy_off(L) + 1;
    return v;
}

struct BwdCoef {
    float invB, cu, cp, cn;
};

// Sum the contributions of plan entries [e0, e1) of ONE gradient row into acc (this lane's 16-B column slice),
// in list order.  The entries' metadata (triple id -> score difference, the other rows' ids) is fetched G at a
// time, one entry per lane, and broadcast by shuffles; the rows the entries point at are fetched 4 at a time
// before the dependent adds, so the chain of a row costs one latency per 4 entries instead of 4 per entry.
//   user row u:      d/du  = g (p - n) + cu u          (own = u; two gathered rows per entry)
//   item row, pos:   d/dp  = g u + cp p                (own = p; one gathered row per entry)
//   item row, neg:   d/dn  = -g u + cn n               (own = n)
template <int G>
__device__ __forceinline__ void grad_row_entries(const BprArgs& a, const BwdCoef& k, bool user_side,
                                                 const int32_t* list, int e0, int e1, int c, bool on, int lig,
                                                 const f32x4& own, f32x4& acc) {
    for (int base = e0; base < e1; base += G) {
        const int e = base + lig;
        float g = 0.f;
        int ra = 0, rb = 0, role = 0;
        if (e < e1) {
            const int ent = list[e];
            const int b = ent & 0x3fffffff;
            role = ent >> 30;
            const float x = a.xbuf[b];
            const float sig = 1.0f / (1.0f + expf(-x));
            g = -k.invB * sig * (1.0f - sig) / (1e-5f + sig);
            if (user_side) {
                ra = a.ip[b];
                rb = a.in_[b];
            } else {
                ra = a.iu[b];
            }
        }
        const int cnt = (e1 - base) < G ? (e1 - base) : G;
        int t = 0;
        for (; t + 4 <= cnt; t += 4) {
            float gg[4];
            int aa[4], bb[4], rr[4];
            f32x4 xa[4], xb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gg[q] = __shfl(g, t + q, G);
                aa[q] = __shfl(ra, t + q, G);
                bb[q] = __shfl(rb, t + q, G);
                rr[q] = __shfl(role, t + q, G);
            }
            if (on) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (user_side) {
                        xa[q] = reinterpret_cast<const f32x4*>(a.tp + (int64_t)aa[q] * a.d)[c];
                        xb[q] = reinterpret_cast<const f32x4*>(a.tn + (int64_t)bb[q] * a.d)[c];
                    } else {
                        xa[q] = reinterpret_cast<const f32x4*>(a.tu + (int64_t)aa[q] * a.d)[c];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (user_side) {
                        acc.x += gg[q] * (xa[q].x - xb[q].x) + k.cu * own.x;
                        acc.y += gg[q] * (xa[q].y - xb[q].y) + k.cu * own.y;
                        acc.z += gg[q] * (xa[q].z - xb[q].z) + k.cu * own.z;
                        acc.w += gg[q] * (xa[q].w - xb[q].w) + k.cu * own.w;
                    } else {
                        const float sg = rr[q] == 0 ? gg[q] : -gg[q];
                        const float cc = rr[q] == 0 ? k.cp : k.cn;
                        acc.x += sg * xa[q].x + cc * own.x;
                        acc.y += sg * xa[q].y + cc * own.y;
                        acc.z += sg * xa[q].z + cc * own.z;
                        acc.w += sg * xa[q].w + cc * own.w;
                    }
                }
            }
        }
        for (; t < cnt; ++t) {
            const float gq = __shfl(g, t, G);
            const int aq = __shfl(ra, t, G), bq = __shfl(rb, t, G), rq = __shfl(role, t, G);
            if (on) {
                if (user_side) {
                    const f32x4 p = reinterpret_cast<const f32x4*>(a.tp + (int64_t)aq * a.d)[c];
                    const f32x4 n = reinterpret_cast<const f32x4*>(a.tn + (int64_t)bq * a.d)[c];
                    acc.x += gq * (p.x - n.x) + k.cu * own.x;
                    acc.y += gq * (p.y - n.y) + k.cu * own.y;
                    acc.z += gq * (p.z - n.z) + k.cu * own.z;
                    acc.w += gq * (p.w - n.w) + k.cu * own.w;
                } else {
                    const f32x4 u = reinterpret_cast<const f32x4*>(a.tu + (int64_t)aq * a.d)[c];
                    const float sg = rq == 0 ? gq : -gq;
                    const float cc = rq == 0 ? k.cp : k.cn;
                    acc.x += sg * u.x + cc * own.x;
                    acc.y += sg * u.y + cc * own.y;
                    acc.z += sg * u.z + cc * own.z;
                    acc.w += sg * u.w + cc * own.w;
                }
            }
        }
    }
}

__device__ __forceinline__ BwdCoef bwd_coef(const BprArgs& a, const float* tot) {
    BwdCoef k;
    k.invB = 1.0f / (float)a.B_global;
    const float nu_ = sqrtf(tot[0]), np_ = sqrtf(tot[1]), nn = sqrtf(tot[2]);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.loss_out) {
        a.loss_out[0] = tot[3] * k.invB;
        a.loss_out[1] = a.reg * (nu_ * k.invB + np_ * k.invB + nn * k.invB);
    }
    // d(reg*|X|_F/B)/dX = reg * X / (B*|X|_F)   (0 at X = 0, as autograd returns)
    k.cu = nu_ > 0.f ? a.reg * k.invB / nu_ : 0.f;
    k.cp = np_ > 0.f ? a.reg * k.invB / np_ : 0.f;
    k.cn = nn > 0.f ? a.reg * k.invB / nn : 0.f;
    return k;
}

// Deterministic backward, ONE launch.  Blocks [0, light_blocks): one lane group per touched row with <= BPR_HEAVY
// entries -- the row is summed in list order and STORED.  Blocks beyond: one block per heavy row of the plan's
// heavy list; its lane groups split the row's entry list into contiguous chunks and the partial sums meet in a
// fixed order (shuffle tree inside a wave, the 4 waves through LDS).  d <= 256 (one 16-B slice per lane).
template <int G>
__global__ __launch_bounds__(BPR_THREADS) void bpr_bwd_rows_kernel(BprArgs a, int light_blocks) {
    __shared__ float red[4];
    __shared__ float tot[4];
    __shared__ f32x4 wsum[4][G];
    // the plan's header and (heavy workgroups) its heavy count are in flight while the batch sums are reduced
    const PlanView pv = plan_view(a.plan);
    batch_totals(a, tot, red);
    if ((int)blockIdx.x >= light_blocks && (int)blockIdx.x - light_blocks >= pv.n_heavy) return;   // surplus block
    const BwdCoef k = bwd_coef(a, tot);
    const int lig = threadIdx.x % G;
    const int nvec = a.d >> 2;
    const bool on = lig < nvec;
    if ((int)blockIdx.x < light_blocks) {
        const int64_t gid = (int64_t)blockIdx.x * (BPR_THREADS / G) + threadIdx.x / G;
        const int64_t gstride = (int64_t)light_blocks * (BPR_THREADS / G);
        for (int64_t w = gid; w < (int64_t)pv.n_u + pv.n_i; w += gstride) {
            const bool user_side = w < pv.n_u;
            const int64_t row = user_side ? pv.urow[w] : pv.irow[w - pv.n_u];
            const int e0 = user_side ? pv.uptr[w] : pv.iptr[w - pv.n_u];
            const int e1 = user_side ? pv.uptr[w + 1] : pv.iptr[w - pv.n_u + 1];
            if (e1 - e0 > BPR_HEAVY) continue;                    // on the heavy list
            if (a.own_mod > 1 && (int)(w % a.own_mod) != a.own_rem) continue;   // another rank's row
            f32x4 own = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
            if (on) own = reinterpret_cast<const f32x4*>((user_side ? a.tu : a.tp) + row * a.d)[lig];
            grad_row_entries<G>(a, k, user_side, user_side ? pv.ulist : pv.ilist, e0, e1, lig, on, lig, own, acc);
            if (on) *reinterpret_cast<f32x4*>((user_side ? a.gu : a.gp) + row * a.d + lig * 4) = acc;
        }
        return;
    }
    constexpr int NGB = BPR_THREADS / G;
    const int gg = threadIdx.x / G;
    const int heavy_blocks = (int)gridDim.x - light_blocks;
    for (int h = (int)blockIdx.x - light_blocks; h < pv.n_heavy; h += heavy_blocks) {
        const int64_t w = pv.heavy[h];
        if (a.own_mod > 1 && (int)(w % a.own_mod) != a.own_rem) continue;       // (block-uniform: no barrier is skipped by a part of the block)
        const bool user_side = w < pv.n_u;
        const int64_t row = user_side ? pv.urow[w] : pv.irow[w - pv.n_u];
        const int r0 = user_side ? pv.uptr[w] : pv.iptr[w - pv.n_u];
        const int r1 = user_side ? pv.uptr[w + 1] : pv.iptr[w - pv.n_u + 1];
        int chunk = (r1 - r0 + NGB - 1) / NGB;
        chunk = (chunk + 3) & ~3;
        const int e0 = r0 + gg * chunk;
        const int e1 = e0 + chunk < r1 ? e0 + chunk : r1;
        f32x4 own = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
        if (on) own = reinterpret_cast<const f32x4*>((user_side ? a.tu : a.tp) + row * a.d)[lig];
        if (e0 < r1) grad_row_entries<G>(a, k, user_side, user_side ? pv.ulist : pv.ilist, e0, e1, lig, on, lig, own, acc);
#pragma unroll
        for (int off = G; off < 64; off <<= 1) {
            acc.x += __shfl_down(acc.x, off);
            acc.y += __shfl_down(acc.y, off);
            acc.z += __shfl_down(acc.z, off);
            acc.w += __shfl_down(acc.w, off);
        }
        __syncthreads();
        if ((threadIdx.x & 63) < G) wsum[threadIdx.x >> 6][lig] = acc;
        __syncthreads();
        if (threadIdx.x < G && on) {
            const f32x4 t0 = wsum[0][lig], t1 = wsum[1][lig], t2 = wsum[2][lig], t3 = wsum[3][lig];
            f32x4 r;
            r.x = (t0.x + t1.x) + (t2.x + t3.x);
            r.y = (t0.y + t1.y) + (t2.y + t3.y);
            r.z = (t0.z + t1.z) + (t2.z + t3.z);
            r.w = (t0.w + t1.w) + (t2.w + t3.w);
            *reinterpret_cast<f32x4*>((user_side ? a.gu : a.gp) + row * a.d + lig * 4) = r;
        }
    }
}

// ---------------------------------------------------------------- plan builder (device)
// One workgroup per batch sorts the batch's (row | entry) keys in LDS (bitonic; see PlanKey for the two widths) and
// emits the reverse index in the layout bpr_bwd_rows_kernel reads.  An epoch of S-ML (159 batches of
// 4096 triples) is one launch of 159 workgroups.
constexpr int PLAN_THREADS = 1024;

// Sort key of a plan entry: row id above the payload (entry index in the batch, role bit for the item side).
// 64 bits in general; 32 bits when every row id of the batch is below 2^17 (MovieLens / CiteULike scale: half the
// LDS traffic of the sort, 32-bit compares): row << 14 | role << 13 | entry, entry < 8192.
template <typename K>
struct PlanKey;
template <>
struct PlanKey<unsigned long long> {
    static constexpr unsigned long long INVALID = ~0ull;
    __device__ static unsigned long long make(uint32_t row, unsigned e, unsigned role) {
        return ((unsigned long long)row << 31) | e | (role << 30);
    }
    __device__ static int32_t row(unsigned long long k) { return (int32_t)(k >> 31); }
    __device__ static int32_t entry(unsigned long long k) { return (int32_t)(k & 0x7fffffffull); }   // e | role << 30
};
template <>
struct PlanKey<uint32_t> {
    static constexpr uint32_t INVALID = ~0u;
    __device__ static uint32_t make(uint32_t row, unsigned e, unsigned role) { return (row << 14) | (role << 13) | e; }
    __device__ static int32_t row(uint32_t k) { return (int32_t)(k >> 14); }
    __device__ static int32_t entry(uint32_t k) { return (int32_t)((k & 0x1fffu) | (((k >> 13) & 1u) << 30)); }
};

// LDS index of key i: eight keys of padding after every 64.  A pass whose eight keys per thread are 8 apart (threads
// t, t+8, .. of a wave then sit 64 keys apart) would put eight lanes on every bank; with the padding the 64 lanes of a
// read fall on 64 different banks for every stride the sort uses (1, 8, 64, 512, 4096), and a thread's eight
// consecutive keys stay contiguous and 32-B aligned.
__device__ __forceinline__ int plan_pk(int i) { return i + ((i >> 6) << 3); }
__host__ __device__ constexpr size_t plan_padded_keys(size_t n) { return n + ((n + 63) / 64) * 8; }

// Bitonic sort of P keys (a power of two) in LDS by one workgroup, ascending.  Written without direction bits: the
// first sub-stage of stage k compares key i with key i ^ (k - 1) (the mirrored position in its block of k), the others i
// with i ^ stride, and every compare puts the smaller key at the lower index -- two instructions (min, max) per
// compare-exchange of 32-bit keys.  A thread takes the 2^NB keys whose indices differ in NB consecutive bit positions,
// runs those NB sub-stages on them in registers and writes them back: a stage of log2(k) sub-stages costs
// ceil(log2(k) / 3) passes over LDS (one barrier each) instead of log2(k), and the stages k = 2, 4, 8 are one pass.
// P = 8192: 33 passes instead of 91.
template <typename K>
__device__ __forceinline__ void bitonic_cx(K& x, K& y) {
    const bool sw = y < x;
    const K lo = sw ? y : x, hi = sw ? x : y;
    x = lo;
    y = hi;
}
template <>
__device__ __forceinline__ void bitonic_cx<uint32_t>(uint32_t& x, uint32_t& y) {
    const uint32_t lo = min(x, y), hi = max(x, y);
    x = lo;
    y = hi;
}

// Sub-stages with strides 2^b, 2^(b-1) .. 2^(b-NB+1).  FIRST: 2^b is the first sub-stage of its stage (k = 2^(b+1)):
// the upper half of a thread's keys then comes from the mirrored low bits, so that register c pairs with register ~c.
template <typename K, int NB, bool FIRST>
__device__ __forceinline__ void bitonic_pass(K* keys, int P, int b) {
    constexpr int N = 1 << NB;
    const int lowbit = b - NB + 1;
    const int lowmask = (1 << lowbit) - 1;
    for (int t = threadIdx.x; t < (P >> NB); t += PLAN_THREADS) {
        const int hi = (t >> lowbit) << (b + 1), lo = t & lowmask;
        const int lo_up = FIRST ? lowmask - lo : lo;
        K v[N];
#pragma unroll
        for (int c = 0; c < N; ++c) v[c] = keys[plan_pk(hi | (c << lowbit) | (c >> (NB - 1) ? lo_up : lo))];
        if (FIRST) {
#pragma unroll
            for (int c = 0; c < N / 2; ++c) bitonic_cx(v[c], v[c ^ (N - 1)]);
        }
#pragma unroll
        for (int s = FIRST ? NB - 2 : NB - 1; s >= 0; --s) {
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (!(c & (1 << s))) bitonic_cx(v[c], v[c | (1 << s)]);
        }
#pragma unroll
        for (int c = 0; c < N; ++c) keys[plan_pk(hi | (c << lowbit) | (c >> (NB - 1) ? lo_up : lo))] = v[c];
    }
    __syncthreads();
}

template <typename K>
__device__ inline void bitonic_sort_lds(K* keys, int P) {
    if (P >= 8) {                                  // stages k = 2, 4, 8 on eight consecutive keys, in registers
        for (int t = threadIdx.x; t < (P >> 3); t += PLAN_THREADS) {
            K v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = keys[plan_pk(t * 8) + c];
#pragma unroll
            for (int m = 1; m <= 3; ++m) {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (!(c & (1 << (m - 1)))) bitonic_cx(v[c], v[c ^ ((1 << m) - 1)]);
#pragma unroll
                for (int s = m - 2; s >= 0; --s) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (!(c & (1 << s))) bitonic_cx(v[c], v[c | (1 << s)]);
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) keys[plan_pk(t * 8) + c] = v[c];
        }
        __syncthreads();
    }
    for (int m = P >= 8 ? 4 : 1; (1 << m) <= P; ++m) {
        // sub-stage strides 2^(m-1) .. 1: the first pass takes what three-at-a-time leaves over, so that the last one
        // works on eight consecutive keys
        const int nb0 = m % 3 ? m % 3 : 3;
        if (nb0 == 3) bitonic_pass<K, 3, true>(keys, P, m - 1);
        else if (nb0 == 2) bitonic_pass<K, 2, true>(keys, P, m - 1);
        else bitonic_pass<K, 1, true>(keys, P, m - 1);
        for (int b = m - 1 - nb0; b >= 0; b -= 3) bitonic_pass<K, 3, false>(keys, P, b);
    }
}

// Exclusive prefix sum of one int per thread over the workgroup (wave scans + PLAN_THREADS / 64 wave totals in LDS).
__device__ inline int plan_block_scan(int v, int* scr, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    if (lane == 63) scr[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < PLAN_THREADS / 64; ++i) {
        const int c = scr[i];
        base += i < w ? c : 0;
        tot += c;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// keys[0..P) sorted, invalid = ~0.  Writes rows[], ptr[], list[] and returns the segment count.
template <typename K>
__device__ inline int plan_emit(const K* keys, int P, int32_t* rows, int32_t* ptr, int32_t* list, int* scan) {
    const int chunk = (P + PLAN_THREADS - 1) / PLAN_THREADS;
    const int e0 = threadIdx.x * chunk, e1 = min(e0 + chunk, P);
    int starts = 0, valid = 0;
    for (int e = e0; e < e1; ++e) {
        const K kx = keys[plan_pk(e)];
        if (kx == PlanKey<K>::INVALID) break;
        ++valid;
        if (e == 0 || PlanKey<K>::row(kx) != PlanKey<K>::row(keys[plan_pk(e - 1)])) ++starts;
    }
    int nseg, nvalid;
    int rank = plan_block_scan(starts, scan, nseg);
    plan_block_scan(valid, scan, nvalid);
    for (int e = e0; e < e1; ++e) {
        const K kx = keys[plan_pk(e)];
        if (kx == PlanKey<K>::INVALID) break;
        if (e == 0 || PlanKey<K>::row(kx) != PlanKey<K>::row(keys[plan_pk(e - 1)])) {
            rows[rank] = PlanKey<K>::row(kx);
            ptr[rank] = e;
            ++rank;
        }
        list[e] = PlanKey<K>::entry(kx);
    }
    if (threadIdx.x == 0) ptr[nseg] = nvalid;
    __syncthreads();
    return nseg;
}

#ifdef CRH_PROFILE
__device__ unsigned long long crh_plan_clk[16];   // phase stamps of workgroup 0 (s_memrealtime, 100 MHz), last launch
#define PLAN_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) crh_plan_clk[8 * blockIdx.y + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PLAN_STAMP(i) do {} while (0)
#endif

// One side of one batch with keys of type K (LDS: P keys + the scan scratch); returns the side's row count.
// side 0: the L user keys (P / 2 slots); side 1: the 2 L item keys, positives then negatives.
template <typename K>
__device__ inline int plan_side(int side, const int32_t* iu, const int32_t* ip, const int32_t* in_, int64_t lo, int cnt,
                                int P, char* smem, int32_t* rows, int32_t* ptr, int32_t* list, int* scan) {
    K* keys = reinterpret_cast<K*>(smem);
    const int Ps = side == 0 && P > 2 ? P >> 1 : P;
    for (int e = threadIdx.x; e < Ps; e += PLAN_THREADS) {
        K kx = PlanKey<K>::INVALID;
        if (side == 0) {
            if (e < cnt) kx = PlanKey<K>::make((uint32_t)iu[lo + e], (unsigned)e, 0u);
        } else if (e < cnt) {
            kx = PlanKey<K>::make((uint32_t)ip[lo + e], (unsigned)e, 0u);
        } else if (e < 2 * cnt) {
            kx = PlanKey<K>::make((uint32_t)in_[lo + e - cnt], (unsigned)(e - cnt), 1u);
        }
        keys[plan_pk(e)] = kx;
    }
    __syncthreads();
    PLAN_STAMP(2);
    bitonic_sort_lds(keys, Ps);
    PLAN_STAMP(3);
    const int n = plan_emit(keys, Ps, rows, ptr, list, scan);
    PLAN_STAMP(4);
    return n;
}

// Heavy rows of one plan, ascending (a fixed order: the one-launch MF step sums per-block partials, and which block
// takes which heavy row follows the list): count per thread chunk, serial scan of the counts, ordered write.
// Also writes the plan header.  scan: PLAN_THREADS ints of LDS.
__device__ inline void plan_emit_heavy(int32_t* pl, int64_t L, int nu, int ni, const int32_t* uptr, const int32_t* iptr,
                                       int* scan) {
    int32_t* hv = pl + plan_heavy_off(L);
    const int rchunk = (nu + ni + PLAN_THREADS - 1) / PLAN_THREADS;
    const int r0 = threadIdx.x * rchunk, r1 = min(r0 + rchunk, nu + ni);
    int nh = 0;
    for (int r = r0; r < r1; ++r)
        nh += (r < nu ? uptr[r + 1] - uptr[r] : iptr[r - nu + 1] - iptr[r - nu]) > BPR_HEAVY;
    __syncthreads();
    int n_heavy;
    int w = plan_block_scan(nh, scan, n_heavy);
    if (threadIdx.x == 0) {
        pl[0] = nu;
        pl[1] = ni;
        pl[2] = (int32_t)L;
        hv[0] = n_heavy;
    }
    for (int r = r0; r < r1; ++r)
        if ((r < nu ? uptr[r + 1] - uptr[r] : iptr[r - nu + 1] - iptr[r - nu]) > BPR_HEAVY) hv[1 + w++] = r;
}

// Heavy lists for plans whose row lists were built elsewhere (crh_bpr_plan_build_large, or a caller's own builder:
// up to 3 x 524288 rows per plan).  Two passes over HV_BLOCKS chunks per plan: count, then ordered write.
constexpr int HV_BLOCKS = 64, HV_THREADS = 256;

__device__ __forceinline__ int plan_row_heavy(const int32_t* uptr, const int32_t* iptr, int nu, int r) {
    return (r < nu ? uptr[r + 1] - uptr[r] : iptr[r - nu + 1] - iptr[r - nu]) > BPR_HEAVY;
}

__global__ __launch_bounds__(HV_THREADS) void plan_heavy_pass_kernel(int32_t* plans, int64_t stride, int32_t* counts,
                                                                      int write) {
    __shared__ int scan[HV_THREADS];
    __shared__ int base_s;
    int32_t* pl = plans + (int64_t)blockIdx.y * stride;
    const int64_t L = pl[2];
    const int nu = pl[0], ni = pl[1], n = nu + ni;
    const int32_t* uptr = pl + 3 + L;
    const int32_t* iptr = pl + 3 + (3 * L + 1) + 2 * L;
    const int per_block = (n + HV_BLOCKS - 1) / HV_BLOCKS;
    const int b0 = blockIdx.x * per_block, b1 = min(b0 + per_block, n);
    const int rc = (max(b1 - b0, 0) + HV_THREADS - 1) / HV_THREADS;
    const int t0 = b0 + threadIdx.x * rc, t1 = min(t0 + rc, b1);
    int nh = 0;
    for (int r = t0; r < t1; ++r) nh += plan_row_heavy(uptr, iptr, nu, r);
    scan[threadIdx.x] = nh;
    __syncthreads();
    int32_t* cnt = counts + (int64_t)blockIdx.y * HV_BLOCKS;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < HV_THREADS; ++i) {
            const int c = scan[i];
            scan[i] = acc;
            acc += c;
        }
        if (!write) {
            cnt[blockIdx.x] = acc;
        } else {
            int base = 0, total = 0;
            for (int i = 0; i < HV_BLOCKS; ++i) {
                if (i < (int)blockIdx.x) base += cnt[i];
                total += cnt[i];
            }
            base_s = base;
            if (blockIdx.x == 0) pl[plan_heavy_off(L)] = total;
        }
    }
    if (!write) return;
    __syncthreads();
    int32_t* hv = pl + plan_heavy_off(L) + 1;
    int w = base_s + scan[threadIdx.x];
    for (int r = t0; r < t1; ++r)
        if (plan_row_heavy(uptr, iptr, nu, r)) hv[w++] = r;
}

__global__ __launch_bounds__(PLAN_THREADS) void bpr_plan_kernel(const int32_t* __restrict__ iu,
                                                                const int32_t* __restrict__ ip,
                                                                const int32_t* __restrict__ in_, int64_t n_rec,
                                                                int L, int P, int32_t* __restrict__ plans,
                                                                int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PLAN_STAMP(0);
    int* scan = reinterpret_cast<int*>(smem + plan_padded_keys(P) * 8);      // behind the (padded) key area, sized for 64-bit keys
    const int side = blockIdx.y;                   // 0: user rows, 1: item rows -- one workgroup each
    const int64_t lo = (int64_t)blockIdx.x * L;
    const int cnt = (int)((n_rec - lo) < L ? (n_rec - lo) : L);
    int32_t* pl = plans + (int64_t)blockIdx.x * stride;
    int32_t* urow = pl + 3;
    int32_t* uptr = urow + L;
    int32_t* ulist = uptr + (L + 1);
    int32_t* irow = ulist + L;
    int32_t* iptr = irow + 2 * L;
    int32_t* ilist = iptr + (2 * L + 1);

    // largest row id of this side decides the key width (block maximum: wave maxima through LDS)
    int mx = 0;
    for (int e = threadIdx.x; e < cnt; e += PLAN_THREADS) mx = max(mx, side == 0 ? iu[lo + e] : max(ip[lo + e], in_[lo + e]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) scan[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = 0;
#pragma unroll
    for (int i = 0; i < PLAN_THREADS / 64; ++i) mx = max(mx, scan[i]);
    const bool narrow = mx < (1 << 17) && L <= 8192;
    __syncthreads();
    PLAN_STAMP(1);
    int32_t* rows = side == 0 ? urow : irow;
    int32_t* ptr = side == 0 ? uptr : iptr;
    int32_t* list = side == 0 ? ulist : ilist;
    const int n = narrow ? plan_side<uint32_t>(side, iu, ip, in_, lo, cnt, P, smem, rows, ptr, list, scan)
                         : plan_side<unsigned long long>(side, iu, ip, in_, lo, cnt, P, smem, rows, ptr, list, scan);
    if (threadIdx.x == 0) pl[side] = n;            // header: rows per side (the batch size follows in the heavy pass)
}

// Second launch of the device builder: header + heavy list of every plan, once both sides' rows are in place.
__global__ __launch_bounds__(PLAN_THREADS) void bpr_plan_heavy_kernel(int L, int32_t* __restrict__ plans, int64_t stride) {
    __shared__ int scan[PLAN_THREADS / 64];
    int32_t* pl = plans + (int64_t)blockIdx.x * stride;
    const int nu = pl[0], ni = pl[1];
    const int32_t* uptr = pl + 3 + L;
    const int32_t* iptr = pl + 3 + (3 * L + 1) + 2 * L;
    plan_emit_heavy(pl, L, nu, ni, uptr, iptr, scan);
}
#ifdef CRH_PROFILE
extern "C" int crh_profile_plan_clocks(unsigned long long* out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(crh_plan_clk), sizeof(crh_plan_clk)) == hipSuccess ? 0 : -1;
}
#endif

// ---------------------------------------------------------------- plan builder for batches beyond the LDS sort
// Batches above 8 192 triples (S-TRAIN-XL: 65 536) do not fit one workgroup's LDS.  Same result, several workgroups
// per batch: (1) chunks of LP_CHUNK keys are sorted in LDS (bitonic, 64-bit keys row << 31 | role << 30 | entry);
// (2) log2(chunks) merge passes in global memory, every thread producing LP_VT consecutive outputs of one merged pair
// from its merge-path split (keys are unique, so there are no ties to order); (3) segment starts are counted per block,
// each block sums the counts of the blocks before it, and rows / offsets / lists are written in place; (4) the heavy
// lists by plan_heavy_pass_kernel.  grid.y = batch, so an epoch's plans are a handful of launches.
constexpr int LP_CHUNK = 8192, LP_VT = 8, LP_EMIT = 4096, LP_MAX_EBLOCKS = 256;
typedef unsigned long long lpkey;

struct LargePlanSide {       // one side (users / items) of every batch of the call
    int width_per_L;         // 1 = user side (L keys per batch), 2 = item side (2 L keys)
    int64_t P;               // padded keys per batch: LP_CHUNK * power of two
};

__device__ __forceinline__ lpkey lp_key(const int32_t* iu, const int32_t* ip, const int32_t* in_, int64_t lo, int cnt,
                                        int item_side, int64_t e) {
    if (!item_side) return e < cnt ? PlanKey<lpkey>::make((uint32_t)iu[lo + e], (unsigned)e, 0u) : PlanKey<lpkey>::INVALID;
    if (e < cnt) return PlanKey<lpkey>::make((uint32_t)ip[lo + e], (unsigned)e, 0u);
    if (e < 2 * (int64_t)cnt) return PlanKey<lpkey>::make((uint32_t)in_[lo + e - cnt], (unsigned)(e - cnt), 1u);
    return PlanKey<lpkey>::INVALID;
}

__global__ __launch_bounds__(PLAN_THREADS) void plan_chunk_sort_kernel(const int32_t* __restrict__ iu,
                                                                        const int32_t* __restrict__ ip,
                                                                        const int32_t* __restrict__ in_, int64_t n_rec,
                                                                        int64_t L, int item_side, int64_t P,
                                                                        lpkey* __restrict__ keys) {
    __shared__ lpkey sk[plan_padded_keys(LP_CHUNK)];
    const int64_t lo = (int64_t)blockIdx.y * L;
    const int cnt = (int)((n_rec - lo) < L ? (n_rec - lo) : L);
    const int64_t c0 = (int64_t)blockIdx.x * LP_CHUNK;
    for (int e = threadIdx.x; e < LP_CHUNK; e += PLAN_THREADS) sk[plan_pk(e)] = lp_key(iu, ip, in_, lo, cnt, item_side, c0 + e);
    __syncthreads();
    bitonic_sort_lds(sk, LP_CHUNK);
    lpkey* out = keys + (int64_t)blockIdx.y * P + c0;
    for (int e = threadIdx.x; e < LP_CHUNK; e += PLAN_THREADS) out[e] = sk[plan_pk(e)];
}

__global__ __launch_bounds__(256) void plan_merge_pass_kernel(const lpkey* __restrict__ src, lpkey* __restrict__ dst,
                                                              int64_t P, int64_t run) {
    const int64_t o0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * LP_VT;
    if (o0 >= P) return;
    const lpkey* a = src + (int64_t)blockIdx.y * P + (o0 / (2 * run)) * (2 * run);
    const lpkey* b = a + run;
    const int64_t d = o0 % (2 * run);
    int64_t lo = d > run ? d - run : 0, hi = d < run ? d : run;
    while (lo < hi) {                       // merge path: the first i with a[i] >= b[d - 1 - i]
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < b[d - 1 - mid]) lo = mid + 1;
        else hi = mid;
    }
    int64_t i = lo, j = d - lo;
    lpkey* out = dst + (int64_t)blockIdx.y * P + o0;
    lpkey av = i < run ? a[i] : PlanKey<lpkey>::INVALID, bv = j < run ? b[j] : PlanKey<lpkey>::INVALID;
#pragma unroll
    for (int k = 0; k < LP_VT; ++k) {
        const bool take_a = j >= run || (i < run && av < bv);
        out[k] = take_a ? av : bv;
        if (take_a) { ++i; av = i < run ? a[i] : PlanKey<lpkey>::INVALID; }
        else { ++j; bv = j < run ? b[j] : PlanKey<lpkey>::INVALID; }
    }
}

// write = 0: counts[batch][block] = segment starts of the block's LP_EMIT keys; write = 1: rows / offsets / list.
__global__ __launch_bounds__(PLAN_THREADS) void plan_emit_large_kernel(const lpkey* __restrict__ keys, int64_t P,
                                                                        int64_t n_rec, int64_t L, int item_side,
                                                                        int32_t* __restrict__ plans, int64_t stride,
                                                                        int32_t* __restrict__ counts, int n_eblocks,
                                                                        int write) {
    __shared__ int scan[PLAN_THREADS];
    __shared__ int base_s, total_s;
    const int64_t lo = (int64_t)blockIdx.y * L;
    const int cnt = (int)((n_rec - lo) < L ? (n_rec - lo) : L);
    const int64_t nvalid = item_side ? 2 * (int64_t)cnt : cnt;
    const lpkey* k = keys + (int64_t)blockIdx.y * P;
    int32_t* pl = plans + (int64_t)blockIdx.y * stride;
    const int64_t width = item_side ? 2 * L : L;
    int32_t* rows = item_side ? pl + 3 + (3 * L + 1) : pl + 3;
    int32_t* ptr = rows + width;
    int32_t* list = ptr + (width + 1);
    constexpr int PER = LP_EMIT / PLAN_THREADS;
    const int64_t e0 = (int64_t)blockIdx.x * LP_EMIT + (int64_t)threadIdx.x * PER;
    int starts = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int64_t e = e0 + q;
        if (e < nvalid && (e == 0 || PlanKey<lpkey>::row(k[e]) != PlanKey<lpkey>::row(k[e - 1]))) ++starts;
    }
    scan[threadIdx.x] = starts;
    __syncthreads();
    for (int off = 1; off < PLAN_THREADS; off <<= 1) {      // inclusive scan of the per-thread counts
        const int v = (int)threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    int32_t* cb = counts + (int64_t)blockIdx.y * LP_MAX_EBLOCKS;
    if (!write) {
        if (threadIdx.x == PLAN_THREADS - 1) cb[blockIdx.x] = scan[PLAN_THREADS - 1];
        return;
    }
    if (threadIdx.x == 0) {
        int base = 0, total = 0;
        for (int b = 0; b < n_eblocks; ++b) {
            if (b < (int)blockIdx.x) base += cb[b];
            total += cb[b];
        }
        base_s = base;
        total_s = total;
    }
    __syncthreads();
    int rank = base_s + scan[threadIdx.x] - starts;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int64_t e = e0 + q;
        if (e >= nvalid) break;
        const lpkey kx = k[e];
        if (e == 0 || PlanKey<lpkey>::row(kx) != PlanKey<lpkey>::row(k[e - 1])) {
            rows[rank] = PlanKey<lpkey>::row(kx);
            ptr[rank] = (int32_t)e;
            ++rank;
        }
        list[e] = PlanKey<lpkey>::entry(kx);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ptr[total_s] = (int32_t)nvalid;
        pl[item_side ? 1 : 0] = total_s;
        pl[2] = (int32_t)L;
    }
}

// ---------------------------------------------------------------- dense Adam (+ zero the gradient)
struct AdamSeg {
    float* p;
    float* g;
    float* m;
    float* v;
    int64_t n4;   // elements / 4
};


__global__ __launch_bounds__(256) void adam_dense_kernel(AdamSeg s0, AdamSeg s1, float one_minus_b1, float b2,
                                                         float one_minus_b2, float bc2_sqrt, float eps,
                                                         float neg_step_size, int zero_grad,
                                                         const float* __restrict__ step_scalars) {
    if (step_scalars) {   // step-dependent factors from memory: lets a captured hipGraph be replayed every epoch
        bc2_sqrt = step_scalars[0];
        neg_step_size = step_scalars[1];
    }
    const int64_t total = s0.n4 + s1.n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const bool first = i < s0.n4;
        const AdamSeg& s = first ? s0 : s1;
        const int64_t j = first ? i : i - s0.n4;
        f32x4* P = reinterpret_cast<f32x4*>(s.p) + j;
        f32x4* G = reinterpret_cast<f32x4*>(s.g) + j;
        f32x4* M = reinterpret_cast<f32x4*>(s.m) + j;
        f32x4* V = reinterpret_cast<f32x4*>(s.v) + j;
        const AdamK k{one_minus_b1, b2, one_minus_b2, eps};
        if (zero_grad & 2) {   // streaming tables (far beyond the caches): nontemporal accesses
            f32x4 p = __builtin_nontemporal_load(P), g = __builtin_nontemporal_load(G);
            f32x4 m = __builtin_nontemporal_load(M), v = __builtin_nontemporal_load(V);
            adam_elem4(p, m, v, g, k, bc2_sqrt, neg_step_size);
            __builtin_nontemporal_store(p, P);
            __builtin_nontemporal_store(m, M);
            __builtin_nontemporal_store(v, V);
            if (zero_grad & 1) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, G);
            continue;
        }
        f32x4 p = *P, g = *G, m = *M, v = *V;
        adam_elem4(p, m, v, g, k, bc2_sqrt, neg_step_size);
        *P = p; *M = m; *V = v;
        if (zero_grad) *G = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---------------------------------------------------------------- plain SGD (north_star's "BPR loss + SGD update")
// torch.optim.SGD defaults (no momentum, no weight decay): p <- p + (-lr) * g, ONE fused multiply-add per element
// (the canonical form of this build: oracle/oracle_np.py sgd_dense; ATen's vectorised add_(g, alpha=-lr) uses
// an fma as well).  SURVEY.md F3: the reference trains with Adam, SGD is the extra mode the north_star names.
// A row whose gradient is zero does not move, so the dense pass and the touched-rows pass give the same tables.
__global__ __launch_bounds__(256) void sgd_dense_kernel(float* __restrict__ p, float* __restrict__ g, int64_t n4,
                                                        float neg_lr, int zero_grad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4* P = reinterpret_cast<f32x4*>(p) + i;
        f32x4* G = reinterpret_cast<f32x4*>(g) + i;
        f32x4 x = *P;
        sgd_elem4(x, *G, neg_lr);
        *P = x;
        if (zero_grad) *G = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---------------------------------------------------------------- touched-rows replay of dense Adam
// torch.optim.Adam moves EVERY row every step: a row whose gradient is zero still decays its
// moments and drifts by its stale momentum.  At catalogue scale (S-TRAIN-XL: 11 M rows, 196 K touched per
// step) the dense pass is 99.5 % of the step's HBM traffic.  But rows are independent and the update is
// elementwise, so a row that is not touched between steps s and t can be brought up to date LATER by
// replaying its zero-gradient steps s+1..t-1 in registers -- the same fp32 operations in the same order,
// hence the same bits as the dense kernel -- provided the step-dependent factors of every step are at hand
// (`table`: [step][2] = {sqrt(1-beta2^step), -lr/(1-beta1^step)}, step 1 at index 1).  `last[row]` is the
// step the row's (p, m, v) are valid for.
//   mode 0  catch-up  rows of the plan -> valid for step `step` - 1      (before the forward pass of `step`)
//   mode 1  step      rows of the plan: apply step `step` with g = G[row], clear G[row]
//   mode 2  flush     ALL rows -> valid for step `step`                   (before the tables are read)
struct AdamRowsArgs {
    float *p, *g, *m, *v;
    int32_t* last;
    int64_t n_rows;        // rows of the table (mode 2)
    int d;
    const int32_t* plan;   // modes 0/1: touched rows (users, then items offset by user_rows)
    int64_t user_rows;
    int64_t step;
    const float* table;
    AdamK k;
    int mode;
};

template <int G>
__global__ __launch_bounds__(256) void adam_rows_kernel(AdamRowsArgs a) {
    const int lig = threadIdx.x % G;
    const int64_t gid = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    const int64_t gstride = (int64_t)gridDim.x * (256 / G);
    const int nvec = a.d >> 2;
    int64_t n_work = a.n_rows;
    PlanView pv{};
    if (a.mode != 2) {
        pv = plan_view(a.plan);
        n_work = (int64_t)pv.n_u + pv.n_i;
    }
    for (int64_t w = gid; w < n_work; w += gstride) {
        int64_t row = w;
        if (a.mode != 2) row = w < pv.n_u ? (int64_t)pv.urow[w] : a.user_rows + pv.irow[w - pv.n_u];
        const int64_t from = (int64_t)a.last[row] + 1;
        const int64_t upto = a.mode == 0 ? a.step - 1 : (a.mode == 1 ? a.step - 1 : a.step);   // zero-gradient part
        if (a.mode != 1 && from > upto) continue;
        for (int c = lig; c < nvec; c += G) {
            const int64_t o = row * a.d + (int64_t)c * 4;
            f32x4 p = *reinterpret_cast<f32x4*>(a.p + o);
            f32x4 m = *reinterpret_cast<f32x4*>(a.m + o);
            f32x4 v = *reinterpret_cast<f32x4*>(a.v + o);
            for (int64_t s = from; s <= upto; ++s) {
                const float bc2_sqrt = a.table[2 * s], nss = a.table[2 * s + 1];
                adam_elem4(p, m, v, f32x4{0.f, 0.f, 0.f, 0.f}, a.k, bc2_sqrt, nss);
            }
            if (a.mode == 1) {
                const f32x4 g = *reinterpret_cast<f32x4*>(a.g + o);
                const float bc2_sqrt = a.table[2 * a.step], nss = a.table[2 * a.step + 1];
                adam_elem4(p, m, v, g, a.k, bc2_sqrt, nss);
                *reinterpret_cast<f32x4*>(a.g + o) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<f32x4*>(a.p + o) = p;
            *reinterpret_cast<f32x4*>(a.m + o) = m;
            *reinterpret_cast<f32x4*>(a.v + o) = v;
        }
        // all lanes of the group have read last[row] before anyone overwrites it (same wave, in order)
        if (lig == 0) a.last[row] = (int32_t)(a.mode == 0 ? a.step - 1 : a.step);
    }
}

// SGD on the rows of a batch's plan only (users, then items offset by user_rows): p[row] += -lr * g[row], g[row] = 0.
// Equal to sgd_dense_kernel over the whole table (untouched rows have a zero gradient), at 3 x 16 B per touched
// element instead of a pass over every row: the update of choice at catalogue scale.
template <int G>
__global__ __launch_bounds__(256) void sgd_rows_kernel(float* __restrict__ p, float* __restrict__ g, int d,
                                                       const int32_t* __restrict__ plan, int64_t user_rows,
                                                       float neg_lr) {
    const PlanView pv = plan_view(plan);
    const int lig = threadIdx.x % G;
    const int nvec = d >> 2;
    const int64_t n_work = (int64_t)pv.n_u + pv.n_i;
    for (int64_t w = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; w < n_work; w += (int64_t)gridDim.x * (256 / G)) {
        const int64_t row = w < pv.n_u ? (int64_t)pv.urow[w] : user_rows + pv.irow[w - pv.n_u];
        for (int c = lig; c < nvec; c += G) {
            const int64_t o = row * d + (int64_t)c * 4;
            f32x4 x = *reinterpret_cast<f32x4*>(p + o);
            sgd_elem4(x, *reinterpret_cast<const f32x4*>(g + o), neg_lr);
            *reinterpret_cast<f32x4*>(p + o) = x;
            *reinterpret_cast<f32x4*>(g + o) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

int pick_group(int d) {
    const int nvec = d / 4;
    int g = 1;
    while (g < nvec && g < 64) g <<= 1;
    return g;
}

template <typename F>
int dispatch_group(int g, F&& f) {
    switch (g) {
        case 1: return f(std::integral_constant<int, 1>());
        case 2: return f(std::integral_constant<int, 2>());
        case 4: return f(std::integral_constant<int, 4>());
        case 8: return f(std::integral_constant<int, 8>());
        case 16: return f(std::integral_constant<int, 16>());
        case 32: return f(std::integral_constant<int, 32>());
        default: return f(std::integral_constant<int, 64>());
    }
}


// ---------------------------------------------------------------- whole MF step in ONE launch
// At MovieLens size a step is latency: three dependent launches (forward, row gradients, dense Adam) of a few
// microseconds each over ~20 MB of cache-resident state.  mf_step_kernel does the step in one launch, organised by
// TABLE ROW (dense Adam touches every row anyway):
//   * a lane group owns a row: if the batch touches it, the row's gradient is summed in plan-list order as in
//     bpr_bwd_rows_kernel, but the score difference x_b of each entry is RECOMPUTED from the rows themselves (user
//     entry: u.(p-n) from the two rows it gathers anyway; item entry: one extra row) -- no forward pass, no xbuf;
//     then Adam is applied to the row in registers and the new row is written to the OTHER parameter buffer
//     (ping-pong: groups still reading the old rows are never overtaken); m, v in place.  No gradient table
//     exists at all (24 instead of 32 B per element of optimiser traffic).
//   * the three Frobenius norms of the NEXT batch -- the only grid-wide dependency of a step -- are accumulated
//     here from the freshly updated rows: sum_b |u_b|^2 = sum over rows of multiplicity(row, next batch) * |row|^2,
//     as per-block partials that every block of the next launch re-reduces in a fixed order.  The first step of
//     an epoch takes its norms from bpr_fwd_kernel's partials (same layout).
//   * the BPR loss of the batch is summed by the user rows (every triple is in exactly one user list), also as
//     block partials: the NEXT launch (or crh_mf_step_finish) turns them into the reported value.
// What bounds the kernel is the CHAIN of dependent memory round trips of a row, so the plan is flattened once per
// epoch (mf_tables_kernel) into per-row arrays the kernel can read at once: range[b][row] = the row's slice of the
// batch's entry array, entry = the two OTHER rows of the triple (ids resolved), mult[b][row] = the row's multiplicities
// in batch b.  A row then costs three round trips: {range, mult, p, m, v} -> entries -> gathered rows.
// Heavy rows (more than BPR_HEAVY entries) are done by the extra blocks, one block per row, exactly as before.
// Deterministic: no atomics, fixed summation orders.  d <= 256, batch < 32768.
#ifndef MF_HEAVY_BLOCKS_N
#define MF_HEAVY_BLOCKS_N 96
#endif
#ifndef MF_ROWS_N
#define MF_ROWS_N 2
#endif
#ifndef MF_NQ_N
#define MF_NQ_N 2          // entries whose gathered rows are in flight together in the one-launch step (2 or 4)
#endif
#ifndef MF_OCC_N
#define MF_OCC_N 4         // waves per SIMD the one-launch step is compiled for
#endif
constexpr int MF_MAX_LIGHT = 2048, MF_HEAVY_BLOCKS = MF_HEAVY_BLOCKS_N, MF_ROWS = MF_ROWS_N;     // (tuning builds: -D..._N)

struct MfStepArgs {
    const float* pin;      // (U + I, d) parameters before the step, users first
    float* pout;           // parameters after the step (another buffer)
    float* m;
    float* v;
    int64_t U, I;
    int d;
    int64_t B;
    float reg;
    const int32_t* plan;   // this batch (heavy list)
    const int2* range;     // (U + I): [e0, e1) of the row in `entries`, (0, 0) if the batch does not touch it
    const int2* entries;   // (3 L): x = first other row | role << 30, y = second other row (table rows, users first)
    const int32_t* mult2;  // (U + I) multiplicities in the NEXT batch: user rows count, item rows pos | neg << 16; or NULL
    const float* part_in;  // [n_in][4]: norms^2 of THIS batch (u, p, n) and the loss sum of the PREVIOUS one
    int n_in;
    float* part_out;       // [gridDim.x][4]: norms^2 of the NEXT batch, loss sum of THIS one
    float* loss_prev;      // [2] of the previous step (its bpr is known only now) or NULL
    float inv_b_prev;
    float* loss_now;       // [2] of this step: l2 is written now
    AdamK k;
    const float* step_scalars;
    float neg_lr;          // SGD variant (OPT = 1): p <- fma(-lr, g, p); m, v, step_scalars unused
    int light_blocks;
    int ablate;            // measurement only (CRH_MF_ABLATE): 1 no entries, 2 no batch sums, 4 no norms, 8 no stores,
};                         // 16 per-block clocks of the launch -> crh_profile_mf_clocks (profile build only)

#ifdef CRH_PROFILE
__device__ unsigned long long crh_mf_clk[2 * 4096];   // [block]{start, end} of s_memrealtime (100 MHz), last launch
#endif

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) {
#pragma clang fp contract(off)      // the same bits wherever a score difference is recomputed
    return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w;
}

// NQ consecutive entries (lanes t .. t+NQ-1 of the group hold their metadata) of one row: acc += d(loss)/d(row)
// contributions, loss += -log(1e-5 + sigmoid(x)) on the user side.
//   user row (own = u): entry rows (p, n);  item row as positive (own = p): (u, n);  as negative (own = n): (u, p)
template <int G, int NQ>
__device__ __forceinline__ void mf_entries_chunk(const MfStepArgs& a, const BwdCoef& k, bool user_side, const int2& en, int t,
                                                 bool on, int lig, const f32x4& own, f32x4& acc, float& loss) {
    int aa[NQ], bb[NQ];
    f32x4 xa[NQ], xb[NQ];
    float dp[NQ], dn[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        aa[q] = __shfl(en.x, t + q, G);
        bb[q] = __shfl(en.y, t + q, G);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        xa[q] = xb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (on) {
            xa[q] = reinterpret_cast<const f32x4*>(a.pin + (int64_t)(aa[q] & 0x3fffffff) * a.d)[lig];
            xb[q] = reinterpret_cast<const f32x4*>(a.pin + (int64_t)bb[q] * a.d)[lig];
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // x = u.p - u.n with the same per-lane products and the same group tree on all three rows of a triple
        if (user_side) {                    // own = u, xa = p, xb = n
            dp[q] = dot4(own, xa[q]);
            dn[q] = dot4(own, xb[q]);
        } else if ((aa[q] >> 30) == 0) {    // own = p, xa = u, xb = n
            dp[q] = dot4(xa[q], own);
            dn[q] = dot4(xa[q], xb[q]);
        } else {                            // own = n, xa = u, xb = p
            dp[q] = dot4(xa[q], xb[q]);
            dn[q] = dot4(xa[q], own);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        dp[q] = group_sum<G>(dp[q]);
        dn[q] = group_sum<G>(dn[q]);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float x = dp[q] - dn[q];
        const float sig = 1.0f / (1.0f + expf(-x));
        const float g = -k.invB * sig * (1.0f - sig) / (1e-5f + sig);
        if (user_side) {
            loss += -logf(1e-5f + sig);
            acc.x += g * (xa[q].x - xb[q].x) + k.cu * own.x;
            acc.y += g * (xa[q].y - xb[q].y) + k.cu * own.y;
            acc.z += g * (xa[q].z - xb[q].z) + k.cu * own.z;
            acc.w += g * (xa[q].w - xb[q].w) + k.cu * own.w;
        } else {
            const bool pos = (aa[q] >> 30) == 0;
            const float sg = pos ? g : -g;
            const float cc = pos ? k.cp : k.cn;
            acc.x += sg * xa[q].x + cc * own.x;
            acc.y += sg * xa[q].y + cc * own.y;
            acc.z += sg * xa[q].z + cc * own.z;
            acc.w += sg * xa[q].w + cc * own.w;
        }
    }
}

// Entries [e0, e1) of one row, in list order: metadata one entry per lane, rows fetched 2 entries at a time (most rows
// of a batch have one or two entries; 4 at a time costs the registers of a fourth wave per SIMD).
template <int G, int NQM = 2>
__device__ __forceinline__ void mf_row_entries(const MfStepArgs& a, const BwdCoef& k, bool user_side, int e0, int e1,
                                               bool on, int lig, const f32x4& own, f32x4& acc, float& loss) {
    for (int base = e0; base < e1; base += G) {
        const int e = base + lig;
        int2 en = {0, 0};
        if (e < e1) en = a.entries[e];
        const int cnt = (e1 - base) < G ? (e1 - base) : G;
        int t = 0;
        if constexpr (NQM >= 4 && G >= 4)
            for (; t + 4 <= cnt; t += 4) mf_entries_chunk<G, 4>(a, k, user_side, en, t, on, lig, own, acc, loss);
        for (; t + 2 <= cnt; t += 2) mf_entries_chunk<G, 2>(a, k, user_side, en, t, on, lig, own, acc, loss);
        if (t < cnt) mf_entries_chunk<G, 1>(a, k, user_side, en, t, on, lig, own, acc, loss);
    }
}

// Adam on one row slice + what the updated row contributes to the norms of the next batch.
template <int G, int OPT>
__device__ __forceinline__ void mf_row_update(const MfStepArgs& a, int64_t row, bool on, int lig, f32x4 p, f32x4 m, f32x4 v,
                                              const f32x4& grad, int mult, float bc2_sqrt, float nss, float& su, float& sp,
                                              float& sn) {
    const int64_t o = row * a.d + lig * 4;
    if (on) {
        if constexpr (OPT == 0) adam_elem4(p, m, v, grad, a.k, bc2_sqrt, nss);
        else sgd_elem4(p, grad, a.neg_lr);
        if (!(CRH_ABLATE(a.ablate) & 8) || p.x == 123.f) {
            *reinterpret_cast<f32x4*>(a.pout + o) = p;
            if constexpr (OPT == 0) {
                *reinterpret_cast<f32x4*>(a.m + o) = m;
                *reinterpret_cast<f32x4*>(a.v + o) = v;
            }
        }
    }
    if (mult == 0) return;                                       // uniform over the lane group
    const float nsq = group_sum<G>(on ? dot4(p, p) : 0.f);
    if (lig == 0) {
        if (row < a.U) {
            su += (float)mult * nsq;
        } else {
            sp += (float)(mult & 0xffff) * nsq;
            sn += (float)(mult >> 16) * nsq;
        }
    }
}

// OPT = 0: torch.optim.Adam (m, v in place); OPT = 1: plain SGD -- no optimiser state at all, a row costs one read and
// one write of p (8 B per element instead of 24)
template <int G, int OPT>
__global__ __launch_bounds__(BPR_THREADS, MF_OCC_N) void mf_step_kernel(MfStepArgs a) {
    __shared__ f32x4 red4[4];
    __shared__ float red[4];
    __shared__ f32x4 wsum[4][G];
#ifdef CRH_PROFILE
    if ((a.ablate & 16) && threadIdx.x == 0 && blockIdx.x < 4096) crh_mf_clk[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    const int lig = threadIdx.x % G;
    const bool on = lig < (a.d >> 2);
    const int64_t R = a.U + a.I;
    // heavy rows take the longest: their blocks come FIRST in the grid so that they start with the launch
    const int heavy_blocks = (int)gridDim.x - a.light_blocks;
    const bool light = (int)blockIdx.x >= heavy_blocks;
    const int64_t gid = (int64_t)((int)blockIdx.x - heavy_blocks) * (BPR_THREADS / G) + threadIdx.x / G;
    const int64_t gstride = (int64_t)a.light_blocks * (BPR_THREADS / G);
    const int64_t n_work = R;                                      // the light work items are the rows of the table
    auto work_row = [&](int64_t w, int2& range_out) -> int64_t {
        if (!(CRH_ABLATE(a.ablate) & 1)) range_out = a.range[w];
        return w;
    };
    // first round trip of the row chain, issued before the batch sums are reduced
    int2 rg = {0, 0};
    int mult = 0;
    int64_t row0 = 0;
    f32x4 own = {0.f, 0.f, 0.f, 0.f}, m0 = own, v0 = own;
    if (light && gid < n_work) {
        row0 = work_row(gid, rg);
        if (a.mult2 && !(CRH_ABLATE(a.ablate) & 4)) mult = a.mult2[row0];
        if (on) {
            const int64_t o = row0 * a.d + lig * 4;
            own = *reinterpret_cast<const f32x4*>(a.pin + o);
            if constexpr (OPT == 0) {
                m0 = *reinterpret_cast<const f32x4*>(a.m + o);
                v0 = *reinterpret_cast<const f32x4*>(a.v + o);
            }
        }
    }
    // batch sums: every block reduces the previous launch's partials in the same order (one 16-B load per partial)
    f32x4 tot;
    {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int i = threadIdx.x; i < ((CRH_ABLATE(a.ablate) & 2) ? 1 : a.n_in); i += BPR_THREADS) {
            const f32x4 x = reinterpret_cast<const f32x4*>(a.part_in)[i];
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s.x += __shfl_xor(s.x, off); s.y += __shfl_xor(s.y, off);
            s.z += __shfl_xor(s.z, off); s.w += __shfl_xor(s.w, off);
        }
        if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = s;
        __syncthreads();
        const f32x4 t0 = red4[0], t1 = red4[1], t2 = red4[2], t3 = red4[3];
        tot.x = (t0.x + t1.x) + (t2.x + t3.x);
        tot.y = (t0.y + t1.y) + (t2.y + t3.y);
        tot.z = (t0.z + t1.z) + (t2.z + t3.z);
        tot.w = (t0.w + t1.w) + (t2.w + t3.w);
    }
    BwdCoef k;
    k.invB = 1.0f / (float)a.B;
    {
        const float nu_ = sqrtf(tot.x), np_ = sqrtf(tot.y), nn = sqrtf(tot.z);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (a.loss_now) a.loss_now[1] = a.reg * (nu_ * k.invB + np_ * k.invB + nn * k.invB);
            if (a.loss_prev) a.loss_prev[0] = tot.w * a.inv_b_prev;
        }
        k.cu = nu_ > 0.f ? a.reg * k.invB / nu_ : 0.f;
        k.cp = np_ > 0.f ? a.reg * k.invB / np_ : 0.f;
        k.cn = nn > 0.f ? a.reg * k.invB / nn : 0.f;
    }
    float bc2_sqrt = 0.f, nss = 0.f;
    if constexpr (OPT == 0) {
        bc2_sqrt = a.step_scalars[0];
        nss = a.step_scalars[1];
    }
    float su = 0.f, sp = 0.f, sn = 0.f, sl = 0.f;
    if (light) {
        for (int64_t w = gid; w < n_work; w += gstride) {
            int64_t row = row0;
            if (w != gid) {                                        // tables beyond MF_MAX_LIGHT blocks of rows
                row = work_row(w, rg);
                mult = a.mult2 ? a.mult2[row] : 0;
                if (on) {
                    const int64_t o = row * a.d + lig * 4;
                    own = *reinterpret_cast<const f32x4*>(a.pin + o);
                    if constexpr (OPT == 0) {
                        m0 = *reinterpret_cast<const f32x4*>(a.m + o);
                        v0 = *reinterpret_cast<const f32x4*>(a.v + o);
                    }
                }
            }
            if (rg.y - rg.x > BPR_HEAVY) continue;                // a heavy block does this row, Adam included
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float loss = 0.f;
            if (rg.y > rg.x) mf_row_entries<G, MF_NQ_N>(a, k, row < a.U, rg.x, rg.y, on, lig, own, acc, loss);
            if (lig == 0) sl += loss;
            mf_row_update<G, OPT>(a, row, on, lig, own, m0, v0, acc, mult, bc2_sqrt, nss, su, sp, sn);
        }
    } else {
        const PlanView pv = plan_view(a.plan);
        constexpr int NGB = BPR_THREADS / G;
        const int gg = threadIdx.x / G;
        for (int h = (int)blockIdx.x; h < pv.n_heavy; h += heavy_blocks) {
            const int64_t w = pv.heavy[h];
            const bool user_side = w < pv.n_u;
            const int64_t row = user_side ? (int64_t)pv.urow[w] : a.U + pv.irow[w - pv.n_u];
            const int2 hr = a.range[row];
            int chunk = (hr.y - hr.x + NGB - 1) / NGB;
            chunk = (chunk + 3) & ~3;
            const int e0 = hr.x + gg * chunk;
            const int e1 = e0 + chunk < hr.y ? e0 + chunk : hr.y;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            own = acc;
            if (on) own = reinterpret_cast<const f32x4*>(a.pin + row * a.d)[lig];
            float loss = 0.f;
            if (e0 < hr.y) mf_row_entries<G, MF_NQ_N>(a, k, user_side, e0, e1, on, lig, own, acc, loss);
            if (lig == 0) sl += loss;
#pragma unroll
            for (int off = G; off < 64; off <<= 1) {
                acc.x += __shfl_down(acc.x, off);
                acc.y += __shfl_down(acc.y, off);
                acc.z += __shfl_down(acc.z, off);
                acc.w += __shfl_down(acc.w, off);
            }
            __syncthreads();
            if ((threadIdx.x & 63) < G) wsum[threadIdx.x >> 6][lig] = acc;
            __syncthreads();
            if (threadIdx.x < G) {
                const f32x4 t0 = wsum[0][lig], t1 = wsum[1][lig], t2 = wsum[2][lig], t3 = wsum[3][lig];
                f32x4 r;
                r.x = (t0.x + t1.x) + (t2.x + t3.x);
                r.y = (t0.y + t1.y) + (t2.y + t3.y);
                r.z = (t0.z + t1.z) + (t2.z + t3.z);
                r.w = (t0.w + t1.w) + (t2.w + t3.w);
                m0 = v0 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (OPT == 0 && on) {
                    m0 = *reinterpret_cast<const f32x4*>(a.m + row * a.d + lig * 4);
                    v0 = *reinterpret_cast<const f32x4*>(a.v + row * a.d + lig * 4);
                }
                mf_row_update<G, OPT>(a, row, on, lig, own, m0, v0, r, a.mult2 ? a.mult2[row] : 0, bc2_sqrt, nss, su, sp, sn);
            }
        }
    }
    su = block_sum(su, red);
    sp = block_sum(sp, red);
    sn = block_sum(sn, red);
    sl = block_sum(sl, red);
    if (threadIdx.x == 0) {
        float* o = a.part_out + (size_t)blockIdx.x * 4;
        o[0] = su; o[1] = sp; o[2] = sn; o[3] = sl;
#ifdef CRH_PROFILE
        if ((a.ablate & 16) && blockIdx.x < 4096) crh_mf_clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    }
}

#ifdef CRH_PROFILE
// profile build only (not part of the ABI): the per-block clocks of the last mf_step launch
extern "C" int crh_profile_mf_clocks(unsigned long long* out_host, int n_blocks) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(crh_mf_clk), (size_t)n_blocks * 16) == hipSuccess ? 0 : -1;
}
#endif

// Flatten the plans of an epoch for mf_step_kernel (grid.y = batch): per table row its entry range and its
// multiplicities, per entry the two other rows of the triple.  range / mult must be zero on entry.
__global__ void mf_tables_kernel(const int32_t* plans, int64_t stride, const int32_t* iu, const int32_t* ip,
                                 const int32_t* in_, int64_t L, int64_t U, int64_t R, int2* range, int32_t* mult,
                                 int2* entries) {
    const int64_t bt = blockIdx.y;
    const PlanView pv = plan_view(plans + bt * stride);
    const int32_t *bu = iu + bt * L, *bp = ip + bt * L, *bn = in_ + bt * L;
    int2* rg = range + bt * R;
    int32_t* mu = mult + bt * R;
    int2* en = entries + bt * 3 * L;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < pv.n_u + pv.n_i; w += gridDim.x * blockDim.x) {
        if (w < pv.n_u) {
            const int e0 = pv.uptr[w], e1 = pv.uptr[w + 1];
            rg[pv.urow[w]] = int2{e0, e1};
            mu[pv.urow[w]] = e1 - e0;
            for (int e = e0; e < e1; ++e) {
                const int b = pv.ulist[e];
                en[e] = int2{(int)(U + bp[b]), (int)(U + bn[b])};
            }
        } else {
            const int wi = w - pv.n_u;
            const int e0 = pv.iptr[wi], e1 = pv.iptr[wi + 1];
            int negs = 0;
            for (int e = e0; e < e1; ++e) {
                const int ent = pv.ilist[e], b = ent & 0x3fffffff, role = ent >> 30;
                negs += role;
                en[L + e] = int2{bu[b] | (role << 30), (int)(U + (role == 0 ? bn[b] : bp[b]))};
            }
            rg[U + pv.irow[wi]] = int2{(int)L + e0, (int)L + e1};
            mu[U + pv.irow[wi]] = (e1 - e0 - negs) | (negs << 16);
        }
    }
}

__global__ void mf_finish_kernel(const float* part_in, int n_in, float inv_b, float* loss_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_in; i += BPR_THREADS) s += part_in[(size_t)i * 4 + 3];
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss_out[0] = s * inv_b;
}

}  // namespace

extern "C" int64_t crh_bpr_plan_ints(int64_t batch) { return batch > 0 ? plan_ints(batch) : 0; }
extern "C" int crh_bpr_heavy_threshold(void) { return BPR_HEAVY; }

// (Re)build the heavy-row lists of n_batches plans whose header and row lists are in place.
static size_t crh_bpr_plan_heavy_workspace_bytes(int64_t n_batches) {   // (internal: crh_bpr_plan_build_large)
    return n_batches > 0 ? (size_t)n_batches * HV_BLOCKS * sizeof(int32_t) : 0;
}

static int crh_bpr_plan_heavy_lists(int32_t* plans, int64_t n_batches, int64_t layout_batch, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    CRH_CHECK_ARG(plans && n_batches > 0 && n_batches <= 65535 && layout_batch > 0, "crh_bpr_plan_heavy_lists: bad arguments");
    if (!workspace || workspace_bytes < crh_bpr_plan_heavy_workspace_bytes(n_batches)) {
        crh_set_error("crh_bpr_plan_heavy_lists: workspace %zu < %zu bytes", workspace_bytes,
                      crh_bpr_plan_heavy_workspace_bytes(n_batches));
        return CRH_ERR_WS;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(plan_heavy_pass_kernel, dim3(HV_BLOCKS, (unsigned)n_batches), dim3(HV_THREADS), 0, st, plans,
                           plan_ints(layout_batch), reinterpret_cast<int32_t*>(workspace), pass);
        CRH_HIP(hipGetLastError());
    }
    return CRH_OK;
}

// HOST function: reverse index of one batch of triples (see bpr_bwd_rows_kernel).  Item-side
// gradients of positives and negatives land in the SAME table (grad_pos == grad_neg).
extern "C" int crh_bpr_plan_build_host(const int32_t* user_idx_host, const int32_t* pos_idx_host,
                                       const int32_t* neg_idx_host, int64_t batch, int64_t layout_batch,
                                       int32_t* plan_out_host) {
    CRH_CHECK_ARG(user_idx_host && pos_idx_host && neg_idx_host && plan_out_host && batch > 0,
                  "crh_bpr_plan_build_host: bad arguments");
    CRH_CHECK_ARG(layout_batch >= batch && layout_batch < (1 << 30), "crh_bpr_plan_build_host: bad layout batch");
    const int64_t B = batch, L = layout_batch;
    int32_t* pl = plan_out_host;
    pl[2] = (int32_t)L;
    int32_t* urow = pl + 3;
    int32_t* uptr = urow + L;
    int32_t* ulist = uptr + (L + 1);
    int32_t* irow = ulist + L;
    int32_t* iptr = irow + 2 * L;
    int32_t* ilist = iptr + (2 * L + 1);
    std::vector<uint64_t> keys((size_t)2 * B);
    for (int64_t b = 0; b < B; ++b) keys[b] = ((uint64_t)(uint32_t)user_idx_host[b] << 32) | (uint32_t)b;
    std::sort(keys.begin(), keys.begin() + B);
    int nu = 0;
    for (int64_t e = 0; e < B; ++e) {
        const int32_t row = (int32_t)(keys[e] >> 32);
        if (e == 0 || row != urow[nu - 1]) { urow[nu] = row; uptr[nu] = (int32_t)e; ++nu; }
        ulist[e] = (int32_t)(keys[e] & 0xffffffffu);
    }
    uptr[nu] = (int32_t)B;
    for (int64_t b = 0; b < B; ++b) {
        keys[2 * b] = ((uint64_t)(uint32_t)pos_idx_host[b] << 32) | (uint32_t)b;
        keys[2 * b + 1] = ((uint64_t)(uint32_t)neg_idx_host[b] << 32) | (uint32_t)b | (1u << 30);
    }
    std::sort(keys.begin(), keys.end());      // inside a row: positives (ascending b), then negatives
    int ni = 0;
    for (int64_t e = 0; e < 2 * B; ++e) {
        const int32_t row = (int32_t)(keys[e] >> 32);
        if (e == 0 || row != irow[ni - 1]) { irow[ni] = row; iptr[ni] = (int32_t)e; ++ni; }
        ilist[e] = (int32_t)(keys[e] & 0xffffffffu);
    }
    iptr[ni] = (int32_t)(2 * B);
    pl[0] = nu;
    pl[1] = ni;
    int32_t* hv = pl + plan_heavy_off(L);
    int nh = 0;
    for (int r = 0; r < nu; ++r)
        if (uptr[r + 1] - uptr[r] > BPR_HEAVY) hv[1 + nh++] = r;
    for (int r = 0; r < ni; ++r)
        if (iptr[r + 1] - iptr[r] > BPR_HEAVY) hv[1 + nh++] = nu + r;
    hv[0] = nh;
    return CRH_OK;
}

// Device: plans of every batch of an epoch in one launch (batch_size <= 8192; the sort runs in LDS).
extern "C" int crh_bpr_plan_build(const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx,
                                  int64_t n_records, int64_t batch_size, int32_t* plans_out, void* stream) {
    CRH_CHECK_ARG(user_idx && pos_idx && neg_idx && plans_out && n_records > 0, "crh_bpr_plan_build: bad arguments");
    CRH_CHECK_ARG(batch_size >= 1 && batch_size <= 8192, "crh_bpr_plan_build: batch_size=%lld outside 1..8192 "
                  "(larger batches: crh_bpr_plan_build_large)", (long long)batch_size);
    int P = 2;
    while (P < 2 * batch_size) P <<= 1;
    const size_t lds = plan_padded_keys(P) * 8 + (2 * PLAN_THREADS + 2) * sizeof(int);
    if (lds > 64 * 1024)
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bpr_plan_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t nb = (n_records + batch_size - 1) / batch_size;
    hipLaunchKernelGGL(bpr_plan_kernel, dim3((unsigned)nb, 2), dim3(PLAN_THREADS), lds,
                       reinterpret_cast<hipStream_t>(stream), user_idx, pos_idx, neg_idx, n_records, (int)batch_size,
                       P, plans_out, plan_ints(batch_size));
    CRH_HIP(hipGetLastError());
    hipLaunchKernelGGL(bpr_plan_heavy_kernel, dim3((unsigned)nb), dim3(PLAN_THREADS), 0,
                       reinterpret_cast<hipStream_t>(stream), (int)batch_size, plans_out, plan_ints(batch_size));
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

namespace {
int64_t lp_padded(int64_t width) {
    const int64_t c = (width + LP_CHUNK - 1) / LP_CHUNK;
    int64_t p = 1;
    while (p < c) p <<= 1;
    return p * LP_CHUNK;
}
size_t lp_align(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

// Workspace of crh_bpr_plan_build_large: two key buffers of the (padded) item side per batch, the per-block segment
// counts, the heavy-list counts.
extern "C" size_t crh_bpr_plan_build_large_workspace_bytes(int64_t n_records, int64_t batch_size) {
    if (n_records <= 0 || batch_size <= 0) return 0;
    const int64_t nb = (n_records + batch_size - 1) / batch_size;
    return 2 * lp_align((size_t)nb * lp_padded(2 * batch_size) * sizeof(lpkey)) +
           lp_align((size_t)nb * LP_MAX_EBLOCKS * sizeof(int32_t)) + lp_align(crh_bpr_plan_heavy_workspace_bytes(nb));
}

// Device: plans of every batch of an epoch for ANY batch size up to 524 288 (several workgroups per batch: chunk sorts in
// LDS, merge passes in global memory; see plan_chunk_sort_kernel).  Same layout and contents as crh_bpr_plan_build.
extern "C" int crh_bpr_plan_build_large(const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx,
                                        int64_t n_records, int64_t batch_size, int32_t* plans_out, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    CRH_CHECK_ARG(user_idx && pos_idx && neg_idx && plans_out && n_records > 0, "crh_bpr_plan_build_large: bad arguments");
    CRH_CHECK_ARG(batch_size >= 1 && lp_padded(2 * batch_size) <= (int64_t)LP_EMIT * LP_MAX_EBLOCKS,
                  "crh_bpr_plan_build_large: batch_size=%lld outside 1..%lld", (long long)batch_size,
                  (long long)LP_EMIT * LP_MAX_EBLOCKS / 2);
    const int64_t L = batch_size, nb = (n_records + L - 1) / L;
    CRH_CHECK_ARG(nb <= 65535, "crh_bpr_plan_build_large: %lld batches in one call (at most 65535)", (long long)nb);
    const size_t need = crh_bpr_plan_build_large_workspace_bytes(n_records, batch_size);
    if (!workspace || workspace_bytes < need) {
        crh_set_error("crh_bpr_plan_build_large: workspace %zu < %zu bytes", workspace_bytes, need);
        return CRH_ERR_WS;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t key_b = lp_align((size_t)nb * lp_padded(2 * L) * sizeof(lpkey));
    char* wsp = reinterpret_cast<char*>(workspace);
    lpkey* buf[2] = {reinterpret_cast<lpkey*>(wsp), reinterpret_cast<lpkey*>(wsp + key_b)};
    int32_t* counts = reinterpret_cast<int32_t*>(wsp + 2 * key_b);
    void* heavy_ws = wsp + 2 * key_b + lp_align((size_t)nb * LP_MAX_EBLOCKS * sizeof(int32_t));
    const int64_t stride = plan_ints(L);
    CRH_HIP(hipMemsetAsync(plans_out, 0, (size_t)nb * stride * sizeof(int32_t), st));
    for (int side = 0; side < 2; ++side) {
        const int64_t P = lp_padded(side ? 2 * L : L);
        hipLaunchKernelGGL(plan_chunk_sort_kernel, dim3((unsigned)(P / LP_CHUNK), (unsigned)nb), dim3(PLAN_THREADS), 0, st,
                           user_idx, pos_idx, neg_idx, n_records, L, side, P, buf[0]);
        CRH_HIP(hipGetLastError());
        int cur = 0;
        for (int64_t run = LP_CHUNK; run < P; run <<= 1) {
            hipLaunchKernelGGL(plan_merge_pass_kernel, dim3((unsigned)((P / LP_VT + 255) / 256), (unsigned)nb), dim3(256), 0,
                               st, buf[cur], buf[cur ^ 1], P, run);
            CRH_HIP(hipGetLastError());
            cur ^= 1;
        }
        const int n_eblocks = (int)((P + LP_EMIT - 1) / LP_EMIT);
        for (int write = 0; write < 2; ++write) {
            hipLaunchKernelGGL(plan_emit_large_kernel, dim3((unsigned)n_eblocks, (unsigned)nb), dim3(PLAN_THREADS), 0, st,
                               buf[cur], P, n_records, L, side, plans_out, stride, counts, n_eblocks, write);
            CRH_HIP(hipGetLastError());
        }
    }
    return crh_bpr_plan_heavy_lists(plans_out, nb, L, heavy_ws, crh_bpr_plan_heavy_workspace_bytes(nb), stream);
}

extern "C" size_t crh_bpr_workspace_bytes(int64_t batch) {
    if (batch <= 0) return 0;
    // forward partials, per-triple score differences
    return (size_t)BPR_MAX_BLOCKS * 16 + (((size_t)batch * 4 + 255) & ~(size_t)255) + 256;
}

namespace {

// phases: 1 = forward (partials + xbuf), 2 = backward (needs the xbuf of a forward over the same workspace)
int bpr_launch(int phases, const float* user_table, const float* pos_table, const float* neg_table, int d,
               const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx, int64_t batch,
               int64_t global_batch, float reg, float* grad_user, float* grad_pos, float* grad_neg,
               float* loss_out, const int32_t* plan, const float* totals, float* sums_out, void* workspace,
               size_t workspace_bytes, void* stream, const char* who, int own_mod = 1, int own_rem = 0) {
    CRH_CHECK_ARG(user_table && pos_table && neg_table, "%s: NULL table", who);
    CRH_CHECK_ARG(batch > 0 && global_batch >= batch, "%s: empty batch / global batch smaller than the local one", who);
    CRH_CHECK_ARG(d >= 4 && d % 4 == 0, "%s: d=%d must be a positive multiple of 4", who, d);
    CRH_CHECK_ARG((grad_user == nullptr) == (grad_pos == nullptr) && (grad_pos == nullptr) == (grad_neg == nullptr),
                  "%s: give all three gradient tables or none", who);
    CRH_CHECK_ARG((((uintptr_t)user_table | (uintptr_t)pos_table | (uintptr_t)neg_table | (uintptr_t)grad_user |
                    (uintptr_t)grad_pos | (uintptr_t)grad_neg) & 15) == 0,
                  "%s: tables must be 16-byte aligned", who);
    CRH_CHECK_ARG(!plan || d <= 256, "%s: the deterministic (plan) backward handles d <= 256, got %d", who, d);
    CRH_CHECK_ARG(!plan || (grad_user && user_idx && pos_idx && neg_idx && grad_pos == grad_neg && pos_table == neg_table),
                  "%s: a plan needs index arrays, gradient tables and one shared item table", who);
    if (!workspace || workspace_bytes < crh_bpr_workspace_bytes(batch)) {
        crh_set_error("%s: workspace %zu < %zu bytes", who, workspace_bytes, crh_bpr_workspace_bytes(batch));
        return CRH_ERR_WS;
    }
    CRH_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "%s: the workspace must be 16-byte aligned", who);
    BprArgs a;
    a.tu = user_table; a.tp = pos_table; a.tn = neg_table;
    a.iu = user_idx; a.ip = pos_idx; a.in_ = neg_idx;
    a.B = batch; a.d = d; a.reg = reg;
    a.gu = grad_user; a.gp = grad_pos; a.gn = grad_neg;
    a.partials = reinterpret_cast<float*>(workspace);
    a.xbuf = a.partials + (size_t)BPR_MAX_BLOCKS * 4;
    a.loss_out = loss_out;
    a.plan = plan;
    a.totals = totals;
    a.sums_out = sums_out;
    a.B_global = global_batch;
    a.own_mod = own_mod;
    a.own_rem = own_rem;
    const int G = pick_group(d);
    const int64_t per_block = BPR_THREADS / G;
    int64_t blocks = (batch + per_block - 1) / per_block;
    if (blocks > BPR_MAX_BLOCKS) blocks = BPR_MAX_BLOCKS;
    a.nblocks_fwd = (int)blocks;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dispatch_group(G, [&](auto gc) -> int {
        constexpr int GG = decltype(gc)::value;
        if (phases & 1) {
            hipLaunchKernelGGL(bpr_fwd_kernel<GG>, dim3((unsigned)blocks), dim3(BPR_THREADS), 0, st, a);
            CRH_HIP(hipGetLastError());
            if (sums_out) {
                BprArgs r = a;
                r.totals = nullptr;
                hipLaunchKernelGGL(bpr_sums_kernel, dim3(1), dim3(BPR_THREADS), 0, st, r);
                CRH_HIP(hipGetLastError());
            }
        }
        if (!(phases & 2)) return CRH_OK;
        const unsigned bwd_blocks = (grad_user || loss_out) ? (grad_user ? (unsigned)blocks : 1u) : 0u;
        if (plan) {
            int64_t hb = 3 * batch / BPR_HEAVY + 1;                     // worst case; surplus blocks exit at once
            if (hb > 512) hb = 512;
            // rows per lane group (grid-stride): the smallest count up to 4 that keeps the launch in one resident
            // round (256 CUs x 5 workgroups at this kernel's ~100 VGPRs); B = 4096: two rows, LightGCN step -4 us
            static const int force_rows = CRH_TUNE_ENV("CRH_BWD_ROWS") ? atoi(CRH_TUNE_ENV("CRH_BWD_ROWS")) : 0;
            int64_t rb = 0;
            for (int r = force_rows > 0 ? force_rows : 1; r <= (force_rows > 0 ? force_rows : 4); ++r) {
                rb = (3 * batch + per_block * r - 1) / (per_block * r);      // <= 3B touched rows
                if (rb + hb <= 256 * 5) break;
            }
            if (rb > BPR_MAX_BLOCKS) rb = BPR_MAX_BLOCKS;
            hipLaunchKernelGGL(bpr_bwd_rows_kernel<GG>, dim3((unsigned)(rb + hb)), dim3(BPR_THREADS), 0, st, a, (int)rb);
            CRH_HIP(hipGetLastError());
        } else if (bwd_blocks) {
            hipLaunchKernelGGL(bpr_bwd_kernel<GG>, dim3(bwd_blocks), dim3(BPR_THREADS), 0, st, a);
            CRH_HIP(hipGetLastError());
        }
        return CRH_OK;
    });
}

}  // namespace

extern "C" int crh_bpr_fwd_bwd_f32(const float* user_table, const float* pos_table, const float* neg_table,
                                   int d, const int32_t* user_idx, const int32_t* pos_idx,
                                   const int32_t* neg_idx, int64_t batch, float reg, float* grad_user,
                                   float* grad_pos, float* grad_neg, float* loss_out, const int32_t* plan,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    return bpr_launch(3, user_table, pos_table, neg_table, d, user_idx, pos_idx, neg_idx, batch, batch, reg,
                      grad_user, grad_pos, grad_neg, loss_out, plan, nullptr, nullptr, workspace, workspace_bytes,
                      stream, "crh_bpr_fwd_bwd_f32");
}

// Data-parallel split: forward over the rank's slice -> sums_out[4]; the caller all-reduces
// them; backward over the same slice with the global sums and the global batch size.
extern "C" int crh_bpr_fwd_f32(const float* user_table, const float* pos_table, const float* neg_table, int d,
                               const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx,
                               int64_t batch, float* sums_out, void* workspace, size_t workspace_bytes,
                               void* stream) {
    CRH_CHECK_ARG(sums_out, "crh_bpr_fwd_f32: NULL sums_out");
    return bpr_launch(1, user_table, pos_table, neg_table, d, user_idx, pos_idx, neg_idx, batch, batch, 0.f,
                      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sums_out, workspace, workspace_bytes,
                      stream, "crh_bpr_fwd_f32");
}

extern "C" int crh_bpr_bwd_f32(const float* user_table, const float* pos_table, const float* neg_table, int d,
                               const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx,
                               int64_t batch, int64_t global_batch, float reg, const float* sums,
                               float* grad_user, float* grad_pos, float* grad_neg, float* loss_out,
                               const int32_t* plan, void* workspace, size_t workspace_bytes, void* stream) {
    CRH_CHECK_ARG(sums, "crh_bpr_bwd_f32: NULL sums");
    return bpr_launch(2, user_table, pos_table, neg_table, d, user_idx, pos_idx, neg_idx, batch, global_batch, reg,
                      grad_user, grad_pos, grad_neg, loss_out, plan, sums, nullptr, workspace, workspace_bytes,
                      stream, "crh_bpr_bwd_f32");
}

// Row-ownership split of the deterministic backward (data-parallel touched-rows step): every rank holds the
// whole batch, its plan, the score differences of a forward over the whole batch (same workspace) and the batch sums; rank r
// sums and stores only the gradient rows of the plan's row slots w with w % own_mod == own_rem -- each row by ONE rank, in
// the plan's entry order, i.e. the very bits of the single-GPU launch -- and the ranks then exchange whole rows
// (crh_rows_pack_f32 -> all-gather -> crh_rows_unpack_f32).
extern "C" int crh_bpr_bwd_owned_f32(const float* user_table, const float* item_table, int d, const int32_t* user_idx,
                                     const int32_t* pos_idx, const int32_t* neg_idx, int64_t batch, float reg,
                                     const float* sums, float* grad_user, float* grad_item, float* loss_out,
                                     const int32_t* plan, int own_mod, int own_rem, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    CRH_CHECK_ARG(sums && plan, "crh_bpr_bwd_owned_f32: NULL sums / plan");
    CRH_CHECK_ARG(own_mod >= 1 && own_rem >= 0 && own_rem < own_mod, "crh_bpr_bwd_owned_f32: own_rem=%d outside 0..%d", own_rem,
                  own_mod - 1);
    return bpr_launch(2, user_table, item_table, item_table, d, user_idx, pos_idx, neg_idx, batch, batch, reg, grad_user,
                      grad_item, grad_item, loss_out, plan, sums, nullptr, workspace, workspace_bytes, stream,
                      "crh_bpr_bwd_owned_f32", own_mod, own_rem);
}

namespace {
// slot w of a plan -> table row (users first), or -1 past the plan's rows
__device__ __forceinline__ int64_t plan_slot_row(const PlanView& pv, int64_t w, int64_t user_rows) {
    if (w < pv.n_u) return pv.urow[w];
    if (w < (int64_t)pv.n_u + pv.n_i) return user_rows + pv.irow[w - pv.n_u];
    return -1;
}

// out_ids[m], out_rows[m] = the table row of plan slot own_rem + m * own_mod (m < cap; -1 / untouched past the plan's rows)
template <int G>
__global__ __launch_bounds__(256) void rows_pack_kernel(const float* __restrict__ table, const int32_t* __restrict__ plan,
                                                        int64_t user_rows, int own_mod, int own_rem, int64_t cap, int d,
                                                        int32_t* __restrict__ out_ids, float* __restrict__ out_rows) {
    const PlanView pv = plan_view(plan);
    const int lig = threadIdx.x % G;
    const int64_t m = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (m >= cap) return;
    const int64_t row = plan_slot_row(pv, own_rem + m * own_mod, user_rows);
    if (lig == 0) out_ids[m] = (int32_t)row;
    if (row < 0) return;
    for (int c = lig; c < (d >> 2); c += G)
        reinterpret_cast<f32x4*>(out_rows + m * d)[c] = reinterpret_cast<const f32x4*>(table + row * d)[c];
}

template <int G>
__global__ __launch_bounds__(256) void rows_unpack_kernel(float* __restrict__ table, const int32_t* __restrict__ ids,
                                                          const float* __restrict__ rows, int64_t n, int d) {
    const int lig = threadIdx.x % G;
    const int64_t m = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (m >= n) return;
    const int64_t row = ids[m];
    if (row < 0) return;
    for (int c = lig; c < (d >> 2); c += G)
        reinterpret_cast<f32x4*>(table + row * d)[c] = reinterpret_cast<const f32x4*>(rows + m * d)[c];
}
}  // namespace

// The exchange format of the row-ownership split: `cap` = crh_rows_pack_cap(batch, own_mod) slots of (row id, d floats), the
// gradient rows a rank owns, in slot order; ids past the plan's rows are -1.  pack reads `table` (the gradient table) at the
// owned rows; unpack STORES rows into it (every row has exactly one owner: nothing is added), ids < 0 skipped.
extern "C" int64_t crh_rows_pack_cap(int64_t batch, int own_mod) {
    return batch > 0 && own_mod >= 1 ? (3 * batch + own_mod - 1) / own_mod : 0;
}

extern "C" int crh_rows_pack_f32(const float* table, const int32_t* plan, int64_t batch, int64_t user_rows, int d, int own_mod,
                                 int own_rem, int32_t* out_ids, float* out_rows, void* stream) {
    CRH_CHECK_ARG(table && plan && out_ids && out_rows, "crh_rows_pack_f32: NULL pointer");
    CRH_CHECK_ARG(batch > 0 && d >= 4 && d % 4 == 0 && d <= 256, "crh_rows_pack_f32: batch=%lld d=%d", (long long)batch, d);
    CRH_CHECK_ARG(own_mod >= 1 && own_rem >= 0 && own_rem < own_mod, "crh_rows_pack_f32: own_rem=%d outside 0..%d", own_rem, own_mod - 1);
    const int64_t cap = crh_rows_pack_cap(batch, own_mod);
    const int G = pick_group(d);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dispatch_group(G, [&](auto gc) -> int {
        constexpr int GG = decltype(gc)::value;
        const int64_t per_block = 256 / GG;
        hipLaunchKernelGGL(rows_pack_kernel<GG>, dim3((unsigned)((cap + per_block - 1) / per_block)), dim3(256), 0, st, table, plan,
                           user_rows, own_mod, own_rem, cap, d, out_ids, out_rows);
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    });
}

extern "C" int crh_rows_unpack_f32(float* table, const int32_t* ids, const float* rows, int64_t n, int d, void* stream) {
    CRH_CHECK_ARG(table && ids && rows, "crh_rows_unpack_f32: NULL pointer");
    CRH_CHECK_ARG(n >= 0 && d >= 4 && d % 4 == 0 && d <= 256, "crh_rows_unpack_f32: n=%lld d=%d", (long long)n, d);
    if (n == 0) return CRH_OK;
    const int G = pick_group(d);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dispatch_group(G, [&](auto gc) -> int {
        constexpr int GG = decltype(gc)::value;
        const int64_t per_block = 256 / GG;
        hipLaunchKernelGGL(rows_unpack_kernel<GG>, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), 0, st, table, ids, rows,
                           n, d);
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    });
}

extern "C" int crh_adam_dense_f32(float* p0, float* g0, float* m0, float* v0, int64_t n0, float* p1, float* g1,
                                  float* m1, float* v1, int64_t n1, double lr, double beta1, double beta2,
                                  double eps, int64_t step, int zero_grad, const float* step_scalars,
                                  void* stream) {
    CRH_CHECK_ARG(p0 && g0 && m0 && v0 && n0 > 0, "crh_adam_dense_f32: NULL / empty first tensor");
    CRH_CHECK_ARG(n1 == 0 || (p1 && g1 && m1 && v1), "crh_adam_dense_f32: NULL second tensor");
    CRH_CHECK_ARG(n0 % 4 == 0 && n1 % 4 == 0, "crh_adam_dense_f32: element counts must be multiples of 4");
    CRH_CHECK_ARG(step >= 1 || step_scalars, "crh_adam_dense_f32: step starts at 1");
    if (step < 1) step = 1;
    // scalar factors in double like torch (python floats), then cast
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    AdamSeg s0{p0, g0, m0, v0, n0 / 4}, s1{p1, g1, m1, v1, n1 / 4};
    const int64_t total = s0.n4 + s1.n4;
    int64_t blocks = (total + 255) / 256;
    // measured at S-TRAIN-XL (45 GB per launch): 2048 blocks 8.73 ms, 16384 blocks + nontemporal accesses 8.03 ms
    static const int max_blocks = CRH_TUNE_ENV("CRH_ADAM_BLOCKS") ? atoi(CRH_TUNE_ENV("CRH_ADAM_BLOCKS")) : 16384;
    if (blocks > max_blocks) blocks = max_blocks;
    static const int nt_mode = CRH_TUNE_ENV("CRH_ADAM_NT") ? atoi(CRH_TUNE_ENV("CRH_ADAM_NT")) : 1;
    const int zg = (zero_grad ? 1 : 0) | ((nt_mode && total * 16 > ((int64_t)256 << 20)) ? 2 : 0);
    hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       s0, s1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                       (float)(-(lr / bc1)), zg, step_scalars);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}


// torch.optim.SGD(lr) defaults on a dense tensor: p <- fma(-lr, g, p); zero_grad != 0 also clears g.
extern "C" int crh_sgd_dense_f32(float* p, float* g, int64_t n, double lr, int zero_grad, void* stream) {
    CRH_CHECK_ARG(p && g && n > 0 && n % 4 == 0, "crh_sgd_dense_f32: NULL tensor / element count not a multiple of 4");
    CRH_CHECK_ARG((((uintptr_t)p | (uintptr_t)g) & 15) == 0, "crh_sgd_dense_f32: tensors must be 16-byte aligned");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(sgd_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g,
                       n / 4, (float)(-lr), zero_grad);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// The same update on the rows of one batch's plan only (bit-identical tables: a zero gradient moves nothing);
// clears the gradient rows it consumed.  p, g: (n_rows, d) tables, users first, item rows offset by user_rows.
extern "C" int crh_sgd_rows_f32(float* p, float* g, int d, const int32_t* plan, int64_t batch, int64_t user_rows,
                                double lr, void* stream) {
    CRH_CHECK_ARG(p && g && plan && batch > 0, "crh_sgd_rows_f32: NULL pointer / empty batch");
    CRH_CHECK_ARG(d >= 4 && d % 4 == 0, "crh_sgd_rows_f32: d=%d must be a positive multiple of 4", d);
    const int G = pick_group(d);
    int64_t blocks = (3 * batch + (256 / G) - 1) / (256 / G);
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dispatch_group(G, [&](auto gc) -> int {
        constexpr int GG = decltype(gc)::value;
        hipLaunchKernelGGL(sgd_rows_kernel<GG>, dim3((unsigned)blocks), dim3(256), 0, st, p, g, d, plan, user_rows,
                           (float)(-lr));
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    });
}

// Touched-rows replay of dense Adam (see adam_rows_kernel).  p, g, m, v: (n_rows, d) fp32 tables, users first;
// last_step (n_rows) int32, all zero at the start (every row valid for "step 0"); plan: a batch's reverse index
// (crh_bpr_plan_build*), its item rows are offset by user_rows; scalar_table: device floats [2*(max_step+1)],
// entry s = {sqrt(1-beta2^s), -lr/(1-beta1^s)} (crh_adam_step_scalars_host); mode 0/1/2 = catch-up / step /
// flush.  Interleaving  catch-up(t) -> forward/backward(t) -> step(t)  for t = 1, 2, ... and a flush(T)
// before the tables are read reproduces crh_adam_dense_f32 applied T times bit for bit.
extern "C" int crh_adam_rows_f32(float* p, float* g, float* m, float* v, int32_t* last_step, int64_t n_rows, int d,
                                 const int32_t* plan, int64_t batch, int64_t user_rows, int64_t step,
                                 const float* scalar_table, double beta1, double beta2, double eps, int mode,
                                 void* stream) {
    CRH_CHECK_ARG(p && m && v && last_step && scalar_table && n_rows > 0, "crh_adam_rows_f32: NULL / empty table");
    CRH_CHECK_ARG(d >= 4 && d % 4 == 0, "crh_adam_rows_f32: d=%d must be a positive multiple of 4", d);
    CRH_CHECK_ARG(mode >= 0 && mode <= 2, "crh_adam_rows_f32: mode %d outside 0..2", mode);
    CRH_CHECK_ARG(mode == 2 || (plan && batch > 0), "crh_adam_rows_f32: modes 0/1 need the batch's plan");
    CRH_CHECK_ARG(mode != 1 || g, "crh_adam_rows_f32: mode 1 needs the gradient table");
    CRH_CHECK_ARG(step >= (mode == 2 ? 0 : 1), "crh_adam_rows_f32: step starts at 1");
    AdamRowsArgs a{p, g, m, v, last_step, n_rows, d, plan, user_rows, step, scalar_table,
                   AdamK{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps}, mode};
    const int G = pick_group(d);
    const int64_t work = mode == 2 ? n_rows : 3 * batch;
    int64_t blocks = (work + (256 / G) - 1) / (256 / G);
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dispatch_group(G, [&](auto gc) -> int {
        constexpr int GG = decltype(gc)::value;
        hipLaunchKernelGGL(adam_rows_kernel<GG>, dim3((unsigned)blocks), dim3(256), 0, st, a);
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    });
}

// HOST helper: the two step-dependent Adam factors {sqrt(1-beta2^step), -lr/(1-beta1^step)} exactly as
// crh_adam_dense_f32 derives them, for callers that keep them in device memory (graph replay).
// HOST helper: the same two factors for steps first_step .. first_step + n - 1 (out: n pairs).
extern "C" void crh_adam_step_scalars_range_host(double lr, double beta1, double beta2, int64_t first_step, int64_t n,
                                                 float* out_host) {
    for (int64_t i = 0; i < n; ++i) {
        const double bc1 = 1.0 - pow(beta1, (double)(first_step + i));
        const double bc2 = 1.0 - pow(beta2, (double)(first_step + i));
        out_host[2 * i] = (float)sqrt(bc2);
        out_host[2 * i + 1] = (float)(-(lr / bc1));
    }
}

extern "C" void crh_adam_step_scalars_host(double lr, double beta1, double beta2, int64_t step, float* out2_host) {
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    out2_host[0] = (float)sqrt(bc2);
    out2_host[1] = (float)(-(lr / bc1));
}


// ---------------------------------------------------------------- MF step in one launch (see mf_step_kernel)
extern "C" int crh_mf_step_parts(int64_t n_rows, int d) {
    if (n_rows <= 0 || d < 4 || d > 256 || d % 4) return 0;
    // MF_ROWS table rows per lane group (the second through the grid-stride loop): all blocks of a MovieLens-sized
    // step are then resident at once -- one round of 706 blocks instead of 1315 in two: 22.5 -> 19.5 us per step;
    // three rows: 21.4; fetching both rows' first round trip up front: no further gain
    const int64_t per_block = (BPR_THREADS / pick_group(d)) * MF_ROWS;
    int64_t light = (n_rows + per_block - 1) / per_block;
    if (light > MF_MAX_LIGHT) light = MF_MAX_LIGHT;
    return (int)light + MF_HEAVY_BLOCKS;
}

// number of [4]-float partial sums crh_bpr_fwd_f32 leaves at the start of its workspace for a batch
extern "C" int crh_bpr_fwd_parts(int64_t batch, int d) {
    if (batch <= 0 || d < 4 || d % 4) return 0;
    const int64_t per_block = BPR_THREADS / pick_group(d);
    const int64_t blocks = (batch + per_block - 1) / per_block;
    return (int)(blocks > BPR_MAX_BLOCKS ? BPR_MAX_BLOCKS : blocks);
}

// Flattened per-epoch tables for crh_mf_step_f32 from the epoch's plans and triples (n_records triples in batches of
// batch_size, the last one short): range_out (n_batches, rows) int2, mult_out (n_batches, rows) int32, entries_out
// (n_batches, 3 * batch_size) int2.  batch_size < 32768; user-row ids < 2^30.
extern "C" int crh_mf_step_tables(const int32_t* plans, const int32_t* user_idx, const int32_t* pos_idx,
                                  const int32_t* neg_idx, int64_t n_records, int64_t batch_size, int64_t user_rows,
                                  int64_t item_rows, int32_t* range_out, int32_t* mult_out, int32_t* entries_out,
                                  void* stream) {
    CRH_CHECK_ARG(plans && user_idx && pos_idx && neg_idx && range_out && mult_out && entries_out,
                  "crh_mf_step_tables: NULL pointer");
    CRH_CHECK_ARG(n_records > 0 && batch_size > 0 && batch_size < 32768 && user_rows > 0 && item_rows > 0 &&
                  user_rows + item_rows < ((int64_t)1 << 30), "crh_mf_step_tables: bad sizes");
    const int64_t nb = (n_records + batch_size - 1) / batch_size, R = user_rows + item_rows;
    CRH_CHECK_ARG(nb <= 65535, "crh_mf_step_tables: more than 65535 batches");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    CRH_HIP(hipMemsetAsync(range_out, 0, (size_t)(nb * R) * sizeof(int2), st));
    CRH_HIP(hipMemsetAsync(mult_out, 0, (size_t)(nb * R) * sizeof(int32_t), st));
    const unsigned bx = (unsigned)std::min<int64_t>((3 * batch_size + 255) / 256, 64);
    hipLaunchKernelGGL(mf_tables_kernel, dim3(bx, (unsigned)nb), dim3(256), 0, st, plans, plan_ints(batch_size), user_idx,
                       pos_idx, neg_idx, batch_size, user_rows, R, reinterpret_cast<int2*>(range_out), mult_out,
                       reinterpret_cast<int2*>(entries_out));
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// One optimiser step of model/MF.py:19-27 (gather, bpr_loss + l2_reg_loss, backward, dense Adam) in one launch.
// table_in/table_out: (user_rows + item_rows, d) parameters before / after (two different buffers); m, v in place.
// plan: this batch's plan (heavy-row list); range/entries: this batch's rows of crh_mf_step_tables' outputs;
// mult_next: the NEXT batch's row of mult_out, NULL for the last step of the epoch.  part_in: [n_parts_in][4]
// partial sums left by the previous call's part_out (n_parts_in = crh_mf_step_parts) or, for the first step of an
// epoch, by crh_bpr_fwd_f32's workspace (crh_bpr_fwd_parts x 4 floats).  loss_out[1] (l2) is written by this call,
// loss_out[0] (bpr) by the next call through loss_prev_out (batch_prev = that step's batch size) or by
// crh_mf_step_finish.
namespace {
int mf_step_run(const char* who, int opt, const float* table_in, float* table_out, float* m, float* v, int64_t user_rows,
                int64_t item_rows, int d, int64_t batch, float reg, const int32_t* plan, const int32_t* range,
                const int32_t* entries, const int32_t* mult_next, const float* part_in, int n_parts_in, float* part_out,
                float* loss_prev_out, int64_t batch_prev, float* loss_out, double beta1, double beta2, double eps,
                const float* step_scalars, double lr, void* stream) {
    CRH_CHECK_ARG(table_in && table_out && table_in != table_out, "%s: NULL / aliased tables", who);
    CRH_CHECK_ARG(opt != 0 || (m && v && step_scalars), "%s: NULL optimiser state / step scalars", who);
    CRH_CHECK_ARG(user_rows > 0 && item_rows > 0 && batch > 0, "%s: empty table or batch", who);
    CRH_CHECK_ARG(d >= 4 && d % 4 == 0 && d <= 256, "%s: d=%d must be a multiple of 4, at most 256", who, d);
    CRH_CHECK_ARG(plan && range && entries, "%s: NULL plan / step tables", who);
    CRH_CHECK_ARG(part_in && n_parts_in > 0 && part_out, "%s: NULL partial sums", who);
    CRH_CHECK_ARG(!loss_prev_out || batch_prev > 0, "%s: loss_prev_out needs batch_prev", who);
    CRH_CHECK_ARG((((uintptr_t)table_in | (uintptr_t)table_out | (uintptr_t)m | (uintptr_t)v | (uintptr_t)part_in) & 15) == 0,
                  "%s: tables and partial sums must be 16-byte aligned", who);
    MfStepArgs a;
    a.pin = table_in; a.pout = table_out; a.m = m; a.v = v;
    a.U = user_rows; a.I = item_rows; a.d = d;
    a.B = batch; a.reg = reg;
    a.plan = plan;
    a.range = reinterpret_cast<const int2*>(range);
    a.entries = reinterpret_cast<const int2*>(entries);
    a.mult2 = mult_next;
    a.part_in = part_in; a.n_in = n_parts_in; a.part_out = part_out;
    a.loss_prev = loss_prev_out; a.inv_b_prev = loss_prev_out ? 1.0f / (float)batch_prev : 0.f;
    a.loss_now = loss_out;
    a.k = AdamK{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps};
    a.step_scalars = step_scalars;
    a.neg_lr = (float)(-lr);
    const int parts = crh_mf_step_parts(user_rows + item_rows, d);
    a.light_blocks = parts - MF_HEAVY_BLOCKS;
    static const int ablate = CRH_PROFILE_ENV("CRH_MF_ABLATE");
    a.ablate = ablate;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dispatch_group(pick_group(d), [&](auto gc) -> int {
        constexpr int GG = decltype(gc)::value;
        if (opt == 1) hipLaunchKernelGGL((mf_step_kernel<GG, 1>), dim3((unsigned)parts), dim3(BPR_THREADS), 0, st, a);
        else hipLaunchKernelGGL((mf_step_kernel<GG, 0>), dim3((unsigned)parts), dim3(BPR_THREADS), 0, st, a);
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    });
}
}  // namespace

extern "C" int crh_mf_step_f32(const float* table_in, float* table_out, float* m, float* v, int64_t user_rows,
                               int64_t item_rows, int d, int64_t batch, float reg, const int32_t* plan,
                               const int32_t* range, const int32_t* entries, const int32_t* mult_next,
                               const float* part_in, int n_parts_in, float* part_out, float* loss_prev_out,
                               int64_t batch_prev, float* loss_out, double beta1, double beta2, double eps,
                               const float* step_scalars, void* stream) {
    return mf_step_run("crh_mf_step_f32", 0, table_in, table_out, m, v, user_rows, item_rows, d, batch, reg, plan, range,
                       entries, mult_next, part_in, n_parts_in, part_out, loss_prev_out, batch_prev, loss_out, beta1, beta2,
                       eps, step_scalars, 0.0, stream);
}

// The same one-launch step with torch.optim.SGD(lr) (no momentum, no weight decay) instead of Adam: no optimiser state,
// every row is read once and written once to the other buffer (north_star: "BPR loss + SGD update").
extern "C" int crh_mf_step_sgd_f32(const float* table_in, float* table_out, int64_t user_rows, int64_t item_rows, int d,
                                   int64_t batch, float reg, const int32_t* plan, const int32_t* range,
                                   const int32_t* entries, const int32_t* mult_next, const float* part_in,
                                   int n_parts_in, float* part_out, float* loss_prev_out, int64_t batch_prev,
                                   float* loss_out, double lr, void* stream) {
    return mf_step_run("crh_mf_step_sgd_f32", 1, table_in, table_out, nullptr, nullptr, user_rows, item_rows, d, batch, reg,
                       plan, range, entries, mult_next, part_in, n_parts_in, part_out, loss_prev_out, batch_prev, loss_out,
                       0.9, 0.999, 1e-8, nullptr, lr, stream);
}

// bpr loss of the LAST step of an epoch from its partial sums.
extern "C" int crh_mf_step_finish(const float* part_in, int n_parts_in, int64_t batch, float* loss_out, void* stream) {
    CRH_CHECK_ARG(part_in && n_parts_in > 0 && batch > 0 && loss_out, "crh_mf_step_finish: bad arguments");
    hipLaunchKernelGGL(mf_finish_kernel, dim3(1), dim3(BPR_THREADS), 0, reinterpret_cast<hipStream_t>(stream), part_in,
                       n_parts_in, 1.0f / (float)batch, loss_out);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
