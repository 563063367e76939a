// CSR SpMM for the LightGCN propagation on gfx950, with the layer sum fused in.
//
// Replaces model/LightGCN.py:86-96 per layer:  ego = torch.sparse.mm(A_hat, ego)  and the trailing
// torch.stack(...) / torch.mean(dim=1), plus the transposed SpMMs autograd replays in backward
// (A_hat is symmetric).  One launch computes
//        P   = A * X                       (written to Y when Y != NULL: the next layer's input)
//        acc = (acc_in * s_in + P) * s_out (written when acc_out != NULL: running layer sum / Horner step)
// so a forward pass of L layers is L launches and no other elementwise kernel: layer 1 uses
// acc_in = E0, the last layer uses s_out = 1/(L+1) and skips Y.
//
// HBM/L2-bound: a group of G = d/4 lanes (half a wave at d=128) owns one output row; the row's edge
// list is read coalesced (one (col,val) per lane) and broadcast by shuffles; each neighbour row is one
// 16-B-per-lane read; products are accumulated in edge (ascending column) order with fmaf, which is
// the oracle's order (oracle/topk_oracle.c orc_spmm_csr), so results are reproducible bit for bit.
// Catalogues are Zipf-shaped (a popular item has thousands of edges, a user a few dozen), so the
// caller may pass a SCHEDULE built once per graph: rows of <= 64 edges are one work item of a lane
// group; the few heavier rows get a whole wave each (its lane groups take the row's 64-edge segments
// round robin and their partial sums are combined by shuffles in a fixed order: deterministic, the
// association differs from the single chain).  One launch, no partials in memory.
//
// XCD-aware column slicing: for graphs whose dense operand does not fit one XCD's 4 MiB L2 but a column
// slice of it does (CiteULike: 22.5 K rows x 512 B = 11.5 MB; a quarter = 2.9 MB), the feature columns
// are cut into CS slices and block b works on slice (b % 8) % CS, so every XCD (blocks b % 8) gathers
// from ONE slice only and its random row reads hit its own L2 instead of going out to the fabric.
#include <math.h>
#include <stdlib.h>

#include "crh_common.h"

namespace {

struct SpmmArgs {
    const int64_t* rowptr;
    const int32_t* col;
    const float* val;
    int64_t n_rows;
    const float* X;
    int d;
    float* Y;
    float* acc_in;
    float* acc_out;
    float s_in, s_out;
    crh_spmm_sched sched;   // n_seg == 0: one lane group per row
    float* partial;         // unused (kept for the ABI's workspace argument)
    int cs;                 // column slices (1, 2 or 4)
    int64_t light_blocks;   // blocks per slice working on light work items; heavy rows follow
    int skip;               // measurement only (CRH_SPMM_SKIP): 1 = no heavy rows, 2 = no light rows
    // optional optimiser epilogue (crh_spmm_csr_adam_f32): acc is the gradient of p -> one Adam step on (p, m, v)
    float *adam_p, *adam_m, *adam_v;
    AdamK k;
    float bc2_sqrt, neg_step_size;
    const float* step_scalars;
    int zero_acc_in;        // clear acc_in's row once it has been consumed (ready for the next step's scatter)
    int sgd;                // optimiser epilogue is plain SGD: p <- fma(neg_step_size, g, p), no m / v
    int use_slab;           // light rows from the schedule's record stream (sched.slab, laid out for this launch's G)
};

__device__ __forceinline__ void fma4(f32x4& acc, float v, const f32x4& x) {
    acc.x = fmaf(v, x.x, acc.x);
    acc.y = fmaf(v, x.y, acc.y);
    acc.z = fmaf(v, x.z, acc.z);
    acc.w = fmaf(v, x.w, acc.w);
}

// What a row's epilogue reads, requested EARLY by the record path (before the gathers) so that the loads are in flight
// beside them: the row's acc_in and, with the optimiser in the epilogue, its p / m / v.
struct RowPre {
    f32x4 z, p, m, v;
};

__device__ __forceinline__ RowPre row_prefetch(const SpmmArgs& a, int64_t o, bool live, int64_t row) {
    RowPre r;
    r.z = r.p = r.m = r.v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!live) return r;
    if ((a.acc_out || a.adam_p) && a.acc_in) r.z = *reinterpret_cast<const f32x4*>(a.acc_in + o);
    if (a.adam_p) {
        r.p = *reinterpret_cast<const f32x4*>(a.adam_p + o);
        if (!a.sgd) {
            r.m = *reinterpret_cast<const f32x4*>(a.adam_m + o);
            r.v = *reinterpret_cast<const f32x4*>(a.adam_v + o);
        }
    }
    return r;
}

__device__ __forceinline__ void store_row_pre(const SpmmArgs& a, int64_t o, const f32x4& acc, RowPre q, int64_t row) {
    if (a.Y) *reinterpret_cast<f32x4*>(a.Y + o) = acc;
    if (a.acc_out || a.adam_p) {
        f32x4 r;
        r.x = (q.z.x * a.s_in + acc.x) * a.s_out;
        r.y = (q.z.y * a.s_in + acc.y) * a.s_out;
        r.z = (q.z.z * a.s_in + acc.z) * a.s_out;
        r.w = (q.z.w * a.s_in + acc.w) * a.s_out;
        if (a.acc_out) *reinterpret_cast<f32x4*>(a.acc_out + o) = r;
        if (a.adam_p && a.sgd) {
            sgd_elem4(q.p, r, a.neg_step_size);
            *reinterpret_cast<f32x4*>(a.adam_p + o) = q.p;
        } else if (a.adam_p) {
            const float b2s = a.step_scalars ? a.step_scalars[0] : a.bc2_sqrt;
            const float nss = a.step_scalars ? a.step_scalars[1] : a.neg_step_size;
            adam_elem4(q.p, q.m, q.v, r, a.k, b2s, nss);
            *reinterpret_cast<f32x4*>(a.adam_p + o) = q.p;
            *reinterpret_cast<f32x4*>(a.adam_m + o) = q.m;
            *reinterpret_cast<f32x4*>(a.adam_v + o) = q.v;
        }
        if (a.zero_acc_in && a.acc_in) *reinterpret_cast<f32x4*>(a.acc_in + o) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

__device__ __forceinline__ void store_row(const SpmmArgs& a, int64_t o, const f32x4& acc, int64_t row) {
    store_row_pre(a, o, acc, row_prefetch(a, o, true, row), row);
}

// Accumulate edges [e0, e1) of one row into acc for this lane's 16-B column slice, in edge order.
// The edge list is read G entries at a time (one per lane) and broadcast by shuffles; neighbour rows
// are fetched 8 (then 4) at a time before the dependent fma chain so several loads are in flight per group.
template <int G>
__device__ __forceinline__ void row_edges(const SpmmArgs& a, int64_t e0, int64_t e1, int c, bool on, int lig,
                                          f32x4& acc) {
    for (int64_t base = e0; base < e1; base += G) {
        const int64_t e = base + lig;
        const int my_col = e < e1 ? a.col[e] : 0;
        const float my_val = e < e1 ? a.val[e] : 0.f;
        const int cnt = (int)((e1 - base) < G ? (e1 - base) : G);
        int t = 0;
        for (; t + 8 <= cnt; t += 8) {
            int cc[8];
            float vv[8];
            f32x4 x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                cc[q] = __shfl(my_col, t + q, G);
                vv[q] = __shfl(my_val, t + q, G);
            }
            if (on) {
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = reinterpret_cast<const f32x4*>(a.X + (int64_t)cc[q] * a.d)[c];
#pragma unroll
                for (int q = 0; q < 8; ++q) fma4(acc, vv[q], x[q]);
            }
        }
        for (; t + 4 <= cnt; t += 4) {
            int cc[4];
            float vv[4];
            f32x4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cc[q] = __shfl(my_col, t + q, G);
                vv[q] = __shfl(my_val, t + q, G);
            }
            if (on) {
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = reinterpret_cast<const f32x4*>(a.X + (int64_t)cc[q] * a.d)[c];
#pragma unroll
                for (int q = 0; q < 4; ++q) fma4(acc, vv[q], x[q]);
            }
        }
        for (; t < cnt; ++t) {
            const int cc = __shfl(my_col, t, G);
            const float vv = __shfl(my_val, t, G);
            if (on) fma4(acc, vv, reinterpret_cast<const f32x4*>(a.X + (int64_t)cc * a.d)[c]);
        }
    }
}

// Block b -> (slice, work block): slice = (b % 8) % cs keeps every XCD on one column slice; the blocks of a
// slice are numbered j = (b / 8) * (8 / cs) + (b % 8) / cs.
// Light path: work item = one row (no schedule) or one single-segment row; a lane group of G lanes owns the
// row's 16*G-byte column slice and finishes it (bit-identical to the oracle's edge-order chain).
// Heavy path (rows with several segments): one wave per (row, slice).
// The same for lane groups narrower than 8 lanes (the column cuts of a giant row: 4 or 2 lanes per group): a lane holds
// 8 / G entries of an 8-edge chunk, so a group still keeps 8 gathers in flight per round trip -- with one entry per lane a
// 4-lane group walked its chunk of a 2 950-edge row in twice as many dependent round trips as the uncut row needed.
template <int G>
__device__ __forceinline__ void row_edges_narrow(const SpmmArgs& a, int64_t e0, int64_t e1, int c, bool on, int lig,
                                                 f32x4& acc) {
    static_assert(G >= 1 && G < 8 && 8 % G == 0, "narrow lane groups only");
    constexpr int EPL = 8 / G;
    for (int64_t base = e0; base < e1; base += 8) {
        int my_col[EPL];
        float my_val[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const int64_t e = base + s * G + lig;
            my_col[s] = e < e1 ? a.col[e] : 0;
            my_val[s] = e < e1 ? a.val[e] : 0.f;
        }
        f32x4 x[8];
        float vv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cc = __shfl(my_col[q / G], q % G, G);
            vv[q] = __shfl(my_val[q / G], q % G, G);
            x[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on && base + q < e1) x[q] = reinterpret_cast<const f32x4*>(a.X + (int64_t)cc * a.d)[c];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (base + q < e1) fma4(acc, vv[q], x[q]);
    }
}

// ---- record stream ("slab") light path, round 3 --------------------------------------------------------------------
// One lane group = one record (see crh_spmm_sched::slab).  A lane holds ONE pair of each unit; pair q of a unit is
// broadcast to the lane group by shuffles; the fma chain runs in edge order, so the row's bits are those of the
// descriptor path and of the oracle's chain.
#define CRH_SLAB_LD(Q)                                                                                         \
    float v##Q = 0.f;                                                                                          \
    f32x4 x##Q = {0.f, 0.f, 0.f, 0.f};                                                                         \
    if constexpr (Q0 + Q < G) {                                                                                \
        const int cc = __shfl((int)u.x, Q0 + Q, G);                                                            \
        v##Q = __int_as_float(__shfl((int)u.y, Q0 + Q, G));                                                    \
        if (on && ebase + Q < cnt) x##Q = reinterpret_cast<const f32x4*>(a.X + (int64_t)cc * a.d)[c];          \
    }
#define CRH_SLAB_FM(Q) \
    if constexpr (Q0 + Q < G) { if (ebase + Q < cnt) fma4(acc, v##Q, x##Q); }

// lanes Q0 .. Q0 + 7 of a unit (as far as the lane group goes); lane Q0 holds edge `ebase` of the row
template <int G, int Q0>
__device__ __forceinline__ void slab_batch(const SpmmArgs& a, const uint2& u, int ebase, int cnt, int c, bool on, f32x4& acc) {
    CRH_SLAB_LD(0) CRH_SLAB_LD(1) CRH_SLAB_LD(2) CRH_SLAB_LD(3) CRH_SLAB_LD(4) CRH_SLAB_LD(5) CRH_SLAB_LD(6) CRH_SLAB_LD(7)
    CRH_SLAB_FM(0) CRH_SLAB_FM(1) CRH_SLAB_FM(2) CRH_SLAB_FM(3) CRH_SLAB_FM(4) CRH_SLAB_FM(5) CRH_SLAB_FM(6) CRH_SLAB_FM(7)
}

// all lanes from START on; e_lane0 = edge index lane 0 would hold (-1 in unit 0, whose lane 0 is the header)
template <int G, int START>
__device__ __forceinline__ void slab_unit(const SpmmArgs& a, const uint2& u, int e_lane0, int cnt, int c, bool on, f32x4& acc) {
    if constexpr (START < G) {
        slab_batch<G, START>(a, u, e_lane0 + START, cnt, c, on, acc);
        slab_unit<G, START + 8>(a, u, e_lane0, cnt, c, on, acc);
    }
}

template <int G>
__device__ __forceinline__ void slab_light(const SpmmArgs& a, int64_t jl, int c, bool on, int lig) {
    const crh_spmm_sched& sc = a.sched;
    const uint2* stream = reinterpret_cast<const uint2*>(sc.slab);
    const int64_t w = jl * (256 / G) + threadIdx.x / G;
    // the record's address is arithmetic: bucket by comparing against the buckets' first work items (unused: INT32_MAX)
    int64_t bs = sc.slab_base[0];
    int fs = 0, un = sc.slab_units[0];
#pragma unroll
    for (int b = 1; b < CRH_SPMM_SLAB_BUCKETS; ++b)
        if (w >= sc.slab_first[b]) {
            bs = sc.slab_base[b];
            fs = sc.slab_first[b];
            un = sc.slab_units[b];
        }
    if (w >= sc.n_slab) un = 0;
    const int64_t base = bs + (w - fs) * (int64_t)un * G;
    // header + first edges and the second unit are requested together (a read past a one-unit record lands in the next
    // record or in the stream's tail padding)
    uint2 u0 = {0u, 0u}, u1 = {0u, 0u};
    if (un > 0) {
        u0 = stream[base + lig];
        u1 = stream[base + G + lig];
    }
    int mu = un;                                           // the wave's longest record decides the trip count
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
        const int o = __shfl_xor(mu, off);
        mu = o > mu ? o : mu;
    }
    if (mu == 0) return;
    const int64_t row = __shfl((int)u0.x, 0, G);
    const int cnt = un > 0 ? __shfl((int)u0.y, 0, G) : 0;
    const int64_t o = row * a.d + (int64_t)c * 4;
    const RowPre pre = row_prefetch(a, o, un > 0 && on, row);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    slab_unit<G, 1>(a, u0, -1, cnt, c, on, acc);
    if (mu > 1) {
        uint2 cur = u1;
        for (int k = 1; k < mu; ++k) {
            uint2 nxt = {0u, 0u};
            if (k + 1 < un) nxt = stream[base + (int64_t)(k + 1) * G + lig];
            slab_unit<G, 0>(a, cur, (G - 1) + (k - 1) * G, cnt, c, on, acc);
            cur = nxt;
        }
    }
    if (un > 0 && on) store_row_pre(a, o, acc, pre, row);
}

// One heavy row (more than the schedule's segment length of edges) by a whole workgroup: its 256 / GG lane groups of GG
// lanes split the edge list into contiguous chunks (multiples of 8 edges); partial sums meet in a fixed order: shuffle
// tree inside a wave, then the 4 waves through LDS.  c0 = first float4 column of this workgroup's column range.
// (One WAVE per heavy row of up to 256 / 512 / 1024 edges instead, four rows per block, measured no gain: LightGCN step
// 158.7 -> 159.4 / 161.8 / 165.7 us.)
template <int GG>
__device__ __forceinline__ void heavy_row(const SpmmArgs& a, int64_t row, int c0, int c_end, f32x4 (*wsum)[64]) {
    constexpr int NGB = 256 / GG;
    const int lig = threadIdx.x % GG, gg = threadIdx.x / GG;
    const int c = c0 + lig;
    const bool on = c < c_end;
    const int64_t r0 = a.rowptr[row], r1 = a.rowptr[row + 1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        int64_t chunk = (r1 - r0 + NGB - 1) / NGB;
        chunk = (chunk + 7) & ~(int64_t)7;
        const int64_t e0 = r0 + (int64_t)gg * chunk;
        const int64_t e1 = e0 + chunk < r1 ? e0 + chunk : r1;
        if (e0 < r1) {
            if constexpr (GG < 8) row_edges_narrow<GG>(a, e0, e1, c, on, lig, acc);
            else row_edges<GG>(a, e0, e1, c, on, lig, acc);
        }
    }
#pragma unroll
    for (int off = GG; off < 64; off <<= 1) {
        acc.x += __shfl_down(acc.x, off);
        acc.y += __shfl_down(acc.y, off);
        acc.z += __shfl_down(acc.z, off);
        acc.w += __shfl_down(acc.w, off);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < GG) wsum[wv][lig] = acc;
    __syncthreads();
    if (threadIdx.x < GG && on) {
        f32x4 t0 = wsum[0][lig], t1 = wsum[1][lig], t2 = wsum[2][lig], t3 = wsum[3][lig];
        f32x4 r;
        r.x = (t0.x + t1.x) + (t2.x + t3.x);
        r.y = (t0.y + t1.y) + (t2.y + t3.y);
        r.z = (t0.z + t1.z) + (t2.z + t3.z);
        r.w = (t0.w + t1.w) + (t2.w + t3.w);
        store_row(a, row * a.d + (int64_t)c * 4, r, row);
    }
}

// Block b -> (slice, work block): slice = (b % 8) % cs keeps every XCD on one column slice; the blocks of a
// slice are numbered j = (b / 8) * (8 / cs) + (b % 8) / cs.
// HEAVY blocks lead the grid (j < n_multi; round 3: they trailed it): a heavy row is the longest dependent chain of the
// launch, so it has to start with the first wave of workgroups, whatever the number of rounds the light rows take.  The
// schedule lists one entry per heavy BLOCK in descending row length: multi_row[m] = row, multi_count[m] = n_sub | sub << 8:
// rows above 1024 edges are cut into n_sub = 2 (4 above 4096) COLUMN ranges of the slice, one workgroup each, with G / n_sub
// lanes per lane group -- twice (four times) the lane groups per row, so the chain of the longest row of a Zipf-shaped
// catalogue is half (a quarter) as long; columns are independent, so nothing is combined across workgroups and the result
// stays deterministic (CiteULike shape: three rows of 1 150 - 2 950 edges set the launch's critical path).
// Light path: work item = one row (no schedule) or one single-segment row; a lane group of G lanes owns the
// row's 16*G-byte column slice and finishes it (bit-identical to the oracle's edge-order chain).
template <int G>
__device__ __forceinline__ void spmm_body(const SpmmArgs& a) {
    const int lig = threadIdx.x % G;
    const int xcd = blockIdx.x & 7;
    const int slice = xcd % a.cs;
    const int64_t j = (int64_t)(blockIdx.x >> 3) * (8 / a.cs) + xcd / a.cs;
    const int c = slice * G + lig;                   // this lane's float4 column
    const bool on = c < (a.d >> 2);                  // d/4 not a power of two: the lane group is padded
    const bool seg = a.sched.n_seg > 0;
    const int64_t heavy_blocks = seg ? a.sched.n_multi : 0;
    if (j >= heavy_blocks) {
        const int64_t jl = j - heavy_blocks;
        if (jl >= a.light_blocks || (CRH_ABLATE(a.skip) & 2)) return;
        if constexpr (G == 8) {       // (wider lane groups: 16 - 32 gathers of a unit in flight cost the kernel its occupancy)
            if (a.use_slab) {
                slab_light<G>(a, jl, c, on, lig);
                return;
            }
        }
        const int64_t n_work = seg ? a.sched.n_seg : a.n_rows;
        // rows_per_group work items per lane group, a whole "grid" apart: with the schedule's descending-length order a
        // long row is paired with a short one, and fewer, fully resident workgroups replace a second round of them
        for (int64_t w = jl * (256 / G) + threadIdx.x / G; w < n_work; w += a.light_blocks * (256 / G)) {
            int64_t row, e0, e1;
            if (seg && a.sched.seg_desc) {
                // ONE 16-byte descriptor per work item instead of three dependent loads (seg_row / seg_slot, then
                // rowptr[row], rowptr[row + 1]): tools/probes/spmm_steps_probe.hip prices the schedule indirection at
                // 2.4 us and the rowptr reads at 1.0 us of a 15 us launch
                const i32x4 dsc = reinterpret_cast<const i32x4*>(a.sched.seg_desc)[w];
                if (dsc.w >= 0) continue;                             // a heavy row: done by its own workgroup(s)
                row = dsc.x;
                e0 = dsc.y;
                e1 = e0 + dsc.z;
            } else {
                if (seg && a.sched.seg_slot[w] >= 0) continue;        // a heavy row
                row = seg ? a.sched.seg_row[w] : w;
                e0 = a.rowptr[row];
                e1 = a.rowptr[row + 1];                               // a light work item is a whole row
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            row_edges<G>(a, e0, e1, c, on, lig, acc);
            if (on) store_row(a, row * a.d + (int64_t)c * 4, acc, row);
        }
        return;
    }
    if (CRH_ABLATE(a.skip) & 1) return;
    __shared__ f32x4 wsum[4][64];
    const int64_t row = a.sched.multi_row[j];
    const int cut = a.sched.multi_count ? a.sched.multi_count[j] : 1;
    const int n_sub = cut & 255, sub = cut >> 8;
    const int c_end = (a.d >> 2) < (slice + 1) * G ? (a.d >> 2) : (slice + 1) * G;
    if (G >= 4 && n_sub == 4) heavy_row<(G >= 4 ? G / 4 : 1)>(a, row, slice * G + sub * (G / 4), c_end, wsum);
    else if (G >= 2 && n_sub == 2) heavy_row<(G >= 2 ? G / 2 : 1)>(a, row, slice * G + sub * (G / 2), c_end, wsum);
    else if (sub == 0) heavy_row<G>(a, row, slice * G, c_end, wsum);   // (a cut wider than the lane group: one workgroup does the slice)
}

template <int G>
__global__ __launch_bounds__(256) void spmm_csr_kernel(SpmmArgs a) {
    spmm_body<G>(a);
}
// The same body under its own name when the optimiser rides in the epilogue (crh_spmm_csr_adam_f32 / _sgd_f32): the launch also
// reads and writes p, m, v, and a profile that averages both flavours under one name describes neither.
template <int G>
__global__ __launch_bounds__(256) void spmm_csr_opt_kernel(SpmmArgs a) {
    spmm_body<G>(a);
}

// Grid of a launch (sets use_slab / light_blocks in `a`; a.cs and a.sched must be final).
// Rows per lane group.  Rounds 1-2 took the smallest count (up to 4) that made the whole launch resident at once, because
// the heavy workgroups TRAILED the grid and a second round of workgroups delayed the launch's longest chains (CiteULike
// shape, 4 slices: 3 rows -> 1612 workgroups, LightGCN step 0.182 -> 0.164 ms).  With the heavy workgroups leading the
// grid one row per group is the faster schedule -- the light rows' chains are a third as long and the workgroups of the
// later rounds fill the slots the early ones free (same shape: 1 row 141.6 us per step, 2 rows 142.1, 3 rows 154.5;
// profiles/r03_lgcn_sweep1.log).  CRH_SPMM_ROWS forces a count.
int64_t spmm_grid(SpmmArgs& a, int G) {
    static const int force_rows = CRH_TUNE_ENV("CRH_SPMM_ROWS") ? atoi(CRH_TUNE_ENV("CRH_SPMM_ROWS")) : 0;
    const int64_t n_work = a.sched.n_seg > 0 ? a.sched.n_seg : a.n_rows;
    const int64_t heavy_blocks = a.sched.n_seg > 0 ? (int64_t)a.sched.n_multi : 0;
    const int rows_per_group = force_rows > 0 ? force_rows : 1;
    static const int no_slab = getenv("CRH_SPMM_SLAB") ? !atoi(getenv("CRH_SPMM_SLAB")) : 0;
    a.use_slab = !no_slab && G == 8 && a.sched.n_seg > 0 && a.sched.slab && a.sched.slab_lanes == G && a.sched.n_slab >= 0;
    const int64_t per_block = a.use_slab ? (256 / G) : (256 / G) * rows_per_group;
    a.light_blocks = ((a.use_slab ? (int64_t)a.sched.n_slab : n_work) + per_block - 1) / per_block;
    const int64_t per_slice = a.light_blocks + heavy_blocks;
    const int64_t groups8 = (per_slice + (8 / a.cs) - 1) / (8 / a.cs);      // grid in units of 8 blocks
    return groups8 * 8;
}

template <int G>
int launch_spmm(SpmmArgs a, hipStream_t st) {
    const int64_t grid = spmm_grid(a, G);
    if (a.adam_p) hipLaunchKernelGGL(spmm_csr_opt_kernel<G>, dim3((unsigned)grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(spmm_csr_kernel<G>, dim3((unsigned)grid), dim3(256), 0, st, a);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

}  // namespace

// 64 measured best at CiteULike size (LightGCN step 0.197 ms; 32: 0.213, 16: 0.406, 128: 0.203 -- CRH_SPMM_SEG)
extern "C" int crh_spmm_segment_edges(void) {
    static const int seg = getenv("CRH_SPMM_SEG") ? atoi(getenv("CRH_SPMM_SEG")) : 64;
    return seg;
}

extern "C" size_t crh_spmm_workspace_bytes(const crh_spmm_sched* sched, int d) {
    (void)sched;
    (void)d;
    return 0;   // heavy rows are combined inside their wave: no partial sums in memory any more
}

namespace {
// Column slices and lanes per lane group of a launch.  Slices: as few as make one slice of the dense operand fit an XCD's
// L2 (leaving room for the edge stream), at most 4, and only while a slice keeps >= 4 lanes (64 B) per row; only with a
// schedule (nnz >= 0), and only while re-reading the edge list once per slice stays below the operand.
void spmm_shape(int64_t n_rows, int d, int64_t nnz, int* cs_out, int* g_out) {
    const int nvec = d / 4;
    static const int force_cs = CRH_TUNE_ENV("CRH_SPMM_SLICES") ? atoi(CRH_TUNE_ENV("CRH_SPMM_SLICES")) : 0;
    int cs = 1;
    const double bytes = (double)n_rows * d * 4;
    if (nnz >= 0 && bytes > 3.0e6 && bytes <= 4 * 3.2e6) {
        while (cs < 4 && bytes / cs > 3.2e6) cs <<= 1;
        while (cs > 1 && (double)nnz * 8.0 * cs > bytes) cs >>= 1;
    }
    if (force_cs) cs = force_cs;
    while (cs > 1 && (nvec % cs != 0 || nvec / cs < 4)) cs >>= 1;
    int G = 1;
    while (G < nvec / cs && G < 64) G <<= 1;
    *cs_out = cs;
    *g_out = G;
}

int spmm_run(const char* who, const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, const float* x,
             int d, float* y, float* acc_in, float s_in, float* acc_out, float s_out, const crh_spmm_sched* sched,
             float* adam_p, float* adam_m, float* adam_v, AdamK k, float bc2_sqrt, float nss, const float* step_scalars,
             int zero_acc_in, void* stream, int sgd = 0) {
    CRH_CHECK_ARG(rowptr && x && n_rows > 0, "%s: NULL pointer / empty matrix", who);
    CRH_CHECK_ARG(d >= 4 && d % 4 == 0, "%s: d=%d must be a positive multiple of 4", who, d);
    CRH_CHECK_ARG(y || acc_out || adam_p, "%s: nothing to write (y and acc_out both NULL)", who);
    CRH_CHECK_ARG(y != x && acc_out != x && adam_p != x && !(zero_acc_in && acc_in == x),
                  "%s: outputs must not alias x", who);
    CRH_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)acc_in | (uintptr_t)acc_out | (uintptr_t)adam_p |
                    (uintptr_t)adam_m | (uintptr_t)adam_v) & 15) == 0,
                  "%s: dense operands must be 16-byte aligned", who);
    static const int skip = CRH_PROFILE_ENV("CRH_SPMM_SKIP");
    SpmmArgs a{rowptr, col, val, n_rows, x, d, y, acc_in, acc_out, s_in, s_out, {}, nullptr, 1, 0, skip,
               adam_p, adam_m, adam_v, k, bc2_sqrt, nss, step_scalars, zero_acc_in, sgd, 0};
    if (sched && sched->n_seg > 0) {
        CRH_CHECK_ARG(sched->seg_row && sched->seg_ptr && sched->seg_slot, "%s: incomplete schedule", who);
        CRH_CHECK_ARG(sched->n_multi == 0 || sched->multi_row, "%s: incomplete schedule (heavy rows)", who);
        CRH_CHECK_ARG(sched->version == CRH_SPMM_SCHED_VERSION,
                      "%s: crh_spmm_sched.version = %d, this library reads layout %d (multi_count = n_sub | sub << 8)", who,
                      (int)sched->version, CRH_SPMM_SCHED_VERSION);
        a.sched = *sched;
        static const int use_desc = CRH_TUNE_ENV("CRH_SPMM_DESC") ? atoi(CRH_TUNE_ENV("CRH_SPMM_DESC")) : 1;
        if (!use_desc) a.sched.seg_desc = nullptr;
    } else {
        a.sched.n_seg = 0;
        a.sched.n_multi = 0;
        a.sched.seg_desc = nullptr;
        a.sched.slab = nullptr;
    }
    int cs, G;
    spmm_shape(n_rows, d, sched && sched->n_seg > 0 ? sched->nnz : -1, &cs, &G);
    const int nvec = d / 4;
    CRH_CHECK_ARG(G * cs == nvec || cs == 1, "%s: internal slice error", who);
    a.cs = G * cs != nvec ? 1 : cs;   // d/4 not a power of two (or > 64 lanes): one slice, padded lane group
    CRH_CHECK_ARG(a.cs > 1 || G >= nvec, "%s: d=%d above 256 is not supported", who, d);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (G) {
        case 1: return launch_spmm<1>(a, st);
        case 2: return launch_spmm<2>(a, st);
        case 4: return launch_spmm<4>(a, st);
        case 8: return launch_spmm<8>(a, st);
        case 16: return launch_spmm<16>(a, st);
        case 32: return launch_spmm<32>(a, st);
        default: return launch_spmm<64>(a, st);
    }
}
}  // namespace

extern "C" int crh_spmm_lane_group(int64_t n_rows, int d, int64_t nnz) {
    if (n_rows <= 0 || d < 4 || d % 4) return 0;
    int cs, G;
    spmm_shape(n_rows, d, nnz, &cs, &G);
    return G;
}

extern "C" int crh_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                const float* x, int d, float* y, const float* acc_in, float s_in,
                                float* acc_out, float s_out, const crh_spmm_sched* sched, void* workspace,
                                size_t workspace_bytes, void* stream) {
    (void)workspace;
    (void)workspace_bytes;
    return spmm_run("crh_spmm_csr_f32", rowptr, col, val, n_rows, x, d, y, const_cast<float*>(acc_in), s_in, acc_out,
                    s_out, sched, nullptr, nullptr, nullptr, AdamK{}, 0.f, 0.f, nullptr, 0, stream);
}

// The last SpMM of LightGCN's backward pass with the optimiser fused into its epilogue (model/LightGCN.py:26-28):
//   g = (acc_in * s_in + A x) * s_out   is the gradient of the embedding table p (also stored to acc_out if given);
//   one Adam step (torch.optim.Adam defaults, crh_adam_dense_f32's arithmetic, same bits) on (p, m, v) in place;
//   zero_acc_in != 0: acc_in's row is cleared after it was consumed, ready for the next step's gradient scatter
//   (acc_in must then not be the gathered operand x).
// step >= 1 gives the bias corrections; step_scalars (device, {sqrt(1-beta2^step), -lr/(1-beta1^step)}) overrides
// them for graph replay.
extern "C" int crh_spmm_csr_adam_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                     const float* x, int d, float* acc_in, float s_in, float* acc_out, float s_out,
                                     const crh_spmm_sched* sched, float* p, float* m, float* v, double lr,
                                     double beta1, double beta2, double eps, int64_t step,
                                     const float* step_scalars, int zero_acc_in, void* stream) {
    CRH_CHECK_ARG(p && m && v, "crh_spmm_csr_adam_f32: NULL optimiser state");
    CRH_CHECK_ARG(step >= 1 || step_scalars, "crh_spmm_csr_adam_f32: step starts at 1");
    if (step < 1) step = 1;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    return spmm_run("crh_spmm_csr_adam_f32", rowptr, col, val, n_rows, x, d, nullptr, acc_in, s_in, acc_out, s_out, sched,
                    p, m, v, AdamK{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps},
                    (float)sqrt(bc2), (float)(-(lr / bc1)), step_scalars, zero_acc_in, stream);
}

// The same fused epilogue with torch.optim.SGD(lr) defaults instead of Adam: p <- fma(-lr, g, p), no optimiser state.
extern "C" int crh_spmm_csr_sgd_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                    const float* x, int d, float* acc_in, float s_in, float* acc_out, float s_out,
                                    const crh_spmm_sched* sched, float* p, double lr, int zero_acc_in, void* stream) {
    CRH_CHECK_ARG(p, "crh_spmm_csr_sgd_f32: NULL parameter table");
    return spmm_run("crh_spmm_csr_sgd_f32", rowptr, col, val, n_rows, x, d, nullptr, acc_in, s_in, acc_out, s_out, sched,
                    p, nullptr, nullptr, AdamK{}, 0.f, (float)(-lr), nullptr, zero_acc_in, stream, 1);
}
