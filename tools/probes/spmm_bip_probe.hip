// Round-3 probe: variants of the LightGCN SpMM on the real CiteULike-shaped graph (tools/probes/data/citeulike.bin).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o spmm_bip_probe spmm_bip_probe.hip
// Every variant computes  P = A X;  Y = P;  acc_out = (acc_in * s_in + P) * s_out  (forward layer 1 / 2 of the step) and is
// checked against a CPU chain on sampled rows.  What varies:
//   map 0: block b works on column slice (b % 8) % 4 over ALL rows (the round-2 schedule)
//   map 1: bipartite pairing: XCD s (0..3) takes the USER rows of slice s, XCD s + 4 the ITEM rows of slice s -- every byte of
//          X is gathered by exactly one XCD (user rows gather the item half, item rows the user half)
//   R    : rows per lane group;  pipe 0: one after the other (round 2), pipe 1: descriptors / edge chunks / acc_in of all R
//          rows requested together, gathers of all R rows in flight together
//   heavy-first: the workgroup-per-heavy-row blocks lead the grid instead of trailing it
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
    const int* col; const float* val; const float* X; float* Y; const float* acc_in; float* acc_out; float s_in, s_out;
    int d;
    const i32x4* desc[2];   // per class: {row, e0, cnt, heavy?}
    int n_light[2];         // light work items per class
    const i32x4* hdesc[2];  // heavy rows per class
    int n_heavy[2];
    int light_blocks[2];
    int map, heavy_first, cs;
    unsigned long long* clk;   // per block {start, end} of s_memrealtime (100 MHz) when non-null
    int abl;   // ablation bits: 1 no stores, 2 no acc_in, 4 gathers hit row 0, 8 no heavy rows, 16 no light rows
};

__device__ __forceinline__ void fma4(f32x4& a, float v, const f32x4& x) {
    a.x = fmaf(v, x.x, a.x); a.y = fmaf(v, x.y, a.y); a.z = fmaf(v, x.z, a.z); a.w = fmaf(v, x.w, a.w);
}
__device__ __forceinline__ void store_row(const Args& a, long o, const f32x4& acc, const f32x4& z) {
    if (a.abl & 1) { if (acc.x == 12345.f) a.Y[0] = z.x; return; }
    *reinterpret_cast<f32x4*>(a.Y + o) = acc;
    f32x4 r;
    r.x = (z.x * a.s_in + acc.x) * a.s_out; r.y = (z.y * a.s_in + acc.y) * a.s_out;
    r.z = (z.z * a.s_in + acc.z) * a.s_out; r.w = (z.w * a.s_in + acc.w) * a.s_out;
    *reinterpret_cast<f32x4*>(a.acc_out + o) = r;
}

// heavy rows: 16 gathers in flight (two edge chunks of G requested together)
template <int G>
__device__ __forceinline__ void row_edges_wide(const Args& a, long e0, long e1, int c, int lig, f32x4& acc) {
    const long dmul = (a.abl & 4) ? 0 : a.d;
    for (long base = e0; base < e1; base += 2 * G) {
        const long ea = base + lig, eb = base + G + lig;
        const int ca = ea < e1 ? a.col[ea] : 0, cb = eb < e1 ? a.col[eb] : 0;
        const float va = ea < e1 ? a.val[ea] : 0.f, vb = eb < e1 ? a.val[eb] : 0.f;
        f32x4 x[16]; float vv[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cc = __shfl(ca, q, G); vv[q] = __shfl(va, q, G);
            x[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (base + q < e1) x[q] = reinterpret_cast<const f32x4*>(a.X + (long)cc * dmul)[c];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cc = __shfl(cb, q, G); vv[8 + q] = __shfl(vb, q, G);
            x[8 + q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (base + G + q < e1) x[8 + q] = reinterpret_cast<const f32x4*>(a.X + (long)cc * dmul)[c];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) if (base + q < e1) fma4(acc, vv[q], x[q]);
    }
}

template <int G>
__device__ __forceinline__ void row_edges_seq(const Args& a, long e0, long e1, int c, int lig, f32x4& acc) {
    const long dmul = (a.abl & 4) ? 0 : a.d;
    for (long base = e0; base < e1; base += G) {
        const long e = base + lig;
        const int my_col = e < e1 ? a.col[e] : 0;
        const float my_val = e < e1 ? a.val[e] : 0.f;
        const int cnt = (int)((e1 - base) < G ? (e1 - base) : G);
        int t = 0;
        for (; t + 8 <= cnt; t += 8) {
            int cc[8]; float vv[8]; f32x4 x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { cc[q] = __shfl(my_col, t + q, G); vv[q] = __shfl(my_val, t + q, G); }
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = reinterpret_cast<const f32x4*>(a.X + (long)cc[q] * dmul)[c];
#pragma unroll
            for (int q = 0; q < 8; ++q) fma4(acc, vv[q], x[q]);
        }
        for (; t + 4 <= cnt; t += 4) {
            int cc[4]; float vv[4]; f32x4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { cc[q] = __shfl(my_col, t + q, G); vv[q] = __shfl(my_val, t + q, G); }
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = reinterpret_cast<const f32x4*>(a.X + (long)cc[q] * dmul)[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) fma4(acc, vv[q], x[q]);
        }
        for (; t < cnt; ++t) {
            const int cc = __shfl(my_col, t, G); const float vv = __shfl(my_val, t, G);
            fma4(acc, vv, reinterpret_cast<const f32x4*>(a.X + (long)cc * dmul)[c]);
        }
    }
}

// G lanes per row slice (cs * G * 4 = d), R rows per lane group, PIPE: rows in flight together, H: gathers per row and round
template <int G, int R, int PIPE, int H>
__global__ __launch_bounds__(256) void spmm_k(Args a);

struct ClkScope {
    unsigned long long* p;
    __device__ ClkScope(unsigned long long* clk) : p(clk ? clk + 2 * (size_t)blockIdx.x : nullptr) {
        if (p && threadIdx.x == 0) p[0] = __builtin_amdgcn_s_memrealtime();
    }
    __device__ ~ClkScope() {
        if (p) { __syncthreads(); if (threadIdx.x == 0) p[1] = __builtin_amdgcn_s_memrealtime(); }
    }
};

template <int G, int R, int PIPE, int H>
__global__ __launch_bounds__(256) void spmm_k(Args a) {
    ClkScope clk_scope(a.clk);
    const int lig = threadIdx.x % G;
    const int xcd = blockIdx.x & 7;
    int slice, cls; long j;
    if (a.map == 0) { slice = xcd % a.cs; cls = 0; j = (long)(blockIdx.x >> 3) * (8 / a.cs) + xcd / a.cs; }
    else { slice = xcd % a.cs; cls = xcd / a.cs; j = blockIdx.x >> 3; if (cls > 1) return; }
    const int c = slice * G + lig;
    constexpr int GPB = 256 / G;
    const long heavy_blocks = a.n_heavy[cls];
    long jl = a.heavy_first ? j - heavy_blocks : j;          // light block index
    long jh = a.heavy_first ? j : j - a.light_blocks[cls];   // heavy block index
    if (jl >= 0 && jl < a.light_blocks[cls]) {
        if (a.abl & 16) return;
        const i32x4* desc = a.desc[cls];
        const long n = a.n_light[cls];
        const long stride = (long)a.light_blocks[cls] * GPB;
        const long w0 = jl * GPB + threadIdx.x / G;
        if (!PIPE) {
            for (long w = w0; w < n; w += stride) {
                const i32x4 dsc = desc[w];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                row_edges_seq<G>(a, dsc.y, (long)dsc.y + dsc.z, c, lig, acc);
                const long o = (long)dsc.x * a.d + (long)c * 4;
                store_row(a, o, acc, (a.abl & 2) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.acc_in + o));
            }
            return;
        }
        // ---- pipelined: the R rows of this lane group advance together
        const long dmul = (a.abl & 4) ? 0 : a.d;
        i32x4 dsc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long w = w0 + r * stride;
            dsc[r] = w < n ? desc[w] : i32x4{0, 0, 0, 0};
            if (w >= n) dsc[r].x = -1;
        }
        int mc[R]; float mv[R]; f32x4 z[R], acc[R];
        int maxcnt = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool in = lig < dsc[r].z;
            mc[r] = in ? a.col[dsc[r].y + lig] : 0;
            mv[r] = in ? a.val[dsc[r].y + lig] : 0.f;
            acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            maxcnt = dsc[r].z > maxcnt ? dsc[r].z : maxcnt;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            z[r] = (dsc[r].x >= 0 && !(a.abl & 2)) ? *reinterpret_cast<const f32x4*>(a.acc_in + (long)dsc[r].x * a.d + (long)c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        // the longest row of the wave decides the trip count (wave-uniform loop keeps the shuffles legal)
#pragma unroll
        for (int off = G; off < 64; off <<= 1) { const int o = __shfl_xor(maxcnt, off); maxcnt = o > maxcnt ? o : maxcnt; }
        for (int base = 0; base < maxcnt; base += G) {
            int nc[R]; float nv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {      // next chunk of every row, requested before this chunk's gathers are consumed
                const bool in = base + G + lig < dsc[r].z;
                nc[r] = in ? a.col[dsc[r].y + base + G + lig] : 0;
                nv[r] = in ? a.val[dsc[r].y + base + G + lig] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < G; t += H) {
                f32x4 x[R][H]; float vv[R][H];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int q = 0; q < H; ++q) {
                        const int cc = __shfl(mc[r], t + q, G);
                        vv[r][q] = __shfl(mv[r], t + q, G);
                        x[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (base + t + q < dsc[r].z) x[r][q] = reinterpret_cast<const f32x4*>(a.X + (long)cc * dmul)[c];
                    }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int q = 0; q < H; ++q)
                        if (base + t + q < dsc[r].z) fma4(acc[r], vv[r][q], x[r][q]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) { mc[r] = nc[r]; mv[r] = nv[r]; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (dsc[r].x >= 0) store_row(a, (long)dsc[r].x * a.d + (long)c * 4, acc[r], z[r]);
        return;
    }
    if (jh < 0 || jh >= heavy_blocks || (a.abl & 8)) return;
    __shared__ f32x4 wsum[4][G];
    const i32x4 hd = a.hdesc[cls][jh];
    const long r0 = hd.y, r1 = (long)hd.y + hd.z;
    const int gg = threadIdx.x / G;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        long chunk = (r1 - r0 + GPB - 1) / GPB;
        chunk = (chunk + 7) & ~(long)7;
        const long e0 = r0 + (long)gg * chunk;
        const long e1 = e0 + chunk < r1 ? e0 + chunk : r1;
        if (e0 < r1) { if (G == 8 && (a.abl & 32)) row_edges_wide<G>(a, e0, e1, c, lig, acc); else row_edges_seq<G>(a, e0, e1, c, lig, acc); }
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
        acc.x += __shfl_down(acc.x, off); acc.y += __shfl_down(acc.y, off);
        acc.z += __shfl_down(acc.z, off); acc.w += __shfl_down(acc.w, off);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < G) wsum[wv][lig] = acc;
    __syncthreads();
    if (threadIdx.x < G) {
        f32x4 t0 = wsum[0][lig], t1 = wsum[1][lig], t2 = wsum[2][lig], t3 = wsum[3][lig], r;
        r.x = (t0.x + t1.x) + (t2.x + t3.x); r.y = (t0.y + t1.y) + (t2.y + t3.y);
        r.z = (t0.z + t1.z) + (t2.z + t3.z); r.w = (t0.w + t1.w) + (t2.w + t3.w);
        const long o = (long)hd.x * a.d + (long)c * 4;
        store_row(a, o, r, *reinterpret_cast<const f32x4*>(a.acc_in + o));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// "Slab" light path: the schedule is a STREAM of 8-byte pairs in work-item order.  Record of a row with cnt edges:
// pair 0 = {row, cnt}, pairs 1.. = (col, val) in edge order, padded with {0, 0.f} to `units` 64-byte units
// (units = 1 for cnt <= 7, else 1 + ceil((cnt - 7) / 8)).  Rows are sorted by descending length, so all records of one
// unit count are contiguous ("bucket") and the address of work item w is ARITHMETIC: no descriptor hop, no rowptr hop --
// the dependent chain of a row is {record} -> {gathers} -> {store} instead of {descriptor} -> {edges} -> {gathers} -> ...
// A lane of the group of 8 loads ONE pair per unit (coalesced 64 B per group), broadcast by shuffles as before.
constexpr int NBK = 10;
struct SlabArgs {
    const uint2* stream; const float* X; float* Y; const float* acc_in; float* acc_out; float s_in, s_out;
    int d, n_light, light_blocks, n_heavy;
    int first[NBK];      // first work index of bucket b (buckets in descending unit count); first[0] = 0
    int units[NBK];      // units per record in bucket b
    int base[NBK];       // pair offset of the bucket's first record
    const i32x4* hdesc; const int* col; const float* val;     // heavy rows: as before
    unsigned long long* clk;
    int abl;
};

// broadcast lane Q of every group of 8 lanes to the group WITHOUT the LDS crossbar: two DPP moves (row_newbcast, gfx90a+)
template <int Q> __device__ __forceinline__ int bcast8(int v) {
#ifdef USE_DPP
    int r = __builtin_amdgcn_update_dpp(0, v, 0x150 + Q, 0xF, 0x3, false);      // lanes 0-7 of each row of 16 <- lane Q
    return __builtin_amdgcn_update_dpp(r, v, 0x150 + Q + 8, 0xF, 0xC, false);   // lanes 8-15 <- lane Q + 8
#else
    return __shfl(v, Q, 8);
#endif
}
#define SLAB_LD(Q)                                                                                   \
    float v##Q = 0.f; f32x4 x##Q = {0.f, 0.f, 0.f, 0.f};                                             \
    if (Q < NQ) {                                                                                    \
        const int cc = bcast8<(Q0 + Q) & 7>((int)u.x);                                               \
        v##Q = __int_as_float(bcast8<(Q0 + Q) & 7>((int)u.y));                                       \
        if (ebase + Q < cnt) x##Q = reinterpret_cast<const f32x4*>(a.X + (long)cc * dmul)[c];        \
    }
#define SLAB_FM(Q) if (Q < NQ && ebase + Q < cnt) fma4(acc, v##Q, x##Q);
template <int Q0, int NQ, int EOFF> struct Batch {
    // gathers of edges ebase + q (q < NQ) whose (col, val) sit in lane Q0 + q of `u`, then the fma chain in edge order
    static __device__ __forceinline__ void run(const SlabArgs& a, const uint2& u, int cnt, long dmul, int c, f32x4& acc, int ebase) {
        SLAB_LD(0) SLAB_LD(1) SLAB_LD(2) SLAB_LD(3) SLAB_LD(4) SLAB_LD(5) SLAB_LD(6) SLAB_LD(7)
        SLAB_FM(0) SLAB_FM(1) SLAB_FM(2) SLAB_FM(3) SLAB_FM(4) SLAB_FM(5) SLAB_FM(6) SLAB_FM(7)
    }
};

template <int R, int PF, int OCC>
__global__ __launch_bounds__(256, OCC) void spmm_slab(SlabArgs a) {
    ClkScope clk_scope(a.clk);
    constexpr int G = 8, GPB = 32;
    const int lig = threadIdx.x & 7;
    const int xcd = blockIdx.x & 7;
    const int slice = xcd & 3;
    const long j = (long)(blockIdx.x >> 3) * 2 + (xcd >> 2);
    const int c = slice * G + lig;
    if (j >= a.n_heavy) {
        const long jl = j - a.n_heavy;
        if (jl >= a.light_blocks || (a.abl & 16)) return;
        const long dmul = (a.abl & 4) ? 0 : a.d;
        const int stride = a.light_blocks * GPB;
        const int w0 = (int)jl * GPB + (threadIdx.x >> 3);
        int base[R], units[R];
        uint2 u[R][PF];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int w = w0 + r * stride;
            int bs = a.base[0], fs = 0, un = a.units[0];
#pragma unroll
            for (int b = 1; b < NBK; ++b)
                if (w >= a.first[b]) { bs = a.base[b]; fs = a.first[b]; un = a.units[b]; }
            base[r] = bs + (w - fs) * un * 8;
            units[r] = w < a.n_light ? un : 0;
#pragma unroll
            for (int k = 0; k < PF; ++k) u[r][k] = k < units[r] ? a.stream[base[r] + k * 8 + lig] : uint2{0u, 0u};
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int mu = units[r];
#pragma unroll
            for (int off = G; off < 64; off <<= 1) { const int o = __shfl_xor(mu, off); mu = o > mu ? o : mu; }
            if (mu == 0) continue;                                   // wave-uniform
            const int row = bcast8<0>((int)u[r][0].x), cnt = bcast8<0>((int)u[r][0].y);
            const long o = (long)row * a.d + (long)c * 4;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (units[r] && !(a.abl & 2)) z = *reinterpret_cast<const f32x4*>(a.acc_in + o);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // units beyond the prefetched ones: requested now, consumed after the first PF units' gathers
            Batch<1, 7, 0>::run(a, u[r][0], cnt, dmul, c, acc, 0);   // unit 0: edges 0..6 sit in lanes 1..7
#pragma unroll
            for (int k = 1; k < PF; ++k)
                if (k < mu) Batch<0, 8, 0>::run(a, u[r][k], cnt, dmul, c, acc, 7 + (k - 1) * 8);      // wave-uniform
            if (mu > PF) {
                uint2 nxt = PF < units[r] ? a.stream[base[r] + PF * 8 + lig] : uint2{0u, 0u};
                for (int k = PF; k < mu; ++k) {
                    const uint2 cur = nxt;
                    if (k + 1 < mu) nxt = k + 1 < units[r] ? a.stream[base[r] + (k + 1) * 8 + lig] : uint2{0u, 0u};
                    Batch<0, 8, 0>::run(a, cur, cnt, dmul, c, acc, 7 + (k - 1) * 8);
                }
            }
            if (units[r]) {
                if (a.abl & 1) { if (acc.x == 12345.f) a.Y[0] = z.x; }
                else {
                    *reinterpret_cast<f32x4*>(a.Y + o) = acc;
                    f32x4 rr;
                    rr.x = (z.x * a.s_in + acc.x) * a.s_out; rr.y = (z.y * a.s_in + acc.y) * a.s_out;
                    rr.z = (z.z * a.s_in + acc.z) * a.s_out; rr.w = (z.w * a.s_in + acc.w) * a.s_out;
                    *reinterpret_cast<f32x4*>(a.acc_out + o) = rr;
                }
            }
        }
        return;
    }
    if (a.abl & 8) return;
    // heavy rows as in the lane-group kernel
    __shared__ f32x4 wsum[4][G];
    const i32x4 hd = a.hdesc[j];
    const long r0 = hd.y, r1 = (long)hd.y + hd.z;
    const int gg = threadIdx.x / G;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        Args aa{}; aa.col = a.col; aa.val = a.val; aa.X = a.X; aa.d = a.d; aa.abl = a.abl;
        long chunk = (r1 - r0 + GPB - 1) / GPB;
        chunk = (chunk + 7) & ~(long)7;
        const long e0 = r0 + (long)gg * chunk;
        const long e1 = e0 + chunk < r1 ? e0 + chunk : r1;
        if (e0 < r1) row_edges_seq<G>(aa, e0, e1, c, lig, acc);
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
        acc.x += __shfl_down(acc.x, off); acc.y += __shfl_down(acc.y, off);
        acc.z += __shfl_down(acc.z, off); acc.w += __shfl_down(acc.w, off);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < G) wsum[wv][lig] = acc;
    __syncthreads();
    if (threadIdx.x < G) {
        f32x4 t0 = wsum[0][lig], t1 = wsum[1][lig], t2 = wsum[2][lig], t3 = wsum[3][lig], r;
        r.x = (t0.x + t1.x) + (t2.x + t3.x); r.y = (t0.y + t1.y) + (t2.y + t3.y);
        r.z = (t0.z + t1.z) + (t2.z + t3.z); r.w = (t0.w + t1.w) + (t2.w + t3.w);
        const long o = (long)hd.x * a.d + (long)c * 4;
        const f32x4 z = *reinterpret_cast<const f32x4*>(a.acc_in + o);
        *reinterpret_cast<f32x4*>(a.Y + o) = r;
        f32x4 rr;
        rr.x = (z.x * a.s_in + r.x) * a.s_out; rr.y = (z.y * a.s_in + r.y) * a.s_out;
        rr.z = (z.z * a.s_in + r.z) * a.s_out; rr.w = (z.w * a.s_in + r.w) * a.s_out;
        *reinterpret_cast<f32x4*>(a.acc_out + o) = rr;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// "Octet" light path.  What the probes above showed: every variant of the lane-group kernel lands at 17 us however its
// dependent chain or its cache behaviour is changed, and gathers pointed at one cached row with no stores still take 12 us:
// the kernel is VALU-ISSUE bound (~140 wave-instructions per 8 gathers: 16 cross-lane broadcasts with their index
// arithmetic, 64-bit address arithmetic, per-gather predicates, 32 scalar fmas).  Here:
//  * the stream holds BYTE OFFSETS (col * d * 4), so a gather address is one 32-bit add and the load takes the table
//    base from SGPRs (global_load ... v_off, s[base]);
//  * rows are padded to whole 64-byte units with (offset of a real neighbour, 0.f): fma(0, x, acc) == acc exactly (acc is
//    never -0, X finite), so there are no per-gather predicates;
//  * a WAVE works on an "octet": 8 consecutive records of ONE bucket (same unit count; buckets padded to 8 records), so
//    bucket lookup, unit count and loop bounds are scalar;
//  * a unit's pairs are parked in LDS by the wave and every lane reads its group's pair q with ONE ds_read_b64 (the LDS
//    port, not the VALU): no DPP / bpermute broadcasts;
//  * fmas are packed (v_pk_fma_f32): 2 per gather.
// Per 8 gathers: 8 ds_read_b64 + 8 v_add + 8 global_load + 16 v_pk_fma.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct OctArgs {
    const uint2* stream; const char* X; char* Y; const char* acc_in; char* acc_out; float s_in, s_out;
    int d, n_oct, light_blocks, n_heavy;
    int first[NBK];      // first octet of bucket b
    int units[NBK];
    int base[NBK];       // pair offset of the bucket's first record
    const i32x4* hdesc; const int* col; const float* val;
    unsigned long long* clk;
    int abl;
};
constexpr int OCT_MAXU = 9;

__device__ __forceinline__ void pkfma(f32x4& acc, float v, const f32x4& x) {
    f32x2 lo = {acc.x, acc.y}, hi = {acc.z, acc.w};
    const f32x2 vv = {v, v};
    lo = __builtin_elementwise_fma(vv, f32x2{x.x, x.y}, lo);
    hi = __builtin_elementwise_fma(vv, f32x2{x.z, x.w}, hi);
    acc = f32x4{lo.x, lo.y, hi.x, hi.y};
}

#define OCT_LD(Q)                                                                                    \
    const uint2 p##Q = lds_rec[(k * 8 + Q)];                                                         \
    const f32x4 x##Q = *reinterpret_cast<const f32x4*>(a.X + (p##Q.x + lane_off));
#define OCT_FM(Q) pkfma(acc, __uint_as_float(p##Q.y), x##Q);

template <int R, int OCC>
__global__ __launch_bounds__(256, OCC) void spmm_oct(OctArgs a) {
    ClkScope clk_scope(a.clk);
    constexpr int G = 8;
    __shared__ uint2 lds[4][8][OCT_MAXU * 8];            // [wave][group][unit * 8 + pair]
    const int lig = threadIdx.x & 7, grp = (threadIdx.x >> 3) & 7;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7;
    const int slice = xcd & 3;
    const int j = (int)(blockIdx.x >> 3) * 2 + (xcd >> 2);
    const unsigned lane_off = (unsigned)(slice * G + lig) * 16u;
    if (j >= a.n_heavy) {
        const int jl = j - a.n_heavy;
        if (jl >= a.light_blocks || (a.abl & 16)) return;
        uint2* lds_rec = &lds[wv][grp][0];
        const int n_waves = a.light_blocks * 4;
#pragma unroll 1
        for (int o = jl * 4 + wv; o < a.n_oct; o += n_waves) {
            int bs = a.base[0], fs = 0, un = a.units[0];
#pragma unroll
            for (int b = 1; b < NBK; ++b)
                if (o >= a.first[b]) { bs = a.base[b]; fs = a.first[b]; un = a.units[b]; }
            const uint2* rec = a.stream + bs + (long)((o - fs) * 8 + grp) * un * 8;
            // the whole record -> LDS (one pair per lane and unit)
            for (int k = 0; k < un; ++k) lds_rec[k * 8 + lig] = rec[k * 8 + lig];
            // wave-private region, LDS ops of a wave execute in order: no barrier
            const uint2 hdr = lds_rec[0];
            const bool live = (int)hdr.y >= 0;                       // dummy records pad a bucket to whole octets
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (live && !(a.abl & 2)) z = *reinterpret_cast<const f32x4*>(a.acc_in + (hdr.x + lane_off));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            {
                constexpr int k = 0;
                OCT_LD(1) OCT_LD(2) OCT_LD(3) OCT_LD(4) OCT_LD(5) OCT_LD(6) OCT_LD(7)
                OCT_FM(1) OCT_FM(2) OCT_FM(3) OCT_FM(4) OCT_FM(5) OCT_FM(6) OCT_FM(7)
            }
            for (int k = 1; k < un; ++k) {
                OCT_LD(0) OCT_LD(1) OCT_LD(2) OCT_LD(3) OCT_LD(4) OCT_LD(5) OCT_LD(6) OCT_LD(7)
                OCT_FM(0) OCT_FM(1) OCT_FM(2) OCT_FM(3) OCT_FM(4) OCT_FM(5) OCT_FM(6) OCT_FM(7)
            }
            if (live) {
                if (a.abl & 1) { if (acc.x == 12345.f) *reinterpret_cast<float*>(a.Y) = z.x; }
                else {
                    *reinterpret_cast<f32x4*>(a.Y + (hdr.x + lane_off)) = acc;
                    f32x4 rr;
                    rr.x = (z.x * a.s_in + acc.x) * a.s_out; rr.y = (z.y * a.s_in + acc.y) * a.s_out;
                    rr.z = (z.z * a.s_in + acc.z) * a.s_out; rr.w = (z.w * a.s_in + acc.w) * a.s_out;
                    *reinterpret_cast<f32x4*>(a.acc_out + (hdr.x + lane_off)) = rr;
                }
            }
        }
        return;
    }
    if (a.abl & 8) return;
    __shared__ f32x4 wsum[4][G];
    const int c = slice * G + lig;
    const i32x4 hd = a.hdesc[j];
    const long r0 = hd.y, r1 = (long)hd.y + hd.z;
    const int gg = threadIdx.x / G;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        Args aa{}; aa.col = a.col; aa.val = a.val; aa.X = reinterpret_cast<const float*>(a.X); aa.d = a.d; aa.abl = a.abl;
        long chunk = (r1 - r0 + 31) / 32;
        chunk = (chunk + 7) & ~(long)7;
        const long e0 = r0 + (long)gg * chunk;
        const long e1 = e0 + chunk < r1 ? e0 + chunk : r1;
        if (e0 < r1) row_edges_seq<G>(aa, e0, e1, c, lig, acc);
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
        acc.x += __shfl_down(acc.x, off); acc.y += __shfl_down(acc.y, off);
        acc.z += __shfl_down(acc.z, off); acc.w += __shfl_down(acc.w, off);
    }
    if ((threadIdx.x & 63) < G) wsum[wv][lig] = acc;
    __syncthreads();
    if (threadIdx.x < G) {
        f32x4 t0 = wsum[0][lig], t1 = wsum[1][lig], t2 = wsum[2][lig], t3 = wsum[3][lig], r;
        r.x = (t0.x + t1.x) + (t2.x + t3.x); r.y = (t0.y + t1.y) + (t2.y + t3.y);
        r.z = (t0.z + t1.z) + (t2.z + t3.z); r.w = (t0.w + t1.w) + (t2.w + t3.w);
        const long o = ((long)hd.x * a.d + (long)c * 4) * 4;
        const f32x4 z = *reinterpret_cast<const f32x4*>(a.acc_in + o);
        *reinterpret_cast<f32x4*>(a.Y + o) = r;
        f32x4 rr;
        rr.x = (z.x * a.s_in + r.x) * a.s_out; rr.y = (z.y * a.s_in + r.y) * a.s_out;
        rr.z = (z.z * a.s_in + r.z) * a.s_out; rr.w = (z.w * a.s_in + r.w) * a.s_out;
        *reinterpret_cast<f32x4*>(a.acc_out + o) = rr;
    }
}

struct Graph { long n_u, n_i, nnz; std::vector<long> rp; std::vector<int> col; std::vector<float> val; };

static Graph load(const char* path) {
    Graph g; FILE* f = fopen(path, "rb"); if (!f) { printf("cannot open %s\n", path); exit(1); }
    long h[3]; if (fread(h, 8, 3, f) != 3) exit(1);
    g.n_u = h[0]; g.n_i = h[1]; g.nnz = h[2];
    const long n = g.n_u + g.n_i;
    g.rp.resize(n + 1); g.col.resize(g.nnz); g.val.resize(g.nnz);
    if (fread(g.rp.data(), 8, n + 1, f) != (size_t)(n + 1)) exit(1);
    if (fread(g.col.data(), 4, g.nnz, f) != (size_t)g.nnz) exit(1);
    if (fread(g.val.data(), 4, g.nnz, f) != (size_t)g.nnz) exit(1);
    fclose(f); return g;
}

template <typename T> T* up(const std::vector<T>& v) {
    T* p; CK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "tools/probes/data/citeulike.bin";
    const int seg = argc > 2 ? atoi(argv[2]) : 64;
    Graph g = load(path);
    const long N = g.n_u + g.n_i; const int d = 128;
    printf("graph %s: %ld users + %ld items, %ld edges, heavy threshold %d\n", path, g.n_u, g.n_i, g.nnz, seg);
    // work lists: class 0 = user rows, class 1 = item rows (map 1); class "all" for map 0 -- descending length, heavy apart
    auto build = [&](long lo, long hi, std::vector<i32x4>& light, std::vector<i32x4>& heavy) {
        std::vector<int> rows(hi - lo); std::iota(rows.begin(), rows.end(), (int)lo);
        std::stable_sort(rows.begin(), rows.end(), [&](int x, int y) { return g.rp[x + 1] - g.rp[x] > g.rp[y + 1] - g.rp[y]; });
        for (int r : rows) {
            const int cnt = (int)(g.rp[r + 1] - g.rp[r]);
            i32x4 dsc = {r, (int)g.rp[r], cnt, cnt > seg};
            static const int hsplit = getenv("HSPLIT") ? atoi(getenv("HSPLIT")) : 0;
            if (cnt > seg && hsplit > 0 && cnt > hsplit) {       // timing experiment: parts of a heavy row as separate workgroups
                const int parts = (cnt + hsplit - 1) / hsplit, per = ((cnt + parts - 1) / parts + 7) & ~7;
                for (int e = 0; e < cnt; e += per) heavy.push_back(i32x4{r, (int)g.rp[r] + e, std::min(per, cnt - e), 1});
                continue;
            }
            (cnt > seg ? heavy : light).push_back(dsc);
        }
        std::sort(heavy.begin(), heavy.end(), [](const i32x4& x, const i32x4& y) { return x.z > y.z; });
    };
    std::vector<i32x4> lA, hA, lU, hU, lI, hI;
    build(0, N, lA, hA); build(0, g.n_u, lU, hU); build(g.n_u, N, lI, hI);
    printf("light/heavy: all %zu/%zu  users %zu/%zu  items %zu/%zu\n", lA.size(), hA.size(), lU.size(), hU.size(), lI.size(), hI.size());
    int* dcol = up(g.col); float* dval = up(g.val);
    std::vector<float> hX((size_t)N * d), hZ((size_t)N * d);
    srand(7);
    for (auto& v : hX) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hZ) v = (rand() % 2001 - 1000) * 1e-3f;
    float *X = up(hX), *Z = up(hZ), *Y, *O;
    CK(hipMalloc(&Y, (size_t)N * d * 4)); CK(hipMalloc(&O, (size_t)N * d * 4));
    i32x4 *dlA = up(lA), *dhA = up(hA), *dlU = up(lU), *dhU = up(hU), *dlI = up(lI), *dhI = up(hI);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* clkbuf; CK(hipMalloc(&clkbuf, 1 << 20));

    auto run = [&](const char* name, int map, int R, int pipe, int H, int heavy_first, int abl) {
        Args a{dcol, dval, X, Y, Z, O, 1.0f, 0.25f, d, {nullptr, nullptr}, {0, 0}, {nullptr, nullptr}, {0, 0}, {0, 0}, map, heavy_first, 4, nullptr, abl};
        constexpr int GPB = 32;
        long grid;
        if (map == 0) {
            a.desc[0] = dlA; a.n_light[0] = (int)lA.size(); a.hdesc[0] = dhA; a.n_heavy[0] = (int)hA.size();
            a.light_blocks[0] = (int)((lA.size() + GPB * R - 1) / (GPB * R));
            const long per_slice = a.light_blocks[0] + a.n_heavy[0];
            grid = ((per_slice + 1) / 2) * 8;
        } else {
            a.desc[0] = dlU; a.n_light[0] = (int)lU.size(); a.hdesc[0] = dhU; a.n_heavy[0] = (int)hU.size();
            a.desc[1] = dlI; a.n_light[1] = (int)lI.size(); a.hdesc[1] = dhI; a.n_heavy[1] = (int)hI.size();
            for (int k = 0; k < 2; ++k) a.light_blocks[k] = (int)((a.n_light[k] + GPB * R - 1) / (GPB * R));
            grid = 8L * std::max(a.light_blocks[0] + a.n_heavy[0], a.light_blocks[1] + a.n_heavy[1]);
        }
        auto launch = [&]() {
#define LK(RR, PP, HH) hipLaunchKernelGGL((spmm_k<8, RR, PP, HH>), dim3((unsigned)grid), dim3(256), 0, 0, a)
            if (!pipe) { LK(1, 0, 4); return; }
            if (H == 4) { switch (R) { case 1: LK(1, 1, 4); break; case 2: LK(2, 1, 4); break; case 3: LK(3, 1, 4); break; default: LK(4, 1, 4); } }
            else        { switch (R) { case 1: LK(1, 1, 8); break; case 2: LK(2, 1, 8); break; case 3: LK(3, 1, 8); break; default: LK(4, 1, 8); } }
        };
        CK(hipMemset(Y, 0xff, (size_t)N * d * 4)); CK(hipMemset(O, 0xff, (size_t)N * d * 4));
        launch(); CK(hipDeviceSynchronize());
        // check sampled light rows bitwise against the CPU chain, heavy rows loosely
        std::vector<float> y((size_t)N * d), o((size_t)N * d);
        CK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), O, o.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0, badh = 0;
        for (long r = 0; r < N; r += 37) {
            const bool heavy = g.rp[r + 1] - g.rp[r] > seg;
            for (int k = 0; k < d; k += 5) {
                float s = 0.f;
                for (long e = g.rp[r]; e < g.rp[r + 1]; ++e) s = fmaf(g.val[e], hX[(size_t)g.col[e] * d + k], s);
                const float w = (hZ[r * d + k] * 1.0f + s) * 0.25f;
                if (!heavy) { if (memcmp(&s, &y[r * d + k], 4) || memcmp(&w, &o[r * d + k], 4)) ++bad; }
                else if (fabsf(s - y[r * d + k]) > 1e-4f * (1.f + fabsf(s))) ++badh;
            }
        }
        for (int w = 0; w < 5; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, tot = 0.f; const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); tot += ms;
        }
        // back to back (what a captured step sees): 30 launches between two events
        CK(hipEventRecord(e0)); for (int rep = 0; rep < 30; ++rep) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms30; CK(hipEventElapsedTime(&ms30, e0, e1));
        if (clkbuf) {
            a.clk = clkbuf;
            CK(hipMemset(clkbuf, 0, (size_t)grid * 16));
            launch(); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)grid * 2);
            CK(hipMemcpy(h.data(), clkbuf, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (long b = 0; b < grid; ++b) if (h[2 * b]) { t0 = std::min(t0, h[2 * b]); t1 = std::max(t1, h[2 * b + 1]); }
            // histogram over 0.5 us buckets: blocks started, blocks finished, blocks alive at the bucket's start
            const int nb = (int)((t1 - t0) / 50) + 1;
            std::vector<int> st(nb, 0), fi(nb, 0);
            double dur = 0; long cntb = 0; unsigned long long dmax = 0;
            for (long b = 0; b < grid; ++b) if (h[2 * b]) {
                st[(h[2 * b] - t0) / 50]++; fi[(h[2 * b + 1] - t0) / 50]++;
                dur += (double)(h[2 * b + 1] - h[2 * b]); ++cntb; dmax = std::max(dmax, h[2 * b + 1] - h[2 * b]);
            }
            printf("   clocks: first start -> last end %.2f us, mean block life %.2f us, max %.2f us; per 0.5 us [started/finished/alive]:\n   ",
                   (t1 - t0) * 0.01, dur / cntb * 0.01, dmax * 0.01);
            int alive = 0;
            for (int k = 0; k < nb; ++k) { printf(" %d/%d/%d", st[k], fi[k], alive + st[k]); alive += st[k] - fi[k]; }
            printf("\n");
            a.clk = nullptr;
        }
        printf("%-46s grid %5ld  best %.1f us  avg %.1f us  back-to-back %.1f us  %s\n", name, grid, best * 1e3, tot / reps * 1e3,
               ms30 / 30 * 1e3, abl ? "(ablated)" : bad || badh ? "MISMATCH" : "ok");
        if (!abl && (bad || badh)) printf("   mismatches: light %ld heavy %ld\n", bad, badh);
    };

    // ---- slab stream over the light rows (lA is sorted by descending length)
    auto units_of = [](int cnt) { return cnt <= 7 ? 1 : 1 + (cnt - 7 + 7) / 8; };
    std::vector<uint2> stream;
    SlabArgs sa{};
    {
        int nb = 0;
        for (size_t w = 0; w < lA.size(); ++w) {
            const int cnt = lA[w].z, un = units_of(cnt);
            if (nb == 0 || sa.units[nb - 1] != un) {
                if (nb == NBK) { printf("too many buckets\n"); return 1; }
                sa.first[nb] = (int)w; sa.units[nb] = un; sa.base[nb] = (int)stream.size(); ++nb;
            }
            const size_t at = stream.size();
            stream.resize(at + (size_t)un * 8, uint2{0u, 0u});
            stream[at] = uint2{(unsigned)lA[w].x, (unsigned)cnt};
            for (int e = 0; e < cnt; ++e) {
                float v = g.val[lA[w].y + e]; unsigned vb; memcpy(&vb, &v, 4);
                stream[at + 1 + e] = uint2{(unsigned)g.col[lA[w].y + e], vb};
            }
        }
        for (int b = nb; b < NBK; ++b) { sa.first[b] = 0x7fffffff; sa.units[b] = 1; sa.base[b] = 0; }
        stream.resize(stream.size() + 16 * 8, uint2{0u, 0u});
        printf("slab stream: %zu pairs = %.2f MB (CSR col+val: %.2f MB), %d buckets:", stream.size(), stream.size() * 8e-6, g.nnz * 8e-6, nb);
        for (int b = 0; b < nb; ++b) printf(" u%d@%d", sa.units[b], sa.first[b]);
        printf("\n");
    }
    uint2* dstream = up(stream);
    auto run_slab = [&](const char* name, int R, int PF, int abl, int occ = 6) {
        SlabArgs a = sa;
        a.stream = dstream; a.X = X; a.Y = Y; a.acc_in = Z; a.acc_out = O; a.s_in = 1.f; a.s_out = 0.25f; a.d = d;
        a.n_light = (int)lA.size(); a.n_heavy = (int)hA.size(); a.hdesc = dhA; a.col = dcol; a.val = dval; a.clk = nullptr; a.abl = abl;
        a.light_blocks = (a.n_light + 32 * R - 1) / (32 * R);
        const long grid = ((a.n_heavy + a.light_blocks + 1) / 2) * 8;
        auto launch = [&]() {
#define LS(RR, PP, OO) hipLaunchKernelGGL((spmm_slab<RR, PP, OO>), dim3((unsigned)grid), dim3(256), 0, 0, a)
            if (occ == 6) {
                if (PF == 2) { switch (R) { case 1: LS(1, 2, 6); break; case 2: LS(2, 2, 6); break; case 3: LS(3, 2, 6); break; default: LS(4, 2, 6); } }
                else         { switch (R) { case 1: LS(1, 3, 6); break; case 2: LS(2, 3, 6); break; case 3: LS(3, 3, 6); break; default: LS(4, 3, 6); } }
            } else {
                if (PF == 2) { switch (R) { case 1: LS(1, 2, 4); break; case 2: LS(2, 2, 4); break; case 3: LS(3, 2, 4); break; default: LS(4, 2, 4); } }
                else         { switch (R) { case 1: LS(1, 3, 4); break; case 2: LS(2, 3, 4); break; case 3: LS(3, 3, 4); break; default: LS(4, 3, 4); } }
            }
        };
        CK(hipMemset(Y, 0xff, (size_t)N * d * 4)); CK(hipMemset(O, 0xff, (size_t)N * d * 4));
        launch(); CK(hipDeviceSynchronize());
        std::vector<float> y((size_t)N * d), o((size_t)N * d);
        CK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), O, o.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0, badh = 0;
        for (long r = 0; r < N; r += 7) {
            const bool heavy = g.rp[r + 1] - g.rp[r] > seg;
            for (int k = 0; k < d; k += 5) {
                float sacc = 0.f;
                for (long e = g.rp[r]; e < g.rp[r + 1]; ++e) sacc = fmaf(g.val[e], hX[(size_t)g.col[e] * d + k], sacc);
                const float w = (hZ[r * d + k] * 1.0f + sacc) * 0.25f;
                if (!heavy) { if (memcmp(&sacc, &y[r * d + k], 4) || memcmp(&w, &o[r * d + k], 4)) ++bad; }
                else if (fabsf(sacc - y[r * d + k]) > 1e-4f * (1.f + fabsf(sacc))) ++badh;
            }
        }
        for (int w = 0; w < 5; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        CK(hipEventRecord(e0)); for (int rep = 0; rep < 30; ++rep) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms30; CK(hipEventElapsedTime(&ms30, e0, e1));
        {
            a.clk = clkbuf;
            CK(hipMemset(clkbuf, 0, (size_t)grid * 16));
            launch(); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)grid * 2);
            CK(hipMemcpy(h.data(), clkbuf, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (long b = 0; b < grid; ++b) if (h[2 * b]) { t0 = std::min(t0, h[2 * b]); t1 = std::max(t1, h[2 * b + 1]); }
            const int nbk = (int)((t1 - t0) / 50) + 1;
            std::vector<int> st(nbk, 0), fi(nbk, 0);
            double dur = 0; long cntb = 0; unsigned long long dmax = 0;
            for (long b = 0; b < grid; ++b) if (h[2 * b]) {
                st[(h[2 * b] - t0) / 50]++; fi[(h[2 * b + 1] - t0) / 50]++;
                dur += (double)(h[2 * b + 1] - h[2 * b]); ++cntb; dmax = std::max(dmax, h[2 * b + 1] - h[2 * b]);
            }
            printf("   clocks: first start -> last end %.2f us, mean block life %.2f us, max %.2f us; per 0.5 us [started/finished/alive]:\n   ",
                   (t1 - t0) * 0.01, dur / cntb * 0.01, dmax * 0.01);
            int alive = 0;
            for (int k = 0; k < nbk; ++k) { printf(" %d/%d/%d", st[k], fi[k], alive + st[k]); alive += st[k] - fi[k]; }
            printf("\n");
            a.clk = nullptr;
        }
        printf("%-46s grid %5ld  best %.1f us  back-to-back %.1f us  %s\n", name, grid, best * 1e3, ms30 / 30 * 1e3,
               abl ? "(ablated)" : bad || badh ? "MISMATCH" : "ok");
        if (!abl && (bad || badh)) printf("   mismatches: light %ld heavy %ld\n", bad, badh);
    };


    // ---- two half-launches (user rows / item rows): one stream vs two concurrent streams
    {
        hipStream_t sA, sB; CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
        hipEvent_t fork, joinB; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&joinB, hipEventDisableTiming));
        auto half_args = [&](int cls, int R) {
            Args a{dcol, dval, X, Y, Z, O, 1.0f, 0.25f, d, {nullptr, nullptr}, {0, 0}, {nullptr, nullptr}, {0, 0}, {0, 0}, 0, 1, 4, nullptr, 0};
            a.desc[0] = cls ? dlI : dlU; a.n_light[0] = (int)(cls ? lI.size() : lU.size());
            a.hdesc[0] = cls ? dhI : dhU; a.n_heavy[0] = (int)(cls ? hI.size() : hU.size());
            a.light_blocks[0] = (a.n_light[0] + 32 * R - 1) / (32 * R);
            return a;
        };
        for (int R : {1, 2, 3}) {
            Args aU = half_args(0, R), aI = half_args(1, R);
            const long gU = ((aU.light_blocks[0] + aU.n_heavy[0] + 1) / 2) * 8, gI = ((aI.light_blocks[0] + aI.n_heavy[0] + 1) / 2) * 8;
            auto lU_ = [&](hipStream_t st) { hipLaunchKernelGGL((spmm_k<8, 1, 0, 4>), dim3((unsigned)gU), dim3(256), 0, st, aU); };
            auto lI_ = [&](hipStream_t st) { hipLaunchKernelGGL((spmm_k<8, 1, 0, 4>), dim3((unsigned)gI), dim3(256), 0, st, aI); };
            auto timeit = [&](const char* name, auto body) {
                for (int w = 0; w < 5; ++w) body();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, sA)); for (int rep = 0; rep < 30; ++rep) body(); CK(hipEventRecord(e1, sA)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("halves R%d  %-52s grids %ld + %ld  %.1f us per pair\n", R, name, gU, gI, ms / 30 * 1e3);
            };
            timeit("user rows only", [&] { lU_(sA); });
            timeit("item rows only", [&] { lI_(sA); });

            // item rows at FULL WIDTH (512-B gathers, no column slices: the user half they gather is 2.8 MB)
            {
                Args aF = aI; aF.cs = 1;
                for (int RF : {1, 2, 4}) {
                    aF.light_blocks[0] = (aF.n_light[0] + 8 * RF - 1) / (8 * RF);
                    const long gF = ((aF.light_blocks[0] + aF.n_heavy[0] + 7) / 8) * 8;
                    char nm[96]; snprintf(nm, sizeof nm, "item rows only, full width, %d rows per group (grid %ld)", RF, gF);
                    timeit(nm, [&] { hipLaunchKernelGGL((spmm_k<32, 1, 0, 4>), dim3((unsigned)gF), dim3(256), 0, sA, aF); });
                }
                Args aG = aU; aG.cs = 1;
                aG.light_blocks[0] = (aG.n_light[0] + 8 * 1 - 1) / 8;
                const long gG = ((aG.light_blocks[0] + aG.n_heavy[0] + 7) / 8) * 8;
                timeit("user rows only, full width", [&] { hipLaunchKernelGGL((spmm_k<32, 1, 0, 4>), dim3((unsigned)gG), dim3(256), 0, sA, aG); });
            }
            timeit("both, one stream", [&] { lU_(sA); lI_(sA); });
            timeit("both, two streams (fork / join per pair)", [&] {
                CK(hipEventRecord(fork, sA)); CK(hipStreamWaitEvent(sB, fork, 0));
                lU_(sA); lI_(sB);
                CK(hipEventRecord(joinB, sB)); CK(hipStreamWaitEvent(sA, joinB, 0));
            });
            // two independent chains of 3 dependent half-launches each (the bipartite structure of L = 3 layers)
            timeit("3 layers: 6 half-launches, one stream", [&] { for (int l = 0; l < 3; ++l) { lU_(sA); lI_(sA); } });
            timeit("3 layers: two chains on two streams", [&] {
                CK(hipEventRecord(fork, sA)); CK(hipStreamWaitEvent(sB, fork, 0));
                for (int l = 0; l < 3; ++l) { if (l & 1) { lI_(sA); lU_(sB); } else { lU_(sA); lI_(sB); } }
                CK(hipEventRecord(joinB, sB)); CK(hipStreamWaitEvent(sA, joinB, 0));
            });
        }
    }
    if (getenv("RUN_OCT")) {
    // ---- octet stream: byte offsets, padded units, buckets padded to whole octets
    std::vector<uint2> ostream;
    OctArgs oa{};
    {
        int nb = 0, n_oct = 0;
        size_t w = 0;
        while (w < lA.size()) {
            const int un = units_of(lA[w].z);
            size_t w1 = w;
            while (w1 < lA.size() && units_of(lA[w1].z) == un) ++w1;
            if (nb == NBK) { printf("too many buckets\n"); return 1; }
            oa.first[nb] = n_oct; oa.units[nb] = un; oa.base[nb] = (int)ostream.size(); ++nb;
            const size_t recs = (w1 - w + 7) / 8 * 8;
            for (size_t k = 0; k < recs; ++k) {
                const size_t at = ostream.size();
                ostream.resize(at + (size_t)un * 8, uint2{0u, 0u});
                if (w + k < w1) {
                    const i32x4 ds = lA[w + k];
                    const unsigned self = (unsigned)ds.x * d * 4;
                    const unsigned pad_off = ds.z ? (unsigned)g.col[ds.y] * d * 4 : self;
                    ostream[at] = uint2{self, (unsigned)ds.z};
                    for (int e = 0; e < un * 8 - 1; ++e) {
                        if (e < ds.z) { float v = g.val[ds.y + e]; unsigned vb; memcpy(&vb, &v, 4); ostream[at + 1 + e] = uint2{(unsigned)g.col[ds.y + e] * d * 4, vb}; }
                        else ostream[at + 1 + e] = uint2{pad_off, 0u};
                    }
                } else ostream[at] = uint2{0u, 0xffffffffu};          // dummy: nothing stored
            }
            n_oct += (int)(recs / 8);
            w = w1;
        }
        for (int b = nb; b < NBK; ++b) { oa.first[b] = 0x7fffffff; oa.units[b] = 1; oa.base[b] = 0; }
        oa.n_oct = n_oct;
        printf("octet stream: %zu pairs = %.2f MB, %d octets, %d buckets\n", ostream.size(), ostream.size() * 8e-6, n_oct, nb);
    }
    uint2* dostream = up(ostream);
    auto run_oct = [&](const char* name, int R, int abl, int occ) {
        OctArgs a = oa;
        a.stream = dostream; a.X = (const char*)X; a.Y = (char*)Y; a.acc_in = (const char*)Z; a.acc_out = (char*)O; a.s_in = 1.f; a.s_out = 0.25f; a.d = d;
        a.n_heavy = (int)hA.size(); a.hdesc = dhA; a.col = dcol; a.val = dval; a.clk = nullptr; a.abl = abl;
        a.light_blocks = (a.n_oct + 4 * R - 1) / (4 * R);
        const long grid = ((a.n_heavy + a.light_blocks + 1) / 2) * 8;
        auto launch = [&]() {
#define LO(RR, OO) hipLaunchKernelGGL((spmm_oct<RR, OO>), dim3((unsigned)grid), dim3(256), 0, 0, a)
            if (occ == 6) LO(1, 6); else if (occ == 4) LO(1, 4); else LO(1, 8);
        };
        CK(hipMemset(Y, 0xff, (size_t)N * d * 4)); CK(hipMemset(O, 0xff, (size_t)N * d * 4));
        launch(); CK(hipDeviceSynchronize());
        std::vector<float> y((size_t)N * d), o((size_t)N * d);
        CK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), O, o.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0, badh = 0;
        for (long r = 0; r < N; r += 3) {
            const bool heavy = g.rp[r + 1] - g.rp[r] > seg;
            for (int k = 0; k < d; k += 5) {
                float sacc = 0.f;
                for (long e = g.rp[r]; e < g.rp[r + 1]; ++e) sacc = fmaf(g.val[e], hX[(size_t)g.col[e] * d + k], sacc);
                const float w = (hZ[r * d + k] * 1.0f + sacc) * 0.25f;
                if (!heavy) { if (memcmp(&sacc, &y[r * d + k], 4) || memcmp(&w, &o[r * d + k], 4)) ++bad; }
                else if (fabsf(sacc - y[r * d + k]) > 1e-4f * (1.f + fabsf(sacc))) ++badh;
            }
        }
        for (int w = 0; w < 5; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        CK(hipEventRecord(e0)); for (int rep = 0; rep < 30; ++rep) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms30; CK(hipEventElapsedTime(&ms30, e0, e1));
        printf("%-46s grid %5ld  best %.1f us  back-to-back %.1f us  %s\n", name, grid, best * 1e3, ms30 / 30 * 1e3,
               abl ? "(ablated)" : bad || badh ? "MISMATCH" : "ok");
        if (!abl && (bad || badh)) printf("   mismatches: light %ld heavy %ld\n", bad, badh);
    };
    for (int occ : {8, 6, 4})
        for (int R : {1, 2, 3, 4}) { char nm[64]; snprintf(nm, sizeof nm, "octet R%d occ%d", R, occ); run_oct(nm, R, 0, occ); }
    run_oct("octet R2 occ6 no heavy", 2, 8, 6);
    run_oct("octet R2 occ6 no heavy no stores no acc_in", 2, 11, 6);
    run_oct("octet R2 occ6 no light", 2, 16, 6);
    if (getenv("RUN_SLAB"))
    for (int occ : {6, 4}) {
        printf("---- occupancy target %d waves / SIMD\n", occ);
        run_slab("slab R1 PF2", 1, 2, 0, occ);
        run_slab("slab R2 PF2", 2, 2, 0, occ);
        run_slab("slab R3 PF2", 3, 2, 0, occ);
        run_slab("slab R4 PF2", 4, 2, 0, occ);
        run_slab("slab R2 PF3", 2, 3, 0, occ);
        run_slab("slab R3 PF3", 3, 3, 0, occ);
        run_slab("slab R3 PF2 no heavy", 3, 2, 8, occ);
        run_slab("slab R3 PF2 no heavy no stores no acc_in", 3, 2, 11, occ);
        run_slab("slab R3 PF2 no heavy, nothing", 3, 2, 15, occ);
    }
    }
    for (int pass = 0; pass < 1; ++pass) {
        run("map0 R3 seq (round 2)", 0, 3, 0, 4, 0, 0);
        run("map0 R2 pipe", 0, 2, 1, 4, 1, 0);
        run("map0 R1 seq heavy-first", 0, 1, 0, 4, 1, 0);
        run("map0 R2 pipe, no heavy no stores no acc_in", 0, 2, 1, 4, 1, 11);
        run("map0 R2 pipe, nothing (launch + desc + edges)", 0, 2, 1, 4, 1, 15);
    }
    return 0;
}
