// EXPERIMENT RECORD (round 2) -- NOT part of the library build.  A persistent one-launch LightGCN propagation (8 column
// slices pinned to XCDs, per-XCD barriers, register / LDS resident row state).  Correct (bit-identical light rows, six
// training steps equal to the layer-by-layer path, placement flag 0) but SLOWER on an MI355X at CiteULike size:
// 174 us per launch (3 layers) against 3 x 22.6 us for the separate SpMM launches.  Ablations (profile build): the two
// per-XCD barriers of a launch with NO row work cost 36 us (128 workgroups per counter at 4 workgroups per CU), the row
// work itself 27 us per layer -- halving the gather width to 64 B (8 slices) doubles the L2 requests per useful byte,
// and tools/probes/gather_probe.hip shows bare 64-B gathers moving half the bytes of 128-B gathers in the same time.
// Kept as a record of what was tried; it needs crh_common.h of coldrec_amd/csrc to compile.
// LightGCN propagation, ALL layers of a direction in ONE persistent launch (model/LightGCN.py:86-96 and its autograd).
//
// Why: at CiteULike size a step is six SpMM launches of ~22 us each, and the profile (profiles/r01_train_e_pmc.json) shows
// what they wait for -- every launch starts with cold L2s, so each XCD pulls its 2.9 MB column slice of the dense operand
// through the fabric again (26 % L2 misses, 2.9x the algorithmic bytes), on the critical path of the row gathers.
//
// The propagation acts on feature columns independently: Y[:, s] = A X[:, s].  So the d/4 float4 columns are cut into 8
// slices, slice s is given to the workgroups of ONE XCD (block b -> XCD b % 8, checked at run time, see below), and that XCD
// carries its slice through all L layers: what layer l+1 gathers was written by the same XCD in layer l and is still in its
// L2 (input + output slice: 2 x 1.4 MB).  Between layers only the workgroups of one XCD meet (an arrival counter in their
// own L2, no fabric round trip, no L2 write-back); the neighbour rows are read with sc1 loads (served by the L2, never by a
// CU's stale L1).  Rows are assigned STATICALLY (same lane group, same row, every layer), so the running layer sum (forward)
// or the row of dOUT (backward) and the head of the row's edge list stay in registers for the whole launch.
//
//   forward   OUT = (E0 + A E0 + ... + A^L E0) / (L + 1)                                  one launch instead of L
//   backward  g = c (dOUT + A dOUT + ... + A^L dOUT) by Horner, then torch.optim.Adam / SGD on E in the epilogue, dOUT
//             cleared for the next step                                                    one launch instead of L
//
// Arithmetic: per row the same edge-order fma chain and the same epilogue expressions as crh_spmm_csr_f32 /
// crh_spmm_csr_adam_f32 (light rows: bit-identical to the layer-by-layer path); rows above 64 edges are split over the
// lane groups of a wave (above 64 x groups: of a workgroup) and combined in a fixed order (deterministic).
//
// Placement: "block b runs on XCD b % 8" is what the dispatcher does, not a contract.  Every workgroup compares
// HW_REG_XCC_ID with b % 8 and raises a flag in `status` if they differ (results are then not trustworthy: a consumer on
// another XCD may read a stale L2 line); the host layer probes this once per engine and checks the flag with the losses.
// All spins are bounded (a timed-out wait also raises the flag), so a wrong assumption can never hang the GPU.
#include <math.h>
#include <stdlib.h>

#include "crh_common.h"

namespace {

constexpr int LG_THREADS = 256, LG_SLICES = 8, LG_MAXR = 3, LG_MAXL = 8, LG_BATCH = 8, LG_HEAD = 16;
constexpr int LG_LIGHT_MAX = 64;          // edges one lane group sums on its own

struct LgcnArgs {
    const int64_t* rowptr;
    const int32_t* col;
    const float* val;
    int64_t n_rows;
    int d, L;
    const int32_t *light_rows, *wave_rows, *wg_rows;     // rows by class, each list by descending degree
    int n_light, n_wave, n_wg;
    const float* X0;          // layer-0 operand: E0 (forward) or dOUT (backward)
    float* Xa;                // intermediate layer outputs (ping-pong)
    float* Xb;
    int backward;
    float* OUT;               // forward: mean of the layers
    float c;                  // 1 / (L + 1)
    float* dOUT;              // backward: == X0, its rows are cleared at the end when zero_dout
    int zero_dout;
    float* G_out;             // backward: also store the gradient of E (tests), or NULL
    float *p, *m, *v;         // backward: optimiser state (m, v NULL for SGD)
    AdamK k;
    float bc2_sqrt, nss;      // Adam: sqrt(1 - b2^t), -lr / (1 - b1^t); SGD: nss = -lr
    const float* step_scalars;
    int sgd;
    unsigned* sync;           // [LG_SLICES][LG_MAXL] arrival counters (zeroed by the launch function), then [1] STICKY status
    int debug;                // measurement only (CRH_PROFILE build, CRH_LGCN_DEBUG): 1 no barrier, 2 plain loads, 4 no light
                              // rows, 8 no split rows
};

__device__ __forceinline__ void lg_fma4(f32x4& acc, float v, const f32x4& x) {
    acc.x = fmaf(v, x.x, acc.x);
    acc.y = fmaf(v, x.y, acc.y);
    acc.z = fmaf(v, x.z, acc.z);
    acc.w = fmaf(v, x.w, acc.w);
}

// 16 bytes of row `r`, float4 column `c`, served by the XCD's L2 (sc1: never from this CU's L1, which may hold the line
// as it was before another CU of the XCD rewrote it in the previous layer)
__device__ __forceinline__ f32x4 lg_load_row(__amdgpu_buffer_rsrc_t rs, int r, int d, int c, int plain = 0) {
    const unsigned off = ((unsigned)r * (unsigned)d + (unsigned)c * 4u) * 4u;
    if (CRH_ABLATE(plain)) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, /*sc1*/ 16));
}

// n edges starting at s0 of the edge list, summed into acc in order: the (col, val) pairs are read G per lane group and
// broadcast by shuffles, LG_BATCH neighbour rows are in flight before the dependent fma chain.
template <int G>
__device__ __forceinline__ void lg_range_edges(const LgcnArgs& a, __amdgpu_buffer_rsrc_t rs, int64_t s0, int n, int c, int lig,
                                               f32x4& acc) {
    constexpr int SPB = LG_BATCH / G;
    for (int base = 0; base < n; base += LG_BATCH) {
        int mc[SPB];
        float mv[SPB];
#pragma unroll
        for (int s = 0; s < SPB; ++s) {
            const int t = base + s * G + lig;
            mc[s] = t < n ? a.col[s0 + t] : 0;
            mv[s] = t < n ? a.val[s0 + t] : 0.f;
        }
        f32x4 x[LG_BATCH];
        float vv[LG_BATCH];
#pragma unroll
        for (int q = 0; q < LG_BATCH; ++q) {
            const int cc = __shfl(mc[q / G], q % G, G);
            vv[q] = __shfl(mv[q / G], q % G, G);
            if (base + q < n) x[q] = lg_load_row(rs, cc, a.d, c, a.debug & 2);
        }
#pragma unroll
        for (int q = 0; q < LG_BATCH; ++q)
            if (base + q < n) lg_fma4(acc, vv[q], x[q]);
    }
}

// A light row: the head of its edge list (LG_HEAD entries) sits in LDS since the start of the launch (every layer walks
// the same list), the tail -- rows above 16 edges -- is read from the L2.
template <int G>
__device__ __forceinline__ void lg_row_edges(const LgcnArgs& a, __amdgpu_buffer_rsrc_t rs, int64_t e0, int n, const int* hc,
                                             const float* hv, int c, int lig, f32x4& acc) {
#pragma unroll
    for (int b = 0; b < LG_HEAD / LG_BATCH; ++b) {
        if (b * LG_BATCH >= n) break;
        const i32x4 c0 = *reinterpret_cast<const i32x4*>(hc + b * LG_BATCH), c1 = *reinterpret_cast<const i32x4*>(hc + b * LG_BATCH + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(hv + b * LG_BATCH), v1 = *reinterpret_cast<const f32x4*>(hv + b * LG_BATCH + 4);
        const int cc[LG_BATCH] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float vv[LG_BATCH] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        f32x4 x[LG_BATCH];
#pragma unroll
        for (int q = 0; q < LG_BATCH; ++q)
            if (b * LG_BATCH + q < n) x[q] = lg_load_row(rs, cc[q], a.d, c, a.debug & 2);
#pragma unroll
        for (int q = 0; q < LG_BATCH; ++q)
            if (b * LG_BATCH + q < n) lg_fma4(acc, vv[q], x[q]);
    }
    if (n > LG_HEAD) lg_range_edges<G>(a, rs, e0 + LG_HEAD, n - LG_HEAD, c, lig, acc);
}

// What a row does with P = (A X_{l-1})[row] in layer l (1-based).  `st` is the row's register state: the running layer sum
// (forward) or the row of dOUT (backward).  Returns through memory: X_l (l < L), OUT / the optimiser step (l == L).
__device__ __forceinline__ void lg_epilogue(const LgcnArgs& a, int l, int64_t o, const f32x4& P, f32x4& st, float* Xout) {
    if (!a.backward) {
        if (l < a.L) {
            *reinterpret_cast<f32x4*>(Xout + o) = P;
            st.x = st.x * 1.0f + P.x; st.y = st.y * 1.0f + P.y; st.z = st.z * 1.0f + P.z; st.w = st.w * 1.0f + P.w;
        } else {
            f32x4 r;
            r.x = (st.x * 1.0f + P.x) * a.c; r.y = (st.y * 1.0f + P.y) * a.c;
            r.z = (st.z * 1.0f + P.z) * a.c; r.w = (st.w * 1.0f + P.w) * a.c;
            *reinterpret_cast<f32x4*>(a.OUT + o) = r;
        }
        return;
    }
    // Horner: H1 = (z + A z) c ; H_{l} = z c + A H_{l-1}
    f32x4 H;
    if (l == 1) {
        H.x = (st.x * 1.0f + P.x) * a.c; H.y = (st.y * 1.0f + P.y) * a.c;
        H.z = (st.z * 1.0f + P.z) * a.c; H.w = (st.w * 1.0f + P.w) * a.c;
    } else {
        H.x = (st.x * a.c + P.x) * 1.0f; H.y = (st.y * a.c + P.y) * 1.0f;
        H.z = (st.z * a.c + P.z) * 1.0f; H.w = (st.w * a.c + P.w) * 1.0f;
    }
    if (l < a.L) {
        *reinterpret_cast<f32x4*>(Xout + o) = H;
        return;
    }
    if (a.G_out) *reinterpret_cast<f32x4*>(a.G_out + o) = H;
    f32x4 p = *reinterpret_cast<const f32x4*>(a.p + o);
    if (a.sgd) {
        sgd_elem4(p, H, a.nss);
        *reinterpret_cast<f32x4*>(a.p + o) = p;
    } else {
        f32x4 m = *reinterpret_cast<const f32x4*>(a.m + o), v = *reinterpret_cast<const f32x4*>(a.v + o);
        const float b2s = a.step_scalars ? a.step_scalars[0] : a.bc2_sqrt;
        const float nss = a.step_scalars ? a.step_scalars[1] : a.nss;
        adam_elem4(p, m, v, H, a.k, b2s, nss);
        *reinterpret_cast<f32x4*>(a.p + o) = p;
        *reinterpret_cast<f32x4*>(a.m + o) = m;
        *reinterpret_cast<f32x4*>(a.v + o) = v;
    }
    if (a.zero_dout) *reinterpret_cast<f32x4*>(a.dOUT + o) = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ void lg_flag(const LgcnArgs& a, unsigned code) {
    __hip_atomic_fetch_or(a.sync + LG_SLICES * LG_MAXL, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The workgroups of one slice (= one XCD) meet: everything they stored in this layer is in their L2.
__device__ __forceinline__ void lg_slice_barrier(const LgcnArgs& a, int slice, int l, unsigned n_wg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* cnt = a.sync + slice * LG_MAXL + l;
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_wg) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 4000000) { lg_flag(a, 2u); break; }     // ~seconds: give up loudly instead of hanging
        }
    }
    __syncthreads();
}

template <int G>
__global__ __launch_bounds__(LG_THREADS, 4) void lgcn_fused_kernel(LgcnArgs a) {
    constexpr int NGW = 64 / G;                 // lane groups per wave
    constexpr int NGB = LG_THREADS / G;         // per workgroup
    __shared__ f32x4 wsum[4][G];
    __shared__ __attribute__((aligned(16))) int head_c[NGB][LG_MAXR][LG_HEAD];     // heads of the light rows' edge lists
    __shared__ __attribute__((aligned(16))) float head_v[NGB][LG_MAXR][LG_HEAD];
    const int lig = threadIdx.x % G;
    const int slice = blockIdx.x & (LG_SLICES - 1);
    const int rank = blockIdx.x >> 3;                         // workgroup index inside the slice
    const int n_wg = gridDim.x >> 3;
    const int c = slice * G + lig;                            // this lane's float4 column
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // HW_REG_XCC_ID
        if (xcc != (unsigned)slice) lg_flag(a, 1u);
    }
    const int wave = threadIdx.x >> 6, gw = (threadIdx.x & 63) / G;
    // ---- static work: light rows by lane group (reversed rank: the groups that also carry a split row get the
    //      shortest light rows), split rows by wave / workgroup
    const int64_t NG = (int64_t)n_wg * NGB;
    const int64_t g_rank = NG - 1 - ((int64_t)rank * NGB + threadIdx.x / G);
    // per light row: its id, edge range and register-like state live in LDS (read and written by their own lane group
    // only), so the row loop is a real loop and the kernel stays at 4 workgroups per CU
    __shared__ f32x4 st_lds[LG_MAXR][LG_THREADS];
    __shared__ int row_lds[LG_MAXR][NGB], ne_lds[LG_MAXR][NGB];
    __shared__ int64_t e0_lds[LG_MAXR][NGB];
    const int gq = threadIdx.x / G;
#pragma unroll 1
    for (int k = 0; k < LG_MAXR; ++k) {
        const int64_t w = g_rank + (int64_t)k * NG;
        const int r = w < a.n_light ? a.light_rows[w] : -1;
        const int64_t e0 = r >= 0 ? a.rowptr[r] : 0;
        const int ne = r >= 0 ? (int)(a.rowptr[r + 1] - e0) : 0;
        for (int t = lig; t < LG_HEAD; t += G) {               // (only this lane group reads these LDS words: no barrier)
            head_c[gq][k][t] = t < ne ? a.col[e0 + t] : 0;
            head_v[gq][k][t] = t < ne ? a.val[e0 + t] : 0.f;
        }
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
        if (r >= 0) s0 = *reinterpret_cast<const f32x4*>(a.X0 + (int64_t)r * a.d + (int64_t)c * 4);
        st_lds[k][threadIdx.x] = s0;
        if (lig == 0) {
            row_lds[k][gq] = r;
            ne_lds[k][gq] = ne;
            e0_lds[k][gq] = e0;
        }
    }
    // one split row per wave and per workgroup at most (the launch function checks the list lengths)
    const int64_t w_rank = (int64_t)rank * 4 + wave;
    const int vrow = w_rank < a.n_wave ? a.wave_rows[w_rank] : -1;
    const int brow = rank < a.n_wg ? a.wg_rows[rank] : -1;
    f32x4 vst = {0.f, 0.f, 0.f, 0.f}, bst = vst;
    int64_t ve0 = 0, be0 = 0;
    int vn = 0, bn = 0;
    if (vrow >= 0) {
        ve0 = a.rowptr[vrow];
        vn = (int)(a.rowptr[vrow + 1] - ve0);
        if (gw == 0) vst = *reinterpret_cast<const f32x4*>(a.X0 + (int64_t)vrow * a.d + (int64_t)c * 4);
    }
    if (brow >= 0) {
        be0 = a.rowptr[brow];
        bn = (int)(a.rowptr[brow + 1] - be0);
        if (threadIdx.x < G) bst = *reinterpret_cast<const f32x4*>(a.X0 + (int64_t)brow * a.d + (int64_t)c * 4);
    }
    const unsigned xbytes = (unsigned)((size_t)a.n_rows * a.d * 4);

    for (int l = 1; l <= a.L; ++l) {
        const float* Xin = l == 1 ? a.X0 : ((l & 1) ? a.Xb : a.Xa);
        float* Xout = (l & 1) ? a.Xa : a.Xb;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xin), 0, xbytes, 0x00020000);
        // split rows first (they are the long poles), then the light rows
        if (brow >= 0 && !(CRH_ABLATE(a.debug) & 8)) {                 // whole workgroup: NGB lane groups cut the edge list
            int chunk = (bn + NGB - 1) / NGB;
            chunk = (chunk + LG_BATCH - 1) & ~(LG_BATCH - 1);
            const int b0 = gq * chunk, b1 = min(b0 + chunk, bn);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (b0 < bn) lg_range_edges<G>(a, rs, be0 + b0, b1 - b0, c, lig, acc);
#pragma unroll
            for (int off = G; off < 64; off <<= 1) {
                acc.x += __shfl_down(acc.x, off); acc.y += __shfl_down(acc.y, off);
                acc.z += __shfl_down(acc.z, off); acc.w += __shfl_down(acc.w, off);
            }
            __syncthreads();
            if ((threadIdx.x & 63) < G) wsum[wave][lig] = acc;
            __syncthreads();
            if (threadIdx.x < G) {
                const f32x4 t0 = wsum[0][lig], t1 = wsum[1][lig], t2 = wsum[2][lig], t3 = wsum[3][lig];
                f32x4 P;
                P.x = (t0.x + t1.x) + (t2.x + t3.x); P.y = (t0.y + t1.y) + (t2.y + t3.y);
                P.z = (t0.z + t1.z) + (t2.z + t3.z); P.w = (t0.w + t1.w) + (t2.w + t3.w);
                lg_epilogue(a, l, (int64_t)brow * a.d + (int64_t)c * 4, P, bst, Xout);
            }
        }
        if (vrow >= 0 && !(CRH_ABLATE(a.debug) & 8)) {                 // one wave: its NGW lane groups cut the edge list
            int chunk = (vn + NGW - 1) / NGW;
            chunk = (chunk + LG_BATCH - 1) & ~(LG_BATCH - 1);
            const int b0 = gw * chunk, b1 = min(b0 + chunk, vn);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (b0 < vn) lg_range_edges<G>(a, rs, ve0 + b0, b1 - b0, c, lig, acc);
#pragma unroll
            for (int off = G; off < 64; off <<= 1) {
                acc.x += __shfl_down(acc.x, off); acc.y += __shfl_down(acc.y, off);
                acc.z += __shfl_down(acc.z, off); acc.w += __shfl_down(acc.w, off);
            }
            if (gw == 0) lg_epilogue(a, l, (int64_t)vrow * a.d + (int64_t)c * 4, acc, vst, Xout);
        }
#pragma unroll 1
        for (int k = 0; k < LG_MAXR; ++k) {
            const int r = row_lds[k][gq];
            if (r < 0 || (CRH_ABLATE(a.debug) & 4)) continue;
            f32x4 P = {0.f, 0.f, 0.f, 0.f};
            lg_row_edges<G>(a, rs, e0_lds[k][gq], ne_lds[k][gq], head_c[gq][k], head_v[gq][k], c, lig, P);
            f32x4 stv = st_lds[k][threadIdx.x];
            lg_epilogue(a, l, (int64_t)r * a.d + (int64_t)c * 4, P, stv, Xout);
            st_lds[k][threadIdx.x] = stv;
        }
        if (l < a.L && !(CRH_ABLATE(a.debug) & 1)) lg_slice_barrier(a, slice, l, (unsigned)n_wg);
    }
}

template <int G>
int lg_launch(const LgcnArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(lgcn_fused_kernel<G>, dim3(256 * 4), dim3(LG_THREADS), 0, st, a);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

int lg_run(const char* who, LgcnArgs a, void* stream) {
    CRH_CHECK_ARG(a.rowptr && a.col && a.val && a.X0 && a.Xa && a.Xb && a.sync, "%s: NULL pointer", who);
    CRH_CHECK_ARG(a.L >= 1 && a.L <= LG_MAXL, "%s: n_layers=%d outside 1..%d", who, a.L, LG_MAXL);
    CRH_CHECK_ARG(a.d == 128 || a.d == 256, "%s: d=%d (the fused propagation is built for 128 and 256)", who, a.d);
    CRH_CHECK_ARG(a.n_rows > 0 && (size_t)a.n_rows * a.d * 4 < ((size_t)1 << 32), "%s: table too large for 32-bit offsets", who);
    const int G = a.d / 32;
    const int64_t n_wg = 128;                                     // workgroups per slice (grid 1024)
    CRH_CHECK_ARG(a.n_light >= 0 && a.n_light <= LG_MAXR * n_wg * (LG_THREADS / G),
                  "%s: %d light rows exceed what one resident launch holds in registers (%lld)", who, a.n_light,
                  (long long)(LG_MAXR * n_wg * (LG_THREADS / G)));
    CRH_CHECK_ARG(a.n_wave >= 0 && a.n_wave <= n_wg * 4 && a.n_wg >= 0 && a.n_wg <= n_wg,
                  "%s: too many split rows (%d per wave, %d per workgroup)", who, a.n_wave, a.n_wg);
    CRH_CHECK_ARG((a.n_light == 0 || a.light_rows) && (a.n_wave == 0 || a.wave_rows) && (a.n_wg == 0 || a.wg_rows),
                  "%s: NULL row list", who);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const int debug = CRH_PROFILE_ENV("CRH_LGCN_DEBUG");
    a.debug = debug;
    CRH_CHECK_ARG(!(a.backward && a.zero_dout && a.L < 2), "%s: dOUT can only be cleared in the epilogue when L >= 2 "
                  "(with one layer it is the gathered operand)", who);
    CRH_HIP(hipMemsetAsync(a.sync, 0, LG_SLICES * LG_MAXL * sizeof(unsigned), st));   // (the status word stays: sticky)
    return G == 4 ? lg_launch<4>(a, st) : lg_launch<8>(a, st);
}

}  // namespace

extern "C" size_t crh_lgcn_sync_bytes(void) { return (LG_SLICES * LG_MAXL + 1) * sizeof(unsigned); }
extern "C" int crh_lgcn_light_edges(void) { return LG_LIGHT_MAX; }
// 1 when the fused propagation can hold this graph (d in {128, 256}; row classes: light <= 64 edges, wave-split <=
// 64 x (2048 / d) edges, workgroup-split above) in one resident launch
extern "C" int crh_lgcn_fused_supported(int64_t n_rows, int d, int64_t n_light, int64_t n_wave, int64_t n_wg) {
    if (!(d == 128 || d == 256) || n_rows <= 0 || (size_t)n_rows * d * 4 >= ((size_t)1 << 32)) return 0;
    const int G = d / 32;
    return n_light <= (int64_t)LG_MAXR * 128 * (LG_THREADS / G) && n_wave <= 512 && n_wg <= 128;
}

extern "C" int crh_lgcn_propagate_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, int d,
                                      int n_layers, const int32_t* light_rows, int n_light, const int32_t* wave_rows,
                                      int n_wave, const int32_t* wg_rows, int n_wg, const float* e0, float* xa, float* xb,
                                      float* out, void* sync, void* stream) {
    CRH_CHECK_ARG(out, "crh_lgcn_propagate_f32: NULL output");
    LgcnArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val; a.n_rows = n_rows; a.d = d; a.L = n_layers;
    a.light_rows = light_rows; a.n_light = n_light; a.wave_rows = wave_rows; a.n_wave = n_wave; a.wg_rows = wg_rows; a.n_wg = n_wg;
    a.X0 = e0; a.Xa = xa; a.Xb = xb; a.backward = 0; a.OUT = out; a.c = 1.0f / (float)(n_layers + 1);
    a.sync = reinterpret_cast<unsigned*>(sync);
    return lg_run("crh_lgcn_propagate_f32", a, stream);
}

extern "C" int crh_lgcn_backprop_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, int d,
                                     int n_layers, const int32_t* light_rows, int n_light, const int32_t* wave_rows,
                                     int n_wave, const int32_t* wg_rows, int n_wg, float* dout, float* xa, float* xb,
                                     float* grad_out, float* p, float* m, float* v, int sgd, double lr, double beta1,
                                     double beta2, double eps, int64_t step, const float* step_scalars, int zero_dout,
                                     void* sync, void* stream) {
    CRH_CHECK_ARG(dout && p && (sgd || (m && v)), "crh_lgcn_backprop_f32: NULL gradient / optimiser state");
    CRH_CHECK_ARG(sgd || step >= 1 || step_scalars, "crh_lgcn_backprop_f32: step starts at 1");
    if (step < 1) step = 1;
    LgcnArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val; a.n_rows = n_rows; a.d = d; a.L = n_layers;
    a.light_rows = light_rows; a.n_light = n_light; a.wave_rows = wave_rows; a.n_wave = n_wave; a.wg_rows = wg_rows; a.n_wg = n_wg;
    a.X0 = dout; a.Xa = xa; a.Xb = xb; a.backward = 1; a.c = 1.0f / (float)(n_layers + 1);
    a.dOUT = dout; a.zero_dout = zero_dout; a.G_out = grad_out; a.p = p; a.m = m; a.v = v; a.sgd = sgd;
    a.k = AdamK{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps};
    if (sgd) {
        a.nss = (float)(-lr);
    } else {
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        a.bc2_sqrt = (float)sqrt(bc2);
        a.nss = (float)(-(lr / bc1));
    }
    a.step_scalars = sgd ? nullptr : step_scalars;
    a.sync = reinterpret_cast<unsigned*>(sync);
    return lg_run("crh_lgcn_backprop_f32", a, stream);
}
