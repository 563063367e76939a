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
// caller may pass a SCHEDULE built once per graph: rows are cut into segments of <= 64 edges, every
// segment is one work item, and the few rows with several segments are finished by a small combine
// kernel that adds their partials in segment order (still deterministic, association differs).
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
    const float* acc_in;
    float* acc_out;
    float s_in, s_out;
    crh_spmm_sched sched;   // n_seg == 0: one lane group per row
    float* partial;         // [n_partial][d] partial sums of the rows cut into several segments
};

__device__ __forceinline__ void fma4(f32x4& acc, float v, const f32x4& x) {
    acc.x = fmaf(v, x.x, acc.x);
    acc.y = fmaf(v, x.y, acc.y);
    acc.z = fmaf(v, x.z, acc.z);
    acc.w = fmaf(v, x.w, acc.w);
}

__device__ __forceinline__ void store_row(const SpmmArgs& a, int64_t o, const f32x4& acc) {
    if (a.Y) *reinterpret_cast<f32x4*>(a.Y + o) = acc;
    if (a.acc_out) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (a.acc_in) z = *reinterpret_cast<const f32x4*>(a.acc_in + o);
        f32x4 r;
        r.x = (z.x * a.s_in + acc.x) * a.s_out;
        r.y = (z.y * a.s_in + acc.y) * a.s_out;
        r.z = (z.z * a.s_in + acc.z) * a.s_out;
        r.w = (z.w * a.s_in + acc.w) * a.s_out;
        *reinterpret_cast<f32x4*>(a.acc_out + o) = r;
    }
}

// Accumulate edges [e0, e1) of one row into acc for this lane's 16-B column slice, in edge order.
// The edge list is read G entries at a time (one per lane) and broadcast by shuffles; neighbour rows
// are fetched 4 at a time before the dependent fma chain so several loads are in flight per group.
template <int G>
__device__ __forceinline__ void row_edges(const SpmmArgs& a, int64_t e0, int64_t e1, int c, bool on, int lig,
                                          f32x4& acc) {
    for (int64_t base = e0; base < e1; base += G) {
        const int64_t e = base + lig;
        const int my_col = e < e1 ? a.col[e] : 0;
        const float my_val = e < e1 ? a.val[e] : 0.f;
        const int cnt = (int)((e1 - base) < G ? (e1 - base) : G);
        int t = 0;
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

// Work item = one row (no schedule) or one segment of <= crh_spmm_segment_edges() edges of a row.
// A row in one piece is finished here (bit-identical to the oracle's edge-order chain); a row cut
// into several segments leaves one partial per segment and is finished by spmm_combine_kernel.
template <int G>
__global__ __launch_bounds__(256) void spmm_csr_kernel(SpmmArgs a) {
    const int lig = threadIdx.x % G;
    const int64_t w0 = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    const int64_t wstride = (int64_t)gridDim.x * (256 / G);
    const int nvec = a.d >> 2;
    const bool seg = a.sched.n_seg > 0;
    const int64_t n_work = seg ? a.sched.n_seg : a.n_rows;
    for (int64_t w = w0; w < n_work; w += wstride) {
        const int64_t row = seg ? a.sched.seg_row[w] : w;
        const int64_t e0 = seg ? a.sched.seg_ptr[w] : a.rowptr[row];
        const int64_t e1 = seg ? a.sched.seg_ptr[w + 1] : a.rowptr[row + 1];
        const int slot = seg ? a.sched.seg_slot[w] : -1;
        for (int c0 = 0; c0 < nvec; c0 += G) {                  // one pass when d <= 4*G
            const int c = c0 + lig;
            const bool on = c < nvec;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            row_edges<G>(a, e0, e1, c, on, lig, acc);
            if (on) {
                if (slot < 0) store_row(a, row * a.d + (int64_t)c * 4, acc);
                else *reinterpret_cast<f32x4*>(a.partial + (int64_t)slot * a.d + (int64_t)c * 4) = acc;
            }
        }
    }
}

// Rows cut into several segments: sum the partials in segment order (deterministic), then epilogue.
__global__ __launch_bounds__(256) void spmm_combine_kernel(SpmmArgs a) {
    const int nvec = a.d >> 2;
    const int64_t total = (int64_t)a.sched.n_multi * nvec;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nvec), c = (int)(i - (int64_t)m * nvec);
        const int first = a.sched.multi_first[m], cnt = a.sched.multi_count[m];
        f32x4 acc = *reinterpret_cast<const f32x4*>(a.partial + (int64_t)first * a.d + c * 4);
        for (int q = 1; q < cnt; ++q) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(a.partial + (int64_t)(first + q) * a.d + c * 4);
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        store_row(a, (int64_t)a.sched.multi_row[m] * a.d + (int64_t)c * 4, acc);
    }
}

template <int G>
int launch_spmm(const SpmmArgs& a, hipStream_t st) {
    const int64_t per_block = 256 / G;
    const int64_t n_work = a.sched.n_seg > 0 ? a.sched.n_seg : a.n_rows;
    int64_t blocks = (n_work + per_block - 1) / per_block;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    hipLaunchKernelGGL(spmm_csr_kernel<G>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    CRH_HIP(hipGetLastError());
    if (a.sched.n_seg > 0 && a.sched.n_multi > 0) {
        int64_t cb = ((int64_t)a.sched.n_multi * (a.d / 4) + 255) / 256;
        if (cb > 4096) cb = 4096;
        hipLaunchKernelGGL(spmm_combine_kernel, dim3((unsigned)cb), dim3(256), 0, st, a);
        CRH_HIP(hipGetLastError());
    }
    return CRH_OK;
}

}  // namespace

extern "C" int crh_spmm_segment_edges(void) { return 64; }

extern "C" size_t crh_spmm_workspace_bytes(const crh_spmm_sched* sched, int d) {
    return sched && sched->n_seg > 0 ? (size_t)sched->n_partial * (size_t)d * 4 : 0;
}

extern "C" int crh_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                const float* x, int d, float* y, const float* acc_in, float s_in,
                                float* acc_out, float s_out, const crh_spmm_sched* sched, void* workspace,
                                size_t workspace_bytes, void* stream) {
    CRH_CHECK_ARG(rowptr && x && n_rows > 0, "crh_spmm_csr_f32: NULL pointer / empty matrix");
    CRH_CHECK_ARG(d >= 4 && d % 4 == 0, "crh_spmm_csr_f32: d=%d must be a positive multiple of 4", d);
    CRH_CHECK_ARG(y || acc_out, "crh_spmm_csr_f32: nothing to write (y and acc_out both NULL)");
    CRH_CHECK_ARG(y != x && acc_out != x, "crh_spmm_csr_f32: outputs must not alias x");
    CRH_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)acc_in | (uintptr_t)acc_out | (uintptr_t)workspace) & 15) == 0,
                  "crh_spmm_csr_f32: dense operands must be 16-byte aligned");
    SpmmArgs a{rowptr, col, val, n_rows, x, d, y, acc_in, acc_out, s_in, s_out, {}, nullptr};
    if (sched && sched->n_seg > 0) {
        CRH_CHECK_ARG(sched->seg_row && sched->seg_ptr && sched->seg_slot, "crh_spmm_csr_f32: incomplete schedule");
        CRH_CHECK_ARG(sched->n_multi == 0 || (sched->multi_row && sched->multi_first && sched->multi_count),
                      "crh_spmm_csr_f32: incomplete schedule (multi-segment rows)");
        const size_t need = crh_spmm_workspace_bytes(sched, d);
        if (need && (!workspace || workspace_bytes < need)) {
            crh_set_error("crh_spmm_csr_f32: workspace %zu < %zu bytes", workspace_bytes, need);
            return CRH_ERR_WS;
        }
        a.sched = *sched;
        a.partial = reinterpret_cast<float*>(workspace);
    } else {
        a.sched.n_seg = 0;
        a.sched.n_multi = 0;
    }
    int G = 1;
    while (G < d / 4 && G < 64) G <<= 1;
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
