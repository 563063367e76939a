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
};

template <int G>
__global__ __launch_bounds__(256) void spmm_csr_kernel(SpmmArgs a) {
    const int lig = threadIdx.x % G;
    const int64_t row0 = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    const int64_t rstride = (int64_t)gridDim.x * (256 / G);
    const int nvec = a.d >> 2;
    for (int64_t row = row0; row < a.n_rows; row += rstride) {
        const int64_t e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
        for (int c0 = 0; c0 < nvec; c0 += G) {          // one pass when d <= 4*G
            const int c = c0 + lig;
            const bool on = c < nvec;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int64_t base = e0; base < e1; base += G) {
                const int64_t e = base + lig;
                const int my_col = e < e1 ? a.col[e] : 0;
                const float my_val = e < e1 ? a.val[e] : 0.f;
                const int cnt = (int)((e1 - base) < G ? (e1 - base) : G);
                for (int t = 0; t < cnt; ++t) {
                    const int cc = __shfl(my_col, t, G);
                    const float vv = __shfl(my_val, t, G);
                    if (on) {
                        const f32x4 x = reinterpret_cast<const f32x4*>(a.X + (int64_t)cc * a.d)[c];
                        acc.x = fmaf(vv, x.x, acc.x);
                        acc.y = fmaf(vv, x.y, acc.y);
                        acc.z = fmaf(vv, x.z, acc.z);
                        acc.w = fmaf(vv, x.w, acc.w);
                    }
                }
            }
            if (on) {
                const int64_t o = row * a.d + (int64_t)c * 4;
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
        }
    }
}

}  // namespace

extern "C" int crh_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                const float* x, int d, float* y, const float* acc_in, float s_in,
                                float* acc_out, float s_out, void* stream) {
    CRH_CHECK_ARG(rowptr && x && n_rows > 0, "crh_spmm_csr_f32: NULL pointer / empty matrix");
    CRH_CHECK_ARG(d >= 4 && d % 4 == 0, "crh_spmm_csr_f32: d=%d must be a positive multiple of 4", d);
    CRH_CHECK_ARG(y || acc_out, "crh_spmm_csr_f32: nothing to write (y and acc_out both NULL)");
    CRH_CHECK_ARG(y != x && acc_out != x, "crh_spmm_csr_f32: outputs must not alias x");
    CRH_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)acc_in | (uintptr_t)acc_out) & 15) == 0,
                  "crh_spmm_csr_f32: dense operands must be 16-byte aligned");
    SpmmArgs a{rowptr, col, val, n_rows, x, d, y, acc_in, acc_out, s_in, s_out};
    int G = 1;
    while (G < d / 4 && G < 64) G <<= 1;
    const int64_t rows_per_block = 256 / G;
    int64_t blocks = (n_rows + rows_per_block - 1) / rows_per_block;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (G) {
        case 1: hipLaunchKernelGGL(spmm_csr_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(spmm_csr_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(spmm_csr_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL(spmm_csr_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 16: hipLaunchKernelGGL(spmm_csr_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case 32: hipLaunchKernelGGL(spmm_csr_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(spmm_csr_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
    }
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
