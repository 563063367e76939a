// l2_reg_loss of the reference (util/utils.py:44-48):  reg * sum_e |e|_F / rows(e)  over any number of embedding
// tensors (model files pass 2..6 of them, of different row counts), forward and backward, for the autograd Function
// in coldrec_amd/util/utils.py.  HBM-bound streaming kernels; the Frobenius norm is reduced deterministically
// (per-block partials in the workspace, combined by one block in a fixed order).
#include <math.h>

#include "crh_common.h"

namespace {

constexpr int L2_MAX_BLOCKS = 1024;

__device__ __forceinline__ float l2_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void l2_sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += stride) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) s += x[i] * x[i];
    } else {
        for (int64_t i = tid; i < n; i += stride) s += x[i] * x[i];
    }
    s = l2_block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void l2_finish_kernel(const float* __restrict__ partials, int nb, float* __restrict__ norm_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partials[i];
    s = l2_block_sum(s, red);
    if (threadIdx.x == 0) norm_out[0] = sqrtf(s);
}

// gx (+)= x * reg * gout / (rows * |x|_F)     (0 where |x|_F == 0, as autograd's norm backward returns)
__global__ __launch_bounds__(256) void l2_bwd_kernel(const float* __restrict__ x, int64_t n, float coef, const float* __restrict__ norm,
                                                     const float* __restrict__ gout, float* __restrict__ gx, int accumulate) {
    const float nm = norm[0];
    const float c = nm > 0.f ? coef * (gout ? gout[0] : 1.0f) / nm : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        gx[i] = accumulate ? gx[i] + c * x[i] : c * x[i];
}

}  // namespace

extern "C" size_t crh_l2_workspace_bytes(void) { return (size_t)L2_MAX_BLOCKS * sizeof(float); }

// norm_out[0] = |x|_F of the n contiguous fp32 values at x (device scalar).
extern "C" int crh_l2_norm_f32(const float* x, int64_t n, float* norm_out, void* workspace, size_t workspace_bytes,
                               void* stream) {
    CRH_CHECK_ARG(x && norm_out && n > 0, "crh_l2_norm_f32: NULL pointer / empty tensor");
    if (!workspace || workspace_bytes < crh_l2_workspace_bytes()) {
        crh_set_error("crh_l2_norm_f32: workspace %zu < %zu bytes", workspace_bytes, crh_l2_workspace_bytes());
        return CRH_ERR_WS;
    }
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > L2_MAX_BLOCKS) blocks = L2_MAX_BLOCKS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* partials = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(l2_sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, partials);
    hipLaunchKernelGGL(l2_finish_kernel, dim3(1), dim3(256), 0, st, partials, (int)blocks, norm_out);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// Backward of reg * |x|_F / rows w.r.t. x: gx (+)= x * reg * grad_out / (rows * norm); norm = crh_l2_norm_f32's output,
// grad_out a device scalar (NULL = 1).
extern "C" int crh_l2_reg_bwd_f32(const float* x, int64_t n, int64_t rows, float reg, const float* norm,
                                  const float* grad_out, float* gx, int accumulate, void* stream) {
    CRH_CHECK_ARG(x && norm && gx && n > 0 && rows > 0, "crh_l2_reg_bwd_f32: NULL pointer / empty tensor");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(l2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n,
                       reg / (float)rows, norm, grad_out, gx, accumulate);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
