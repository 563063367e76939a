// Shared helpers for the gfx950 kernels behind include/coldrec_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/coldrec_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

void crh_set_error(const char* fmt, ...);

#define CRH_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            crh_set_error(__VA_ARGS__);   \
            return CRH_ERR_ARG;           \
        }                                 \
    } while (0)

#define CRH_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            crh_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                          __LINE__);                                                      \
            return CRH_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

// canonical ranking key: score descending, global item index ascending
__device__ __forceinline__ bool crh_better(float sa, int ia, float sb, int ib) {
    return sa > sb || (sa == sb && ia < ib);
}

#define CRH_NEG_INF (-__builtin_inff())

// Measurement hooks (parts of a kernel switched off, per-wave clocks) exist only in the -DCRH_PROFILE build that
// tools/profile_*.sh load (make profile -> lib/libcoldrec_hip_profile.so); in the shipped library they fold to 0.
// CRH_TUNE_ENV: tuning switches (thresholds, alternative schedules) that only the measurement build reads from the environment;
// the shipped library compiles their defaults in (the few switches a user or a test can want stay plain getenv: CRH_SCORE_WG,
// CRH_SCORE_DMA, CRH_SCORE_SEED, CRH_SCORE_SEED_MAX_ITEMS, CRH_SCORE_DENSE_BLOCK_MB, CRH_SPMM_SEG, CRH_SPMM_SLAB).
#ifdef CRH_PROFILE
#define CRH_ABLATE(x) (x)
#define CRH_PROFILE_ENV(name) (getenv(name) ? atoi(getenv(name)) : 0)
#define CRH_TUNE_ENV(name) getenv(name)
#else
#define CRH_ABLATE(x) 0
#define CRH_PROFILE_ENV(name) 0
static inline const char* crh_no_env(const char*) { return nullptr; }
#define CRH_TUNE_ENV(name) crh_no_env(name)
#endif

// One Adam step of one element, op for op what torch/optim/adam.py _single_tensor_adam does.  Shared by the
// dense kernel, the touched-rows replay, the one-launch MF step and the SpMM epilogue so that all produce the same bits.
struct AdamK {
    float one_minus_b1, b2, one_minus_b2, eps;
};
__device__ __forceinline__ void adam_elem4(f32x4& p, f32x4& m, f32x4& v, const f32x4& g, const AdamK& k,
                                           float bc2_sqrt, float neg_step_size) {
    // every product and sum rounded on its own, as the separate ATen ops of the reference do -- and so that
    // the compiler cannot fuse differently in the two kernels that share this function
#pragma clang fp contract(off)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float mc = m[c] + k.one_minus_b1 * (g[c] - m[c]);        // exp_avg.lerp_(grad, 1-b1)
        const float vc = v[c] * k.b2 + (k.one_minus_b2 * g[c]) * g[c];  // mul_(b2).addcmul_(g, g, 1-b2)
        const float denom = sqrtf(vc) / bc2_sqrt + k.eps;               // sqrt()/bc2_sqrt + eps
        p[c] = p[c] + neg_step_size * (mc / denom);                     // addcdiv_(m, denom, -step_size)
        m[c] = mc;
        v[c] = vc;
    }
}

// One plain-SGD step of four elements (torch.optim.SGD defaults): p <- fma(-lr, g, p).  Shared by the dense kernel,
// the touched-rows kernel, the one-launch MF step and the SpMM epilogue.
__device__ __forceinline__ void sgd_elem4(f32x4& p, const f32x4& g, float neg_lr) {
    p.x = fmaf(neg_lr, g.x, p.x); p.y = fmaf(neg_lr, g.y, p.y);
    p.z = fmaf(neg_lr, g.z, p.z); p.w = fmaf(neg_lr, g.w, p.w);
}
