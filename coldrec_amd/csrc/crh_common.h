// Shared helpers for the gfx950 kernels behind include/coldrec_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/coldrec_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
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
