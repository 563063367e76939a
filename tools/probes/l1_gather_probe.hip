// Round-3 probe: what the vector memory pipeline of a CU sustains for 16-B-per-lane loads from a table that is L1/L2
// resident, by address pattern: (a) 1 KB contiguous per wave-instruction, (b) 8 random 128-B lines per instruction (the
// SpMM's gathers: 8 lane groups x 128 B), (c) 2 random 512-B rows, (d) 16 random 64-B lines.  Also the shader clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// table: n_lines x 128 B.  lanes_per_line: 8 (128 B), 32 (512-B rows = 4 lines), 4 (64 B), 64 (1 KB contiguous)
template <int LPL>
__global__ __launch_bounds__(256) void k(const char* __restrict__ T, unsigned n_units, int iters, float* out, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPL, lig = lane % LPL;
    unsigned s = (blockIdx.x * 256 + threadIdx.x / LPL * 977u + 12345u) * 2654435761u;
    f32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s = s * 1664525u + 1013904223u;
            const unsigned unit = (s >> 8) % n_units;                       // one random unit per lane group
            x[q] = *reinterpret_cast<const f32x4*>(T + (size_t)unit * (LPL * 16) + lig * 16);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc.x += x[q].x; acc.y += x[q].y; acc.z += x[q].z; acc.w += x[q].w; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (acc.x == 12345.f) out[0] = acc.y + grp;
    if (threadIdx.x == 0 && clk) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    char* T; float* out; unsigned long long* clk;
    const size_t cap = 64u << 20;
    CK(hipMalloc(&T, cap)); CK(hipMemset(T, 0, cap)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 1 << 20));
    const int iters = 200;
    for (size_t table : {(size_t)8 << 10, (size_t)2 << 20, (size_t)11 << 20}) {
        for (int wgs_per_cu : {1, 4, 6}) {
            const int grid = 256 * wgs_per_cu;
            auto run = [&](const char* name, int lpl) {
                const unsigned n_units = (unsigned)(table / (lpl * 16));
                auto launch = [&]() {
                    switch (lpl) {
                        case 64: hipLaunchKernelGGL(k<64>, dim3(grid), dim3(256), 0, 0, T, n_units, iters, out, clk); break;
                        case 32: hipLaunchKernelGGL(k<32>, dim3(grid), dim3(256), 0, 0, T, n_units, iters, out, clk); break;
                        case 8: hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, T, n_units, iters, out, clk); break;
                        default: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, T, n_units, iters, out, clk); break;
                    }
                };
                launch(); launch(); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
                const double bytes = (double)grid * 4 * iters * 8 * 1024;
                const double ghz = (double)h[0] / ((double)h[1] * 10.0);      // cycles per ns (s_memtime vs 100 MHz realtime)
                printf("table %6zu KB  %d WG/CU  %-34s %8.1f us  %6.2f TB/s  %5.1f B/clk/CU at %.2f GHz (cycle counter / realtime)\n",
                       table >> 10, wgs_per_cu, name, ms * 1e3, bytes / ms / 1e9, bytes / 256 / ((double)h[0]), ghz);
            };
            run("1 KB contiguous per instruction", 64);
            run("2 random 512-B rows", 32);
            run("8 random 128-B lines (SpMM gathers)", 8);
            run("16 random 64-B half lines", 4);
        }
    }
    return 0;
}
