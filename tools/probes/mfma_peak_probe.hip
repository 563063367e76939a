// What does this chip sustain on a bare v_mfma_f32_32x32x16_f16 stream with random operands?  8 waves per CU
// (two per SIMD) x 256 CUs, two accumulators alternating per wave (the shape of the scoring kernel's inner loop),
// nothing else.  Prints TFLOP/s and the effective shader clock (s_memtime ticks / wall time).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe mfma_peak_probe.hip && ./mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NB>
__global__ __launch_bounds__(512, 1) void k(const f16x8* in, float* out, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[NB][2];
    for (int i = 0; i < 4; ++i) a[i] = in[(blockIdx.x * 7 + i) * 64 % 4096 + lane];
    for (int i = 0; i < NB; ++i)
        for (int u = 0; u < 2; ++u) b[i][u] = in[((threadIdx.x >> 6) * 31 + i * 2 + u) * 64 % 4096 + lane];
    f32x16 acc0 = {}, acc1 = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i][1], acc1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int blocks = 256;
    std::vector<_Float16> h(4096 * 64 * 8);
    srand(1);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
    f16x8* din;
    float* dout;
    unsigned long long* dclk;
    hipMalloc(&din, h.size() * 2);
    hipMalloc(&dout, blocks * 512 * 4);
    hipMalloc(&dclk, blocks * 8);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        if (zero) hipMemset(din, 0, h.size() * 2);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, din, dout, iters, dclk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c;
            hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost);
            const double flop = 2.0 * 32 * 32 * 16 * 32.0 * iters * 8 * blocks;
            printf("%s operands: %.2f ms  %.0f TFLOP/s (%.3f of 2.5 PF)  s_memtime %.3f GHz-equivalent, %.1f cycles per MFMA per SIMD\n",
                   zero ? "zero  " : "random", ms, flop / ms / 1e9, flop / ms / 1e9 / 2500, c / (ms * 1e6),
                   (double)c / (32.0 * iters * 2));
        }
    }
    return 0;
}
