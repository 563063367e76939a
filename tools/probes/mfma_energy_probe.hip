// Which ingredient of the scoring loop costs matrix-pipe throughput on a power-limited chip?  Variants of a bare
// v_mfma_f32_32x32x16_f16 stream (random operands, 256 CUs):
//   NW   waves per CU (8 = two per SIMD, 4 = one per SIMD)
//   NACC accumulators per wave = MFMAs fed by one A fragment (2 = 64 users per wave, 4 = 128)
//   LDSR 1: the A fragment of every MFMA group comes from LDS (ds_read_b128, 1 KiB per wave), 0: from registers
// Prints TFLOP/s and the effective clock per variant, interleaved so DVFS drift hits all alike.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_energy_probe mfma_energy_probe.hip && ./mfma_energy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NW, int NACC, int LDSR, int NB>
__global__ __launch_bounds__(64 * NW, 1) void k(const f16x8* in, float* out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 16 chunks x 1 KiB of A fragments
    const int lane = threadIdx.x & 63;
    f16x8 b[NB][NACC];
    for (int i = 0; i < NB; ++i)
        for (int u = 0; u < NACC; ++u) b[i][u] = in[((threadIdx.x >> 6) * 31 + i * NACC + u) * 64 % 4096 + lane];
    for (int i = threadIdx.x; i < 16 * 64; i += 64 * NW) reinterpret_cast<f16x8*>(smem)[i] = in[(blockIdx.x * 7 * 64 + i) % (4096 * 64)];
#ifdef PIN_B_AGPR    // the DMA scoring kernel's register split: B fragments in AGPRs, accumulators in VGPRs (build with
    for (int i = 0; i < NB; ++i)   // -mllvm -amdgpu-mfma-vgpr-form -DPIN_B_AGPR)
        for (int u = 0; u < NACC; ++u) asm volatile("" : "+a"(b[i][u]));
#endif
    __syncthreads();
    f16x8 areg[4];
    for (int i = 0; i < 4; ++i) areg[i] = reinterpret_cast<f16x8*>(smem)[i * 64 + lane];
    f32x16 acc[NACC];
    for (int u = 0; u < NACC; ++u) acc[u] = f32x16{};
    const unsigned long long t0 = __builtin_readcyclecounter();
    const f16x8* sm = reinterpret_cast<const f16x8*>(smem);
    int off = lane;   // opaque to the compiler each iteration: the reads stay inside the loop
    f16x8 a_cur = sm[off];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            f16x8 a_nxt;
            asm volatile("" : "+v"(off));
            if (LDSR) {   // fragment of the NEXT group, in flight under this group's MFMAs (as the scoring loop does)
                a_nxt = sm[((i + 1) % NB) * 64 + off];
            } else {
                a_nxt = areg[(i + 1) & 3];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur, b[i][u], acc[u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int u = 0; u < NACC; ++u)
        for (int r = 0; r < 16; ++r) s += acc[u][r];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NW, int NACC, int LDSR, int NB = 16>
void run(const f16x8* din, float* dout, unsigned long long* dclk, int iters, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int it = iters * 2 / NACC * 8 / NW * 16 / NB;      // same MFMA count per CU for every variant
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NW, NACC, LDSR, NB>), dim3(256), dim3(64 * NW), 16 * 1024, 0, din, dout, it, dclk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost);
    const double flop = 2.0 * 32 * 32 * 16 * (double)NB * NACC * it * NW * 256;
    printf("%-34s %7.2f ms  %5.0f TFLOP/s (%.3f of 2.5 PF)  clock %.3f GHz  %.1f cycles per MFMA of one wave\n", name, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 2500, c / (ms * 1e6), (double)c / ((double)NB * NACC * it));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<_Float16> h(4096 * 64 * 8);
    srand(1);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
    f16x8* din;
    float* dout;
    unsigned long long* dclk;
    hipMalloc(&din, h.size() * 2);
    hipMalloc(&dout, 256 * 512 * 4);
    hipMalloc(&dclk, 256 * 8);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        run<8, 2, 0>(din, dout, dclk, iters, "8 waves, 2 acc, A in registers");
        run<8, 2, 1>(din, dout, dclk, iters, "8 waves, 2 acc, A from LDS");
        run<4, 4, 0>(din, dout, dclk, iters, "4 waves, 4 acc, A in registers");
        run<4, 4, 1>(din, dout, dclk, iters, "4 waves, 4 acc, A from LDS");
        run<4, 2, 1>(din, dout, dclk, iters, "4 waves, 2 acc, A from LDS");
        run<8, 4, 1, 8>(din, dout, dclk, iters, "8 waves, 4 acc, A from LDS (8 chunks)");
        run<8, 2, 1, 8>(din, dout, dclk, iters, "8 waves, 2 acc, A from LDS (8 chunks)");
    }
    return 0;
}
