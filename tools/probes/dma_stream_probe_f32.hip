// fp32 edition of dma_stream_probe.hip (round 6): where do the cycles of the LDS-DMA scoring kernel go at fp32 d=64 / d=128?
// The kernel's MFMA stream (4 waves x 128 users, NCH = 8 / 16 chunks per 32-item tile, v_mfma_f32_32x32x2_f32 in k order, every
// B component pinned in its own AGPR, accumulators in VGPRs) rebuilt bare, with the kernel's other ingredients switched on one
// by one (FEAT bits):
//   1  item tiles streamed from global memory into a 4-slot LDS ring by global_load_lds + one s_barrier per tile
//      (else: one static tile in LDS, read again and again)
//   2  the threshold test's VALU work (4 x max-of-16 over the OTHER accumulator set) between the MFMAs of group 1
//   4  two accumulator sets, zero-initialised by the first MFMA of a tile (else one set, accumulating for ever)
//   8  A fragments by inline-asm ds_read_b128 two groups ahead with counted lgkmcnt (else compiler loads, one group ahead)
//  16  the tile-bits DMA piece (256 B per tile and wave)
// hipcc --offload-arch=gfx950 -O3 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form -o dma_stream_probe_f32 dma_stream_probe_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <type_traits>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float max16_chain(const f32x16& v) {      // two-operand maxima only, depth 4 tree
    const float a0 = fmaxf(v[0], v[1]), a1 = fmaxf(v[2], v[3]), a2 = fmaxf(v[4], v[5]), a3 = fmaxf(v[6], v[7]);
    const float a4 = fmaxf(v[8], v[9]), a5 = fmaxf(v[10], v[11]), a6 = fmaxf(v[12], v[13]), a7 = fmaxf(v[14], v[15]);
    float r;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a0), "v"(a1));   // (asm: keeps hipcc from fusing pairs into v_max3)
    float r2, r3, r4;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r2) : "v"(a2), "v"(a3));
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r3) : "v"(a4), "v"(a5));
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r4) : "v"(a6), "v"(a7));
    return fmaxf(fmaxf(r, r2), fmaxf(r3, r4));
}

__device__ __forceinline__ float max16(const f32x16& v) {
    const float a0 = fmaxf(fmaxf(v[0], v[1]), v[2]), a1 = fmaxf(fmaxf(v[3], v[4]), v[5]);
    const float a2 = fmaxf(fmaxf(v[6], v[7]), v[8]), a3 = fmaxf(fmaxf(v[9], v[10]), v[11]);
    const float a4 = fmaxf(fmaxf(v[12], v[13]), v[14]);
    return fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), v[15]));
}

template <int FEAT, int NCH>
__global__ __launch_bounds__(256, 1) void k(const char* __restrict__ tiles, int n_tiles_buf, const unsigned* __restrict__ bits,
                                           float* out, int n_tiles, unsigned long long* clk) {
    constexpr int UW = 4, TILE_B = NCH * 1024, RING = 4, GR = 2, NG = NCH / GR, BG = NG / 2 - 1, CPW = NCH / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    f32x4 b[NCH][UW];
    for (int q = 0; q < NCH; ++q)
        for (int u = 0; u < UW; ++u)
            b[q][u] = *reinterpret_cast<const f32x4*>(tiles + ((size_t)((wave * 31 + q * UW + u) * 64 % 4096) + lane) * 16);
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            asm volatile("" : "+a"(b[q][u].x));
            asm volatile("" : "+a"(b[q][u].y));
            asm volatile("" : "+a"(b[q][u].z));
            asm volatile("" : "+a"(b[q][u].w));
        }
    char* ring = smem;
    unsigned* tb = reinterpret_cast<unsigned*>(smem + RING * TILE_B);
    const char* mine = tiles + (wave * CPW) * 1024 + lane * 16;
    auto dma_tile = [&](int j, int slot) __attribute__((always_inline)) {
        const int t = (blockIdx.x / 32 * 131 + j) % n_tiles_buf;      // the workgroups of an XCD-sized group walk the same tiles
        const char* g = mine + (size_t)t * TILE_B;
        char* l = ring + slot * TILE_B + (wave * CPW) * 1024;
        if (FEAT & 16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bits + (t & ~63) + lane),
                                             (__attribute__((address_space(3))) void*)(tb + ((t >> 6) & 1) * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 1024, 0);
        if constexpr (CPW > 2) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 2048, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 3072, 0);
        }
    };
    f32x16 acc[2][UW];
    for (int p = 0; p < 2; ++p)
        for (int u = 0; u < UW; ++u) acc[p][u] = f32x16{};
    float thr[UW] = {1e30f, 1e30f, 1e30f, 1e30f};
    f32x4 c[4][GR];
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    auto rd = [&](f32x4(&dst)[GR], uint32_t addr, auto Gc) __attribute__((always_inline)) {
        if constexpr (FEAT & 8) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(addr), "i"((decltype(Gc)::value * GR + 0) * 1024));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(addr), "i"((decltype(Gc)::value * GR + 1) * 1024));
        } else {
            const char* p = smem + (addr - (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
            dst[0] = *reinterpret_cast<const f32x4*>(p + (decltype(Gc)::value * GR + 0) * 1024);
            dst[1] = *reinterpret_cast<const f32x4*>(p + (decltype(Gc)::value * GR + 1) * 1024);
        }
    };
    auto wt = [&](f32x4(&x)[GR]) __attribute__((always_inline)) {
        if constexpr (FEAT & 8) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[0]), "+v"(x[1]));
    };
    constexpr int AH = (FEAT & 8) ? 2 : 1;       // groups the fragment reads run ahead
    unsigned hits = 0;
    auto body = [&](auto Pc, int j, int s_cur, int s_nxt, int s_fill) __attribute__((always_inline)) {
        constexpr int P = (FEAT & 4) ? decltype(Pc)::value : 0, Q = (FEAT & 4) ? 1 - P : 0;
        const uint32_t src = ring_lds + s_cur * TILE_B, srcn = ring_lds + s_nxt * TILE_B;
        float m[UW];
        auto group = [&](auto Gc) __attribute__((always_inline)) {
            constexpr int g = decltype(Gc)::value;
            if constexpr (g + AH < NG) rd(c[(g + AH) & 3], src, std::integral_constant<int, g + AH>{});
            else rd(c[(g + AH) & 3], srcn, std::integral_constant<int, g + AH - NG>{});
            if constexpr (g == BG && (FEAT & 1)) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"((FEAT & 16) ? CPW + 1 : CPW) : "memory");
                if constexpr (!(FEAT & 256)) __builtin_amdgcn_s_barrier();
            }
            if constexpr (g == BG && (FEAT & 512)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            wt(c[g & 3]);
            if constexpr (g == BG && (FEAT & 1)) dma_tile(j + 3, s_fill);
#pragma unroll
            for (int jj = 0; jj < GR; ++jj) {
                if constexpr ((FEAT & 4) != 0 && g == 0) {
                    if (jj == 0) {
#pragma unroll
                        for (int u = 0; u < UW; ++u) acc[P][u] = f32x16{};
                    }
                }
                const f32x4 cc = c[g & 3][jj];
#pragma unroll
                for (int u = 0; u < UW; ++u) acc[P][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc.x, b[g * GR + jj][u].x, acc[P][u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < UW; ++u) acc[P][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc.z, b[g * GR + jj][u].z, acc[P][u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < UW; ++u) acc[P][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc.y, b[g * GR + jj][u].y, acc[P][u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < UW; ++u) acc[P][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc.w, b[g * GR + jj][u].w, acc[P][u], 0, 0, 0);
            }
            if constexpr (g == 1 && (FEAT & 2) && !(FEAT & 64)) {
#pragma unroll
                for (int u = 0; u < UW; ++u) m[u] = (FEAT & 32) ? max16_chain(acc[Q][u]) : max16(acc[Q][u]);
                if constexpr ((FEAT & 128) != 0) {      // spread: SPREAD VALU per MFMA gap over the group's 32 MFMAs
#pragma unroll
                    for (int q = 0; q < 32; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, (FEAT & 1024) ? 1 : 2, 0);
                    }
                }
            }
            if constexpr (g >= 1 && g <= 4 && NG > 4 && (FEAT & 2) && (FEAT & 64)) {      // one user tile per group
                m[g - 1] = (FEAT & 32) ? max16_chain(acc[Q][g - 1]) : max16(acc[Q][g - 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g == ((FEAT & 64) ? 4 : 1) && (FEAT & 2)) {
                bool hit = false;
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    asm volatile("" : "+v"(m[u]));           // the maxima are computed in the MFMA region above, not here
                    hit = hit | (m[u] > thr[u]);
                }
                if (__ballot(hit) != 0ull) hits += 1;
            }
        };
        [&]<int... Gs>(std::integer_sequence<int, Gs...>) __attribute__((always_inline)) {
            (group(std::integral_constant<int, Gs>{}), ...);
        }(std::make_integer_sequence<int, NG>{});
    };
    if (FEAT & 1) {
        dma_tile(0, 0); dma_tile(1, 1); dma_tile(2, 2);
    } else {
        for (int i = threadIdx.x; i < RING * TILE_B / 16; i += 256)
            reinterpret_cast<f32x4*>(smem)[i] = *reinterpret_cast<const f32x4*>(tiles + ((size_t)blockIdx.x * 7 * 1024 + (size_t)i * 16) % ((size_t)n_tiles_buf * TILE_B));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    rd(c[0], ring_lds, std::integral_constant<int, 0>{});
    if (AH == 2) rd(c[1], ring_lds, std::integral_constant<int, 1>{});
    const unsigned long long t0 = __builtin_readcyclecounter();
    int s0 = 0;
    for (int j = 0; j < n_tiles; j += 2) {
        const int s1 = (s0 + 1) & 3, s2 = (s0 + 2) & 3, s3 = (s0 + 3) & 3;
        body(std::integral_constant<int, 0>{}, j, s0, s1, s3);
        body(std::integral_constant<int, 1>{}, j + 1, s1, s2, s0);
        s0 = s2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[1][0]), "+v"(c[1][1]), "+v"(c[2][0]), "+v"(c[2][1]),
                 "+v"(c[3][0]), "+v"(c[3][1]));
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)hits;
    for (int p = 0; p < 2; ++p)
        for (int u = 0; u < UW; ++u)
            for (int r = 0; r < 16; ++r) s += acc[p][u][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int FEAT, int NCH>
void run(const char* dt, int ntb, const unsigned* bits, float* dout, unsigned long long* dclk, int n_tiles, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<FEAT, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * NCH * 1024 + 512);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FEAT, NCH>), dim3(256), dim3(256), 4 * NCH * 1024 + 512, 0, dt, ntb * (16 / NCH), bits, dout, n_tiles, dclk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    (void)hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost);
    const double n_mfma = NCH * 4.0 * 4 * n_tiles;       // per wave
    const double flop = 2.0 * 32 * 32 * 2 * n_mfma * 4 * 256;
    printf("d=%-3d %-58s %7.2f ms  %.3f of 157.3 TF  clock %.3f GHz  %.1f cycles per MFMA (64 = peak)  (%s)\n", NCH * 8, name, ms,
           flop / ms / 1e9 / 157.3, c / (ms * 1e6), (double)c / n_mfma, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int n_tiles = argc > 1 ? atoi(argv[1]) : 4000;
    const int ntb = 65536;                      // 1 GiB of tiles: the stream does not fit any cache
    std::vector<float> h((size_t)4096 * 64 * 4);
    srand(1);
    for (auto& x : h) x = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    char* dt;
    unsigned* bits;
    float* dout;
    unsigned long long* dclk;
    (void)hipMalloc(&dt, (size_t)ntb * 16384);
    for (size_t off = 0; off < (size_t)ntb * 16384; off += h.size() * 4) (void)hipMemcpy(dt + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&bits, (size_t)(ntb + 64) * 4);
    (void)hipMemset(bits, 0x55, (size_t)(ntb + 64) * 4);
    (void)hipMalloc(&dout, 256 * 256 * 4);
    (void)hipMalloc(&dclk, 256 * 8);
    for (int rep = 0; rep < 2; ++rep) {
#define LADDER(N)                                                                                                        \
        run<0, N>(dt, ntb, bits, dout, dclk, n_tiles, "bare: static tile, compiler loads");                              \
        run<8, N>(dt, ntb, bits, dout, dclk, n_tiles, "+ asm reads two groups ahead");                                   \
        run<8 | 4, N>(dt, ntb, bits, dout, dclk, n_tiles, "+ two accumulator sets, zero-init by MFMA");                  \
        run<8 | 4 | 2, N>(dt, ntb, bits, dout, dclk, n_tiles, "+ threshold test VALU");                                  \
        run<8 | 4 | 1, N>(dt, ntb, bits, dout, dclk, n_tiles, "+ DMA stream and barrier (no threshold test)");           \
        run<8 | 4 | 2 | 1, N>(dt, ntb, bits, dout, dclk, n_tiles, "+ DMA stream, barrier, threshold test");              \
        run<8 | 4 | 2 | 1 | 16, N>(dt, ntb, bits, dout, dclk, n_tiles, "+ tile-bits piece (the kernel's stream)");       \
        run<8 | 4 | 1 | 256, N>(dt, ntb, bits, dout, dclk, n_tiles, "DMA stream WITHOUT the barrier (no test)");         \
        run<8 | 4 | 512, N>(dt, ntb, bits, dout, dclk, n_tiles, "static tile + barrier per tile (no DMA, no test)");          \
        run<8 | 4 | 2 | 128, N>(dt, ntb, bits, dout, dclk, n_tiles, "threshold test VALU spread 2 per MFMA gap (static tile)");     \
        run<8 | 4 | 2 | 128 | 1024, N>(dt, ntb, bits, dout, dclk, n_tiles, "threshold test VALU spread 1 per MFMA gap (static tile)"); \
        run<8 | 4 | 2 | 32, N>(dt, ntb, bits, dout, dclk, n_tiles, "threshold test as two-operand max chain (static tile)");
        LADDER(8)
        LADDER(16)
    }
    return 0;
}
