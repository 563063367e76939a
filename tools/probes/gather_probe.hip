// Micro-benchmark: random row gathers of one column slice, as the LightGCN SpMM issues them.  Is the L2 channel mapping
// hurt by a power-of-two row stride combined with a fixed column offset?  (build: hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int G>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ X, const int* __restrict__ idx, int n_groups_edges,
                                                     int stride_f, int off_f, int per_xcd_offset, float* out) {
    const int lig = threadIdx.x % G;
    const int g = (blockIdx.x * 256 + threadIdx.x) / G;
    const int ng = gridDim.x * 256 / G;
    const int xoff = per_xcd_offset ? (blockIdx.x & 7) * per_xcd_offset : 0;
    f32x4 acc = {0, 0, 0, 0};
    for (int e = g * 8; e + 8 <= n_groups_edges; e += ng * 8) {
        f32x4 x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const f32x4*>(X + (size_t)idx[e + q] * stride_f + off_f + xoff + lig * 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc.x += x[q].x; acc.y += x[q].y; acc.z += x[q].z; acc.w += x[q].w; }
    }
    if (acc.x == 12345.f) out[0] = acc.y + acc.z + acc.w;
}

int main() {
    const int N = 22531, E = 260000 * 4;        // edges x column slices of one SpMM
    std::vector<int> h(E);
    srand(1);
    for (auto& v : h) v = rand() % N;
    int* idx; float *X, *out;
    hipMalloc(&idx, E * 4); hipMemcpy(idx, h.data(), E * 4, hipMemcpyHostToDevice);
    hipMalloc(&X, (size_t)N * 256 * 4); hipMemset(X, 0, (size_t)N * 256 * 4); hipMalloc(&out, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { const char* name; int G, stride, off, xcdoff; } cfgs[] = {
        {"G=8 (128 B), stride 512 B, one fixed column offset (all XCDs the same slice)", 8, 128, 32, 0},
        {"G=8 (128 B), stride 512 B, XCD-pinned slices (offset = (b%8)%4 * 128 B)    ", 8, 128, 0, -4},
        {"G=8 (128 B), stride 640 B, fixed offset                                     ", 8, 160, 32, 0},
        {"G=8 (128 B), stride 128 B (slice-major layout)                              ", 8, 32, 0, 0},
        {"G=4 ( 64 B), stride 512 B, XCD-pinned (offset = b%8 * 64 B)                 ", 4, 128, 0, 16},
        {"G=4 ( 64 B), stride  64 B (slice-major)                                     ", 4, 16, 0, 0},
        {"G=32 (512 B whole rows), stride 512 B                                       ", 32, 128, 0, 0},
    };
    for (auto& c : cfgs) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            const int edges = c.G == 32 ? E / 4 : (c.G == 4 ? E * 2 : E);    // same bytes: 133 MB
            int xo = c.xcdoff;
#define LAUNCH(GG) hipLaunchKernelGGL(gather_kernel<GG>, dim3(1792), dim3(256), 0, 0, X, idx, edges > E ? E : edges, c.stride, c.off, xo == -4 ? 0 : xo, out)
            if (c.xcdoff == -4) {   // emulate (b%8)%4 * 32 floats
                hipLaunchKernelGGL(gather_kernel<8>, dim3(1792), dim3(256), 0, 0, X, idx, E, c.stride, 0, 32, out);   // b%8*128B: 8 offsets over 2 rows' worth
            } else if (c.G == 8) LAUNCH(8); else if (c.G == 4) LAUNCH(4); else LAUNCH(32);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double bytes = (c.G == 4 ? (double)(E) * 64 : (c.G == 32 ? (double)(E / 4) * 512 : (double)E * 128));
        printf("%s  %.1f us  %.2f TB/s\n", c.name, best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
