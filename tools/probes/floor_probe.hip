// Round-3 probe: floors on this chip for a CiteULike-sized launch (N = 22 531 rows x 512 B = 11.5 MB per array):
// empty launches, a streaming pass with the SpMM epilogue's traffic (read X-row-sized acc_in, write Y and acc_out), pure
// reads, pure writes, at several grid shapes; back-to-back in one stream (what a captured step sees).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_empty() {}
template <int MODE>   // 0: O = Z * 2 (1 read 1 write)  1: Y = Z, O = Z*2 (1 read 2 writes)  2: read only  3: write only  4: 2 reads 2 writes
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ Z, const f32x4* __restrict__ X, f32x4* __restrict__ Y, f32x4* __restrict__ O, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    f32x4 s = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE == 3) { O[i] = f32x4{1, 2, 3, 4}; continue; }
        f32x4 z = Z[i];
        if (MODE == 4) { const f32x4 x = X[i]; z.x += x.x; z.y += x.y; z.z += x.z; z.w += x.w; }
        if (MODE == 2) { s.x += z.x; s.y += z.y; s.z += z.z; s.w += z.w; continue; }
        if (MODE == 1 || MODE == 4) Y[i] = z;
        z.x *= 2; z.y *= 2; z.z *= 2; z.w *= 2;
        O[i] = z;
    }
    if (MODE == 2 && s.x == 12345.f) O[0] = s;
}
int main() {
    const long N = 22531, n = N * 32;
    f32x4 *Z, *X, *Y, *O;
    CK(hipMalloc(&Z, n * 16)); CK(hipMalloc(&X, n * 16)); CK(hipMalloc(&Y, n * 16)); CK(hipMalloc(&O, n * 16));
    CK(hipMemset(Z, 0, n * 16)); CK(hipMemset(X, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int w = 0; w < 20; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        CK(hipEventRecord(e0)); for (int rep = 0; rep < 100; ++rep) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-60s single %.2f us   back-to-back %.2f us\n", name, best * 1e3, ms / 100 * 1e3);
    };
    for (int pass = 0; pass < 2; ++pass) {
        for (int grid : {256, 1024, 2048, 5632}) {
            char nm[128];
            snprintf(nm, sizeof nm, "empty kernel, grid %d x 256", grid);
            time(nm, [&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0); });
        }
        time("empty kernel, grid 512 x 1024", [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(1024), 0, 0); });
        for (int grid : {512, 1024, 2048, 2816}) {
            char nm[128];
            snprintf(nm, sizeof nm, "read 11.5 MB only, grid %d", grid);
            time(nm, [&] { hipLaunchKernelGGL(k_stream<2>, dim3(grid), dim3(256), 0, 0, Z, X, Y, O, n); });
            snprintf(nm, sizeof nm, "write 11.5 MB only, grid %d", grid);
            time(nm, [&] { hipLaunchKernelGGL(k_stream<3>, dim3(grid), dim3(256), 0, 0, Z, X, Y, O, n); });
            snprintf(nm, sizeof nm, "1 read + 1 write (23 MB), grid %d", grid);
            time(nm, [&] { hipLaunchKernelGGL(k_stream<0>, dim3(grid), dim3(256), 0, 0, Z, X, Y, O, n); });
            snprintf(nm, sizeof nm, "1 read + 2 writes (34.5 MB, the SpMM epilogue), grid %d", grid);
            time(nm, [&] { hipLaunchKernelGGL(k_stream<1>, dim3(grid), dim3(256), 0, 0, Z, X, Y, O, n); });
            snprintf(nm, sizeof nm, "2 reads + 2 writes (46 MB), grid %d", grid);
            time(nm, [&] { hipLaunchKernelGGL(k_stream<4>, dim3(grid), dim3(256), 0, 0, Z, X, Y, O, n); });
        }
    }
    return 0;
}
