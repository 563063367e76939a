// Ceiling for the S-TRAIN-XL SpMM's gathers: random 512-byte rows (32 lanes x 16 B, eight rows in flight per lane group, the
// SpMM's access shape at d = 128) out of tables that do / do not fit the 256 MB Infinity Cache, uniform and Zipf(0.8)-like
// indices, nothing else in the kernel.  If this loop cannot go faster out of a cache-resident table than out of HBM, a
// column-blocked SpMM pass cannot either.   hipcc --offload-arch=gfx950 -O3 -o xl_gather_probe xl_gather_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gather_rows(const float* __restrict__ X, const int* __restrict__ idx, long n_edges, float* out) {
    const int lig = threadIdx.x & 31;
    const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 5;
    const long ng = (long)gridDim.x * 8;
    f32x4 acc = {0, 0, 0, 0};
    for (long e = g * 8; e + 8 <= n_edges; e += ng * 8) {
        int id = lig < 8 ? idx[e + lig] : 0;
        f32x4 x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const f32x4*>(X + (size_t)__shfl(id, q, 32) * 128 + lig * 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc.x += x[q].x; acc.y += x[q].y; acc.z += x[q].z; acc.w += x[q].w; }
    }
    if (acc.x == 12345.f) out[0] = acc.y + acc.z + acc.w;
}

int main() {
    const long E = 100L * 1000 * 1000;                      // 51 GB of gathered rows per launch
    int* idx; float *X, *out;
    hipMalloc(&idx, E * 4); hipMalloc(&out, 16);
    const long max_rows = 10L * 1000 * 1000;
    hipMalloc(&X, (size_t)max_rows * 512); hipMemset(X, 0, (size_t)max_rows * 512);
    std::vector<int> h(E);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { const char* name; long rows; double zipf; } cfgs[] = {
        {"uniform over    250 000 rows (128 MB: Infinity-Cache resident)", 250000, 0.0},
        {"uniform over  1 000 000 rows (512 MB: the user table of S-TRAIN-XL)", 1000000, 0.0},
        {"uniform over 10 000 000 rows (5.1 GB: the item table)", 10000000, 0.0},
        {"Zipf(0.8) over 10 000 000 rows, shuffled ids (what a user row gathers)", 10000000, 0.8},
    };
    for (auto& c : cfgs) {
        unsigned long long s = 88172645463325252ULL;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
        if (c.zipf == 0.0) {
            for (long i = 0; i < E; ++i) h[i] = (int)(rnd() % (unsigned long long)c.rows);
        } else {                                            // inverse CDF of the continuous power law, ids scrambled by a multiplier
            const double ex = 1.0 - c.zipf, top = pow((double)c.rows, ex);
            for (long i = 0; i < E; ++i) {
                const double u = (double)(rnd() >> 11) / 9007199254740992.0;
                long k = (long)pow(1.0 + u * (top - 1.0), 1.0 / ex) - 1;
                if (k >= c.rows) k = c.rows - 1;
                h[i] = (int)((unsigned long long)k * 7919ULL % (unsigned long long)c.rows);
            }
        }
        hipMemcpy(idx, h.data(), E * 4, hipMemcpyHostToDevice);
        for (int grid : {2048, 8192}) {
            gather_rows<<<grid, 256>>>(X, idx, E, out);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 3; ++r) gather_rows<<<grid, 256>>>(X, idx, E, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
            printf("%-72s grid %5d: %7.2f ms  %6.2f TB/s of gathered rows\n", c.name, grid, ms, E * 512.0 / ms / 1e9);
        }
    }
    return 0;
}
