// Round-3 probe, design W: ONE WAVE PER ROW, everything about the row uniform -> descriptor and edge list come through the
// SCALAR path (s_load), the gather address is SGPR base + lane offset, the edge weight an SGPR operand of v_fma: per edge
// 1 VMEM + D/64 VALU instead of ~20 wave-instructions in the lane-group kernel (which is ISSUE-bound: gathers pointed at
// one cached row and no stores still take 12 us).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o spmm_w_probe spmm_w_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Edge { int col; float val; };

// desc: {row, cnt (padded to 4), first pair, real cnt}; pairs: (col, val) in work-item order, rows padded with (0, 0.f)
template <int WPB, int UN>
__global__ __launch_bounds__(WPB * 64) void spmm_w(const i32x4* __restrict__ desc, int n_light, const Edge* __restrict__ pairs,
                                                   const float* __restrict__ X, float* __restrict__ Y,
                                                   const float* __restrict__ acc_in, float* __restrict__ acc_out, float s_in,
                                                   float s_out, int light_blocks, const i32x4* __restrict__ hdesc, int n_heavy,
                                                   int abl, int xcd_user, int n_user_light) {
    constexpr int D = 128;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ f32x2 part[WPB][64];
    int b = blockIdx.x;
    if (b < n_heavy) {
        if (abl & 8) return;
        const i32x4 hd = hdesc[b];
        const int cnt = hd.y;                 // padded to a multiple of 8 * WPB?  no: chunk per wave below
        int chunk = (cnt + WPB - 1) / WPB;
        chunk = (chunk + 7) & ~7;
        const int e0 = wv * chunk, e1 = min(cnt, e0 + chunk);
        f32x2 acc = {0.f, 0.f};
        const Edge* p = pairs + hd.z;
        for (int e = e0; e < e1; e += 8) {    // heavy streams are padded to a multiple of 8
            f32x2 x[8]; float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const Edge ed = p[e + q];
                v[q] = ed.val;
                x[q] = reinterpret_cast<const f32x2*>(X + (size_t)ed.col * D)[lane];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc.x = fmaf(v[q], x[q].x, acc.x); acc.y = fmaf(v[q], x[q].y, acc.y); }
        }
        part[wv][lane] = acc;
        __syncthreads();
        if (wv == 0) {
            f32x2 r = part[0][lane];
#pragma unroll
            for (int k = 1; k < WPB; ++k) { r.x += part[k][lane].x; r.y += part[k][lane].y; }
            const size_t o = (size_t)hd.x * D + lane * 2;
            const f32x2 z = *reinterpret_cast<const f32x2*>(acc_in + o);
            *reinterpret_cast<f32x2*>(Y + o) = r;
            f32x2 w; w.x = (z.x * s_in + r.x) * s_out; w.y = (z.y * s_in + r.y) * s_out;
            *reinterpret_cast<f32x2*>(acc_out + o) = w;
        }
        return;
    }
    b -= n_heavy;
    if (b >= light_blocks || (abl & 16)) return;
    // light rows: wave w takes work items w, w + n_waves, ...   (xcd_user > 0: XCDs [0, xcd_user) take the user rows, the others
    // the item rows -- block b sits on XCD b % 8)
    long w, stride, end;
    if (xcd_user > 0) {
        const int xcd = b & 7, j = b >> 3;
        const int per8 = light_blocks >> 3;
        if (xcd < xcd_user) { w = ((long)j * xcd_user + xcd) * WPB + wv; stride = (long)per8 * xcd_user * WPB; end = n_user_light; }
        else { w = n_user_light + ((long)j * (8 - xcd_user) + (xcd - xcd_user)) * WPB + wv; stride = (long)per8 * (8 - xcd_user) * WPB; end = n_light; }
    } else { w = (long)b * WPB + wv; stride = (long)light_blocks * WPB; end = n_light; }
    for (; w < end; w += stride) {
        const i32x4 dsc = desc[w];
        const int row = dsc.x, cnt = dsc.y;
        const Edge* p = pairs + dsc.z;
        const size_t o = (size_t)row * D + lane * 2;
        f32x2 z = {0.f, 0.f};
        if (!(abl & 2)) z = *reinterpret_cast<const f32x2*>(acc_in + o);
        f32x2 acc = {0.f, 0.f};
        for (int e = 0; e < cnt; e += UN) {
            f32x2 x[UN]; float v[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const Edge ed = p[e + q];
                v[q] = ed.val;
                x[q] = reinterpret_cast<const f32x2*>(X + (size_t)((abl & 4) ? 0 : ed.col) * D)[lane];
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) { acc.x = fmaf(v[q], x[q].x, acc.x); acc.y = fmaf(v[q], x[q].y, acc.y); }
        }
        if (abl & 1) { if (acc.x == 12345.f) Y[0] = z.x; continue; }
        *reinterpret_cast<f32x2*>(Y + o) = acc;
        f32x2 r; r.x = (z.x * s_in + acc.x) * s_out; r.y = (z.y * s_in + acc.y) * s_out;
        *reinterpret_cast<f32x2*>(acc_out + o) = r;
    }
}

struct Graph { long n_u, n_i, nnz; std::vector<long> rp; std::vector<int> col; std::vector<float> val; };
static Graph load(const char* path) {
    Graph g; FILE* f = fopen(path, "rb"); if (!f) { printf("cannot open %s\n", path); exit(1); }
    long h[3]; if (fread(h, 8, 3, f) != 3) exit(1);
    g.n_u = h[0]; g.n_i = h[1]; g.nnz = h[2];
    const long n = g.n_u + g.n_i;
    g.rp.resize(n + 1); g.col.resize(g.nnz); g.val.resize(g.nnz);
    if (fread(g.rp.data(), 8, n + 1, f) != (size_t)(n + 1)) exit(1);
    if (fread(g.col.data(), 4, g.nnz, f) != (size_t)g.nnz) exit(1);
    if (fread(g.val.data(), 4, g.nnz, f) != (size_t)g.nnz) exit(1);
    fclose(f); return g;
}
template <typename T> T* up(const std::vector<T>& v) {
    T* p; CK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "tools/probes/data/citeulike.bin";
    const int seg = 64;
    Graph g = load(path);
    const long N = g.n_u + g.n_i; const int d = 128;
    // work list: user rows first (descending length), then item rows (descending length); heavy rows apart
    struct Built { std::vector<i32x4> light, heavy; std::vector<Edge> pairs; int n_user_light; };
    auto build = [&](int pad) {
        Built B;
        auto add_class = [&](long lo, long hi) {
            std::vector<int> rows(hi - lo); std::iota(rows.begin(), rows.end(), (int)lo);
            std::stable_sort(rows.begin(), rows.end(), [&](int x, int y) { return g.rp[x + 1] - g.rp[x] > g.rp[y + 1] - g.rp[y]; });
            for (int r : rows) {
                const int cnt = (int)(g.rp[r + 1] - g.rp[r]);
                const bool heavy = cnt > seg;
                const int p = heavy ? 8 : pad;
                const int cp = (cnt + p - 1) / p * p;
                i32x4 dsc = {r, cp, (int)B.pairs.size(), cnt};
                for (long e = g.rp[r]; e < g.rp[r + 1]; ++e) B.pairs.push_back({g.col[e], g.val[e]});
                for (int k = cnt; k < cp; ++k) B.pairs.push_back({0, 0.f});
                (heavy ? B.heavy : B.light).push_back(dsc);
            }
        };
        add_class(0, g.n_u); B.n_user_light = (int)B.light.size(); add_class(g.n_u, N);
        std::sort(B.heavy.begin(), B.heavy.end(), [](const i32x4& x, const i32x4& y) { return x.w > y.w; });
        for (int k = 0; k < 64; ++k) B.pairs.push_back({0, 0.f});
        return B;
    };
    std::vector<float> hX((size_t)N * d), hZ((size_t)N * d);
    srand(7);
    for (auto& v : hX) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hZ) v = (rand() % 2001 - 1000) * 1e-3f;
    float *X = up(hX), *Z = up(hZ), *Y, *O;
    CK(hipMalloc(&Y, (size_t)N * d * 4)); CK(hipMalloc(&O, (size_t)N * d * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pad : {4, 8}) {
        Built B = build(pad);
        printf("pad %d: light %zu (users %d) heavy %zu pairs %zu (%.2f x edges)\n", pad, B.light.size(), B.n_user_light, B.heavy.size(),
               B.pairs.size(), (double)B.pairs.size() / g.nnz);
        i32x4 *dl = up(B.light), *dh = up(B.heavy); Edge* dp = up(B.pairs);
        // interleave the two classes for the unpinned mapping: sort light by padded length overall
        std::vector<i32x4> mixed = B.light;
        std::stable_sort(mixed.begin(), mixed.end(), [](const i32x4& x, const i32x4& y) { return x.y > y.y; });
        i32x4* dm = up(mixed);
        auto run = [&](const char* name, int wpb, int R, int abl, int xcd_user) {
            const int n_light = (int)B.light.size(), n_heavy = (int)B.heavy.size();
            int light_blocks = (n_light + wpb * R - 1) / (wpb * R);
            light_blocks = (light_blocks + 7) & ~7;
            const int grid = n_heavy + light_blocks;
            const i32x4* dd = xcd_user > 0 ? dl : dm;
            auto launch = [&]() {
                if (pad == 4) {
                    if (wpb == 4) hipLaunchKernelGGL((spmm_w<4, 4>), dim3(grid), dim3(256), 0, 0, dd, n_light, dp, X, Y, Z, O, 1.f, 0.25f, light_blocks, dh, n_heavy, abl, xcd_user, B.n_user_light);
                    else hipLaunchKernelGGL((spmm_w<16, 4>), dim3(grid), dim3(1024), 0, 0, dd, n_light, dp, X, Y, Z, O, 1.f, 0.25f, light_blocks, dh, n_heavy, abl, xcd_user, B.n_user_light);
                } else {
                    if (wpb == 4) hipLaunchKernelGGL((spmm_w<4, 8>), dim3(grid), dim3(256), 0, 0, dd, n_light, dp, X, Y, Z, O, 1.f, 0.25f, light_blocks, dh, n_heavy, abl, xcd_user, B.n_user_light);
                    else hipLaunchKernelGGL((spmm_w<16, 8>), dim3(grid), dim3(1024), 0, 0, dd, n_light, dp, X, Y, Z, O, 1.f, 0.25f, light_blocks, dh, n_heavy, abl, xcd_user, B.n_user_light);
                }
            };
            CK(hipMemset(Y, 0xff, (size_t)N * d * 4)); CK(hipMemset(O, 0xff, (size_t)N * d * 4));
            launch(); CK(hipDeviceSynchronize());
            std::vector<float> y((size_t)N * d), o((size_t)N * d);
            CK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), O, o.size() * 4, hipMemcpyDeviceToHost));
            long bad = 0, badh = 0;
            for (long r = 0; r < N; r += 37) {
                const bool heavy = g.rp[r + 1] - g.rp[r] > seg;
                for (int k = 0; k < d; k += 5) {
                    float s = 0.f;
                    for (long e = g.rp[r]; e < g.rp[r + 1]; ++e) s = fmaf(g.val[e], hX[(size_t)g.col[e] * d + k], s);
                    const float w = (hZ[r * d + k] * 1.0f + s) * 0.25f;
                    if (!heavy) { if (memcmp(&s, &y[r * d + k], 4) || memcmp(&w, &o[r * d + k], 4)) ++bad; }
                    else if (fabsf(s - y[r * d + k]) > 1e-4f * (1.f + fabsf(s))) ++badh;
                }
            }
            for (int w = 0; w < 5; ++w) launch();
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
            }
            CK(hipEventRecord(e0)); for (int rep = 0; rep < 30; ++rep) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms30; CK(hipEventElapsedTime(&ms30, e0, e1));
            printf("  %-44s wpb %2d R %d grid %5d  best %.1f us  back-to-back %.1f us  %s\n", name, wpb, R, grid, best * 1e3, ms30 / 30 * 1e3,
                   abl ? "(ablated)" : bad || badh ? "MISMATCH" : "ok");
            if (!abl && (bad || badh)) printf("     mismatches: light %ld heavy %ld\n", bad, badh);
        };
        for (int wpb : {4, 16}) {
            for (int R : {1, 2, 3, 4, 6}) run("all XCDs, mixed rows", wpb, R, 0, 0);
            run("user rows on XCD 0-3, item rows on 4-7", wpb, 3, 0, 4);
            run("user rows on XCD 0-4, item rows on 5-7", wpb, 3, 0, 5);
            run("user rows on XCD 0-5, item rows on 6-7", wpb, 3, 0, 6);
            run("no heavy", wpb, 3, 8, 0);
            run("no light", wpb, 3, 16, 0);
            run("no heavy, gathers hit row 0", wpb, 3, 8 | 4, 0);
            run("no heavy, no stores, no acc_in", wpb, 3, 8 | 3, 0);
            run("no heavy, row 0, no stores, no acc_in", wpb, 3, 8 | 7, 0);
        }
        hipFree(dl); hipFree(dh); hipFree(dp); hipFree(dm);
    }
    return 0;
}
