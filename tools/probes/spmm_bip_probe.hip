// Round-3 probe: variants of the LightGCN SpMM on the real CiteULike-shaped graph (tools/probes/data/citeulike.bin).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o spmm_bip_probe spmm_bip_probe.hip
// Every variant computes  P = A X;  Y = P;  acc_out = (acc_in * s_in + P) * s_out  (forward layer 1 / 2 of the step) and is
// checked against a CPU chain on sampled rows.  What varies:
//   map 0: block b works on column slice (b % 8) % 4 over ALL rows (the round-2 schedule)
//   map 1: bipartite pairing: XCD s (0..3) takes the USER rows of slice s, XCD s + 4 the ITEM rows of slice s -- every byte of
//          X is gathered by exactly one XCD (user rows gather the item half, item rows the user half)
//   R    : rows per lane group;  pipe 0: one after the other (round 2), pipe 1: descriptors / edge chunks / acc_in of all R
//          rows requested together, gathers of all R rows in flight together
//   heavy-first: the workgroup-per-heavy-row blocks lead the grid instead of trailing it
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
    const int* col; const float* val; const float* X; float* Y; const float* acc_in; float* acc_out; float s_in, s_out;
    int d;
    const i32x4* desc[2];   // per class: {row, e0, cnt, heavy?}
    int n_light[2];         // light work items per class
    const i32x4* hdesc[2];  // heavy rows per class
    int n_heavy[2];
    int light_blocks[2];
    int map, heavy_first, cs;
};

__device__ __forceinline__ void fma4(f32x4& a, float v, const f32x4& x) {
    a.x = fmaf(v, x.x, a.x); a.y = fmaf(v, x.y, a.y); a.z = fmaf(v, x.z, a.z); a.w = fmaf(v, x.w, a.w);
}
__device__ __forceinline__ void store_row(const Args& a, long o, const f32x4& acc, const f32x4& z) {
    *reinterpret_cast<f32x4*>(a.Y + o) = acc;
    f32x4 r;
    r.x = (z.x * a.s_in + acc.x) * a.s_out; r.y = (z.y * a.s_in + acc.y) * a.s_out;
    r.z = (z.z * a.s_in + acc.z) * a.s_out; r.w = (z.w * a.s_in + acc.w) * a.s_out;
    *reinterpret_cast<f32x4*>(a.acc_out + o) = r;
}

template <int G>
__device__ __forceinline__ void row_edges_seq(const Args& a, long e0, long e1, int c, int lig, f32x4& acc) {
    for (long base = e0; base < e1; base += G) {
        const long e = base + lig;
        const int my_col = e < e1 ? a.col[e] : 0;
        const float my_val = e < e1 ? a.val[e] : 0.f;
        const int cnt = (int)((e1 - base) < G ? (e1 - base) : G);
        int t = 0;
        for (; t + 8 <= cnt; t += 8) {
            int cc[8]; float vv[8]; f32x4 x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { cc[q] = __shfl(my_col, t + q, G); vv[q] = __shfl(my_val, t + q, G); }
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = reinterpret_cast<const f32x4*>(a.X + (long)cc[q] * a.d)[c];
#pragma unroll
            for (int q = 0; q < 8; ++q) fma4(acc, vv[q], x[q]);
        }
        for (; t + 4 <= cnt; t += 4) {
            int cc[4]; float vv[4]; f32x4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { cc[q] = __shfl(my_col, t + q, G); vv[q] = __shfl(my_val, t + q, G); }
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = reinterpret_cast<const f32x4*>(a.X + (long)cc[q] * a.d)[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) fma4(acc, vv[q], x[q]);
        }
        for (; t < cnt; ++t) {
            const int cc = __shfl(my_col, t, G); const float vv = __shfl(my_val, t, G);
            fma4(acc, vv, reinterpret_cast<const f32x4*>(a.X + (long)cc * a.d)[c]);
        }
    }
}

// G lanes per row slice (cs * G * 4 = d), R rows per lane group, PIPE: rows in flight together, H: gathers per row and round
template <int G, int R, int PIPE, int H>
__global__ __launch_bounds__(256) void spmm_k(Args a) {
    const int lig = threadIdx.x % G;
    const int xcd = blockIdx.x & 7;
    int slice, cls; long j;
    if (a.map == 0) { slice = xcd % a.cs; cls = 0; j = (long)(blockIdx.x >> 3) * (8 / a.cs) + xcd / a.cs; }
    else { slice = xcd % a.cs; cls = xcd / a.cs; j = blockIdx.x >> 3; if (cls > 1) return; }
    const int c = slice * G + lig;
    constexpr int GPB = 256 / G;
    const long heavy_blocks = a.n_heavy[cls];
    long jl = a.heavy_first ? j - heavy_blocks : j;          // light block index
    long jh = a.heavy_first ? j : j - a.light_blocks[cls];   // heavy block index
    if (jl >= 0 && jl < a.light_blocks[cls]) {
        const i32x4* desc = a.desc[cls];
        const long n = a.n_light[cls];
        const long stride = (long)a.light_blocks[cls] * GPB;
        const long w0 = jl * GPB + threadIdx.x / G;
        if (!PIPE) {
            for (long w = w0; w < n; w += stride) {
                const i32x4 dsc = desc[w];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                row_edges_seq<G>(a, dsc.y, (long)dsc.y + dsc.z, c, lig, acc);
                const long o = (long)dsc.x * a.d + (long)c * 4;
                store_row(a, o, acc, *reinterpret_cast<const f32x4*>(a.acc_in + o));
            }
            return;
        }
        // ---- pipelined: the R rows of this lane group advance together
        i32x4 dsc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long w = w0 + r * stride;
            dsc[r] = w < n ? desc[w] : i32x4{0, 0, 0, 0};
            if (w >= n) dsc[r].x = -1;
        }
        int mc[R]; float mv[R]; f32x4 z[R], acc[R];
        int maxcnt = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool in = lig < dsc[r].z;
            mc[r] = in ? a.col[dsc[r].y + lig] : 0;
            mv[r] = in ? a.val[dsc[r].y + lig] : 0.f;
            acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            maxcnt = dsc[r].z > maxcnt ? dsc[r].z : maxcnt;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            z[r] = dsc[r].x >= 0 ? *reinterpret_cast<const f32x4*>(a.acc_in + (long)dsc[r].x * a.d + (long)c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        // the longest row of the wave decides the trip count (wave-uniform loop keeps the shuffles legal)
#pragma unroll
        for (int off = G; off < 64; off <<= 1) { const int o = __shfl_xor(maxcnt, off); maxcnt = o > maxcnt ? o : maxcnt; }
        for (int base = 0; base < maxcnt; base += G) {
            int nc[R]; float nv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {      // next chunk of every row, requested before this chunk's gathers are consumed
                const bool in = base + G + lig < dsc[r].z;
                nc[r] = in ? a.col[dsc[r].y + base + G + lig] : 0;
                nv[r] = in ? a.val[dsc[r].y + base + G + lig] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < G; t += H) {
                f32x4 x[R][H]; float vv[R][H];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int q = 0; q < H; ++q) {
                        const int cc = __shfl(mc[r], t + q, G);
                        vv[r][q] = __shfl(mv[r], t + q, G);
                        x[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (base + t + q < dsc[r].z) x[r][q] = reinterpret_cast<const f32x4*>(a.X + (long)cc * a.d)[c];
                    }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int q = 0; q < H; ++q)
                        if (base + t + q < dsc[r].z) fma4(acc[r], vv[r][q], x[r][q]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) { mc[r] = nc[r]; mv[r] = nv[r]; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (dsc[r].x >= 0) store_row(a, (long)dsc[r].x * a.d + (long)c * 4, acc[r], z[r]);
        return;
    }
    if (jh < 0 || jh >= heavy_blocks) return;
    __shared__ f32x4 wsum[4][G];
    const i32x4 hd = a.hdesc[cls][jh];
    const long r0 = hd.y, r1 = (long)hd.y + hd.z;
    const int gg = threadIdx.x / G;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        long chunk = (r1 - r0 + GPB - 1) / GPB;
        chunk = (chunk + 7) & ~(long)7;
        const long e0 = r0 + (long)gg * chunk;
        const long e1 = e0 + chunk < r1 ? e0 + chunk : r1;
        if (e0 < r1) row_edges_seq<G>(a, e0, e1, c, lig, acc);
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
        acc.x += __shfl_down(acc.x, off); acc.y += __shfl_down(acc.y, off);
        acc.z += __shfl_down(acc.z, off); acc.w += __shfl_down(acc.w, off);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < G) wsum[wv][lig] = acc;
    __syncthreads();
    if (threadIdx.x < G) {
        f32x4 t0 = wsum[0][lig], t1 = wsum[1][lig], t2 = wsum[2][lig], t3 = wsum[3][lig], r;
        r.x = (t0.x + t1.x) + (t2.x + t3.x); r.y = (t0.y + t1.y) + (t2.y + t3.y);
        r.z = (t0.z + t1.z) + (t2.z + t3.z); r.w = (t0.w + t1.w) + (t2.w + t3.w);
        const long o = (long)hd.x * a.d + (long)c * 4;
        store_row(a, o, r, *reinterpret_cast<const f32x4*>(a.acc_in + o));
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
    const int seg = argc > 2 ? atoi(argv[2]) : 64;
    Graph g = load(path);
    const long N = g.n_u + g.n_i; const int d = 128;
    printf("graph %s: %ld users + %ld items, %ld edges, heavy threshold %d\n", path, g.n_u, g.n_i, g.nnz, seg);
    // work lists: class 0 = user rows, class 1 = item rows (map 1); class "all" for map 0 -- descending length, heavy apart
    auto build = [&](long lo, long hi, std::vector<i32x4>& light, std::vector<i32x4>& heavy) {
        std::vector<int> rows(hi - lo); std::iota(rows.begin(), rows.end(), (int)lo);
        std::stable_sort(rows.begin(), rows.end(), [&](int x, int y) { return g.rp[x + 1] - g.rp[x] > g.rp[y + 1] - g.rp[y]; });
        for (int r : rows) {
            const int cnt = (int)(g.rp[r + 1] - g.rp[r]);
            i32x4 dsc = {r, (int)g.rp[r], cnt, cnt > seg};
            (cnt > seg ? heavy : light).push_back(dsc);
        }
        std::sort(heavy.begin(), heavy.end(), [](const i32x4& x, const i32x4& y) { return x.z > y.z; });
    };
    std::vector<i32x4> lA, hA, lU, hU, lI, hI;
    build(0, N, lA, hA); build(0, g.n_u, lU, hU); build(g.n_u, N, lI, hI);
    printf("light/heavy: all %zu/%zu  users %zu/%zu  items %zu/%zu\n", lA.size(), hA.size(), lU.size(), hU.size(), lI.size(), hI.size());
    int* dcol = up(g.col); float* dval = up(g.val);
    std::vector<float> hX((size_t)N * d), hZ((size_t)N * d);
    srand(7);
    for (auto& v : hX) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hZ) v = (rand() % 2001 - 1000) * 1e-3f;
    float *X = up(hX), *Z = up(hZ), *Y, *O;
    CK(hipMalloc(&Y, (size_t)N * d * 4)); CK(hipMalloc(&O, (size_t)N * d * 4));
    i32x4 *dlA = up(lA), *dhA = up(hA), *dlU = up(lU), *dhU = up(hU), *dlI = up(lI), *dhI = up(hI);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    auto run = [&](const char* name, int map, int R, int pipe, int H, int heavy_first, int resident_cap) {
        Args a{dcol, dval, X, Y, Z, O, 1.0f, 0.25f, d, {nullptr, nullptr}, {0, 0}, {nullptr, nullptr}, {0, 0}, {0, 0}, map, heavy_first, 4};
        constexpr int GPB = 32;
        long grid;
        if (map == 0) {
            a.desc[0] = dlA; a.n_light[0] = (int)lA.size(); a.hdesc[0] = dhA; a.n_heavy[0] = (int)hA.size();
            a.light_blocks[0] = (int)((lA.size() + GPB * R - 1) / (GPB * R));
            const long per_slice = a.light_blocks[0] + a.n_heavy[0];
            grid = ((per_slice + 1) / 2) * 8;
        } else {
            a.desc[0] = dlU; a.n_light[0] = (int)lU.size(); a.hdesc[0] = dhU; a.n_heavy[0] = (int)hU.size();
            a.desc[1] = dlI; a.n_light[1] = (int)lI.size(); a.hdesc[1] = dhI; a.n_heavy[1] = (int)hI.size();
            for (int k = 0; k < 2; ++k) a.light_blocks[k] = (int)((a.n_light[k] + GPB * R - 1) / (GPB * R));
            grid = 8L * std::max(a.light_blocks[0] + a.n_heavy[0], a.light_blocks[1] + a.n_heavy[1]);
        }
        (void)resident_cap;
        auto launch = [&]() {
#define LK(RR, PP, HH) hipLaunchKernelGGL((spmm_k<8, RR, PP, HH>), dim3((unsigned)grid), dim3(256), 0, 0, a)
            if (!pipe) { LK(1, 0, 4); return; }
            if (H == 4) { switch (R) { case 1: LK(1, 1, 4); break; case 2: LK(2, 1, 4); break; case 3: LK(3, 1, 4); break; default: LK(4, 1, 4); } }
            else        { switch (R) { case 1: LK(1, 1, 8); break; case 2: LK(2, 1, 8); break; case 3: LK(3, 1, 8); break; default: LK(4, 1, 8); } }
        };
        CK(hipMemset(Y, 0xff, (size_t)N * d * 4)); CK(hipMemset(O, 0xff, (size_t)N * d * 4));
        launch(); CK(hipDeviceSynchronize());
        // check sampled light rows bitwise against the CPU chain, heavy rows loosely
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
        float best = 1e9f, tot = 0.f; const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); tot += ms;
        }
        // back to back (what a captured step sees): 30 launches between two events
        CK(hipEventRecord(e0)); for (int rep = 0; rep < 30; ++rep) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms30; CK(hipEventElapsedTime(&ms30, e0, e1));
        printf("%-46s grid %5ld  best %.1f us  avg %.1f us  back-to-back %.1f us  %s\n", name, grid, best * 1e3, tot / reps * 1e3,
               ms30 / 30 * 1e3, bad || badh ? "MISMATCH" : "ok");
        if (bad || badh) printf("   mismatches: light %ld heavy %ld\n", bad, badh);
    };
    for (int pass = 0; pass < 2; ++pass) {
        run("map0 R3 seq (round 2)", 0, 3, 0, 4, 0, 0);
        run("map0 R3 seq heavy-first", 0, 3, 0, 4, 1, 0);
        run("map0 R1 seq heavy-first", 0, 1, 0, 4, 1, 0);
        run("map0 R2 pipe H4", 0, 2, 1, 4, 1, 0);
        run("map0 R3 pipe H4", 0, 3, 1, 4, 1, 0);
        run("map0 R4 pipe H4", 0, 4, 1, 4, 1, 0);
        run("map0 R2 pipe H8", 0, 2, 1, 8, 1, 0);
        run("map0 R3 pipe H8", 0, 3, 1, 8, 1, 0);
        run("map1 R1 pipe H8", 1, 1, 1, 8, 1, 0);
        run("map1 R2 pipe H4", 1, 2, 1, 4, 1, 0);
        run("map1 R3 pipe H4", 1, 3, 1, 4, 1, 0);
        run("map1 R4 pipe H4", 1, 4, 1, 4, 1, 0);
        run("map1 R2 pipe H8", 1, 2, 1, 8, 1, 0);
        run("map1 R3 pipe H8", 1, 3, 1, 8, 1, 0);
    }
    return 0;
}
