// Where the host sampler's time goes on the machine it runs on: the two loops of crh_sampler_epoch (sampler.hip) that
// carry an epoch -- the Fisher-Yates shuffle walked per raw draw and the membership test of the drawn negatives --
// rebuilt bare with the variants that were tried against them.  Raw draws come from a pre-generated buffer, so the
// generator's cost is not in these numbers.
//   clang++ -O3 -o sampler_host_probe sampler_host_probe.cpp && ./sampler_host_probe [n_records n_users n_items]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <random>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Out { int32_t *u, *p; };
static inline void emit(const Out& o, int64_t slot, int64_t e) {
    const uint32_t pk = (uint32_t)((uint64_t)e >> 32);
    o.u[slot] = (int32_t)(pk >> 16);
    o.p[slot] = (int32_t)(pk & 0xffffu);
}

// S0: the shipped loop (one pass, rejected draw = self swap)
static int64_t shuffle_s0(int64_t* __restrict__ ord, int64_t n, const uint32_t* __restrict__ w, Out o) {
    int64_t used = 0, i = n - 1;
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);
        const int64_t stop = (int64_t)(mask >> 1);
        int64_t ii = i;
        while (ii > stop) {
            const uint32_t v = w[used++] & mask;
            const bool ok = v <= (uint32_t)ii;
            const int64_t j = ok ? (int64_t)v : ii;
            const int64_t a = ord[ii], b = ord[j];
            ord[ii] = b;
            ord[j] = a;
            emit(o, ii, b);
            ii -= ok;
        }
        i = ii;
    }
    emit(o, 0, ord[0]);
    return used;
}

// S1: two phases per window of WIN raw draws: compact the accepted targets (ALU only), then swap without rejected draws
template <int WIN, int PF>
static int64_t shuffle_s1(int64_t* __restrict__ ord, int64_t n, const uint32_t* __restrict__ w, Out o) {
    int64_t used = 0, i = n - 1;
    uint32_t jl[WIN + 64];
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);
        const int64_t stop = (int64_t)(mask >> 1);
        while (i > stop) {
            // phase A: at most WIN raw draws, at most i - stop accepted
            const int64_t room = i - stop;
            int64_t m = 0;
            int q = 0;
            const uint32_t* ww = w + used;
            while (q < WIN && m < room) {
                const uint32_t v = ww[q++] & mask;
                jl[m] = v;
                m += v <= (uint32_t)(i - m);
            }
            used += q;
            // phase B
            for (int64_t t = 0; t < m; ++t) {
                if (PF > 0 && t + PF < m) __builtin_prefetch(&ord[jl[t + PF]], 1, 3);
                const int64_t ii = i - t, j = jl[t];
                const int64_t a = ord[ii], b = ord[j];
                ord[ii] = b;
                ord[j] = a;
                emit(o, ii, b);
            }
            i -= m;
        }
    }
    emit(o, 0, ord[0]);
    return used;
}

// S2: as S1, but the accept test of a group of 8 draws does not ride on the chain through ii: a draw at most ii - 8 is accepted
// wherever in the group it stands, one above ii is rejected; only a draw in between (8 values out of ii) needs the exact walk
template <int WIN, int PF>
static int64_t shuffle_s2(int64_t* __restrict__ ord, int64_t n, const uint32_t* __restrict__ w, Out o) {
    int64_t used = 0, i = n - 1;
    uint32_t jl[WIN + 64];
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);
        const int64_t stop = (int64_t)(mask >> 1);
        while (i > stop) {
            int64_t ii = i;
            int64_t m = 0;
            int q = 0;
            const uint32_t* ww = w + used;
            while (q < WIN && ii > stop) {
                if (q + 8 <= WIN && ii - stop >= 8) {
                    const uint32_t lo = (uint32_t)(ii - 8), hi = (uint32_t)ii;
                    uint32_t v[8];
                    uint32_t amb = 0;
                    for (int t = 0; t < 8; ++t) {
                        v[t] = ww[q + t] & mask;
                        amb |= (uint32_t)(v[t] > lo) & (uint32_t)(v[t] <= hi);
                    }
                    if (__builtin_expect(amb == 0, 1)) {
                        int64_t mm = m;
                        for (int t = 0; t < 8; ++t) {
                            jl[mm] = v[t];
                            mm += v[t] <= lo;
                        }
                        ii -= mm - m;
                        m = mm;
                        q += 8;
                        continue;
                    }
                }
                const uint32_t v = ww[q++] & mask;
                jl[m] = v;
                const bool ok = v <= (uint32_t)ii;
                m += ok;
                ii -= ok;
            }
            used += q;
            for (int64_t t = 0; t < m; ++t) {
                if (PF > 0 && t + PF < m) __builtin_prefetch(&ord[jl[t + PF]], 1, 3);
                const int64_t pi = i - t, j = jl[t];
                const int64_t a = ord[pi], b = ord[j];
                ord[pi] = b;
                ord[j] = a;
                emit(o, pi, b);
            }
            i = ii;
        }
    }
    emit(o, 0, ord[0]);
    return used;
}

// S3: S2 with phase A in AVX-512: 16 draws masked, compared and compacted (vpcompressd, register form) per step
#include <immintrin.h>
template <int WIN, int PF>
__attribute__((target("avx512f"))) static int64_t shuffle_s3(int64_t* __restrict__ ord, int64_t n, const uint32_t* __restrict__ w, Out o) {
    int64_t used = 0, i = n - 1;
    uint32_t jl[WIN + 64];
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);
        const int64_t stop = (int64_t)(mask >> 1);
        const __m512i vmask = _mm512_set1_epi32((int)mask);
        while (i > stop) {
            int64_t ii = i;
            int64_t m = 0;
            int q = 0;
            const uint32_t* ww = w + used;
            while (q < WIN && ii > stop) {
                if (q + 16 <= WIN && ii - stop >= 16) {
                    const __m512i v = _mm512_and_si512(_mm512_loadu_si512((const void*)(ww + q)), vmask);
                    const __mmask16 acc = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32((int)(uint32_t)(ii - 16)));
                    const __mmask16 rej = _mm512_cmpgt_epu32_mask(v, _mm512_set1_epi32((int)(uint32_t)ii));
                    if (__builtin_expect((__mmask16)(acc | rej) == (__mmask16)0xffff, 1)) {
                        _mm512_storeu_si512((void*)(jl + m), _mm512_maskz_compress_epi32(acc, v));
                        const int c = __builtin_popcount((unsigned)acc);
                        m += c;
                        ii -= c;
                        q += 16;
                        continue;
                    }
                }
                const uint32_t v = ww[q++] & mask;
                jl[m] = v;
                const bool ok = v <= (uint32_t)ii;
                m += ok;
                ii -= ok;
            }
            used += q;
            for (int64_t t = 0; t < m; ++t) {
                if (PF > 0 && t + PF < m) __builtin_prefetch(&ord[jl[t + PF]], 1, 3);
                const int64_t pi = i - t, j = jl[t];
                const int64_t a = ord[pi], b = ord[j];
                ord[pi] = b;
                ord[j] = a;
                emit(o, pi, b);
            }
            i = ii;
        }
    }
    emit(o, 0, ord[0]);
    return used;
}

// S4: S3 on 32-bit elements that ARE the record (user << 16 | item), no index beside it: half the table
template <int WIN, int PF>
__attribute__((target("avx512f"))) static int64_t shuffle_s4(uint32_t* __restrict__ ord, int64_t n, const uint32_t* __restrict__ w, Out o) {
    int64_t used = 0, i = n - 1;
    uint32_t jl[WIN + 64];
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);
        const int64_t stop = (int64_t)(mask >> 1);
        const __m512i vmask = _mm512_set1_epi32((int)mask);
        while (i > stop) {
            int64_t ii = i;
            int64_t m = 0;
            int q = 0;
            const uint32_t* ww = w + used;
            while (q < WIN && ii > stop) {
                if (q + 16 <= WIN && ii - stop >= 16) {
                    const __m512i v = _mm512_and_si512(_mm512_loadu_si512((const void*)(ww + q)), vmask);
                    const __mmask16 acc = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32((int)(uint32_t)(ii - 16)));
                    const __mmask16 rej = _mm512_cmpgt_epu32_mask(v, _mm512_set1_epi32((int)(uint32_t)ii));
                    if (__builtin_expect((__mmask16)(acc | rej) == (__mmask16)0xffff, 1)) {
                        _mm512_storeu_si512((void*)(jl + m), _mm512_maskz_compress_epi32(acc, v));
                        const int c = __builtin_popcount((unsigned)acc);
                        m += c;
                        ii -= c;
                        q += 16;
                        continue;
                    }
                }
                const uint32_t v = ww[q++] & mask;
                jl[m] = v;
                const bool ok = v <= (uint32_t)ii;
                m += ok;
                ii -= ok;
            }
            used += q;
            for (int64_t t = 0; t < m; ++t) {
                if (PF > 0 && t + PF < m) __builtin_prefetch(&ord[jl[t + PF]], 1, 3);
                const int64_t pi = i - t, j = jl[t];
                const uint32_t a = ord[pi], b = ord[j];
                ord[pi] = b;
                ord[j] = a;
                o.u[pi] = (int32_t)(b >> 16); o.p[pi] = (int32_t)(b & 0xffffu);
            }
            i = ii;
        }
    }
    o.u[0] = (int32_t)(ord[0] >> 16); o.p[0] = (int32_t)(ord[0] & 0xffffu);
    return used;
}

// the masked-rejection draws of item ids: D0 shipped (write, advance if accepted), D1 AVX-512 compaction
static int64_t draws_d0(const uint32_t* __restrict__ w, uint32_t imask, uint32_t imax, int32_t* __restrict__ dst, int64_t cnt) {
    int64_t w_ = 0, used = 0;
    while (w_ < cnt) {
        const uint32_t v = w[used++] & imask;
        dst[w_] = (int32_t)v;
        w_ += v <= imax;
    }
    return used;
}
__attribute__((target("avx512f"))) static int64_t draws_d1(const uint32_t* __restrict__ w, uint32_t imask, uint32_t imax,
                                                           int32_t* __restrict__ dst, int64_t cnt) {
    int64_t w_ = 0, used = 0;
    const __m512i vm = _mm512_set1_epi32((int)imask), vx = _mm512_set1_epi32((int)imax);
    while (w_ + 16 <= cnt) {                              // (a full store of 16 stays inside dst)
        const __m512i v = _mm512_and_si512(_mm512_loadu_si512((const void*)(w + used)), vm);
        const __mmask16 acc = _mm512_cmple_epu32_mask(v, vx);
        _mm512_storeu_si512((void*)(dst + w_), _mm512_maskz_compress_epi32(acc, v));
        w_ += __builtin_popcount((unsigned)acc);
        used += 16;
    }
    while (w_ < cnt) {
        const uint32_t v = w[used++] & imask;
        dst[w_] = (int32_t)v;
        w_ += v <= imax;
    }
    return used;
}

// M2: 64 tests into one mask word (no store per test), then the set bits are appended
static int64_t member_m2(const uint64_t* __restrict__ bits, int64_t wpu, const int32_t* __restrict__ u,
                         const int32_t* __restrict__ neg, int64_t n, int64_t batch, int32_t* __restrict__ chk) {
    int64_t tot = 0;
    for (int64_t lo = 0; lo < n; lo += batch) {
        const int64_t hi = lo + batch < n ? lo + batch : n;
        int64_t nc = 0;
        for (int64_t t0 = lo; t0 < hi; t0 += 64) {
            const int64_t t1 = t0 + 64 < hi ? t0 + 64 : hi;
            uint64_t mw = 0;
            for (int64_t t = t0; t < t1; ++t)
                mw |= ((bits[(size_t)u[t] * wpu + (neg[t] >> 6)] >> (neg[t] & 63)) & 1ull) << (t - t0);
            while (mw) {
                chk[nc++] = (int32_t)(t0 + __builtin_ctzll(mw));
                mw &= mw - 1;
            }
        }
        tot += nc;
    }
    return tot;
}

// membership: M0 shipped compaction loop; M1 with a prefetch of the bitmap word PF slots ahead
template <int PF>
static int64_t member(const uint64_t* __restrict__ bits, int64_t wpu, const int32_t* __restrict__ u,
                      const int32_t* __restrict__ neg, int64_t n, int64_t batch, int32_t* __restrict__ chk) {
    int64_t tot = 0;
    for (int64_t lo = 0; lo < n; lo += batch) {
        const int64_t hi = lo + batch < n ? lo + batch : n;
        int64_t nc = 0;
        for (int64_t t = lo; t < hi; ++t) {
            if (PF > 0 && t + PF < hi) __builtin_prefetch(&bits[(size_t)u[t + PF] * wpu + (neg[t + PF] >> 6)], 0, 3);
            chk[nc] = (int32_t)t;
            nc += (bits[(size_t)u[t] * wpu + (neg[t] >> 6)] >> (neg[t] & 63)) & 1u;
        }
        tot += nc;
    }
    return tot;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 650161;
    const int64_t nu = argc > 2 ? atoll(argv[2]) : 6040, ni = argc > 3 ? atoll(argv[3]) : 3706;
    const int reps = 40;
    {   // the cores of the GPU box idle at a third of their clock and take tens of milliseconds to ramp: spin first
        const double t0 = now();
        volatile uint64_t sink = 0;
        while (now() - t0 < 0.5) for (int q = 0; q < 100000; ++q) sink = sink * 6364136223846793005ull + 1442695040888963407ull;
    }
    std::mt19937 g(7);
    std::vector<uint32_t> raw((size_t)(2 * n + 4096));
    for (auto& x : raw) x = g();
    std::vector<int64_t> base((size_t)n), ord((size_t)n), ref;
    for (int64_t t = 0; t < n; ++t) base[t] = t | ((int64_t)(uint32_t)(g() % (nu << 16 | ni)) << 32);
    std::vector<int32_t> uo((size_t)n), po((size_t)n), uref, pref;
    Out o{uo.data(), po.data()};
    auto run = [&](const char* name, auto fn) {
        // checked once from the same start; timed on the permutation as the previous pass left it (as in the library: a copy
        // in front of every pass would leave the table wherever memcpy's streaming stores put it)
        ord = base;
        int64_t used = fn(ord.data(), n, raw.data(), o);
        if (ref.empty()) { ref = ord; uref = uo; pref = po; }
        const bool same = ref == ord && uref == uo && pref == po;
        double best = 1e9;
        for (int r = 0; r < reps; ++r) {
            const double t0 = now();
            used = fn(ord.data(), n, raw.data(), o);
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        printf("%-28s %.3f ms  %.2f ns/raw draw  (%lld raw)  %s\n", name, best * 1e3, best * 1e9 / used, (long long)used,
               same ? "same permutation" : "DIFFERENT");
    };
    printf("records %lld\n", (long long)n);
    run("shuffle S0 shipped", shuffle_s0);
    run("shuffle S1 win 624", shuffle_s1<624, 0>);
    run("shuffle S1 win 624 pf 8", shuffle_s1<624, 8>);
    run("shuffle S1 win 624 pf 24", shuffle_s1<624, 24>);
    run("shuffle S1 win 128", shuffle_s1<128, 0>);
    run("shuffle S1 win 4096 pf 24", shuffle_s1<4096, 24>);
    run("shuffle S2 win 624", shuffle_s2<624, 0>);
    run("shuffle S2 win 624 pf 8", shuffle_s2<624, 8>);
    run("shuffle S2 win 624 pf 24", shuffle_s2<624, 24>);
    run("shuffle S2 win 2496 pf 24", shuffle_s2<2496, 24>);
    const bool avx512 = __builtin_cpu_supports("avx512f");
    if (avx512) {
        run("shuffle S3 avx512 win 624 pf 8", shuffle_s3<624, 8>);
        run("shuffle S3 avx512 win 624 pf 16", shuffle_s3<624, 16>);
        std::vector<uint32_t> o32((size_t)n);
        for (int64_t t = 0; t < n; ++t) o32[t] = (uint32_t)((uint64_t)base[t] >> 32);
        for (int pf : {16, 32}) {
            double best = 1e9;
            int64_t used = 0;
            for (int r = 0; r < reps; ++r) {
                const double t0 = now();
                used = pf == 16 ? shuffle_s4<624, 16>(o32.data(), n, raw.data(), o) : shuffle_s4<624, 32>(o32.data(), n, raw.data(), o);
                const double dt = now() - t0;
                if (dt < best) best = dt;
            }
            printf("shuffle S4 32-bit records pf %d  %.3f ms  %.2f ns/raw draw\n", pf, best * 1e3, best * 1e9 / used);
        }
    }
    {
        std::vector<int32_t> d0((size_t)n + 16), d1((size_t)n + 16);
        const uint32_t imax = (uint32_t)(ni - 1), imask = 0xffffffffu >> __builtin_clz(imax);
        auto rund = [&](const char* name, auto fn, std::vector<int32_t>& dst) {
            double best = 1e9;
            int64_t used = 0;
            for (int r = 0; r < reps; ++r) {
                const double t0 = now();
                // batch by batch, as the library draws them (the compaction restarts per batch)
                used = 0;
                for (int64_t lo = 0; lo < n; lo += 4096) used += fn(raw.data() + used, imask, imax, dst.data() + lo, (lo + 4096 < n ? lo + 4096 : n) - lo);
                const double dt = now() - t0;
                if (dt < best) best = dt;
            }
            printf("%-28s %.3f ms  %.2f ns/raw draw  (%lld raw)\n", name, best * 1e3, best * 1e9 / used, (long long)used);
        };
        rund("draws D0 shipped", draws_d0, d0);
        if (avx512) {
            rund("draws D1 avx512", draws_d1, d1);
            d0.resize((size_t)n); d1.resize((size_t)n);
            printf("draws identical: %s\n", d0 == d1 ? "yes" : "NO");
        }
    }

    const int64_t wpu = (ni + 63) / 64;
    std::vector<uint64_t> bits((size_t)(nu * wpu));
    for (auto& x : bits) x = ((uint64_t)g() << 32 | g()) & ((uint64_t)g() << 32 | g()) & ((uint64_t)g() << 32 | g()) & ((uint64_t)g() << 32 | g());
    std::vector<int32_t> u((size_t)n), neg((size_t)n), chk((size_t)n + 1);
    for (int64_t t = 0; t < n; ++t) { u[t] = (int32_t)(g() % nu); neg[t] = (int32_t)(g() % ni); }
    auto runm = [&](const char* name, auto fn) {
        double best = 1e9;
        int64_t tot = 0;
        for (int r = 0; r < reps; ++r) {
            const double t0 = now();
            tot = fn(bits.data(), wpu, u.data(), neg.data(), n, 4096, chk.data());
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        printf("%-28s %.3f ms  %.2f ns/test  (%lld hits)\n", name, best * 1e3, best * 1e9 / n, (long long)tot);
    };
    runm("membership M0 shipped", member<0>);
    runm("membership pf 8", member<8>);
    runm("membership pf 32", member<32>);
    runm("membership M2 mask words", member_m2);
    {   // the same loops on a bitmap that stays in the first-level cache: what the instructions alone cost
        const int64_t wpu_s = 4;
        std::vector<int32_t> us((size_t)n), negs((size_t)n);
        for (int64_t t = 0; t < n; ++t) { us[t] = u[t] % 64; negs[t] = neg[t] % 256; }
        auto small = [&](const char* name, auto fn) {
            double best = 1e9;
            for (int r = 0; r < reps; ++r) {
                const double t0 = now();
                fn(bits.data(), wpu_s, us.data(), negs.data(), n, 4096, chk.data());
                const double dt = now() - t0;
                if (dt < best) best = dt;
            }
            printf("%-28s %.3f ms  %.2f ns/test  (2 KB bitmap)\n", name, best * 1e3, best * 1e9 / n);
        };
        small("membership M0 shipped", member<0>);
        small("membership M2 mask words", member_m2);
    }
    return 0;
}
