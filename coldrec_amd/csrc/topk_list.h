// Wave-cooperative per-user top-k lists kept in LDS (one wave = 64 lanes owns its lists, so
// there is no cross-wave synchronisation anywhere).  Lists are sorted best-first by the
// canonical key; lane t holds entries t and t + 64 during an insertion, hence k <= 128.
#pragma once
#include "crh_common.h"

// All 64 lanes call with wave-uniform (sc, gi).  lsu/liu: this user's k-entry list, cntp its fill.
// Lane t holds entries t and t + 64 during an insertion (k <= CRH_MAX_K = 128); the second half is only touched
// when K > 64 (wave-uniform branch), so the common k <= 64 case costs what it did with one entry per lane.
__device__ __forceinline__ void wave_list_insert(float* lsu, int* liu, int* cntp, int K, float sc,
                                                 int gi, int lane) {
    const int n = __builtin_amdgcn_readfirstlane(*cntp);
    float es = CRH_NEG_INF, es2 = CRH_NEG_INF;
    int ei = CRH_PAD_IDX, ei2 = CRH_PAD_IDX;
    if (lane < n) {
        es = lsu[lane];
        ei = liu[lane];
    }
    const bool ahead = lane < n && crh_better(es, ei, sc, gi);
    int p = __popcll(__ballot(ahead));
    const bool wide = K > 64;
    if (wide) {
        if (lane + 64 < n) {
            es2 = lsu[lane + 64];
            ei2 = liu[lane + 64];
        }
        p += __popcll(__ballot(lane + 64 < n && crh_better(es2, ei2, sc, gi)));
    }
    if (p < K) {
        // every shifted entry was read into registers above, so the stores cannot overtake a load
        if (lane >= p && lane < n && lane + 1 < K) {
            lsu[lane + 1] = es;
            liu[lane + 1] = ei;
        }
        if (wide && lane + 64 >= p && lane + 64 < n && lane + 65 < K) {
            lsu[lane + 65] = es2;
            liu[lane + 65] = ei2;
        }
        if (lane == 0) {
            lsu[p] = sc;
            liu[p] = gi;
            *cntp = n < K ? n + 1 : K;
        }
    }
}

// Write a finished list (n valid entries of K) to out[0..K), padded with (-inf, PAD); whole wave.
__device__ __forceinline__ void wave_list_store(const float* ls, const int* li, int n, int K, float* out_s, int* out_i,
                                                int lane) {
    for (int e = lane; e < K; e += 64) {
        out_s[e] = e < n ? ls[e] : CRH_NEG_INF;
        out_i[e] = e < n ? li[e] : CRH_PAD_IDX;
    }
}

// The same with a filter on the item id: only entries with lo <= id < hi are written (compacted, order kept), the rest
// of out[0..K) is padding.  Seeded lists under item-range cuts: every cut starts from the same prefix top-k, so a cut
// writes the entries of ITS range only (the first cut also the prefix's) and the merge sees every item once.
__device__ __forceinline__ void wave_list_store_range(const float* ls, const int* li, int n, int K, float* out_s, int* out_i,
                                                      int lane, int lo, int hi) {
    int base = 0;
    for (int e0 = 0; e0 < K; e0 += 64) {
        const int e = e0 + lane;
        float sc = CRH_NEG_INF;
        int gi = CRH_PAD_IDX;
        if (e < n) {
            sc = ls[e];
            gi = li[e];
        }
        const bool keep = e < n && gi >= lo && gi < hi;
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            out_s[pos] = sc;
            out_i[pos] = gi;
        }
        base += __popcll(m);
    }
    for (int e = base + lane; e < K; e += 64) {
        out_s[e] = CRH_NEG_INF;
        out_i[e] = CRH_PAD_IDX;
    }
}

// true when the candidate cannot enter a FULL list even at its best possible value
// (the masked value -1e9 may exceed a raw score below -1e9, hence the max)
__device__ __forceinline__ bool wave_list_rejects(const float* lsu, const int* liu, int n, int K,
                                                  float sc, int gi) {
    if (n < K) return false;
    const float ks = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lsu[K - 1])));
    const int ki = __builtin_amdgcn_readfirstlane(liu[K - 1]);
    const float ub = fmaxf(sc, CRH_MASKED_SCORE);
    return !crh_better(ub, gi, ks, ki);
}

// Is global item gi masked for block slot `slot`?  bitmap bit or membership in the user's
// ascending rated list (wave-parallel scan, 64 entries per step).  Wave-uniform result.
__device__ __forceinline__ bool wave_is_masked(int gi, int64_t slot, const int64_t* rated_rowptr,
                                               const int32_t* rated_col, const uint32_t* bitmap,
                                               int lane) {
    if (bitmap) {
        const uint32_t w = __builtin_amdgcn_readfirstlane(bitmap[gi >> 5]);
        if ((w >> (gi & 31)) & 1u) return true;
    }
    if (rated_rowptr) {
        const int64_t lo = rated_rowptr[slot];
        const int64_t hi = rated_rowptr[slot + 1];
        for (int64_t base = lo; base < hi; base += 64) {
            const int64_t e = base + lane;
            const int v = e < hi ? rated_col[e] : CRH_PAD_IDX;
            if (__ballot(v == gi) != 0ull) return true;
            if (__ballot(v > gi) != 0ull) break;   // ascending: no later match
        }
    }
    return false;
}

// Same test with the user's list bounds [lo, hi) already at hand (score_topk keeps them in LDS): the
// bitmap word and the first probe of the rated list are requested together, so the common case costs ONE
// memory round trip instead of three dependent ones.  Lists longer than 64 entries are searched 64-ary:
// lane l probes the head of sub-block l, then the one sub-block that can hold gi is scanned.
__device__ __forceinline__ bool wave_is_masked_at(int gi, int64_t lo, int64_t hi, const int32_t* rated_col,
                                                  const uint32_t* bitmap, int lane) {
    const int64_t n = hi - lo;
    const int64_t step = (n + 63) >> 6;                 // sub-block length (1 when n <= 64)
    uint32_t w = 0;
    if (bitmap) w = bitmap[gi >> 5];
    int v = CRH_PAD_IDX;
    const int64_t e = lo + (int64_t)lane * step;
    if (n > 0 && e < hi) v = rated_col[e];
    if ((w >> (gi & 31)) & 1u) return true;            // w is wave-uniform (same address in every lane)
    if (n <= 0) return false;
    if (__ballot(v == gi) != 0ull) return true;
    if (step == 1) return false;
    const int j = __popcll(__ballot(v < gi)) - 1;      // last sub-block whose head is below gi
    if (j < 0) return false;
    const int64_t b0 = lo + (int64_t)j * step + 1;     // head already compared
    const int64_t b1 = (lo + (int64_t)(j + 1) * step) < hi ? (lo + (int64_t)(j + 1) * step) : hi;
    for (int64_t base = b0; base < b1; base += 64) {
        const int64_t q = base + lane;
        const int x = q < b1 ? rated_col[q] : CRH_PAD_IDX;
        if (__ballot(x == gi) != 0ull) return true;
        if (__ballot(x > gi) != 0ull) break;
    }
    return false;
}

// fast-path threshold for a list: its k-th score once full, but never above what a masked
// (-1e9) candidate could beat, so that skipping `raw <= tau` stays exact
__device__ __forceinline__ float wave_list_tau(const float* lsu, int n, int K) {
    if (n < K) return CRH_NEG_INF;
    const float t = lsu[K - 1];
    return t >= CRH_MASKED_SCORE ? t : CRH_NEG_INF;
}
