// Device helpers shared by the scoring kernels (score_topk.hip, score_topk_dma.hip): the launch record, MFMA operand
// helpers, the per-wave LDS lists and the slow path of the fused selection.  See score_topk.hip for the design.
#pragma once
#include <math.h>

#include "crh_common.h"
#include "topk_list.h"

namespace crh_score {

struct ScoreArgs {
    const void* user_emb;   // fp32 or fp16 (the kernel's element type)
    const int32_t* users;
    int64_t n_users;
    const void* item_emb;
    const void* packed;    // item tiles in MFMA-fragment order (pack_items_kernel), or NULL
    int64_t n_items;
    const int64_t* rated_rowptr;
    const int32_t* rated_col;
    const uint32_t* bitmap;
    int k;
    int64_t item_base;
    int n_splits;
    int64_t n_ugroups;
    float* out_score;   // [n_splits][n_users][k]
    int32_t* out_idx;
    int ablate;         // measurement only (CRH_SCORE_ABLATE, profile build; results invalid): 1 no selection, 2 every fetch hits one
                        // cached tile, 4 no workgroup barriers, 8 no fragment waits (DMA kernel) / no block stores (block
                        // launch), 16 no DMA wait
    unsigned long long* wave_clock;   // measurement only (CRH_SCORE_TIMING): per-wave clocks, see print_wave_timing / the DMA histogram
    unsigned* xcd_sync;     // [8 XCD][1 + n_windows] zeroed counters, or NULL: keeps the waves of an XCD within
    int sync_window;        // two windows of `sync_window` tiles of each other (see xcd_window_sync)
    int64_t sync_stride;    // counters per XCD
    float* dense;           // small catalogues: write the score tiles here (slot-major, `dense_stride` floats per
    int64_t dense_stride;   // user, a multiple of 32) instead of selecting; crh_mask_topk_f32 ranks the block
    int64_t user_base;      // first table row of the block when `users` is NULL
    // SEEDED lists (score_topk_seeded): every user's list starts as the finished top-k of an item PREFIX that was ranked
    // beforehand ([n_users][k], canonical order, padded with (-inf, PAD)), so the thresholds are tight from the first
    // tile and an item-range cut no longer repeats the top-k warm-up.  The seeds' ids lie below this launch's item_base.
    const float* seed_score;
    const int32_t* seed_idx;
    // DMA kernel: per 32-item tile of THIS launch's shard the candidate-bitmap bits of its items (tile_bits_kernel), or --
    // without a bitmap -- any readable memory of at least 4 bytes per tile (never looked at)
    const uint32_t* tile_bits;
};

// in : lane (i,0) holds k = 8q+0..3 of row i, lane (i,1) holds k = 8q+4..7
// out: .x = {8q | 8q+1}, .z = {8q+2 | 8q+3}, .y = {8q+4 | 8q+5}, .w = {8q+6 | 8q+7}  (low | high half)
// v_permlane32_swap vdst, src exchanges lanes 32-63 of vdst with lanes 0-31 of src.
// NOTE (hipcc 7.2): keep the swap on the integer vector.  A helper taking float& x, float& y and
// bit-casting the two scalar results separately is miscompiled (the second result is replaced
// by the first: v_mov y, x after the swap) -- found by reading the ISA, would fail parity.
__device__ __forceinline__ void chunk_swap(f32x4& cf) {
    u32x4 c = __builtin_bit_cast(u32x4, cf);
    const u32x2 r0 = __builtin_amdgcn_permlane32_swap(c.x, c.y, false, false);
    const u32x2 r1 = __builtin_amdgcn_permlane32_swap(c.z, c.w, false, false);
    u32x4 o;
    o.x = r0[0];
    o.y = r0[1];
    o.z = r1[0];
    o.w = r1[1];
    cf = __builtin_bit_cast(f32x4, o);
}

__device__ __forceinline__ f32x4 load16(const char* p) { return *reinterpret_cast<const f32x4*>(p); }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr bool kSwap = true;      // row-major rows need the k-pair swap (chunk_swap)
    template <int UW>
    static __device__ __forceinline__ void mma(f32x16 (&acc)[UW], const f32x4& c, const f32x4 (&b)[UW]) {
        // four k-pairs per 32-byte chunk, users interleaved so dependent MFMAs are one slot apart
#pragma unroll
        for (int u = 0; u < UW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, b[u].x, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, b[u].z, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, b[u].y, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, b[u].w, acc[u], 0, 0, 0);
    }
};
template <>
struct Elem<_Float16> {
    static constexpr bool kSwap = false;
    template <int UW>
    static __device__ __forceinline__ void mma(f32x16 (&acc)[UW], const f32x4& c, const f32x4 (&b)[UW]) {
        const f16x8 ca = __builtin_bit_cast(f16x8, c);
#pragma unroll
        for (int u = 0; u < UW; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca, __builtin_bit_cast(f16x8, b[u]), acc[u], 0, 0, 0);
    }
};

template <int UPW>
struct WaveLds {
    float* ls;       // [UPW][K]
    int* li;         // [UPW][K]
    int* cnt;        // [UPW]
    int64_t* rlo;    // [UPW] bounds of each user's rated list (rated_rowptr staged once per wave)
    int64_t* rhi;    // [UPW]
    unsigned* rfilter;   // [UPW][8] 256-bit membership filter of the user's rated items in this split's range
};

template <int UPW>
__host__ __device__ constexpr size_t wave_lds_bytes(int K) {
    return (size_t)UPW * K * 8 + (size_t)UPW * 4 + (size_t)UPW * 16 + (size_t)UPW * 32;
}

template <int UPW>
__device__ __forceinline__ void wave_lds_carve(WaveLds<UPW>& w, char* base, int K) {
    w.ls = reinterpret_cast<float*>(base);
    w.li = reinterpret_cast<int*>(w.ls + UPW * K);
    w.cnt = w.li + UPW * K;
    w.rlo = reinterpret_cast<int64_t*>(w.cnt + UPW);
    w.rhi = w.rlo + UPW;
    w.rfilter = reinterpret_cast<unsigned*>(w.rhi + UPW);
}

__device__ __forceinline__ unsigned rated_hash(int gi) { return ((unsigned)gi * 2654435761u) >> 24; }

// Stage the list bounds of this wave's UPW slots and build their membership filters.  A slow-path candidate is tested
// against the filter first (one LDS word): only a hit (a rated item, or a false positive: 1 - exp(-len/256)) pays the
// memory round trip of the exact search.  The lists of consecutive slots are one contiguous run of rated_col, read
// coalesced; ids outside [id0, id1) (other splits, other shards) are left out.
template <int UPW>
__device__ __forceinline__ void wave_lds_init(const WaveLds<UPW>& w, const ScoreArgs& a, int64_t slot0, int id0, int id1,
                                              bool filter, int lane) {
    for (int j = lane; j < UPW; j += 64) {
        w.cnt[j] = 0;
        const int64_t slot = slot0 + j;
        const bool has = a.rated_rowptr && slot < a.n_users;
        w.rlo[j] = has ? a.rated_rowptr[slot] : 0;
        w.rhi[j] = has ? a.rated_rowptr[slot + 1] : 0;
    }
    if (a.seed_score) {
        // seeded lists: copy each slot's prefix top-k into LDS; its fill = the entries before the padding
        const int K = a.k;
        for (int j = 0; j < UPW; ++j) {
            const int64_t slot = slot0 + j;
            if (slot >= a.n_users) break;                         // wave-uniform
            int n = 0;
            for (int e0 = 0; e0 < K; e0 += 64) {
                const int e = e0 + lane;
                int gi = CRH_PAD_IDX;
                if (e < K) {
                    gi = a.seed_idx[slot * K + e];
                    w.ls[j * K + e] = a.seed_score[slot * K + e];
                    w.li[j * K + e] = gi;
                }
                n += __popcll(__ballot(gi != CRH_PAD_IDX));
            }
            if (lane == 0) w.cnt[j] = n;
        }
    }
    if (!filter || !a.rated_rowptr) return;
    for (int j = lane; j < UPW * 8; j += 64) w.rfilter[j] = 0u;
    const int64_t s1 = slot0 + UPW < a.n_users ? slot0 + UPW : a.n_users;
    const int64_t e0 = a.rated_rowptr[slot0], e1 = a.rated_rowptr[s1];
    int u = 0;                                   // this lane's position in the slot sequence (entries ascend)
    for (int64_t base = e0; base < e1; base += 64) {
        const int64_t e = base + lane;
        if (e < e1) {
            const int v = a.rated_col[e];
            while (e >= w.rhi[u]) ++u;           // rhi of the last real slot is e1: terminates
            if (v >= id0 && v < id1) {
                const unsigned hsh = rated_hash(v);
                atomicOr(&w.rfilter[u * 8 + (hsh >> 5)], 1u << (hsh & 31));
            }
        }
    }
}

// maximum of the 16 accumulator registers as a depth-3 tree of 3-input maxima (v_max3_f32): the wave waits
// for this chain between two tiles, and a sequential chain of 15 is ~4x longer
// Written as the eight instructions it is (7 v_max3_f32 + 1 v_max_f32): from fmaxf hipcc first canonicalises accumulator
// registers it cannot prove quiet (v_max_f32 x, x: two more instructions per call), and beside v_mfma_f32_32x32x2_f32 every VALU
// instruction costs its full issue time.  NaNs are ignored by both forms alike.
__device__ __forceinline__ float max16(const f32x16& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    float a0, a1, a2, a3, a4, b0, b1, r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(v[3]), "v"(v[4]), "v"(v[5]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a2) : "v"(v[6]), "v"(v[7]), "v"(v[8]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a3) : "v"(v[9]), "v"(v[10]), "v"(v[11]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a4) : "v"(v[12]), "v"(v[13]), "v"(v[14]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(b0) : "v"(a0), "v"(a1), "v"(a2));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(b1) : "v"(a3), "v"(a4), "v"(v[15]));
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(b0), "v"(b1));
    return r;
#else
    const float a0 = fmaxf(fmaxf(v[0], v[1]), v[2]), a1 = fmaxf(fmaxf(v[3], v[4]), v[5]);
    const float a2 = fmaxf(fmaxf(v[6], v[7]), v[8]), a3 = fmaxf(fmaxf(v[9], v[10]), v[11]);
    const float a4 = fmaxf(fmaxf(v[12], v[13]), v[14]);
    const float b0 = fmaxf(fmaxf(a0, a1), a2), b1 = fmaxf(fmaxf(a3, a4), v[15]);
    return fmaxf(b0, b1);
#endif
}

// bit r = (v[r] > t), r = 0 .. 15: one compare and one add-with-carry per register (m = 2 m + vcc, highest row first)
// instead of compare + select + shift/or
__device__ __forceinline__ unsigned gt_mask16(const f32x16& v, float t) {
    unsigned m = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int r = 15; r >= 0; --r)
        asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(v[r]), "v"(t) : "vcc");
#endif
    return m;
}

// v[r] for a WAVE-UNIFORM r: hipcc indexes the register tuple through the GPR index mode (s_set_gpr_idx_on / v_mov /
// s_set_gpr_idx_off) -- one VALU instruction where the select tree below took fifteen
__device__ __forceinline__ float pick16u(const f32x16& v, int r) { return v[__builtin_amdgcn_readfirstlane(r)]; }

__device__ __forceinline__ float pick16(const f32x16& v, int r) {
    const bool b0 = r & 1, b1 = r & 2, b2 = r & 4, b3 = r & 8;
    const float p0 = b0 ? v[1] : v[0], p1 = b0 ? v[3] : v[2], p2 = b0 ? v[5] : v[4], p3 = b0 ? v[7] : v[6];
    const float p4 = b0 ? v[9] : v[8], p5 = b0 ? v[11] : v[10], p6 = b0 ? v[13] : v[12], p7 = b0 ? v[15] : v[14];
    const float q0 = b1 ? p1 : p0, q1 = b1 ? p3 : p2, q2 = b1 ? p5 : p4, q3 = b1 ? p7 : p6;
    const float s0 = b2 ? q1 : q0, s1 = b2 ? q3 : q2;
    return b3 ? s1 : s0;
}

// Slow path for one 32x32 accumulator tile (rare).  acc[r] = score of item row
// (r&3) + 8*(r>>2) + 4*(lane>>5) for user column lane&31.
// Cost per event is what bounds mid-size catalogues and the fp16 kernel (each event stalls the wave, in the workgroup
// kernel the whole CU), so memory round trips are kept off it: the candidate-bitmap bits of the tile's 32 items are ONE
// wave-uniform 64-bit window requested on entry (it lands while the candidate masks are built), bitmap-masked
// candidates of users whose list is full are dropped before the serial part, and the rated list is only searched when
// the user's LDS filter says the item may be in it.
template <int UPW>
__device__ __forceinline__ void tile_slow_path(const f32x16& acc, float& tau_reg, const WaveLds<UPW>& w,
                                               int K, int ucol0, int64_t slot0, const ScoreArgs& a,
                                               int64_t item0, int64_t split_end, int lane, bool rows_zeroed = false) {
    const int64_t g0 = a.item_base + item0;          // global id of the tile's first item
    unsigned blo = 0u, bhi = 0u;
    // rows_zeroed: the tiles come from the packed copy, whose bitmap-masked rows are zero (pack_row_masked): such an item
    // scores exactly 0 and is a candidate only of a user whose threshold is negative.  No such user in this group of 32 =
    // no masked candidate: the tile's bitmap window (a memory round trip on the event's critical path) is not needed.
    if (a.bitmap && !(rows_zeroed && __ballot(tau_reg < 0.0f) == 0ull)) {
        const int64_t last = (a.item_base + a.n_items - 1) >> 5;
        const int64_t w0 = (g0 >> 5) < last ? (g0 >> 5) : last, w1 = w0 < last ? w0 + 1 : last;
        blo = a.bitmap[w0];
        bhi = a.bitmap[w1];
        __builtin_amdgcn_s_waitcnt(0x0f70);   // landed HERE, inside the branch: a wait behind the join would be paid by every event (see below)
    }
    unsigned cm = gt_mask16(acc, tau_reg);
    // bits of this lane's 16 rows: rows (r&3) + 8*(r>>2) + 4*hh <-> bit r.  (No bitmap word was fetched = no masked row can be a
    // candidate here: the bit shuffling is skipped with it -- a dozen VALU instructions per event.)
    unsigned m16 = 0u;
    if ((blo | bhi) != 0u) {          // wave-uniform
        const unsigned tb = (unsigned)(((((unsigned long long)bhi) << 32) | blo) >> (g0 & 31));
        const unsigned x = tb >> (4 * (lane >> 5));
        m16 = (x & 0xFu) | ((x >> 4) & 0xF0u) | ((x >> 8) & 0xF00u) | ((x >> 12) & 0xF000u);
        // a masked candidate scores -1e9: it can only enter a list that is not full (tau = -inf)
        if (tau_reg > CRH_NEG_INF) cm &= ~m16;
    }
    const unsigned bm = cm & m16;
    unsigned long long lanes = __ballot(cm != 0u);
    const bool wide = K > 64;                         // lane t holds entries t and t + 64 of a list (k <= 128)
    while (lanes) {
        const int L = __builtin_ctzll(lanes);
        lanes &= lanes - 1;
        unsigned cmL = __builtin_amdgcn_readlane(cm, L);
        const unsigned bmL = __builtin_amdgcn_readlane(bm, L);
        const int jl = L & 31, hh = L >> 5;
        const int64_t slot = slot0 + jl;
        if (slot >= a.n_users) continue;
        const int ul = ucol0 + jl;
        float* lsu = w.ls + ul * K;
        int* liu = w.li + ul * K;
        while (cmL) {
            const int r = __builtin_ctz(cmL);
            cmL &= cmL - 1;
            const int64_t il = item0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (il >= split_end) continue;   // clamped duplicate rows of the tail tile
            // r is wave-uniform: pick the accumulator register with a select tree (static indices only),
            // then read lane L
            float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pick16u(acc, r)), L));
            // the user's threshold may have moved since the candidate masks were built (an earlier candidate of this event).
            // Strictly below it = out; a TIE is decided by the ids further down: within a tile the candidates of a user do not
            // come in id order (rows 0-3, 8-11, ... of one half-wave, then 4-7, 12-15, ... of the other)
            if (sc < __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tau_reg), jl))) continue;
            const int gi = (int)(a.item_base + il);
            // ONE batch of LDS reads per candidate: fill, filter word and the whole list (entries past the fill are stale
            // and overridden below); the insert position, the shifted entries and the user's new threshold all come out of
            // these registers
            const unsigned hsh = rated_hash(gi);
            const int n_raw = w.cnt[ul];
            const unsigned fw_raw = a.rated_rowptr ? w.rfilter[ul * 8 + (hsh >> 5)] : 0u;
            float es = CRH_NEG_INF, es2 = CRH_NEG_INF;
            int ei = CRH_PAD_IDX, ei2 = CRH_PAD_IDX;
            if (lane < K) {
                es = lsu[lane];
                ei = liu[lane];
            }
            if (wide && lane + 64 < K) {
                es2 = lsu[lane + 64];
                ei2 = liu[lane + 64];
            }
            const int n = __builtin_amdgcn_readfirstlane(n_raw);
            if (n >= K) {   // full list: the candidate must beat its last entry under the canonical key
                const int q1 = K - 1;
                const float ks = __builtin_bit_cast(float, q1 < 64 ? __builtin_amdgcn_readlane(__builtin_bit_cast(int, es), q1)
                                                                   : __builtin_amdgcn_readlane(__builtin_bit_cast(int, es2), q1 - 64));
                const int ki = q1 < 64 ? __builtin_amdgcn_readlane(ei, q1) : __builtin_amdgcn_readlane(ei2, q1 - 64);
                if (!crh_better(fmaxf(sc, CRH_MASKED_SCORE), gi, ks, ki)) continue;
            }
            bool masked = (bmL >> r) & 1u;
            if (!masked && ((__builtin_amdgcn_readfirstlane(fw_raw) >> (hsh & 31)) & 1u)) {
                masked = wave_is_masked_at(gi, w.rlo[ul], w.rhi[ul], a.rated_col, nullptr, lane);
                // every load of the search has landed before the loop goes round: otherwise hipcc, seeing a load that may still
                // be pending on some exit, guards the next candidate's registers with s_waitcnt vmcnt(0) -- and vmcnt counts in
                // order, so EVERY event then waited for the next tile's whole prefetch (16 loads issued a moment ago): ~2 of an
                // event's ~3 us
                __builtin_amdgcn_s_waitcnt(0x0f70);
            }
            if (masked) sc = CRH_MASKED_SCORE;
            int p = __popcll(__ballot(lane < n && crh_better(es, ei, sc, gi)));
            if (wide) p += __popcll(__ballot(lane + 64 < n && crh_better(es2, ei2, sc, gi)));
            if (p < K) {
                // every shifted entry is in registers, so the stores cannot overtake a load
                if (lane >= p && lane < n && lane + 1 < K) {
                    lsu[lane + 1] = es;
                    liu[lane + 1] = ei;
                }
                if (wide && lane + 64 >= p && lane + 64 < n && lane + 65 < K) {
                    lsu[lane + 65] = es2;
                    liu[lane + 65] = ei2;
                }
                const int n2 = n < K ? n + 1 : K;
                if (lane == 0) {
                    lsu[p] = sc;
                    liu[p] = gi;
                    w.cnt[ul] = n2;
                }
                if (n2 >= K) {   // full: the k-th entry is now the old (k-1)-th, or the candidate itself
                    const int q = K >= 2 ? K - 2 : 0;
                    const float prev = __builtin_bit_cast(float, q < 64 ? __builtin_amdgcn_readlane(__builtin_bit_cast(int, es), q)
                                                                        : __builtin_amdgcn_readlane(__builtin_bit_cast(int, es2), q - 64));
                    const float kth = (p == K - 1) ? sc : prev;
                    // never above what a masked (-1e9) candidate could beat (wave_list_tau)
                    if ((lane & 31) == jl) tau_reg = kth >= CRH_MASKED_SCORE ? kth : CRH_NEG_INF;
                }
            }
        }
    }
    // (no read-back of the thresholds: every insert updated its user's lanes; padding columns keep their +inf)
}

// Soft lockstep of the waves of one XCD.  Every wave streams the same item tiles, but left alone the waves
// drift apart (the older wave of a SIMD pair gets ~80 % of the matrix pipe) until their working set no
// longer fits the XCD's 4 MiB L2: the L2 hit rate drops to ~50 % and every wave's stream goes out to the
// fabric (5.5 TB per launch at S-EVAL against a 5 GB table; the fp16 build is bound by exactly this).
// Protocol: a wave adds itself to done[w] when it finishes window w (= sync_window tiles) and does not start
// window w+2 before every REGISTERED wave of its XCD finished window w.  Only resident waves register, the
// wait is bounded, and a wave that times out stops synchronising, so this can slow a launch down but never
// hang or change its result.  All counters of an XCD are touched by that XCD only (one L2: coherent).
__device__ __forceinline__ bool xcd_window_sync(unsigned* cnt, int64_t win, int lane) {
    bool ok = true;
    if (lane == 0) {
        if (win >= 1) __hip_atomic_fetch_add(cnt + 1 + (win - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (win >= 2) {
            int spins = 0;
            while (__hip_atomic_load(cnt + 1 + (win - 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
                   __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __builtin_amdgcn_s_sleep(32);
                if (++spins > 200000) { ok = false; break; }      // ~0.2 s: give up, run free
            }
        }
    }
    return __builtin_amdgcn_readfirstlane((int)ok) != 0;
}

constexpr int WG_RING = 3;   // item-tile slots in LDS (workgroup kernels)

// score_topk_dma.hip (built with -mllvm -amdgpu-mfma-vgpr-form): the LDS-DMA workgroup kernel for 512- and 256-byte rows
// (mode = CRH_SCORE_DMA: 1 default, 3 = barrier form for fp32 too)
__attribute__((visibility("hidden"))) int launch_score_dma(int esz, int d, int mode, const ScoreArgs& a, hipStream_t stream);
__attribute__((visibility("hidden"))) size_t score_dma_lds_bytes(int row_bytes, int k, int ring_slots, bool flags);
__attribute__((visibility("hidden"))) int score_dma_ring_slots(int esz, int d, int k, int mode);

}  // namespace crh_score
