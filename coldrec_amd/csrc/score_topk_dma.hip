// LDS-DMA form of the fused scoring + top-k workgroup kernel (see the comment on the kernel).  Its own translation unit
// because it is built with -mllvm -amdgpu-mfma-vgpr-form (Makefile): one wave per SIMD owns all 512 registers, the
// accumulators must stay in VGPRs (the threshold test reads them) while the user fragments fill the 256 AGPRs; without
// the flag hipcc puts a one-wave-per-SIMD kernel's accumulators into AGPRs and copies half of the user fragments around.
#include <type_traits>
#include <utility>

#include "score_topk_common.h"

namespace crh_score {
namespace {

constexpr int DMA_NW = 4;           // waves per workgroup (one per SIMD)
constexpr int DMA_UPW = 128;        // users per wave
constexpr int TBITS_B = 2 * 64 * 4; // two blocks of 64 tiles' candidate bits
constexpr int FLAGS_B = 64;         // flag form: ready[<= 8] and done[<= 8] counters of the ring slots

// per-wave LDS of the DMA kernel: the lists (scores, ids, fill) and a 192-bit membership filter of each user's rated list.
// The rated-list bounds are read from memory when a candidate's filter says "maybe rated" (rare).  k = 20: 188 bytes per user,
// 94 KiB per workgroup beside the 64 KiB ring.
constexpr int DMA_FW = 6;           // filter words per user
__host__ __device__ constexpr size_t dma_wave_lds_bytes(int K) {
    return (size_t)DMA_UPW * K * 8 + (size_t)DMA_UPW * 4 + (size_t)DMA_UPW * DMA_FW * 4;
}
__device__ __forceinline__ unsigned rated_hash192(int gi) { return ((((unsigned)gi * 2654435761u) >> 16) * (32u * DMA_FW)) >> 16; }

// The 4 UW MFMAs of one fp32 chunk, as Elem<float>::mma (k pairs in the order x, z, y, w, users interleaved), with
// hook(integral_constant<BASE + i>) run right behind MFMA i: the flag form hangs its few non-MFMA instructions (polls, bumps,
// DMA issue) into single MFMA gaps this way -- an fp32 MFMA runs 64 cycles, a gap hides about a dozen scalar / LDS / VMEM
// instructions, and a block of them in FRONT of a group's MFMAs is paid in full (the first flag form did that: -2 % at
// d=128, -4 % at d=64 on 10 M-item streams against the barrier form).
template <int K>
__device__ __forceinline__ float f4_comp(const f32x4& v) {
    if constexpr (K == 0) return v.x;
    else if constexpr (K == 1) return v.z;
    else if constexpr (K == 2) return v.y;
    else return v.w;
}
template <int UW, int BASE, typename H>
__device__ __forceinline__ void mma_f32_hooked(f32x16 (&acc)[UW], const f32x4& c, const f32x4 (&b)[UW], H&& hook) {
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
        ((acc[I % UW] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_comp<I / UW>(c), f4_comp<I / UW>(b[I % UW]), acc[I % UW], 0, 0, 0),
          hook(std::integral_constant<int, BASE + I>{})),
         ...);
    }(std::make_integer_sequence<int, 4 * UW>{});
}

// Slow path of one 32x32 accumulator tile, as tile_slow_path (score_topk_common.h) with every memory round trip of the
// common case removed: the candidate-bitmap bits of the tile come out of LDS (tb: the tile's 32 bits, fetched by DMA with
// the tile stream, see tile_bits_kernel), the membership filter out of LDS too; only a filter hit goes to memory (list
// bounds + search).  While one wave is in here the other three wait at the next barrier, and a memory wait in here would
// also wait for the tile DMAs in flight.
// The insert hands back the user's new threshold out of the registers it holds (+1.0 % against reading it back, same box).
// (Measured and not kept, round 5: one LDS round trip per candidate -- the whole list, its fill and the filter word
// requested together, the insert and the user's new threshold computed from those registers -- ran 2.7 % SLOWER on the
// same box, 0.5316 vs 0.5457 of 2.5 PF: most events end at the "cannot enter" test after three scalar reads.)
// ONE_TRIP (fp32): the candidate's list rides the first batch of LDS reads (fill, list, filter word: one wait) and the insert works
// out of those registers -- at fp32 a candidate above its user's threshold does enter (no rounding ties), so the second round trip
// of the two-step form (fill and list read again by the insert) is paid by nearly every event; at fp16 most events end at the
// "cannot enter" test and the lean first batch wins (the round-5 measurement above).
#ifdef CRH_DMA_TWO_TRIP                      // (variant build: the two-step form for fp32 too, tools/ab_lib.sh)
constexpr bool DMA_ONE_TRIP_F32 = false;
#else
constexpr bool DMA_ONE_TRIP_F32 = true;
#endif
template <bool ONE_TRIP>
__device__ __forceinline__ void tile_slow_path_dma(const f32x16& acc, float& tau_reg, float* ls, int* li, int* cnt, int K,
                                                   int ucol0, int64_t slot0, const ScoreArgs& a, int64_t item0,
                                                   int64_t split_end, int lane, unsigned tb, const unsigned* rfilter) {
    unsigned cm = gt_mask16(acc, tau_reg);
    // bits of this lane's 16 rows: rows (r&3) + 8*(r>>2) + 4*hh <-> bit r.  Masked rows are packed as zeros: with no negative
    // threshold in the group none of them is a candidate, and the bit shuffling (a dozen VALU instructions) is skipped
    unsigned m16 = 0u;
    if (tb != 0u && __ballot(tau_reg < 0.0f) != 0ull) {          // wave-uniform
        const unsigned x = tb >> (4 * (lane >> 5));
        m16 = (x & 0xFu) | ((x >> 4) & 0xF0u) | ((x >> 8) & 0xF00u) | ((x >> 12) & 0xF000u);
        // a masked candidate scores -1e9: it can only enter a list that is not full (tau = -inf)
        if (tau_reg > CRH_NEG_INF) cm &= ~m16;
    }
    const unsigned bm = cm & m16;
    unsigned long long lanes = __ballot(cm != 0u);
    while (lanes) {
        const int L = __builtin_ctzll(lanes);
        lanes &= lanes - 1;
        unsigned cmL = __builtin_amdgcn_readlane(cm, L);
        const unsigned bmL = __builtin_amdgcn_readlane(bm, L);
        const int jl = L & 31, hh = L >> 5;
        const int64_t slot = slot0 + jl;
        if (slot >= a.n_users) continue;
        const int ul = ucol0 + jl;
        float* lsu = ls + ul * K;
        int* liu = li + ul * K;
        while (cmL) {
            const int r = __builtin_ctz(cmL);
            cmL &= cmL - 1;
            const int64_t il = item0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (il >= split_end) continue;   // clamped duplicate rows of the tail tile
            float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pick16u(acc, r)), L));
            const int gi = (int)(a.item_base + il);
            // one batch of LDS reads: fill, tail entry, filter word
            const unsigned hsh = rated_hash192(gi);
            const int n_raw = cnt[ul];
            const int le = ONE_TRIP ? (lane < K ? lane : K - 1) : K - 1;   // (ONE_TRIP: lane l reads entry l; the others the tail entry)
            const float ks_raw = lsu[le];
            const int ki_raw = liu[le];
            const unsigned fw_raw = a.rated_rowptr ? rfilter[ul * DMA_FW + (hsh >> 5)] : 0u;
            const int n = __builtin_amdgcn_readfirstlane(n_raw);
            if (n >= K) {
                const float ks = ONE_TRIP ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ks_raw), K - 1))
                                          : __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ks_raw)));
                const int ki = ONE_TRIP ? __builtin_amdgcn_readlane(ki_raw, K - 1) : __builtin_amdgcn_readfirstlane(ki_raw);
                if (!crh_better(fmaxf(sc, CRH_MASKED_SCORE), gi, ks, ki)) continue;   // cannot enter a full list
            }
            bool masked = (bmL >> r) & 1u;
            if (!masked && ((__builtin_amdgcn_readfirstlane(fw_raw) >> (hsh & 31)) & 1u)) {
                const int64_t lo = a.rated_rowptr[slot], hi = a.rated_rowptr[slot + 1];
                masked = wave_is_masked_at(gi, lo, hi, a.rated_col, nullptr, lane);
                // every load of the search has landed HERE, inside the rare branch: left pending on this exit, hipcc guards the
                // join with s_waitcnt vmcnt(0) -- paid by EVERY event, and vmcnt counts the tile DMAs in flight too: the
                // flag form issued them one group (0.8 us) earlier, the barrier form at 256-byte rows in this very group, so an
                // event waited for a DMA round trip (round 6, found in the ISA; tile_slow_path has had the same fix since round 4)
                __builtin_amdgcn_s_waitcnt(0x0f70);
            }
            if (masked) sc = CRH_MASKED_SCORE;
            // wave_list_insert (k <= 64 here) that also hands back the user's NEW threshold out of the registers it already
            // holds -- the k-th entry after the insert is the old (k-1)-th or the candidate -- instead of reading it back
            {
                float es = CRH_NEG_INF;
                int ei = CRH_PAD_IDX;
                if constexpr (ONE_TRIP) {
                    if (lane < n) {
                        es = ks_raw;
                        ei = ki_raw;
                    }
                } else {
                    if (lane < n) {
                        es = lsu[lane];
                        ei = liu[lane];
                    }
                }
                const int p = __popcll(__ballot(lane < n && crh_better(es, ei, sc, gi)));
                if (p < K) {
                    if (lane >= p && lane < n && lane + 1 < K) {
                        lsu[lane + 1] = es;
                        liu[lane + 1] = ei;
                    }
                    const int n2 = n < K ? n + 1 : K;
                    if (lane == 0) {
                        lsu[p] = sc;
                        liu[p] = gi;
                        cnt[ul] = n2;
                    }
                    if (n2 >= K) {
                        const float prev = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, es), K >= 2 ? K - 2 : 0));
                        const float kth = (p == K - 1) ? sc : prev;
                        if ((lane & 31) == jl) tau_reg = kth >= CRH_MASKED_SCORE ? kth : CRH_NEG_INF;
                    }
                }
            }
        }
    }
    // (no read-back of the thresholds: every insert above updated its user's lanes; padding columns keep their +inf)
}


// ---------------------------------------------------------------------------------------------------------
// LDS-DMA form of the workgroup kernel for 512-byte rows (fp16 d=256: configs[4]; fp32 d=128: the headline).
// What changes against score_topk_wg_kernel, and why (profiles/r04_z_f16_ceiling.json: two thirds of the matrix
// pipe's cycles were MFMA, the rest ring commits, the per-tile barrier and the threshold test at the end of a tile):
//   * the item stream reaches LDS by DMA (global_load_lds, 16 B per lane, 1 KiB per instruction): no staging
//     registers, no ds_write pass; FOUR ring slots, the fill of a tile is issued two tiles before the barrier that
//     publishes it (one tile of slack measured 7 % slower: the first workgroup of an XCD to ask for a tile waits for the
//     fabric, and the lockstep makes everybody wait with it);
//   * FOUR waves (one per SIMD) of 128 users each instead of eight of 64: an A fragment read from LDS feeds four
//     MFMAs instead of two (half the LDS traffic per flop; the chip is power-limited on fp16 operands).  The B
//     fragments of 128 users are 256 registers: they live in the ACCUMULATOR half of the unified register file
//     (an empty asm with an "a" constraint pins them there; gfx950 MFMAs take A/B operands from AGPRs), the two
//     accumulator SETS, the A fragments and everything else in the 256 VGPRs;
//   * two accumulator sets: the threshold test of tile j-1 (and, rarely, its slow path) is issued between the
//     MFMAs of tile j, so the pipe no longer drains at every tile boundary;
//   * ONE workgroup barrier per tile, in the MIDDLE of the tile's MFMA stream: it publishes tile j+1 (every wave
//     waited for its own DMA pieces first) and frees the slot of tile j-1 for the DMA of tile j+3.  A fragment
//     prefetches run across the tile boundary, so no LDS latency is exposed anywhere in the steady state;
//   * the slow path touches no memory in the common case (tile_slow_path_dma): the tiles' candidate bits ride the DMA
//     stream (64 tiles per 256-byte piece); the rated-list bounds left the LDS and the membership filters shrank to 192
//     bits, which makes room for the fourth ring slot at k <= 20.
// Users, thresholds and lists are per wave exactly as in the other kernels: results are identical.
// The loop runs n_steps + 1 bodies: body j multiplies tile j (the last one a clamped duplicate nobody selects) and
// selects tile j - 1.
//
// FL = the FLAG form (fp32, round 6).  The barrier form above gives the four waves no slack at all: a slow-path event of one
// wave (~0.4 us) is paid by the whole workgroup at the next barrier, and events are what separates the kernel from its bare
// MFMA stream (tools/probes/dma_stream_probe_f32.hip: stream 0.943 / 0.963 of the fp32 peak at d = 64 / 128, kernel 0.884 /
// 0.925; with the barriers ablated the d=64 kernel gets 2.5 of its 3.5 points of event cost back).  Here the ring is
// guarded by two monotonic LDS counters per slot instead:
//   ready[s]  += 1 by every wave when ITS pieces of the tile in slot s have landed (it waits for its own DMA of the body
//                before, a whole tile ago, just before it issues the next one at the head of a body);
//   done[s]   += 1 by every wave when its last fragment read of the tile in slot s has returned.
// A wave reads tile t only once ready = 4 (t / R + 1) and refills the slot of tile t - R only once done = 4 (t / R): all
// dependencies point at earlier tiles, so there is no cycle; the polls are one ds_read_b32 issued a group ahead (it rides
// the counted fragment waits) + v_readfirstlane + a scalar compare, and spin only when a partner really is behind.  With
// R = 4 slots and a prefetch distance of 2 tiles a wave may run (NG - 2) / NG of a tile ahead of the slowest reader and a
// whole tile ahead of the slowest refiller: several events' worth at fp32 tile times (3.4 / 6.8 us).  fp16 tiles last
// 0.85 us -- less than a DMA round trip -- so the fp16 kernel keeps the barrier form.
template <typename T, int D, int R, bool FL>
__global__ __launch_bounds__(256, 1) void score_topk_dma_kernel(ScoreArgs a) {
    constexpr int NW = DMA_NW, UW = 4, UPW = DMA_UPW;
    // body j issues the DMA of tile j + PF; in the flag form it publishes its pieces of tile j + PF - 1 (issued one body ago) first:
    // a wave may lead the slowest reader by (NG - 2) / NG of a tile and the slowest refiller by one tile
    constexpr int PF = FL ? 2 : 3;
    static_assert(R == 4 && (!FL || sizeof(T) == 4), "ring shape");
    constexpr int ROWB = D * (int)sizeof(T);
    constexpr int NCH = ROWB / 32;
    constexpr int TILE_B = NCH * 1024;
    constexpr int CPW = NCH / NW;                      // 1 KiB DMA pieces of a tile each wave issues
    constexpr int GR = 2, NG = NCH / GR;               // A fragments are read in groups of GR chunks, one group ahead
    constexpr int BG = NG / 2 - 1;                     // the group in front of which the ring barrier sits
    static_assert(NCH % NW == 0 && NG % 4 == 0 && BG + 2 < NG, "row width not supported by the DMA kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int S = a.n_splits;
    const int split = (int)(blockIdx.x % S);
    const int64_t ug_raw = (int64_t)(blockIdx.x / S) * NW + wave;
    const bool live = ug_raw < a.n_ugroups;            // a dead wave still fetches and synchronises
    const int64_t ug = live ? ug_raw : a.n_ugroups - 1;
    const int K = a.k;
    const int i = lane & 31, h = lane >> 5;

    char* ring = smem;                                  // [R][TILE_B]
    unsigned* tbits = reinterpret_cast<unsigned*>(smem + R * TILE_B);   // [2][64]
    unsigned* flags = tbits + TBITS_B / 4;             // FL: ready[R] at 0, done[R] at 8; slots of the prologue's tiles start ready
    if constexpr (FL) {
        // (published by the prologue's barrier.  Tiles 0 .. PF-2 of the prologue start complete; tile PF-1 is bumped by body 0 like
        // every tile after it: the prologue waited for all of them)
        if (threadIdx.x < 16) flags[threadIdx.x] = threadIdx.x < PF - 1 ? (unsigned)NW : 0u;
    }
    const int64_t NT = (a.n_items + 31) >> 5;
    const int64_t t0 = NT * split / S, t1 = NT * (split + 1) / S;
    const int64_t split_end = (t1 << 5) < a.n_items ? (t1 << 5) : a.n_items;
    const int64_t slot0w = ug * UPW;

    // ---- this wave's lists
    char* wl = smem + R * TILE_B + TBITS_B + (FL ? FLAGS_B : 0) + (size_t)wave * dma_wave_lds_bytes(K);
    float* ls = reinterpret_cast<float*>(wl);           // [UPW][K]
    int* li = reinterpret_cast<int*>(ls + UPW * K);      // [UPW][K]
    int* cnt = li + UPW * K;                            // [UPW]
    for (int j = lane; j < UPW; j += 64) cnt[j] = 0;
    if (a.seed_score) {
        // seeded lists: copy each slot's prefix top-k into LDS; its fill = the entries before the padding
        for (int j = 0; j < UPW; ++j) {
            const int64_t slot = slot0w + j;
            if (slot >= a.n_users) break;                         // wave-uniform
            int n = 0;
            for (int e0 = 0; e0 < K; e0 += 64) {
                const int e = e0 + lane;
                int gi = CRH_PAD_IDX;
                if (e < K) {
                    gi = a.seed_idx[slot * K + e];
                    ls[j * K + e] = a.seed_score[slot * K + e];
                    li[j * K + e] = gi;
                }
                n += __popcll(__ballot(gi != CRH_PAD_IDX));
            }
            if (lane == 0) cnt[j] = n;
        }
    }
    // ---- membership filters of the rated lists (192 bits per user, ids of this split's range only)
    unsigned* rfilter = reinterpret_cast<unsigned*>(cnt + UPW);      // [UPW][DMA_FW]
    for (int j = lane; j < UPW * DMA_FW; j += 64) rfilter[j] = 0u;
    if (live && a.rated_rowptr) {
        const int id0 = (int)(a.item_base + (t0 << 5)), id1 = (int)(a.item_base + split_end);
        const int64_t s1 = slot0w + UPW < a.n_users ? slot0w + UPW : a.n_users;
        const int64_t e0 = a.rated_rowptr[slot0w], e1 = a.rated_rowptr[s1];
        int64_t nxt = a.rated_rowptr[slot0w + 1];   // end of the list of this lane's current user
        int u = 0;
        for (int64_t base = e0; base < e1; base += 64) {
            const int64_t e = base + lane;
            if (e < e1) {
                const int v = a.rated_col[e];
                while (e >= nxt) { ++u; nxt = a.rated_rowptr[slot0w + u + 1]; }      // entries ascend: terminates at e1
                if (v >= id0 && v < id1) {
                    const unsigned hsh = rated_hash192(v);
                    atomicOr(&rfilter[u * DMA_FW + (hsh >> 5)], 1u << (hsh & 31));
                }
            }
        }
    }

    f32x4 b[NCH][UW];
    float tau[UW];
#pragma unroll
    for (int u = 0; u < UW; ++u) {
        int64_t slot = slot0w + 32 * u + i;
        tau[u] = slot < a.n_users ? CRH_NEG_INF : __builtin_inff();      // padding columns never take a candidate
        if (a.seed_score && slot < a.n_users) tau[u] = wave_list_tau(ls + (32 * u + i) * K, cnt[32 * u + i], K);
        if (slot >= a.n_users) slot = a.n_users - 1;
        const int64_t row = a.users ? (int64_t)a.users[slot] : a.user_base + slot;
        const char* up = reinterpret_cast<const char*>(a.user_emb) + row * ROWB + 16 * h;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            b[q][u] = load16(up + 32 * q);
            if constexpr (Elem<T>::kSwap) chunk_swap(b[q][u]);
        }
    }
    // the user fragments live in AGPRs from here on (and have landed: nothing of theirs is left on the VM counter)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            if constexpr (sizeof(T) == 4) {     // fp32 MFMAs take ONE dword per operand: pinned as a 4-vector hipcc copies every
                asm volatile("" : "+a"(b[q][u].x));   // component to a VGPR first (v_accvgpr_read per MFMA, 23 spills); as four
                asm volatile("" : "+a"(b[q][u].y));   // scalars the MFMA reads a[N] directly
                asm volatile("" : "+a"(b[q][u].z));
                asm volatile("" : "+a"(b[q][u].w));
            } else {
                asm volatile("" : "+a"(b[q][u]));
            }
        }
#endif

    const char* packed = reinterpret_cast<const char*>(a.packed);
    const int n_steps = (int)(t1 - t0);                 // tiles of this split (32-bit loop arithmetic: scalar compares, no VALU)
    // this wave's pieces of step j -> ring slot.  Steps past the split (the extra body, the prefetch distance) read clamped
    // tiles: valid addresses, data nobody selects.
    // One global pointer per lane and ONE LDS base per slot serve the wave's CPW pieces: the instruction's immediate offset
    // moves both addresses (LDS address = M0 + offset + 16 * lane).
    const char* my_pieces = packed + (wave * CPW) * 1024 + lane * 16;
    const int n_tiles32 = (int)NT;                     // item ids are int32, so tiles fit easily
    auto dma_tile = [&](int j, int slot) __attribute__((always_inline)) {
        int t = (int)t0 + j;
        t = t < n_tiles32 ? t : n_tiles32 - 1;
        if (CRH_ABLATE(a.ablate) & 2) t = 0;   // measurement only: every fetch hits the same (cached) tile
        const char* g = my_pieces + (int64_t)t * TILE_B;
        char* l = ring + slot * TILE_B + (wave * CPW) * 1024;
#define CRH_DMA_PIECE(CQ)                                                                                              \
    if constexpr (CQ < CPW)                                                                                            \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,                             \
                                         (__attribute__((address_space(3))) void*)l, 16, (CQ) * 1024, 0)
        CRH_DMA_PIECE(0); CRH_DMA_PIECE(1); CRH_DMA_PIECE(2); CRH_DMA_PIECE(3);
#undef CRH_DMA_PIECE
        static_assert(CPW <= 4, "pieces beyond the immediate offset range");
    };
    // candidate bits of the 64 tiles of block blk (tile >> 6) -> tbits[blk & 1].  Issued by EVERY wave with EVERY tile (a
    // 256-byte piece; the waves write identical bytes): no branch inside the MFMA stream, one more DMA on everybody's counter
    auto dma_tbits = [&](int64_t blk) __attribute__((always_inline)) {
        int t = ((int)blk << 6) + lane;
        t = t < n_tiles32 ? t : n_tiles32 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.tile_bits + (unsigned)t),
                                         (__attribute__((address_space(3))) void*)(tbits + (blk & 1) * 64), 4, 0, 0);
    };
    const bool has_bits = a.bitmap != nullptr;          // without a candidate bitmap tile_bits points at readable filler
    auto dma_wait_all = [&]() __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    };
    auto dma_wait_but_newest_tile = [&]() __attribute__((always_inline)) {   // everything but the CPW + 1 pieces issued last
#if defined(__HIP_DEVICE_COMPILE__)
        if (CRH_ABLATE(a.ablate) & 16) return;   // measurement only: what the DMA wait costs (results invalid)
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(CPW + 1) : "memory");
#endif
    };

    f32x16 acc[2][UW];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int u = 0; u < UW; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][u][r] = 0.0f;
    // A fragments: LDS -> registers by inline asm with COUNTED waits.  Left to hipcc the loop gets "s_waitcnt lgkmcnt(0)" right
    // behind the reads of the NEXT group (seen in the ISA: every second group stalls for a full LDS round trip).  The asm reads
    // are invisible to the compiler's counter bookkeeping; lds_wait names the registers it guards, so no MFMA that consumes
    // them can be scheduled above it (cdna_hip_programming.md 5.7: form (ii)).  LDS operations complete in order, so
    // lgkmcnt(2 GR) = "everything but the 2 GR reads issued last has landed" also covers any list access of the slow path.
    f32x4 c[4][GR];
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    auto lds_group = [&](f32x4(&dst)[GR], uint32_t addr, auto Gc) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(addr), "i"((decltype(Gc)::value * GR + 0) * 1024));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(addr), "i"((decltype(Gc)::value * GR + 1) * 1024));
#endif
    };
    auto lds_wait = [&](f32x4(&x)[GR], auto Nc) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CRH_ABLATE(a.ablate) & 8) {          // measurement only: what the fragment waits cost (results invalid)
            asm volatile("" : "+v"(x[0]), "+v"(x[1]));
            return;
        }
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x[0]), "+v"(x[1]) : "i"(decltype(Nc)::value));
#endif
    };
    static_assert(GR == 2, "lds_group / lds_wait are written for two chunks per group");
    // ---- flag form: counters, polls (see the kernel's comment)
    const uint32_t flags_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned*)flags;
    unsigned rflag = 0, dflag = 0;                       // poll results in flight (landed at the next counted fragment wait)
    auto flag_read = [&](unsigned& dst, int word) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(flags_lds + 4u * (unsigned)word));
#endif
    };
    // (all 64 lanes add to the one counter -- lane 0 a one, the others zeros: one instruction and no exec-mask detour; the
    // compiler's form of `if (lane == 0) atomicAdd` is a dozen instructions with two branches)
    const unsigned bump_val = lane == 0 ? 1u : 0u;
    auto flag_bump = [&](int word) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_add_u32 %0, %1" ::"v"(flags_lds + 4u * (unsigned)word), "v"(bump_val) : "memory");
#endif
    };
    auto flag_land = [&](unsigned& f) __attribute__((always_inline)) {     // everything LDS issued so far has landed (it was long ago)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f));
#endif
    };
    // the counted wait of a group in the flag form: everything but the GR fragment reads issued last has landed -- the
    // fragments of this group AND of the next one (a group is 2048 cycles of MFMAs: they landed long ago), the polls, the bumps
    auto lds_wait_fl = [&](f32x4(&x)[GR], unsigned& f0, unsigned& f1) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(f0), "+v"(f1) : "i"(GR));
#endif
    };
    // spin until counter `word` reaches `want` (rare: a partner wave is behind -- in a slow-path event, or waiting for its XCD)
    auto flag_spin = [&](unsigned have, int word, unsigned want) __attribute__((always_inline)) {
        unsigned v = __builtin_amdgcn_readfirstlane(have);
        if (__builtin_expect(v >= want, 1) || (CRH_ABLATE(a.ablate) & 4)) return;
        unsigned spins = 0;
        do {
            __builtin_amdgcn_s_sleep(1);
            unsigned t;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(flags_lds + 4u * (unsigned)word) : "memory");
#endif
            v = __builtin_amdgcn_readfirstlane(t);
            if (++spins > (1u << 24)) __builtin_trap();      // seconds: a protocol bug must not hang the GPU (nor pass silently)
        } while (v < want);
    };

#ifdef CRH_PROFILE
    unsigned long long ev_ticks = 0, ev_count = 0;       // CRH_SCORE_TIMING
#endif
    // body j: MFMAs of tile j (ring slot s_cur) into accumulator set P, threshold test of tile j-1 out of set 1-P;
    // s_nxt = slot of tile j+1, s_fill = slot of tile j + PF (barrier form: the slot tile j-1 left)
    auto body = [&](auto Pc, int j, int s_cur, int s_nxt, int s_fill) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value, Q = 1 - P;
        const uint32_t src = ring_lds + s_cur * TILE_B, srcn = ring_lds + s_nxt * TILE_B;
        float m[UW];
        // one group of GR chunks; g is a compile-time constant (every register array below is indexed statically)
        auto group = [&](auto Gc) __attribute__((always_inline)) {
            constexpr int g = decltype(Gc)::value;
            // fragments of the group after next (two groups = 16 MFMAs of slack for the LDS round trip); the last two groups
            // of a tile read the first two of tile j+1, visible since this body's barrier (flag form: since the poll above)
            if constexpr (g + 2 < NG) lds_group(c[(g + 2) & 3], src, std::integral_constant<int, g + 2>{});
            else lds_group(c[(g + 2) & 3], srcn, std::integral_constant<int, g + 2 - NG>{});
            if constexpr (!FL && g == BG) {
                dma_wait_but_newest_tile();             // my pieces of tile j+1 (issued two tiles ago) have landed
                if (!(CRH_ABLATE(a.ablate) & 4)) __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FL) {
                lds_wait_fl(c[g & 3], rflag, dflag);
            } else {
                lds_wait(c[g & 3], std::integral_constant<int, 2 * GR>{});   // this group's fragments (read two groups ago) are in
            }
            // barrier form: tile j+3 into the slot tile j-1 left (every wave is past it); the scheduler places the DMA instructions
            // BETWEEN this group's MFMAs (they follow the wait in program order).  With it the candidate bits of the block
            // of 64 tiles that holds tile j + 32: the current block again (same bytes) in its first half, the NEXT block in
            // its second half -- into the buffer of the block before, whose last tile was selected at the head of body
            // 64 b, before every wave's barrier of that body (flag form: 32 tiles behind every wave, which are at most R apart)
            if constexpr (!FL && g == BG) {
                dma_tbits((t0 + j + 32) >> 6);
                dma_tile(j + 3, s_fill);
                // one DMA instruction and a few of its address instructions per MFMA gap instead of all of them in one gap
                // (+0.9 % on the same box: a single wave per SIMD hides about five issue slots per MFMA, not twenty)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);   // SALU
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (the DMA)
                }
            }
            if constexpr (FL) {
                // The flag protocol, one piece per MFMA gap (n = index of the MFMA just issued, 0 .. 8 UW - 1 in this group):
                //   group 0      n=0  my pieces of tile j + PF - 1 (issued a whole tile ago) have landed -> publish them
                //                n=1  may the slot of tile j + PF - R be refilled?  (poll issued in the last group of the body before)
                //                n=2  the tile bits, n=3 tile j + PF
                //   group NG-3   n=0  poll: is tile j+1 complete?   n=8: the poll has landed, n=9: spin if it is not
                //   group NG-2   n=0  every fragment read of tile j has returned (the group's counted wait left only its own reads
                //                     -- of tile j+1 -- outstanding): the slot may be refilled
                //   group NG-1   n=0  poll for the next body's refill, n=8: landed
                auto hook = [&](auto Nc) __attribute__((always_inline)) {
                    constexpr int n = decltype(Nc)::value;
                    constexpr bool any = (g == 0 && n <= 3) || (g == NG - 3 && (n == 0 || n == 8 || n == 9)) || (g == NG - 2 && n == 0) ||
                                         (g == NG - 1 && (n == 0 || n == 8));
                    if constexpr (any) __builtin_amdgcn_sched_barrier(0);
                    if constexpr (g == 0 && n == 0) {
                        dma_wait_all();
                        flag_bump((j + PF - 1) & (R - 1));
                    }
                    if constexpr (g == 0 && n == 1) flag_spin(dflag, 8 + s_fill, (unsigned)NW * (unsigned)((j + PF) / R));
                    if constexpr (g == 0 && n == 2) dma_tbits((t0 + j + 32) >> 6);
                    if constexpr (g == 0 && n == 3) dma_tile(j + PF, s_fill);
                    if constexpr (g == NG - 3 && n == 0) flag_read(rflag, s_nxt);
                    if constexpr (g == NG - 3 && n == 8) flag_land(rflag);
                    if constexpr (g == NG - 3 && n == 9) flag_spin(rflag, s_nxt, (unsigned)NW * (unsigned)((j + 1) / R + 1));
                    if constexpr (g == NG - 2 && n == 0) flag_bump(8 + s_cur);
                    if constexpr (g == NG - 1 && n == 0) flag_read(dflag, 8 + ((j + 1 + PF) & (R - 1)));
                    if constexpr (g == NG - 1 && n == 8) flag_land(dflag);
                    if constexpr (any) __builtin_amdgcn_sched_barrier(0);
                };
                if constexpr (g == 0) {
#pragma unroll
                    for (int u = 0; u < UW; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[P][u][r] = 0.0f;
                }
                if constexpr (sizeof(T) == 4) {
                    [&]<int... JJ>(std::integer_sequence<int, JJ...>) __attribute__((always_inline)) {
                        (mma_f32_hooked<UW, JJ * 4 * UW>(acc[P], c[g & 3][JJ], b[g * GR + JJ], hook), ...);
                    }(std::make_integer_sequence<int, GR>{});
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < GR; ++jj) {
                    if constexpr (g == 0) {
                        if (jj == 0) {
#pragma unroll
                            for (int u = 0; u < UW; ++u)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[P][u][r] = 0.0f;
                        }
                    }
                    Elem<T>::template mma<UW>(acc[P], c[g & 3][jj], b[g * GR + jj]);
                }
            }
            if constexpr (g == 1) {
#pragma unroll
                for (int u = 0; u < UW; ++u) m[u] = max16(acc[Q][u]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g == 1) {
                // four compares straight into scalar masks (OR-ed per lane first, the ballot of the result was a select and a second
                // compare: two VALU instructions per tile beside MFMAs that hide none)
                unsigned long long hitm = 0ull;
#pragma unroll
                for (int u = 0; u < UW; ++u) hitm |= __ballot(m[u] > tau[u]);
                if (hitm != 0ull && live && j > 0 && !(CRH_ABLATE(a.ablate) & 1)) {
#ifdef CRH_PROFILE
                    const unsigned long long ev_t0 = a.wave_clock != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
                    const int64_t ts = t0 + j - 1;                                      // the tile being selected
                    const unsigned tb = has_bits ? tbits[((ts >> 6) & 1) * 64 + (ts & 63)] : 0u;
#pragma unroll
                    for (int u = 0; u < UW; ++u)
                        if (__ballot(m[u] > tau[u]) != 0ull)
                            tile_slow_path_dma<sizeof(T) == 4 && DMA_ONE_TRIP_F32>(acc[Q][u], tau[u], ls, li, cnt, K, 32 * u, slot0w + 32 * u, a, ts << 5,
                                               split_end, lane, tb, rfilter);
                    // the list stores of the inserts are drained HERE: left pending, the compiler parks an lgkmcnt(0) at a
                    // later point of the common path, right behind the fragment reads of the barrier group
                    __builtin_amdgcn_s_waitcnt(0xc07f);
#ifdef CRH_PROFILE
                    if (a.wave_clock != nullptr) {      // CRH_SCORE_TIMING: cycles spent in events and their number, per wave
                        ev_ticks += __builtin_amdgcn_s_memtime() - ev_t0;
                        ev_count += 1;
                    }
#endif
                }
            }
        };
        [&]<int... Gs>(std::integer_sequence<int, Gs...>) __attribute__((always_inline)) {
            (group(std::integral_constant<int, Gs>{}), ...);
        }(std::make_integer_sequence<int, NG>{});
    };

    if (n_steps > 0) {
        dma_tbits(t0 >> 6);
        dma_tbits((t0 >> 6) + 1);
#pragma unroll
        for (int p = 0; p < PF; ++p) dma_tile(p, p);
        dma_wait_all();
        __builtin_amdgcn_s_barrier();                     // (also publishes the flag form's initial counters)
        lds_group(c[0], ring_lds, std::integral_constant<int, 0>{});
        lds_group(c[1], ring_lds, std::integral_constant<int, 1>{});
        if constexpr (FL) flag_read(dflag, 8 + (PF & (R - 1)));      // body 0's refill poll (nothing has been read yet: 0 >= 0)
        int s0 = 0;
        // XCD soft lockstep (see xcd_window_sync): wave 0 reports / waits for the workgroup, the others meet it at the
        // tile's barrier
        unsigned* sync_cnt = nullptr;
        int win_steps = 0, next_sync = n_steps + 4;
        if (a.xcd_sync && wave == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // HW_REG_XCC_ID
            sync_cnt = a.xcd_sync + (int64_t)xcc * a.sync_stride;
            if (lane == 0) __hip_atomic_fetch_add(sync_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            win_steps = a.sync_window & ~1;               // whole loop trips (two bodies each)
            if (win_steps < 2) win_steps = 2;
            next_sync = 0;
        }
#ifdef CRH_PROFILE
        unsigned long long t_prev = 0;                  // CRH_SCORE_TIMING: duration of every PAIR of tiles, histogrammed per wave
#endif
        for (int j = 0; j <= n_steps; j += 2) {
#ifdef CRH_PROFILE
            if (a.wave_clock != nullptr && blockIdx.x < 16) {
                const unsigned long long t_now = __builtin_amdgcn_s_memtime();
                if (j >= 64 && lane == 0) {
                    // two tiles per trip: 256-cycle units = 128 per tile (fp16); fp32 tiles are 4x as long: 512 per tile
                    unsigned long long dt = (t_now - t_prev) >> ((sizeof(T) == 4 ? 10 : 8) - (ROWB == 256 ? 1 : 0));
                    if (dt > 63) dt = 63;
                    atomicAdd(&a.wave_clock[((size_t)blockIdx.x * 4 + wave) * 64 + dt], 1ull);
                }
                t_prev = t_now;
            }
#endif
            if (j >= next_sync && j < n_steps) {       // wave 0 only (next_sync stays beyond the range elsewhere)
                const int win = j / win_steps;
                if (xcd_window_sync(sync_cnt, win, lane)) {
                    next_sync += win_steps;
                } else {                // timed out: run free, and count this workgroup into every window it will not report
                    next_sync = n_steps + 4;
                    const int n_win = (n_steps + win_steps - 1) / win_steps;
                    if (lane == 0)
                        for (int wdw = win; wdw < n_win; ++wdw)
                            __hip_atomic_fetch_add(sync_cnt + 1 + wdw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const int s1 = (s0 + 1) & (R - 1), s2 = (s0 + 2) & (R - 1);
            // barrier form: tile j+3 -> slot of tile j-1 = s0 + 3, tile j+4 -> slot of tile j; flag form: tile j + PF -> its own slot
            body(std::integral_constant<int, 0>{}, j, s0, s1, (s0 + PF) & (R - 1));
            if (j + 1 > n_steps) break;
            body(std::integral_constant<int, 1>{}, j + 1, s1, s2, (s1 + PF) & (R - 1));
            s0 = s2;
        }
        dma_wait_all();   // the prefetches of the last bodies must not outlive the workgroup's LDS ...
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[1][0]), "+v"(c[1][1]), "+v"(c[2][0]), "+v"(c[2][1]),
                     "+v"(c[3][0]), "+v"(c[3][1]), "+v"(rflag), "+v"(dflag));   // ... nor the last fragment reads / polls their registers
#endif
    }

#ifdef CRH_PROFILE
    if (a.wave_clock != nullptr && blockIdx.x < 16 && lane == 0) {
        a.wave_clock[(size_t)16 * 4 * 64 + ((size_t)blockIdx.x * 4 + wave) * 2] = ev_ticks;
        a.wave_clock[(size_t)16 * 4 * 64 + ((size_t)blockIdx.x * 4 + wave) * 2 + 1] = ev_count;
    }
#endif
    if (live) {
        for (int j = 0; j < UPW; ++j) {
            const int64_t slot = slot0w + j;
            if (slot >= a.n_users) break;
            const int n = __builtin_amdgcn_readfirstlane(cnt[j]);
            const int64_t o = ((int64_t)split * a.n_users + slot) * K;
            if (a.seed_score && S > 1)
                wave_list_store_range(ls + j * K, li + j * K, n, K, a.out_score + o, a.out_idx + o, lane,
                                      split == 0 ? INT32_MIN : (int)(a.item_base + (t0 << 5)), (int)(a.item_base + split_end));
            else
                wave_list_store(ls + j * K, li + j * K, n, K, a.out_score + o, a.out_idx + o, lane);
        }
    }
}

template <typename T, int D, int R, bool FL>
int launch_score_dma_t(const ScoreArgs& a, hipStream_t stream) {
    const size_t lds = score_dma_lds_bytes(D * (int)sizeof(T), a.k, R, FL);
    auto kern = score_topk_dma_kernel<T, D, R, FL>;
    CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    const int64_t blocks = ((a.n_ugroups + DMA_NW - 1) / DMA_NW) * a.n_splits;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * DMA_NW), lds, stream, a);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

}  // namespace

size_t score_dma_lds_bytes(int row_bytes, int k, int ring_slots, bool flags) {
    return (size_t)ring_slots * (row_bytes / 32) * 1024 + TBITS_B + (flags ? FLAGS_B : 0) + DMA_NW * dma_wave_lds_bytes(k);
}

// Ring slots of a launch, 0 = the lists do not fit beside the ring.  (A flag form with EIGHT slots for 256-byte rows -- prefetch
// distance 5, publish lag 1 -- was built and measured in round 6: at d=64 the per-wave kernel is as fast or faster wherever that
// form beat the barrier form; HISTORY.md, profiles/r06_flag_vs_barrier_vs_wave_ab.log.)
int score_dma_ring_slots(int esz, int d, int k, int mode) {
    const bool fl = esz == 4 && d == 128 && mode == 2;
    return score_dma_lds_bytes(d * esz, k, 4, fl) <= 160 * 1024 ? 4 : 0;
}

// 512-byte rows (fp32 d=128: the headline; fp16 d=256: configs[4]) and 256-byte rows (fp32 d=64: the reference's default
// width, main.py:97 --emb_size 64 = configs[0]): half the tile, half the B registers, the same loop.  mode: 2 = the flag form
// (fp32 d=128 only), anything else the barrier form.
int launch_score_dma(int esz, int d, int mode, const ScoreArgs& a, hipStream_t stream) {
    if (esz == 2) return launch_score_dma_t<_Float16, 256, 4, false>(a, stream);
    if (d == 64) return launch_score_dma_t<float, 64, 4, false>(a, stream);
    return mode == 2 ? launch_score_dma_t<float, 128, 4, true>(a, stream) : launch_score_dma_t<float, 128, 4, false>(a, stream);
}

}  // namespace crh_score
