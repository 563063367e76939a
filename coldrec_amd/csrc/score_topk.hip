// Fused full-catalogue scoring + masking + per-user top-k for gfx950 (MI355X).
//
// Replaces model/MF.py:58-63 (user_emb[users] @ item_emb.T), model/BaseRecommender.py:175-180
// (the -10e8 masks) and :182 (torch.topk) of the reference without ever writing the
// (users x items) score block to memory.
//
// What every variant shares:
//   * a wave owns 32*UW users.  Their embeddings are loaded once and stay in VGPRs as the
//     B operand of the MFMA for the whole kernel (the "hot user block").
//   * items are walked in tiles of 32 rows.  With A = items -> rows and B = users -> columns a lane
//     holds 16 item scores of ONE user, so the running k-th best score tau of that user is one VGPR and
//     the common case is: max of 16 registers, one compare, one ballot per 32x32 tile.
//   * fp32: v_mfma_f32_32x32x2_f32 issued in k order into ONE accumulator per (item tile, user tile) --
//     bit-identical to the canonical fp32 fma chain of the oracle (oracle/topk_oracle.c).  fp16 tables:
//     v_mfma_f32_32x32x16_f16 with fp32 accumulation.
//   * only tiles that contain a score above tau take the slow path (tile_slow_path): the candidate-bitmap bits of the
//     tile are one wave-uniform load, masked candidates of full lists are dropped up front, the rated list is searched
//     in memory only when the user's LDS membership filter says so, and survivors are inserted into the user's sorted
//     list in LDS by the whole wave (topk_list.h).  Expected slow-path events per user are ~ k*ln(range/k), against
//     range/32 tiles.
//   * the grid is (user groups) x (item-range splits); partial lists are merged by merge_topk with the same
//     canonical key, so the result does not depend on the split count, on the tiling or (after the
//     all-gather) on the GPU count.
//   * pack_items_kernel writes a copy of the item shard in MFMA-fragment order (one pass per call) so the
//     scoring loops issue fully coalesced 1 KiB loads and no cross-lane shuffles.
// Two kernels:
//   * score_topk_kernel      one wave = one workgroup, waves never synchronise; two waves per SIMD hide each
//                            other's stalls; optional soft lockstep of the waves of an XCD (L2 locality).
//                            Small and mid-size catalogues (below 2 M items), every width, the row-major fallback; also
//                            the producer of the dense score block that crh_mask_topk_f32 ranks (small calls).
//   * score_topk_wg_kernel   the 8 waves of a workgroup walk the tiles together through a three-slot LDS ring
//                            (1/8 of the L2 -> CU traffic), 1 / 2 / 4 tiles per slot by row width, the workgroups of an
//                            XCD in soft lockstep: fp16 (where the per-wave stream is the bound) and fp32 d=128
//                            launches of >= 2 M items whose 512-user workgroups fill the CUs.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "score_topk_common.h"

using namespace crh_score;

namespace {

// Item table -> fragment order.  Tile t = rows [32t, 32t+32) (rows past the end repeat the last row),
// chunk q = k in [8q, 8q+8):   packed[((t*NCH + q)*64 + h*32 + i)*4 + c] = V[32t + i][8q + 2*j(c) + h],
// j(x,y,z,w) = 0,2,1,3 -- exactly the f32x4 (chunk_swap's output order) lane (i,h) feeds to the four
// 32x32x2 MFMAs of chunk q (k = 8q+2j in the low half-wave, 8q+2j+1 in the high one), so the scoring loop issues one fully coalesced 1 KiB load per
// chunk and no cross-lane swaps (measured: the two v_permlane32_swap per chunk cost ~9 % of the MFMA
// rate, the 32-B-per-row gather another ~4 %).  One workgroup per tile, staged through LDS.
// Rows of items the candidate bitmap masks are packed as ZEROS (both pack kernels): a masked item is worth -1e9 whatever its
// embedding says, it can only enter a list that is not full yet (threshold -inf, where every item is a candidate anyway), and
// a full list drops it in the slow path by its bitmap bit -- but with its real embedding one masked item in five would first
// beat the threshold and cost an event (a stall of the wave, in the workgroup kernels of the whole CU).  Score 0 never beats a
// positive threshold.  The dense route ranks with crh_mask_topk_f32, which applies the bitmap itself.
__device__ __forceinline__ bool pack_row_masked(const uint32_t* bitmap, int64_t gid) {
    return bitmap != nullptr && ((bitmap[gid >> 5] >> (gid & 31)) & 1u);
}

template <int D>
__global__ __launch_bounds__(256) void pack_items_kernel(const float* __restrict__ v, int64_t n_items,
                                                         float* __restrict__ packed, const uint32_t* __restrict__ bitmap,
                                                         int64_t item_base) {
    constexpr int NCH = D / 8;
    __shared__ float tile[32][D + 4];
    const int64_t t = blockIdx.x;
    for (int e = threadIdx.x; e < 32 * (D / 4); e += 256) {
        const int r = e / (D / 4), c = e % (D / 4);
        int64_t row = (t << 5) + r;
        if (row >= n_items) row = n_items - 1;
        f32x4 x = *reinterpret_cast<const f32x4*>(v + row * D + 4 * c);
        if (pack_row_masked(bitmap, item_base + row)) x = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        tile[r][4 * c + 0] = x.x; tile[r][4 * c + 1] = x.y; tile[r][4 * c + 2] = x.z; tile[r][4 * c + 3] = x.w;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NCH * 64; e += 256) {
        const int q = e >> 6, ln = e & 63, i = ln & 31, h = ln >> 5;
        f32x4 o;
        o.x = tile[i][8 * q + 0 + h];   // same component order as chunk_swap's output: the MFMAs are
        o.z = tile[i][8 * q + 2 + h];   // issued .x, .z, .y, .w = k pairs (0,1) (2,3) (4,5) (6,7)
        o.y = tile[i][8 * q + 4 + h];
        o.w = tile[i][8 * q + 6 + h];
        *reinterpret_cast<f32x4*>(packed + ((t * NCH) * 64 + e) * 4) = o;
    }
}

// fp16 tables (config 5: DropoutNet-style generated embeddings): v_mfma_f32_32x32x16_f16, fp32 accumulate.
// Lane (i,h) feeds k = 8h..8h+7 of row i, which is one 16-B load of a row-major fp16 row: no swaps at all;
// packing only makes the loads contiguous: unit (2q+h) of row i -> slot ((t*NCH+q)*64 + h*32 + i).
template <int D>
__global__ __launch_bounds__(256) void pack_items_f16_kernel(const _Float16* __restrict__ v, int64_t n_items,
                                                             _Float16* __restrict__ packed, const uint32_t* __restrict__ bitmap,
                                                             int64_t item_base) {
    constexpr int NCH = D / 16;
    const int64_t t = blockIdx.x;
    const u32x4* src = reinterpret_cast<const u32x4*>(v);
    u32x4* dst = reinterpret_cast<u32x4*>(packed);
    for (int e = threadIdx.x; e < NCH * 64; e += 256) {
        // consecutive threads read consecutive 16-B units of a row (coalesced), scatter 16-B units
        const int r = e / (2 * NCH), c = e % (2 * NCH);
        int64_t row = (t << 5) + r;
        if (row >= n_items) row = n_items - 1;
        const int q = c >> 1, h = c & 1;
        dst[(t * NCH + q) * 64 + h * 32 + r] = pack_row_masked(bitmap, item_base + row) ? u32x4{0u, 0u, 0u, 0u} : src[row * (2 * NCH) + c];
    }
}

// Candidate-bitmap bits per 32-item tile of a shard (bit r of word t = bit of global item item_base + 32 t + r): what the DMA
// kernel streams beside the item tiles (256 bytes per 64 tiles), so that its slow path finds a tile's bits in LDS instead of
// fetching bitmap words from memory.  Items past the shard's end read as "not masked" (their candidates are dropped anyway).
__global__ __launch_bounds__(256) void tile_bits_kernel(const uint32_t* __restrict__ bitmap, int64_t item_base, int64_t n_items,
                                                        int64_t n_tiles, uint32_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles) return;
    const int64_t g0 = item_base + (t << 5), last = (item_base + n_items - 1) >> 5;
    const int64_t w0 = (g0 >> 5) < last ? (g0 >> 5) : last, w1 = w0 < last ? w0 + 1 : last;
    const unsigned long long both = ((unsigned long long)bitmap[w1] << 32) | bitmap[w0];
    unsigned bits = (unsigned)(both >> (g0 & 31));
    const int64_t left = n_items - (t << 5);
    if (left < 32) bits &= (1u << left) - 1u;
    out[t] = bits;
}

// OCC = waves per SIMD the kernel is built for:
//   1: two register tiles (A double-buffered), ~320 VGPR+AGPR, one wave per SIMD;
//   2: one register tile reloaded chunk by chunk right behind its last use, <= 256 registers, so a
//      second wave on the SIMD fills the matrix pipe while this one selects / waits / inserts.
// DN = a block launch (small catalogues and the seed prefix): the tile goes to a.dense, no lists.  Its own instantiation: in one
// body with the selection the staging code's registers pushed a spill into the slow path of the two-waves-per-SIMD kernels.
template <typename T, int D, int UW, int OCC, bool PK, bool DN>
__global__ __launch_bounds__(64, OCC) void score_topk_kernel(ScoreArgs a) {
    constexpr int WPW = 1;
    constexpr int ROWB = D * (int)sizeof(T);   // bytes per table row
    constexpr int NCH = ROWB / 32;             // 32-byte chunks per row: lane (i,h) owns 16 B of each
    constexpr bool SWAP = Elem<T>::kSwap;
    constexpr int UPW = 32 * UW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t vb = (int64_t)blockIdx.x * WPW + wave;
    const int S = a.n_splits;
    const int split = (int)(vb % S);
    const int64_t ug = vb / S;
    if (ug >= a.n_ugroups) return;
    const int K = a.k;
    const int i = lane & 31, h = lane >> 5;
    if (CRH_ABLATE(a.wave_clock != nullptr) && lane == 0) a.wave_clock[2 * vb] = wall_clock64();
    unsigned* sync_cnt = nullptr;
    if (a.xcd_sync) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // HW_REG_XCC_ID
        sync_cnt = a.xcd_sync + (int64_t)xcc * a.sync_stride;
        if (lane == 0) __hip_atomic_fetch_add(sync_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- item range of this split, in tiles of 32 rows
    const int64_t NT = (a.n_items + 31) >> 5;
    const int64_t t0 = NT * split / S, t1 = NT * (split + 1) / S;
    const int64_t split_end = (t1 << 5) < a.n_items ? (t1 << 5) : a.n_items;

    WaveLds<UPW> w;
    wave_lds_carve<UPW>(w, smem + (size_t)wave * wave_lds_bytes<UPW>(K), K);
    // (a block launch keeps no lists: nothing to stage)
    if constexpr (!DN) wave_lds_init<UPW>(w, a, ug * UPW, (int)(a.item_base + (t0 << 5)), (int)(a.item_base + split_end), true, lane);

    // ---- hot user block -> registers (B fragments, already pair-swapped)
    f32x4 b[NCH][UW];
    float tau[UW];
#pragma unroll
    for (int u = 0; u < UW; ++u) {
        int64_t slot = ug * UPW + 32 * u + i;
        // padding columns of the last group: duplicate the last user, threshold +inf (with -inf every tile of the
        // wave would enter the slow path to find nothing: 100 000 users ran 2.2x slower than 131 072)
        tau[u] = slot < a.n_users ? CRH_NEG_INF : __builtin_inff();
        if (a.seed_score && slot < a.n_users) tau[u] = wave_list_tau(w.ls + (32 * u + i) * K, w.cnt[32 * u + i], K);
        if (slot >= a.n_users) slot = a.n_users - 1;
        const int64_t row = a.users ? (int64_t)a.users[slot] : a.user_base + slot;
        const char* up = reinterpret_cast<const char*>(a.user_emb) + row * ROWB + 16 * h;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            b[q][u] = load16(up + 32 * q);
            if constexpr (SWAP) chunk_swap(b[q][u]);
        }
    }

    // Two register tiles: while tile t is multiplied out of one, tile t+1 lands in the other
    // (issued a whole tile = NCH*4*UW MFMAs ahead, pinned there by sched_barrier; left to
    // itself hipcc sinks the loads behind the MFMAs and waits vmcnt(0) at the loop head).
    // address of this lane's first 16 B of tile t: row-major table -> row (t*32 + i), k offset 4h (+8 per
    // chunk); packed table (pack_items_kernel) -> chunk-major fragments, 1 KiB per wave-load (+256 floats
    // per chunk).  Rows past the split end are clamped / padded: always a valid address.
    const int64_t T_all = (a.n_items + 31) >> 5;
    constexpr int QSTRIDE = PK ? 1024 : 32;    // bytes between this lane's consecutive chunks
    // A tile's address = a wave-uniform base (scalar registers) + a 32-bit lane offset: global_load's saddr form.  As one
    // 64-bit pointer per lane it cost a register pair the 256-register budget of two waves per SIMD does not have (hipcc
    // spilled it and reloaded it from scratch behind an s_waitcnt vmcnt(0) in front of EVERY tile: r04 kernel table, 24 B/lane).
    auto tile_base = [&](int64_t t) -> const char* {
        if (CRH_ABLATE(a.ablate) & 2) t = 0;   // measurement only: every load hits the same (cached) tile
        if constexpr (PK) {
            if (t >= T_all) t = T_all - 1;
            return reinterpret_cast<const char*>(a.packed) + t * (NCH * 1024);
        } else {   // (the tile fetched ahead of the split's last one: clamped to it -- always a valid address)
            const int64_t t_last = (split_end - 1) >> 5;
            return reinterpret_cast<const char*>(a.item_emb) + ((t < t_last ? t : t_last) << 5) * ROWB;
        }
    };
    auto lane_off = [&](int64_t t) -> unsigned {
        if constexpr (PK) {
            unsigned o = (unsigned)lane * 16u;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(o));   // opaque per tile: keeps hipcc from folding it back into one hoisted 64-bit lane pointer
#endif
            return o;
        } else {   // rows past the split end are clamped to its last row: always a valid address
            if (CRH_ABLATE(a.ablate) & 2) t = 0;
            const int64_t t_last = (split_end - 1) >> 5;
            const int64_t left = split_end - ((t < t_last ? t : t_last) << 5);   // wave-uniform, >= 1
            const int row = left > i ? i : (int)(left - 1);
            return (unsigned)row * (unsigned)ROWB + 16u * (unsigned)h;
        }
    };
    auto load_tile = [&](f32x4(&dst)[NCH], int64_t t) {
        const char* vp = tile_base(t);
        const unsigned off = lane_off(t);
#pragma unroll
        for (int q = 0; q < NCH; ++q) dst[q] = load16(vp + QSTRIDE * q + off);
    };
    unsigned long long slow_ticks = 0, slow_events = 0;   // profile build only (CRH_SCORE_TIMING)
    auto do_tile = [&](f32x4(&src)[NCH], int64_t t, const char* vnext, unsigned off_next) {
        f32x16 acc[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            f32x4 c = src[q];
            if constexpr (SWAP && !PK) chunk_swap(c);   // packed tiles are stored in fragment order already
            Elem<T>::template mma<UW>(acc, c, b[q]);
            if (OCC > 1) {   // ring: the chunk just consumed is refilled with the next tile's rows
                src[q] = load16(vnext + QSTRIDE * q + off_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if ((CRH_ABLATE(a.ablate) & 1) && !DN) {   // keep the products live, skip selection (roofline ablation, results invalid)
#pragma unroll
            for (int u = 0; u < UW; ++u) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(acc[u]));
#endif
            }
            return;
        }
        if constexpr (DN) {
            // small catalogue: the tile goes to memory as it is.  acc[r] = item row (r&3) + 8*(r>>2) + 4h of user
            // column i: four 16-B pieces per lane, columns 8g + 4h .. +3 of the tile.  (Turned through LDS so that a store
            // writes whole 128-B rows it ran 3-6 % slower at the validation shapes: the stores are ~10 % of a block launch.)
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int64_t slot = ug * UPW + 32 * u + i;
                if (slot < a.n_users && (!(CRH_ABLATE(a.ablate) & 8) || acc[u][0] == 123.f)) {   // (8: measurement only, no block stores)
                    // uniform base + 32-bit lane offset (the dispatcher keeps 32 rows of the block under 4 GiB)
                    unsigned ro = (unsigned)i * (unsigned)a.dense_stride + 4u * (unsigned)h;
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : "+v"(ro));   // see lane_off
#endif
                    float* row = a.dense + (ug * UPW + 32 * u) * a.dense_stride + (t << 5) + ro;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(row + 8 * g) =
                            f32x4{acc[u][4 * g], acc[u][4 * g + 1], acc[u][4 * g + 2], acc[u][4 * g + 3]};
                }
            }
            return;
        }
        // selection: one compare per lane and accumulator in the common case
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            const float m = max16(acc[u]);
            if (__ballot(m > tau[u]) != 0ull) {
                unsigned long long c0 = 0;
                if (CRH_ABLATE(a.wave_clock != nullptr)) c0 = wall_clock64();
                tile_slow_path<UPW>(acc[u], tau[u], w, K, 32 * u, ug * UPW + 32 * u, a, t << 5, split_end,
                                    lane, PK);
                if (CRH_ABLATE(a.wave_clock != nullptr)) { slow_ticks += wall_clock64() - c0; ++slow_events; }
            }
        }
    };

    if (CRH_ABLATE(a.wave_clock != nullptr) && lane == 0) a.wave_clock[2 * (a.n_ugroups * S) + 2 * vb] = wall_clock64();
    if (t0 < t1) {
        if constexpr (OCC == 1) {
            f32x4 ta[NCH], tb[NCH];
            load_tile(ta, t0);
            for (int64_t t = t0; t < t1; t += 2) {
                load_tile(tb, t + 1);   // rows past the split end are clamped: always a valid address
                __builtin_amdgcn_sched_barrier(0);
                do_tile(ta, t, nullptr, 0u);
                if (t + 1 >= t1) break;
                load_tile(ta, t + 2);
                __builtin_amdgcn_sched_barrier(0);
                do_tile(tb, t + 1, nullptr, 0u);
            }
        } else {
            f32x4 ta[NCH];
            load_tile(ta, t0);
            int64_t next_sync = sync_cnt ? t0 : t1 + 1;
            for (int64_t t = t0; t < t1; ++t) {
                if (t == next_sync) {
                    const int64_t win = (t - t0) / a.sync_window;
                    if (xcd_window_sync(sync_cnt, win, lane)) {
                        next_sync += a.sync_window;
                    } else {   // timed out: run free, and count this wave into every window it will not report
                        next_sync = t1 + 1;
                        const int64_t n_win = (t1 - t0 + a.sync_window - 1) / a.sync_window;
                        if (lane == 0)
                            for (int64_t wdw = win; wdw < n_win; ++wdw)
                                __hip_atomic_fetch_add(sync_cnt + 1 + wdw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                do_tile(ta, t, tile_base(t + 1), lane_off(t + 1));
            }
        }
    }

    if (CRH_ABLATE(a.wave_clock != nullptr) && lane == 0) {
        a.wave_clock[2 * (a.n_ugroups * S) + 2 * vb + 1] = wall_clock64();
        a.wave_clock[4 * (a.n_ugroups * S) + 2 * vb] = slow_ticks;
        a.wave_clock[4 * (a.n_ugroups * S) + 2 * vb + 1] = slow_events;
        if (DN) a.wave_clock[2 * vb + 1] = wall_clock64();
    }
    if constexpr (DN) return;
    // ---- write this split's lists: [split][slot][k], padded with (-inf, PAD)
    for (int j = 0; j < UPW; ++j) {
        const int64_t slot = ug * UPW + j;
        if (slot >= a.n_users) break;
        const int n = __builtin_amdgcn_readfirstlane(w.cnt[j]);
        const int64_t o = ((int64_t)split * a.n_users + slot) * K;
        if (a.seed_score && S > 1)    // a cut keeps its own item range (the first cut: and the seeds below it)
            wave_list_store_range(w.ls + j * K, w.li + j * K, n, K, a.out_score + o, a.out_idx + o, lane,
                                  split == 0 ? INT32_MIN : (int)(a.item_base + (t0 << 5)), (int)(a.item_base + split_end));
        else
            wave_list_store(w.ls + j * K, w.li + j * K, n, K, a.out_score + o, a.out_idx + o, lane);
    }
    if (CRH_ABLATE(a.wave_clock != nullptr) && lane == 0) a.wave_clock[2 * vb + 1] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------------------
// Workgroup-cooperative variant for fp16 tables.  At 16x the fp32 MFMA rate the per-wave item stream of the
// kernel above is what bounds it: every wave pulls every tile through its CU's vector cache, and a CU sustains
// only ~11 B/clk of L2 fills (measured: 17 % of the fp16 peak at d=256).  Here the 8 waves of a workgroup
// (one CU, two per SIMD) walk the item tiles TOGETHER: each wave fetches 1/8 of a packed tile (global ->
// registers, two tiles ahead), drops it into a two-slot LDS ring, and all 8 waves read their A fragments from
// LDS -- 8x less L2 -> CU traffic.  One s_barrier per tile (LDS only: the prefetch loads stay in flight across
// it).  Users, thresholds, lists and the slow path are per wave exactly as above, so results are identical.

template <typename T, int D, int UW, int NW, int TT>
__global__ __launch_bounds__(64 * NW, 8 / NW) void score_topk_wg_kernel(ScoreArgs a) {
    // TT = item tiles per ring slot and barrier ("step").  The cost of a step that is not MFMA (barrier skew, ring
    // commit, the drain of the matrix pipe before the threshold test) is the same for every width, so narrow rows
    // take two tiles per step: d=128 fp16 43 % -> see DESIGN section 5.
    constexpr int ROWB = D * (int)sizeof(T);
    constexpr int NCH = ROWB / 32;
    constexpr int UPW = 32 * UW;
    constexpr int TILE_B = NCH * 1024;                 // one packed tile
    constexpr int SCH = TT * NCH;                      // 1 KiB chunks of a step
    constexpr int STEP_B = TT * TILE_B;
    constexpr int CPW = (SCH + NW - 1) / NW;           // chunks of a step each wave fetches
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

    char* ring = smem;                                  // [RING][STEP_B]
    const int64_t NT = (a.n_items + 31) >> 5;
    const int64_t t0 = NT * split / S, t1 = NT * (split + 1) / S;
    const int64_t split_end = (t1 << 5) < a.n_items ? (t1 << 5) : a.n_items;

    WaveLds<UPW> w;
    wave_lds_carve<UPW>(w, smem + WG_RING * STEP_B + (size_t)wave * wave_lds_bytes<UPW>(K), K);
    wave_lds_init<UPW>(w, a, ug * UPW, (int)(a.item_base + (t0 << 5)), (int)(a.item_base + split_end), live, lane);

    f32x4 b[NCH][UW];
    float tau[UW];
#pragma unroll
    for (int u = 0; u < UW; ++u) {
        int64_t slot = ug * UPW + 32 * u + i;
        // padding columns of the last group: duplicate the last user, threshold +inf (with -inf every tile of the
        // wave would enter the slow path to find nothing: 100 000 users ran 2.2x slower than 131 072)
        tau[u] = slot < a.n_users ? CRH_NEG_INF : __builtin_inff();
        if (a.seed_score && slot < a.n_users) tau[u] = wave_list_tau(w.ls + (32 * u + i) * K, w.cnt[32 * u + i], K);
        if (slot >= a.n_users) slot = a.n_users - 1;
        const int64_t row = a.users ? (int64_t)a.users[slot] : a.user_base + slot;
        const char* up = reinterpret_cast<const char*>(a.user_emb) + row * ROWB + 16 * h;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            b[q][u] = load16(up + 32 * q);
            if constexpr (Elem<T>::kSwap) chunk_swap(b[q][u]);
        }
    }

    // the user fragments must have LANDED before the loop: otherwise the compiler's vmcnt bookkeeping for them
    // (needed in the first iteration only) also drains the tile prefetches of every later iteration
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int u = 0; u < UW; ++u) asm volatile("" : "+v"(b[q][u]));
#endif
    const char* packed = reinterpret_cast<const char*>(a.packed);
    // step j of this split = tiles t0 + j*TT .. +TT-1 (tiles past the split or the table: computed on clamped
    // rows, their candidates dropped by the il >= split_end test of the slow path)
    const int64_t n_steps = (t1 - t0 + TT - 1) / TT;

    // this wave's share of step j: chunks wave, wave+NW, ... (1 KiB each, 16 B per lane)
    auto fetch = [&](f32x4(&st)[CPW], int64_t j) {
        if (j >= n_steps) j = n_steps - 1;
        if (CRH_ABLATE(a.ablate) & 2) j = 0;   // measurement only: every fetch hits the same (cached) tiles
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int qq = wave + c * NW;
            if (SCH % NW == 0 || qq < SCH) {   // branch-free when NW | SCH
                int64_t t = t0 + j * TT + qq / NCH;
                if (t >= NT) t = NT - 1;
                st[c] = load16(packed + t * TILE_B + (qq % NCH) * 1024 + lane * 16);
            }
        }
    };
    // ring of WG_RING = 3 slots: step j lives in slot j % 3.  Step j+2 is committed during iteration j (into the
    // slot step j-1 left before the last barrier), so step j+1 is already visible while step j is multiplied and its
    // first LDS group can be pulled into registers BEFORE the barrier: the next iteration starts its MFMAs without
    // waiting for LDS.
    auto commit = [&](const f32x4(&st)[CPW], int slot) {
        char* dst = ring + slot * STEP_B + lane * 16;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int qq = wave + c * NW;
            if (SCH % NW == 0 || qq < SCH) *reinterpret_cast<f32x4*>(dst + qq * 1024) = st[c];
        }
    };
    auto lds_barrier = [&]() {   // LDS traffic of this wave done, then meet; global prefetches keep flying
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0), vmcnt/expcnt untouched
        if (CRH_ABLATE(a.ablate) & 4) return;   // measurement only (profile build): what the workgroup barriers cost; results invalid
        __builtin_amdgcn_s_barrier();
    };
    // A fragments come from LDS in groups of GR chunks, one group ahead of the MFMAs that consume them
    constexpr int GR = NCH >= 16 ? 2 : (NCH >= 8 ? 4 : NCH / 2);
    static_assert(NCH % GR == 0, "a group never straddles two tiles");
    f32x4 cfirst[GR];            // first group of the step about to be multiplied
    auto load_first = [&](int slot) {
        const char* src = ring + slot * STEP_B + lane * 16;
#pragma unroll
        for (int j = 0; j < GR; ++j) cfirst[j] = *reinterpret_cast<const f32x4*>(src + j * 1024);
    };
    auto compute = [&](int64_t j, int slot, int slot_next, bool has_next) {
        const char* src = ring + slot * STEP_B + lane * 16;
        f32x16 acc[TT][UW];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int u = 0; u < UW; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][u][r] = 0.0f;
        f32x4 c[2][GR];
#pragma unroll
        for (int jj = 0; jj < GR; ++jj) c[0][jj] = cfirst[jj];
#pragma unroll
        for (int g = 0; g < SCH / GR; ++g) {
            if (g + 1 < SCH / GR) {
#pragma unroll
                for (int jj = 0; jj < GR; ++jj)
                    c[(g + 1) & 1][jj] = *reinterpret_cast<const f32x4*>(src + ((g + 1) * GR + jj) * 1024);
            } else if (has_next) {
                load_first(slot_next);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < GR; ++jj)
                Elem<T>::template mma<UW>(acc[(g * GR + jj) / NCH], c[g & 1][jj], b[(g * GR + jj) % NCH]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (CRH_ABLATE(a.ablate) & 1) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int u = 0; u < UW; ++u) {
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" ::"v"(acc[tt][u]));
#endif
                }
            return;
        }
        if (!live) return;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const float m = max16(acc[tt][u]);
                if (__ballot(m > tau[u]) != 0ull)
                    tile_slow_path<UPW>(acc[tt][u], tau[u], w, K, 32 * u, ug * UPW + 32 * u, a,
                                        (t0 + j * TT + tt) << 5, split_end, lane, true);
            }
    };

    if (n_steps > 0) {
        f32x4 sa[CPW], sb[CPW];
        fetch(sa, 0);
        fetch(sb, 1);
        commit(sa, 0);                  // step 0 -> slot 0
        commit(sb, 1);                  // step 1 -> slot 1
        fetch(sa, 2);                   // stage A: step 2
        fetch(sb, 3);                   // stage B: step 3
        lds_barrier();
        load_first(0);
        int s0 = 0;                     // slot of step j
        // XCD soft lockstep (see xcd_window_sync): wave 0 reports / waits for the workgroup, the others meet it at the
        // step's barrier.  Without it the 32 workgroups of an XCD drift apart until their tiles no longer share the L2:
        // 4.5 TB of fabric-side reads per launch at the S-DN shape against a 25.6 GB table (profiles/r02_a_legs_pmc.json).
        unsigned* sync_cnt = nullptr;
        int64_t win_steps = 0, next_sync = n_steps + 2;
        if (a.xcd_sync && wave == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // HW_REG_XCC_ID
            sync_cnt = a.xcd_sync + (int64_t)xcc * a.sync_stride;
            if (lane == 0) __hip_atomic_fetch_add(sync_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            win_steps = (a.sync_window / TT) & ~(int64_t)1;      // whole loop trips (two steps each)
            if (win_steps < 2) win_steps = 2;
            next_sync = 0;
        }
        for (int64_t j = 0; j < n_steps; j += 2) {
            if (j >= next_sync) {       // wave 0 only (next_sync stays beyond the range elsewhere)
                const int64_t win = j / win_steps;
                if (xcd_window_sync(sync_cnt, win, lane)) {
                    next_sync += win_steps;
                } else {                // timed out: run free, and count this workgroup into every window it will not report
                    next_sync = n_steps + 2;
                    const int64_t n_win = (n_steps + win_steps - 1) / win_steps;
                    if (lane == 0)
                        for (int64_t wdw = win; wdw < n_win; ++wdw)
                            __hip_atomic_fetch_add(sync_cnt + 1 + wdw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
            commit(sa, s2);             // step j+2 -> the slot step j-1 left before the last barrier
            fetch(sa, j + 4);
            compute(j, s0, s1, j + 1 < n_steps);
            lds_barrier();
            if (j + 1 >= n_steps) break;
            commit(sb, s0);             // step j+3 -> slot of step j
            fetch(sb, j + 5);
            compute(j + 1, s1, s2, j + 2 < n_steps);
            lds_barrier();
            s0 = s2;
        }
    }

    if (live) {
        for (int j = 0; j < UPW; ++j) {
            const int64_t slot = ug * UPW + j;
            if (slot >= a.n_users) break;
            const int n = __builtin_amdgcn_readfirstlane(w.cnt[j]);
            const int64_t o = ((int64_t)split * a.n_users + slot) * K;
            if (a.seed_score && S > 1)
                wave_list_store_range(w.ls + j * K, w.li + j * K, n, K, a.out_score + o, a.out_idx + o, lane,
                                      split == 0 ? INT32_MIN : (int)(a.item_base + (t0 << 5)), (int)(a.item_base + split_end));
            else
                wave_list_store(w.ls + j * K, w.li + j * K, n, K, a.out_score + o, a.out_idx + o, lane);
        }
    }
}

// item tiles per ring slot of the workgroup kernel: ~32 MFMAs per wave and step whatever the row width
constexpr int wg_tiles_per_step(int row_bytes) { return row_bytes <= 128 ? 4 : (row_bytes <= 256 ? 2 : 1); }

template <typename T, int D, int UW, int NW>
int launch_score_wg(const ScoreArgs& a, hipStream_t stream) {
    constexpr int TT = wg_tiles_per_step(D * (int)sizeof(T));
    const size_t lds = (size_t)WG_RING * TT * (D * sizeof(T) / 32) * 1024 + NW * wave_lds_bytes<32 * UW>(a.k);
    auto kern = score_topk_wg_kernel<T, D, UW, NW, TT>;
    CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    const int64_t blocks = ((a.n_ugroups + NW - 1) / NW) * a.n_splits;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), lds, stream, a);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

template <typename T, int D, int UW, int OCC, bool PK, bool DN>
int launch_score_pk(const ScoreArgs& a, hipStream_t stream);

template <typename T, int D, int UW, int OCC>
int launch_score(const ScoreArgs& a, hipStream_t stream) {
    if (a.dense) return a.packed ? launch_score_pk<T, D, UW, OCC, true, true>(a, stream) : launch_score_pk<T, D, UW, OCC, false, true>(a, stream);
    if (a.packed) return launch_score_pk<T, D, UW, OCC, true, false>(a, stream);
    return launch_score_pk<T, D, UW, OCC, false, false>(a, stream);
}

template <typename T, int D, int UW, int OCC, bool PK, bool DN>
int launch_score_pk(const ScoreArgs& a, hipStream_t stream) {
    constexpr int UPW = 32 * UW;
    constexpr int WPW = 1;
    const size_t lds = DN ? 0 : wave_lds_bytes<UPW>(a.k) * WPW;   // (a block launch keeps no lists)
    auto kern = score_topk_kernel<T, D, UW, OCC, PK, DN>;
    if (lds > 64 * 1024)
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t vblocks = a.n_ugroups * a.n_splits;
    const int64_t blocks = (vblocks + WPW - 1) / WPW;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WPW), lds, stream, a);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// users per wave = 32*UW of the instantiation the dispatcher below picks for (element bytes, d)
constexpr int users_per_wave(int esz, int d) {
    return esz == 4 ? (d >= 256 ? 32 : (d >= 128 ? 64 : 128)) : (d >= 256 ? 64 : 128);
}

// Item-range split count.  Workgroups are single waves that all take the same time, and the
// dispatcher packs them occ-per-SIMD CU by CU (measured: a grid of half the capacity runs on half
// the CUs), so the grid should fill a whole number of "rounds" of capacity = 256 CUs x 4 SIMDs x occ.
// Every extra split costs extra slow-path inserts (~k*ln(range/k) per user and split); the model
// below (5700*S/T relative overhead, fitted on the 10M-item run) trades that against idle slots.
int pick_splits(int64_t n_ugroups, int64_t n_items, int occ) {
    const double cap = 1024.0 * occ;
    const int64_t T = (n_items + 31) / 32;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 64; ++s) {
        if (s > 1 && T / s < 64) break;                       // keep >= 64 tiles per split
        const double w = (double)n_ugroups * s;
        const double rounds = ceil(w / cap);
        const double eff = w / (rounds * cap);
        const double cost = (1.0 + 5700.0 * s / (double)T) / eff;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

// Split count of the workgroup kernel: n_wg workgroups of `slots` per round (one or two per CU).  Measured on the fp16
// d=256 kernel (tools/f16_probe.py): a round filled to 77 % / 50 % takes 0.85 / 0.76 of a full one (the chip is
// power-limited: idle CUs are clock headroom for the others), every extra cut costs ~6 % at 10 M items and k=20 (the
// top-k warm-up of each user repeats per cut; cut_cost = that share x tiles: 18750 fp16, 5700 at the fp32 MFMA rate) and cuts stream different tile ranges through the same L2 (~+10 %).
int pick_splits_wg(int64_t n_wg, int64_t n_items, int k, int slots, double cut_cost) {
    const int64_t T = (n_items + 31) / 32;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 64; ++s) {
        if (s > 1 && T / s < 64) break;
        const double w = (double)n_wg * s;
        const double rounds = ceil(w / slots);
        const double fill = (w - (rounds - 1) * slots) / slots;
        const double cost = ((rounds - 1) + 0.55 + 0.45 * fill) / s * (1.0 + cut_cost * (k / 20.0) * s / (double)T) *
                            (s > 1 ? 1.1 : 1.0);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

}  // namespace

extern "C" int crh_score_topk_supports_dim(int d) {
    return d == 8 || d == 16 || d == 32 || d == 64 || d == 128 || d == 256;
}

extern "C" int crh_score_topk_f16_supports_dim(int d) { return d == 16 || d == 32 || d == 64 || d == 128 || d == 256; }

namespace {
size_t lists_bytes(int64_t n_users, int k) {   // worst case: 64 splits of (score, idx), 256-B aligned
    return (((size_t)64 * (size_t)n_users * (size_t)k * 8) + 255) & ~(size_t)255;
}
size_t packed_bytes(int64_t n_items, int d, int esz) { return (size_t)((n_items + 31) / 32) * 32 * (size_t)d * esz; }
// window counters of the XCD lockstep: 8 XCDs x (1 + one per 8-tile window, the smallest window allowed)
size_t tbits_bytes(int64_t n_items) { return (((size_t)((n_items + 31) / 32) * sizeof(uint32_t)) + 255) & ~(size_t)255; }
size_t sync_bytes(int64_t n_items) { return ((size_t)((n_items + 255) / 256 + 2) * 8 * sizeof(unsigned) + 255) & ~(size_t)255; }

int pack_items(int esz, const void* item_emb, int64_t n_items, int d, void* pk, hipStream_t st, const uint32_t* bitmap,
               int64_t item_base) {
    const unsigned tiles = (unsigned)((n_items + 31) / 32);
    const float* vf = reinterpret_cast<const float*>(item_emb);
    float* pf = reinterpret_cast<float*>(pk);
    const _Float16* vh = reinterpret_cast<const _Float16*>(item_emb);
    _Float16* ph = reinterpret_cast<_Float16*>(pk);
#define CRH_PACK(KERN, V, P) hipLaunchKernelGGL(KERN, dim3(tiles), dim3(256), 0, st, V, n_items, P, bitmap, item_base)
    if (esz == 4) {
        switch (d) {
            case 8: CRH_PACK(pack_items_kernel<8>, vf, pf); break;
            case 16: CRH_PACK(pack_items_kernel<16>, vf, pf); break;
            case 32: CRH_PACK(pack_items_kernel<32>, vf, pf); break;
            case 64: CRH_PACK(pack_items_kernel<64>, vf, pf); break;
            case 128: CRH_PACK(pack_items_kernel<128>, vf, pf); break;
            default: CRH_PACK(pack_items_kernel<256>, vf, pf); break;
        }
    } else {
        switch (d) {
            case 16: CRH_PACK(pack_items_f16_kernel<16>, vh, ph); break;
            case 32: CRH_PACK(pack_items_f16_kernel<32>, vh, ph); break;
            case 64: CRH_PACK(pack_items_f16_kernel<64>, vh, ph); break;
            case 128: CRH_PACK(pack_items_f16_kernel<128>, vh, ph); break;
            default: CRH_PACK(pack_items_f16_kernel<256>, vh, ph); break;
        }
    }
#undef CRH_PACK
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// The per-wave kernel for (element size, d, waves per SIMD).
int launch_score_per_wave(int esz, int d, int occ, const ScoreArgs& a, hipStream_t st) {
    if (esz == 4) {
        switch (d) {
            case 8: return launch_score<float, 8, 4, 1>(a, st);
            case 16: return launch_score<float, 16, 4, 1>(a, st);
            case 32: return launch_score<float, 32, 4, 1>(a, st);
            case 64: return launch_score<float, 64, 4, 1>(a, st);
            case 128: return occ == 2 ? launch_score<float, 128, 2, 2>(a, st) : launch_score<float, 128, 2, 1>(a, st);
            default: return launch_score<float, 256, 1, 1>(a, st);
        }
    }
    switch (d) {
        case 16: return launch_score<_Float16, 16, 4, 2>(a, st);
        case 32: return launch_score<_Float16, 32, 4, 2>(a, st);
        case 64: return launch_score<_Float16, 64, 4, 2>(a, st);
        case 128: return launch_score<_Float16, 128, 4, 2>(a, st);
        default: return launch_score<_Float16, 256, 2, 2>(a, st);
    }
}

// fp32 catalogues from here on take a workgroup kernel (when the users fill its 512-user workgroups), shorter ones the per-wave
// kernel with seeded lists.  Same box, 131 072 users, seeded per-wave / workgroup (LDS-DMA): 2 M items 0.888 / 0.864 (unseeded),
// 3 M 0.8835 / 0.8952 (seeded), 4 M 0.8876 / 0.9052, 10 M 0.898 (unseeded) / 0.9285.
constexpr int64_t FP32_WG_MIN_ITEMS = 2500000;
// Round 6, same box, 131 072 users, seeded, fraction of the fp32 MFMA peak -- LDS-DMA kernel in its flag form / its barrier form /
// per-wave kernel (profiles/r06_flag_vs_barrier_vs_wave_ab.log):
//   d=128   1.25 M 0.880 / 0.857 / 0.876   1.8 M 0.895 / 0.879 / 0.882   2.5 M 0.903 / 0.893 / 0.889   5 M 0.915 / 0.915 / 0.893
//           10 M 0.920 / 0.927 / 0.898
//   d=64    1.25 M 0.810 / 0.772 / 0.823   2.5 M 0.840 / 0.823 / 0.852   5 M 0.858 / 0.859 / 0.868   10 M 0.870 / 0.883 / 0.879
// d=128: the DMA kernel from 1.2 M items (one rank's shard of the 8-GPU split of the headline is 1.25 M), flag form below 6 M
// items -- where a user's ~k ln(N / P) slow-path events are a visible share and the barrier form pays each of them four times --
// barrier form above.  d=64 (one wave of 128 users per SIMD, nothing shared, nothing to wait for): the per-wave kernel is as
// fast or faster up to 5 M items; the DMA kernel (barrier form) from 7.5 M.
constexpr int64_t FP32_DMA_MIN_ITEMS_D128 = 1200000, FP32_DMA_FLAG_MAX_ITEMS = 6000000, FP32_DMA_MIN_ITEMS_D64 = 7500000;
// Small and mid-size blocks are scored into a dense block and ranked by crh_mask_topk_f32 (see score_topk_any).
constexpr int64_t DENSE_MAX_ITEMS = 262144;
// users go in chunks when the block would exceed this: 8 GiB (1 GiB chunks ran the same shapes at 0.37-0.43 of the MFMA peak
// instead of 0.52: too few rows per ranking launch to fill the CUs; tools/midsize_probe.py d<MB> routes)
constexpr size_t DENSE_MAX_BLOCK = (size_t)8 << 30;
// Which shapes.  The dense route costs 8 bytes of HBM traffic per (user, item) pair whatever the shape: 0.38 - 0.44 of
// the fp32 MFMA peak at d=128.  The fused selection pays k (1 + ln(N/k)) slow-path events of ~1.6 us per user (and that
// again per item-range cut when there are too few user groups to fill the chip), against N/32 tiles of MFMA work.
// Measured crossover on one MI355X, d=128, masks on (tools/midsize_probe.py, gpurun_out -> profiles/r02_midsize_probe.log):
//   users x items   dense    fused (per-wave kernel)
//    8192 x 262144   8.8 ms  15.1 ms        16384 x 131072   9.2 ms  15.0 ms       16384 x 524288  43.8 ms  29.6 ms
//   32768 x 131072  18.3 ms  20.5 ms        32768 x 262144  36.1 ms  29.7 ms       65536 x  65536  15.9 ms  20.3 ms
//   65536 x 131072  36.2 ms  29.7 ms       131072 x 131072  72.1 ms  46.3 ms
// i.e. dense up to ~5e9 pairs per call; any user count up to 32 768 items (the trainers' validation shapes; with 8 GiB
// blocks 262 144 / 524 288 users x 32 768 items run at 0.47 dense against 0.44 / 0.45 fused).
size_t dense_block_bytes(int64_t n_users, int64_t n_items) {
    static const int64_t max_items = CRH_TUNE_ENV("CRH_SCORE_DENSE_MAX_ITEMS") ? atoll(CRH_TUNE_ENV("CRH_SCORE_DENSE_MAX_ITEMS")) : DENSE_MAX_ITEMS;
    static const double max_pairs = CRH_TUNE_ENV("CRH_SCORE_DENSE_MAX_PAIRS") ? atof(CRH_TUNE_ENV("CRH_SCORE_DENSE_MAX_PAIRS")) : 5e9;
    if (n_items > max_items || (n_items > 32768 && (double)n_users * (double)n_items > max_pairs)) return 0;
    const size_t row = (size_t)((n_items + 31) / 32) * 32 * sizeof(float);
    const size_t all = (size_t)n_users * row;
    // read per call (like CRH_SCORE_WG): the chunking tests lower it to cut a small block into many user chunks
    const char* mb_env = getenv("CRH_SCORE_DENSE_BLOCK_MB");
    const size_t max_block = mb_env && atoll(mb_env) > 0 ? (size_t)atoll(mb_env) << 20 : DENSE_MAX_BLOCK;
    if (all <= max_block) return (all + 255) & ~(size_t)255;
    const size_t groups = std::max<size_t>(1, max_block / (row * 64));
    return (groups * 64 * row + 255) & ~(size_t)255;       // whole 64-user groups
}

// Prefix length of the seeded route (0 = not applicable): 1/16 of the catalogue in 4 096 .. 16 384 items; up to 16 384 users 1/8 of it,
// capped at 32 768 (65 536 up to 4 096 users); whole tiles.
// After a prefix of P items a user takes ~k ln(N / P) candidates through the slow path instead of k (1 + ln(N / k)), and
// every item-range cut shares that total instead of repeating the warm-up; the prefix itself is ranked by the dense route
// at 8 bytes of traffic per (user, item) pair.
// Few users make the prefix's dense block cheap, so they take a longer one (tools/seed_prefix_sweep.sh, round 4: 8 192 x 262 144
// 0.666 -> 0.679 at 32 768 items, 4 096 x 10 M 0.815 -> 0.827 at 65 536, 16 384 x 1 M 0.787 -> 0.793 at 32 768; 65 536 x 131 072 is
// best at its 8 192).
int64_t seed_prefix_items(int64_t n_items, int64_t n_users) {
    if (n_items < 65536) return 0;
    const char* pe = CRH_TUNE_ENV("CRH_SCORE_SEED_ITEMS");                  // tuning hook (read per call)
    const bool few = n_users <= 16384;
    int64_t p = pe && atoll(pe) >= 1024 ? std::min<int64_t>(atoll(pe), n_items / 4) : n_items / (few ? 8 : 16);
    if (!pe) p = std::max<int64_t>(4096, std::min<int64_t>(n_users <= 4096 ? 65536 : (few ? 32768 : 16384), p));
    // the longer prefixes of small user blocks are worth 0.6-2 % of the peak: not more than 1 GiB of stage-1 score block for
    // them (a trainer keeps the workspace resident beside its training state)
    while (!pe && p > 16384 && (size_t)n_users * (size_t)p * sizeof(float) > ((size_t)1 << 30)) p >>= 1;
    return p & ~(int64_t)31;
}
size_t seed_bytes(int64_t n_users, int k) { return (((size_t)n_users * k * 8) + 255) & ~(size_t)255; }

// Split count under seeded lists: cuts are (nearly) free, so fill whole rounds of wave slots; a small per-cut charge for the
// filter rebuild, the seed copy and the merge.
int pick_splits_seeded(int64_t n_units, int64_t n_items, double cap) {
    const int64_t T = (n_items + 31) / 32;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 64; ++s) {
        if (s > 1 && T / s < 64) break;
        const double w = (double)n_units * s;
        const double rounds = ceil(w / cap);
        const double cost = rounds * cap / w * (1.0 + 0.004 * s);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

// Everything the dispatcher decides about one stage of a call BEFORE it touches the GPU -- shared with crh_score_topk_route,
// so that a caller (bench.py's kernel labels, tests) asks the library which kernels a shape takes instead of guessing.
struct RoutePlan {
    bool dense;        // score block + crh_mask_topk_f32
    bool use_wg;       // a workgroup kernel (register-staged ring, or the LDS-DMA form when use_dma)
    bool use_dma;
    int dma_mode;      // form of the DMA kernel: 2 = flag form (fp32 d=128), 3 = barrier form
    bool can_pack;     // the workspace holds the fragment-ordered copy of the shard
    int wg_waves_l;    // waves of the workgroup that is launched
    int wg_slots_l;    // workgroups resident per round
    int upw;           // users per wave
    int occ;           // waves per SIMD of the per-wave kernel
    size_t dense_b;    // bytes of the dense block (0: not applicable)
    size_t tb_off;     // workspace offset of the tiles' candidate bits (DMA kernel)
    int no_pack;
};

RoutePlan plan_route(int esz, int64_t n_users, int64_t n_items, int d, int k, bool has_workspace, size_t workspace_bytes,
                     bool has_bitmap, int n_splits, bool seeded) {
    RoutePlan r;
    // The workgroup-cooperative kernel (8 waves share the packed item tiles through LDS) once there are enough
    // user groups to fill the CUs and the workspace holds the packed copy.  fp16: d = 64/128/256, 64 users per
    // wave.  fp32: d = 128 (64 users per wave; +1.5 % over the per-wave kernel and 1/8 of its L2 -> CU traffic).
    static const int no_pack = CRH_TUNE_ENV("CRH_SCORE_NO_PACK") ? atoi(CRH_TUNE_ENV("CRH_SCORE_NO_PACK")) : 0;
    const int wg_mode = getenv("CRH_SCORE_WG") ? atoi(getenv("CRH_SCORE_WG")) : 1;   // per call: tests force 2
    const bool can_pack = !no_pack && has_workspace &&
                          workspace_bytes >= lists_bytes(n_users, k) + packed_bytes(n_items, d, esz);
    // one 8-wave workgroup per CU when its lists fit the 160 KiB of LDS, else two 4-wave workgroups (large k);
    // measured at k=20 (fp16): 8 waves 50.6 % / 45.7 % of the fp16 peak at d=256 / 128, two 4-wave groups
    // 46.5 % / 47.4 % (CRH_SCORE_WG=4 forces them)
    const size_t ring_b = (size_t)WG_RING * wg_tiles_per_step(d * esz) * (d * esz / 32) * 1024;
    const size_t wg4_lds = ring_b + 4 * wave_lds_bytes<64>(k), wg8_lds = ring_b + 8 * wave_lds_bytes<64>(k);
    // (Round 4 measured a 128-users-per-wave / one-wave-per-SIMD instantiation of this kernel -- <_Float16, 256, UW = 4, NW = 4>,
    // 256 VGPRs + 193 AGPRs, one LDS read of an A fragment per four MFMAs: +1.4 % over the shipped form on the same box, and
    // 0.70x with a second accumulator set that lets the threshold test ride in the next tile's MFMA shadow (B fragments 256 +
    // accumulators 128 registers: the allocator spills inside the MFMA stream, also with the slow path as a real call):
    // profiles/r04_f16_variant_ab_*.log, DESIGN.md 4.1.  Not shipped.)
    const int wg_waves = (wg_mode != 4 && wg8_lds <= 160 * 1024) ? 8 : 4;
    const int wg_upw = 64;
    const int wg_slots = wg_waves == 8 ? 256 : 512;                   // workgroups resident per round
    // 512- and 256-byte rows (fp16 d=256 / 128, fp32 d=128 / 64): the LDS-DMA form (four waves of 128 users, score_topk_dma_kernel)
    // wherever a workgroup kernel runs and its lists fit beside the four ring slots (k <= 20 at 512-byte rows, k <= 28 at 256).
    // Same box, 131 072 x 10 M: fp16 d=256 0.577 vs 0.535 for the ring kernel, fp32 d=128 0.920 vs 0.910.
    // CRH_SCORE_DMA (read per call): 0 never, anything else (default) on
    const int dma_mode = getenv("CRH_SCORE_DMA") ? atoi(getenv("CRH_SCORE_DMA")) : 1;
    const size_t tb_off = lists_bytes(n_users, k) + packed_bytes(n_items, d, esz) + sync_bytes(n_items);   // the tiles' candidate bits
    // (fp16 d=128 was measured on it too, round 6: 0.482 against 0.479 of 2.5 PF for the ring kernel at 10 M items, 0.305
    // against 0.347 at 1.25 M -- a 1 024-cycle tile is too short for one barrier + one threshold test each: not routed)
    const bool dma_rows = esz == 2 ? d == 256 : (d == 128 || d == 64);
    // the form of the fp32 DMA kernel: CRH_SCORE_DMA = 2 flag form, 3 barrier form, anything else by stream length
    const int dma_form = dma_mode == 2 || dma_mode == 3 ? dma_mode : (esz == 4 && d == 128 && n_items < FP32_DMA_FLAG_MAX_ITEMS ? 2 : 3);
    const bool dma_shape = dma_mode && dma_rows && score_dma_ring_slots(esz, d, k, dma_form) > 0 &&
                           (!has_bitmap || workspace_bytes >= tb_off + tbits_bytes(n_items));
    // fp32 d=64 (the reference's default width) has no register-staged ring kernel: a workgroup kernel only as the DMA form
    const bool wg_shape = esz == 2 ? (d == 64 || d == 128 || d == 256) : ((d == 128 && wg_waves == 8) || (d == 64 && dma_shape));
    // fp32: the lockstep of 8 waves makes every slow-path event a stall of the whole CU, so the workgroup kernel only
    // pays on long streams (FP32_WG_MIN_ITEMS) and when its 512-user workgroups fill the CUs (65 536 users = 128 workgroups ran
    // at 0.33 against 0.60 per wave)
    const int64_t n_wg64 = ((n_users + 63) / 64 + wg_waves - 1) / wg_waves;
    const int64_t fp32_min_items = !dma_shape ? FP32_WG_MIN_ITEMS : (d == 64 ? FP32_DMA_MIN_ITEMS_D64 : FP32_DMA_MIN_ITEMS_D128);
    const bool fp32_wg_ok = n_items >= fp32_min_items && (double)n_wg64 / (double)(((n_wg64 + 255) / 256) * 256) >= 0.9;
    const bool use_wg = wg_mode && can_pack && wg_shape && (dma_shape || wg_waves == 8 || 2 * wg4_lds <= 160 * 1024) &&
                        (n_users + 63) / 64 >= 512 && (esz == 2 || fp32_wg_ok || wg_mode == 2);
    const bool use_dma = use_wg && dma_shape;
    const int wg_waves_l = use_dma ? 4 : wg_waves;                    // waves of the workgroup that is launched
    const int wg_slots_l = use_dma ? 256 : wg_slots;                  // ... and how many of them are resident per round
    const int upw = use_dma ? 128 : (use_wg ? wg_upw : users_per_wave(esz, d));
    // two waves per SIMD hide the selection / slow path behind the partner's MFMAs (+8 % measured at fp32
    // d=128); CRH_SCORE_OCC=1 selects the double-buffered one-wave-per-SIMD fp32 build (tuning hook)
    static const int variant = CRH_TUNE_ENV("CRH_SCORE_OCC") ? atoi(CRH_TUNE_ENV("CRH_SCORE_OCC")) : 2;
    // (fp32 d=64 at two waves per SIMD -- 128 users per wave, <float, 64, 4, 2> -- spills 13 registers into its tile loop: 131 072 x
    // 10 M ran at 0.466 of the fp32 MFMA peak against 0.884 for the one-wave form, 131 072 x 1.25 M 0.442 against 0.824; round 6)
    const int occ = esz == 2 ? 2 : ((variant == 2 && d == 128) ? 2 : 1);
    // Small catalogues (the trainers' per-epoch validation: a few thousand users x a few thousand items).  The fused
    // selection is built for catalogues where a candidate above the running threshold is rare; here every user takes
    // ~k (1 + ln(N/k)) candidates through the wave-serial slow path (5 - 8 ms for 5 K users x 4 K items, a single
    // wave of 64 users being the critical path).  Instead: the same MFMA kernel writes its score tiles to a dense
    // block (bit-identical scores) and crh_mask_topk_f32 -- one wave per user, same masks, same canonical order --
    // ranks it; users go in chunks if the block would pass 8 GiB.  CRH_SCORE_DENSE=0 keeps the fused selection.
    static const int dense_mode = CRH_TUNE_ENV("CRH_SCORE_DENSE") ? atoi(CRH_TUNE_ENV("CRH_SCORE_DENSE")) : 1;
    const size_t dense_b = dense_block_bytes(n_users, n_items);
    r.dense = dense_mode && n_splits == 0 && !seeded && dense_b && has_workspace && workspace_bytes >= dense_b;
    r.use_wg = use_wg;
    r.use_dma = use_dma;
    r.dma_mode = dma_form;
    r.can_pack = can_pack;
    r.wg_waves_l = wg_waves_l;
    r.wg_slots_l = wg_slots_l;
    r.upw = upw;
    r.occ = occ;
    r.dense_b = dense_b;
    r.tb_off = tb_off;
    r.no_pack = no_pack;
    return r;
}

// item-range cut count of a fused stage
int plan_splits(const RoutePlan& r, int esz, int64_t n_users, int64_t n_items, int k, int n_splits, bool seeded) {
    const int64_t T = (n_items + 31) / 32;
    const int64_t n_ug = (n_users + r.upw - 1) / r.upw;
    int s = n_splits > 0 ? n_splits
            : seeded     ? pick_splits_seeded(r.use_wg ? (n_ug + r.wg_waves_l - 1) / r.wg_waves_l : n_ug, n_items,
                                              r.use_wg ? (double)r.wg_slots_l : 1024.0 * r.occ)
            : r.use_wg   ? pick_splits_wg((n_ug + r.wg_waves_l - 1) / r.wg_waves_l, n_items, k, r.wg_slots_l,
                                          esz == 2 ? 18750.0 : 5700.0)
                         : pick_splits(n_ug, n_items, r.occ);
    if (s > T) s = (int)T;
    return s;
}

int score_topk_impl(int esz, const void* user_emb, const int32_t* users, int64_t n_users, const void* item_emb,
                    int64_t n_items, int d, const int64_t* rated_rowptr, const int32_t* rated_col,
                    const uint32_t* cand_bitmap, int k, int64_t item_base, float* out_score, int32_t* out_idx,
                    void* workspace, size_t workspace_bytes, void* stream, int n_splits, void* ev_kernel_start,
                    void* ev_kernel_stop, const char* who, const float* seed_score, const int32_t* seed_idx);

// Does a call of this shape take the seeded route (n_splits == 0), and with which prefix?  0 = no.  ONE predicate for the
// dispatcher (score_topk_any) and for the workspace query (full_workspace_bytes): a shape that never seeds must not be
// charged the prefix's dense block (131 072 x 262 144 asked for 8.6 GB instead of 1.5 GB).
//   cuts      : the users alone do not fill the wave slots (the unseeded picker would cut the item range);
//   small_cat : fp32 catalogues the per-wave kernel ranks (below 2 M items) whatever the user count: the warm-up insertions
//               of the fused selection (three quarters of k (1 + ln(N / k)) fall into the first 20 000 items) are a visible
//               share of a user's work.  With MANY users the prefix stage is what costs (users x prefix scores written and
//               ranked), and modest seeds do: a 4 096-item prefix -- 131 072 x 262 144: 0.749 -> 0.799 of the fp32-MFMA
//               peak (16 384-item prefix: 0.783), x 1 048 576: 0.842 -> 0.859, x 1 250 000 (one rank's shard of the 8-GPU
//               split): 0.853 -> 0.863 (16 384: 0.838).
// CRH_SCORE_SEED (read per call): 0 never, 1 auto (default), 2 whenever possible; CRH_SCORE_SEED_MAX_ITEMS moves the
// small_cat limit, CRH_SCORE_SEED_ITEMS fixes the prefix for both cases.
int64_t seed_route(int esz, int64_t n_users, int64_t n_items, int d) {
    const char* sm = getenv("CRH_SCORE_SEED");
    const int seed_mode = sm ? atoi(sm) : 1;
    int64_t P = seed_prefix_items(n_items, n_users);
    if (!seed_mode || P <= 0 || n_users <= 0) return 0;
    const int upw = users_per_wave(esz, d);
    const int64_t n_ug = (n_users + upw - 1) / upw;
    const int occ_pw = esz == 2 || d == 128 ? 2 : 1;         // waves per SIMD of the per-wave kernel for this width (plan_route)
    const bool cuts = pick_splits(n_ug, n_items, occ_pw) > 1;
    const char* smi = getenv("CRH_SCORE_SEED_MAX_ITEMS");
    const int64_t f32_stream_min = d == 64 ? FP32_DMA_MIN_ITEMS_D64 : FP32_WG_MIN_ITEMS;
    const bool small_cat = !cuts && esz == 4 && n_items <= (smi ? atoll(smi) : f32_stream_min - 1);
    // fp16, 512-byte rows (configs[4]): at 16x the fp32 MFMA rate a slow-path event costs as much as a whole tile and, in
    // the workgroup kernels, stalls the whole CU: 9 % of the launch at 10 M items (profiles/r05_f16_*).  A 4 096-item prefix
    // takes k (1 + ln(P / k)) of every user's k (1 + ln(N / k)) events out of the stream: 282 -> 156 per user at 10 M items
    const bool f16_stream = !cuts && esz == 2 && d == 256 && n_users >= 32768 && n_items >= ((int64_t)1 << 20);
    // fp32, 512-byte rows on the workgroup kernels (the headline): the same, worth less at the fp32 rate -- 131 072 x 3 M 0.883 ->
    // 0.895, x 4 M 0.895 -> 0.905, x 10 M 0.9245 -> 0.9285 (16 384-item prefix; 4 096 / 8 192 at 10 M: +0.35 / +0.38 %).  8 192 items
    // keep the prefix's score block (4.3 GB at 131 072 users) inside what the packed copy of such a shard takes anyway
    const bool f32_stream = !cuts && esz == 4 && (d == 128 || d == 64) && n_users >= 32768 && n_items >= f32_stream_min;
    if ((small_cat || f16_stream) && !CRH_TUNE_ENV("CRH_SCORE_SEED_ITEMS") && P > 4096) P = 4096;
    if (f32_stream && !small_cat && !CRH_TUNE_ENV("CRH_SCORE_SEED_ITEMS") && P > 8192) P = 8192;
    if (!(cuts || small_cat || f16_stream || f32_stream || seed_mode == 2) || dense_block_bytes(n_users, P) == 0) return 0;
    return P;
}

// esz = 4: fp32 tables, exact fp32 MFMA (canonical fma chain).  esz = 2: fp16 tables, fp32 accumulate.
// Route of a call (n_splits == 0; a caller that names a split count gets the plain fused selection):
//   seeded  : a catalogue of >= 65 536 items and either users that do not fill the chip on their own (the unseeded picker
//             would cut the item range; prefix 1/16 of the catalogue, 4 096 .. 16 384 items) or, fp32, fewer than 2 M items
//             (prefix 4 096 items): rank the prefix by the dense route, then the fused selection over the rest, lists seeded;
//   dense   : small catalogues (score block + wave-per-user ranking);
//   fused   : everything else (the headline).
int score_topk_any(int esz, const void* user_emb, const int32_t* users, int64_t n_users, const void* item_emb,
                   int64_t n_items, int d, const int64_t* rated_rowptr, const int32_t* rated_col,
                   const uint32_t* cand_bitmap, int k, int64_t item_base, float* out_score, int32_t* out_idx,
                   void* workspace, size_t workspace_bytes, void* stream, int n_splits, void* ev_kernel_start,
                   void* ev_kernel_stop, const char* who) {
    // the seeded route's own predicate and prefix (shared with the workspace query: seed_route)
    const int64_t P = n_splits == 0 ? seed_route(esz, n_users, n_items, d) : 0;
    if (P > 0 && user_emb && item_emb && out_score && out_idx && n_users > 0 && k >= 1 && k <= CRH_MAX_K && workspace) {
        const size_t sb = seed_bytes(n_users, k);
        const size_t stage1 = dense_block_bytes(n_users, P) + packed_bytes(P, d, esz);
        const size_t stage2 = lists_bytes(n_users, k) + packed_bytes(n_items - P, d, esz);
        if (workspace_bytes >= sb + std::max(stage1, stage2)) {
            float* seed_s = reinterpret_cast<float*>(workspace);
            int32_t* seed_i = reinterpret_cast<int32_t*>(seed_s + (size_t)n_users * k);
            void* ws2 = reinterpret_cast<char*>(workspace) + sb;
            hipStream_t st = reinterpret_cast<hipStream_t>(stream);
            if (ev_kernel_start) CRH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_kernel_start), st));
            int rc = score_topk_impl(esz, user_emb, users, n_users, item_emb, P, d, rated_rowptr, rated_col, cand_bitmap, k,
                                     item_base, seed_s, seed_i, ws2, workspace_bytes - sb, stream, 0, nullptr, nullptr, who,
                                     nullptr, nullptr);
            if (rc != CRH_OK) return rc;
            rc = score_topk_impl(esz, user_emb, users, n_users,
                                 reinterpret_cast<const char*>(item_emb) + (size_t)P * d * esz, n_items - P, d, rated_rowptr,
                                 rated_col, cand_bitmap, k, item_base + P, out_score, out_idx, ws2, workspace_bytes - sb, stream,
                                 0, nullptr, nullptr, who, seed_s, seed_i);
            if (rc != CRH_OK) return rc;
            if (ev_kernel_stop) CRH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_kernel_stop), st));
            return CRH_OK;
        }
    }
    return score_topk_impl(esz, user_emb, users, n_users, item_emb, n_items, d, rated_rowptr, rated_col, cand_bitmap, k,
                           item_base, out_score, out_idx, workspace, workspace_bytes, stream, n_splits, ev_kernel_start,
                           ev_kernel_stop, who, nullptr, nullptr);
}

#ifdef CRH_PROFILE
// CRH_SCORE_TIMING (profile build): per-wave start/end distribution of a per-wave-kernel launch (100 MHz wall clock), its phases
// and what its slow-path events took, printed to stderr; dev = [wave][start,end], [wave][loop start,end], [wave][event ticks,events]
int print_wave_timing(unsigned long long* dev, int64_t n_waves, int timing, hipStream_t st) {
        CRH_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)n_waves * 6);
        CRH_HIP(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
        CRH_HIP(hipFree(dev));
        if (timing == 2)   // raw (start, end) pairs of the LAST launch, wave-major, for offline analysis
            if (FILE* f = fopen("/tmp/crh_wave_clock.bin", "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int64_t wv = 0; wv < n_waves; ++wv) { t0 = std::min(t0, h[2 * wv]); t1 = std::max(t1, h[2 * wv + 1]); }
        std::vector<double> st_(n_waves), en_(n_waves);
        for (int64_t wv = 0; wv < n_waves; ++wv) { st_[wv] = (double)(h[2 * wv] - t0); en_[wv] = (double)(h[2 * wv + 1] - t0); }
        std::sort(st_.begin(), st_.end());
        std::sort(en_.begin(), en_.end());
        const double tot = (double)(t1 - t0);
        fprintf(stderr, "[crh timing] waves=%lld span=%.0f ticks; start quantiles 0/25/50/75/100%%: %.3f %.3f %.3f %.3f %.3f; "
                        "end quantiles 0/10/25/50/75/90/100%%: %.3f %.3f %.3f %.3f %.3f %.3f %.3f (fraction of span)\n",
                (long long)n_waves, tot, st_[0] / tot, st_[n_waves / 4] / tot, st_[n_waves / 2] / tot,
                st_[3 * n_waves / 4] / tot, st_[n_waves - 1] / tot, en_[0] / tot, en_[n_waves / 10] / tot,
                en_[n_waves / 4] / tot, en_[n_waves / 2] / tot, en_[3 * n_waves / 4] / tot, en_[9 * n_waves / 10] / tot,
                en_[n_waves - 1] / tot);
        // phases of a wave (medians, 10 ns ticks) and what its slow-path events took: [2n..4n) loop start/end, [4n..6n) ticks/events
        std::vector<double> in_(n_waves), lp_(n_waves), so_(n_waves);
        double ev_ticks = 0, ev = 0, lp_sum = 0;
        for (int64_t wv = 0; wv < n_waves; ++wv) {
            in_[wv] = (double)(h[2 * n_waves + 2 * wv] - h[2 * wv]);
            lp_[wv] = (double)(h[2 * n_waves + 2 * wv + 1] - h[2 * n_waves + 2 * wv]);
            so_[wv] = (double)(h[2 * wv + 1] - h[2 * n_waves + 2 * wv + 1]);
            ev_ticks += (double)h[4 * n_waves + 2 * wv];
            ev += (double)h[4 * n_waves + 2 * wv + 1];
            lp_sum += lp_[wv];
        }
        std::sort(in_.begin(), in_.end());
        std::sort(lp_.begin(), lp_.end());
        std::sort(so_.begin(), so_.end());
        fprintf(stderr, "[crh timing] median ticks: list setup %.0f, tile loop %.0f, list store %.0f; %.0f slow-path events per wave, "
                        "%.0f ticks each = %.3f of the loop\n",
                in_[n_waves / 2], lp_[n_waves / 2], so_[n_waves / 2], ev / (double)n_waves, ev > 0 ? ev_ticks / ev : 0.0,
                lp_sum > 0 ? ev_ticks / lp_sum : 0.0);
    return CRH_OK;
}
#endif

int score_topk_impl(int esz, const void* user_emb, const int32_t* users, int64_t n_users, const void* item_emb,
                    int64_t n_items, int d, const int64_t* rated_rowptr, const int32_t* rated_col,
                    const uint32_t* cand_bitmap, int k, int64_t item_base, float* out_score, int32_t* out_idx,
                    void* workspace, size_t workspace_bytes, void* stream, int n_splits, void* ev_kernel_start,
                    void* ev_kernel_stop, const char* who, const float* seed_score, const int32_t* seed_idx) {
    CRH_CHECK_ARG(user_emb && item_emb && out_score && out_idx, "%s: NULL table/output pointer", who);
    CRH_CHECK_ARG(n_users > 0 && n_items > 0, "%s: empty block (n_users=%lld, n_items=%lld)", who,
                  (long long)n_users, (long long)n_items);
    CRH_CHECK_ARG(k >= 1 && k <= CRH_MAX_K, "%s: k=%d outside 1..%d", who, k, CRH_MAX_K);
    CRH_CHECK_ARG(esz == 4 ? crh_score_topk_supports_dim(d) : crh_score_topk_f16_supports_dim(d),
                  "%s: d=%d unsupported (pad the tables to %s16/32/64/128/256)", who, d, esz == 4 ? "8/" : "");
    CRH_CHECK_ARG(((uintptr_t)user_emb & 15) == 0 && ((uintptr_t)item_emb & 15) == 0,
                  "%s: tables must be 16-byte aligned", who);
    CRH_CHECK_ARG((rated_rowptr == nullptr) == (rated_col == nullptr) || rated_rowptr != nullptr,
                  "%s: rated_col given without rated_rowptr", who);
    CRH_CHECK_ARG(item_base >= 0 && item_base + n_items < (int64_t)CRH_PAD_IDX, "%s: item ids exceed int32", who);
    CRH_CHECK_ARG(n_splits >= 0 && n_splits <= 64, "%s: n_splits=%d outside 0..64", who, n_splits);

    const RoutePlan rp = plan_route(esz, n_users, n_items, d, k, workspace != nullptr, workspace_bytes, cand_bitmap != nullptr,
                                    n_splits, seed_score != nullptr);
    const bool use_wg = rp.use_wg, use_dma = rp.use_dma, can_pack = rp.can_pack;
    const int wg_waves_l = rp.wg_waves_l, upw = rp.upw, occ = rp.occ, no_pack = rp.no_pack;
    const size_t tb_off = rp.tb_off, dense_b = rp.dense_b;
    ScoreArgs a;
    a.user_emb = user_emb;
    a.users = users;
    a.n_users = n_users;
    a.item_emb = item_emb;
    a.n_items = n_items;
    a.rated_rowptr = rated_rowptr;
    a.rated_col = rated_col;
    a.bitmap = cand_bitmap;
    a.k = k;
    a.item_base = item_base;
    a.n_ugroups = (n_users + upw - 1) / upw;
    static const int ablate = CRH_PROFILE_ENV("CRH_SCORE_ABLATE");
    a.ablate = ablate;
    a.dense = nullptr;
    a.dense_stride = 0;
    a.user_base = 0;
    a.seed_score = seed_score;
    a.seed_idx = seed_idx;
    const int64_t T = (n_items + 31) / 32;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    if (rp.dense) {   // small catalogues: score block + wave-per-user ranking (see plan_route)
        const int64_t stride = T * 32;
        const int upw_pw = users_per_wave(esz, d);
        int64_t chunk = (int64_t)(dense_b / ((size_t)stride * sizeof(float)));
        if (chunk > n_users) chunk = n_users;
        a.packed = nullptr;
        if (!no_pack && workspace_bytes >= dense_b + packed_bytes(n_items, d, esz)) {
            void* pk = reinterpret_cast<char*>(workspace) + dense_b;
            const int prc = pack_items(esz, item_emb, n_items, d, pk, st, cand_bitmap, item_base);
            if (prc != CRH_OK) return prc;
            a.packed = pk;
        }
        a.n_splits = 1;
        a.xcd_sync = nullptr;
        a.sync_window = 0;
        a.sync_stride = 0;
        a.wave_clock = nullptr;
#ifdef CRH_PROFILE
        static const int timing_d = CRH_PROFILE_ENV("CRH_SCORE_TIMING");
#endif
        a.out_score = out_score;      // not written by the kernel on this route
        a.out_idx = out_idx;
        a.dense = reinterpret_cast<float*>(workspace);
        CRH_CHECK_ARG(stride < ((int64_t)1 << 25), "%s: dense block rows of %lld scores exceed the kernel's 32-bit lane offsets", who, (long long)stride);
        a.dense_stride = stride;
        if (ev_kernel_start) CRH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_kernel_start), st));
        for (int64_t u0 = 0; u0 < n_users; u0 += chunk) {
            const int64_t cu = std::min(chunk, n_users - u0);
            a.users = users ? users + u0 : nullptr;
            a.user_base = u0;
            a.n_users = cu;
            a.n_ugroups = (cu + upw_pw - 1) / upw_pw;
            // item-range splits cost nothing here (every wave writes its own columns of the block): as many as
            // fill the wave slots of the chip once
            a.n_splits = (int)std::max<int64_t>(1, std::min<int64_t>(T, (1024 * occ) / a.n_ugroups));
            a.rated_rowptr = rated_rowptr ? rated_rowptr + u0 : nullptr;
#ifdef CRH_PROFILE
            if (timing_d) {
                CRH_HIP(hipMalloc(&a.wave_clock, (size_t)a.n_ugroups * a.n_splits * 48));
                CRH_HIP(hipMemsetAsync(a.wave_clock, 0, (size_t)a.n_ugroups * a.n_splits * 48, st));
            }
#endif
            const int rc1 = launch_score_per_wave(esz, d, occ, a, st);
            if (rc1 != CRH_OK) return rc1;
#ifdef CRH_PROFILE
            if (timing_d) {
                const int trc = print_wave_timing(a.wave_clock, a.n_ugroups * a.n_splits, timing_d, st);
                if (trc != CRH_OK) return trc;
            }
#endif
            const int rc2 = crh_mask_topk_f32(a.dense, cu, n_items, stride, a.rated_rowptr, rated_col, cand_bitmap, k,
                                              item_base, 0, out_score + u0 * k, out_idx + u0 * k, stream);
            if (rc2 != CRH_OK) return rc2;
        }
        if (ev_kernel_stop) CRH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_kernel_stop), st));
        return CRH_OK;
    }

    a.n_splits = plan_splits(rp, esz, n_users, n_items, k, n_splits, seed_score != nullptr);

    if (a.n_splits == 1) {
        a.out_score = out_score;
        a.out_idx = out_idx;
    } else {
        const size_t need = (size_t)a.n_splits * n_users * k * 8;
        if (!workspace || workspace_bytes < need) {
            crh_set_error("%s: workspace %zu < %zu bytes", who, workspace_bytes, need);
            return CRH_ERR_WS;
        }
        a.out_score = reinterpret_cast<float*>(workspace);
        a.out_idx = reinterpret_cast<int32_t*>(a.out_score + (size_t)a.n_splits * n_users * k);
    }
    // fragment-ordered copy of the shard (one HBM pass, ~0.1 % of a 128 K-user block at 10 M items) when the
    // workspace has room for it behind the partial lists
    a.packed = nullptr;
    if (can_pack) {
        void* pk = reinterpret_cast<char*>(workspace) + lists_bytes(n_users, k);
        const int prc = pack_items(esz, item_emb, n_items, d, pk, st, cand_bitmap, item_base);
        if (prc != CRH_OK) return prc;
        a.packed = pk;
    }
    int rc;
    // XCD soft lockstep (xcd_window_sync): only when every wave is resident at once (one round) and all waves
    // walk the same tile range (no item-range cuts); counters sit behind the packed copy in the workspace
    // (an even count of at least 2: the workgroup kernels count whole loop trips of two tiles, and the counters below are sized
    // from THIS value -- an odd window made kernel and host disagree about the number of windows)
    static const int sync_win_raw = CRH_TUNE_ENV("CRH_SCORE_SYNC_WINDOW") ? atoi(CRH_TUNE_ENV("CRH_SCORE_SYNC_WINDOW")) : 128;
    static const int sync_win = sync_win_raw <= 0 ? 0 : std::max(2, sync_win_raw & ~1);
    a.xcd_sync = nullptr;
    a.sync_window = sync_win;
    a.sync_stride = 0;
    const int64_t n_wg_launch = (a.n_ugroups + wg_waves_l - 1) / wg_waves_l;
    // Item-range cuts under SEEDED lists take part too (per-wave kernel): every cut walks its own range, window by window
    // from its own start, and the cuts' window counts differ by at most one, so no wave ever waits for a window another
    // wave will not report; the L2 of an XCD then holds two windows of each cut instead of the cuts' drifting streams.
    const bool cuts_ok = a.n_splits == 1 || (seed_score && !use_wg && T / a.n_splits >= 4 * sync_win);
    const bool sync_pw = !use_wg && occ == 2 && a.n_ugroups * a.n_splits <= 2048 && a.n_ugroups * a.n_splits > 256;
    const bool sync_wg = use_wg && n_wg_launch <= rp.wg_slots_l && n_wg_launch > 8;   // one resident round
    if (sync_win > 0 && (sync_pw || sync_wg) && cuts_ok && a.packed) {
        const int64_t n_win = (T / a.n_splits + 1 + sync_win - 1) / sync_win + 1;
        const size_t need = lists_bytes(n_users, k) + packed_bytes(n_items, d, esz) + sync_bytes(n_items);
        if (workspace_bytes >= need && (size_t)(n_win + 1) * 8 * sizeof(unsigned) <= sync_bytes(n_items)) {
            a.sync_stride = n_win + 1;
            a.xcd_sync = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + lists_bytes(n_users, k) +
                                                     packed_bytes(n_items, d, esz));
            CRH_HIP(hipMemsetAsync(a.xcd_sync, 0, (size_t)a.sync_stride * 8 * sizeof(unsigned), st));
        }
    }
    a.tile_bits = reinterpret_cast<const uint32_t*>(a.packed);      // readable filler when there is no candidate bitmap
    if (use_dma && cand_bitmap) {
        uint32_t* tb = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + tb_off);
        hipLaunchKernelGGL(tile_bits_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, cand_bitmap, item_base, n_items, T, tb);
        CRH_HIP(hipGetLastError());
        a.tile_bits = tb;
    }
    a.wave_clock = nullptr;
#ifdef CRH_PROFILE
    static const int timing = CRH_PROFILE_ENV("CRH_SCORE_TIMING");
    const int64_t n_waves = a.n_ugroups * a.n_splits;
    if (timing && !use_wg) CRH_HIP(hipMalloc(&a.wave_clock, (size_t)n_waves * 48));   // profile build only: [wave][start,end], [wave][loop start,end], [wave][event ticks,events]
    if (timing && use_dma) {   // DMA kernel: per-tile duration histograms of the first 16 workgroups' waves (64 buckets of 128 cycles)
        CRH_HIP(hipMalloc(&a.wave_clock, (size_t)16 * 4 * 66 * 8));    // + [wave][cycles in events, events]
        CRH_HIP(hipMemsetAsync(a.wave_clock, 0, (size_t)16 * 4 * 66 * 8, st));
    }
#endif
    if (ev_kernel_start) CRH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_kernel_start), st));
    if (use_dma) {
        rc = launch_score_dma(esz, d, rp.dma_mode, a, st);
    } else if (esz == 4 && use_wg) {
        rc = launch_score_wg<float, 128, 2, 8>(a, st);
    } else if (use_wg) {
        switch (d) {
            case 64: rc = wg_waves_l == 4 ? launch_score_wg<_Float16, 64, 2, 4>(a, st) : launch_score_wg<_Float16, 64, 2, 8>(a, st); break;
            case 128: rc = wg_waves_l == 4 ? launch_score_wg<_Float16, 128, 2, 4>(a, st) : launch_score_wg<_Float16, 128, 2, 8>(a, st); break;
            default: rc = wg_waves_l == 4 ? launch_score_wg<_Float16, 256, 2, 4>(a, st) : launch_score_wg<_Float16, 256, 2, 8>(a, st); break;
        }
    } else {
        rc = launch_score_per_wave(esz, d, occ, a, st);
    }
    if (rc != CRH_OK) return rc;
    if (ev_kernel_stop) CRH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_kernel_stop), st));
#ifdef CRH_PROFILE
    if (timing && use_dma) {
        CRH_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)16 * 4 * 66);
        CRH_HIP(hipMemcpy(h.data(), a.wave_clock, h.size() * 8, hipMemcpyDeviceToHost));
        CRH_HIP(hipFree(a.wave_clock));
        {
            unsigned long long tk = 0, ne = 0;
            for (int q = 0; q < 64; ++q) { tk += h[(size_t)16 * 4 * 64 + 2 * q]; ne += h[(size_t)16 * 4 * 64 + 2 * q + 1]; }
            fprintf(stderr, "[crh dma timing] events: %.0f per wave, %.0f cycles each (tiles with at least one candidate; 64 waves)\n",
                    (double)ne / 64.0, ne ? (double)tk / (double)ne : 0.0);
        }
        for (int wv = 0; wv < 4; ++wv) {
            unsigned long long tot = 0, wsum = 0;
            std::vector<unsigned long long> b(64, 0);
            for (int blk = 0; blk < 16; ++blk)
                for (int q = 0; q < 64; ++q) b[q] += h[((size_t)blk * 4 + wv) * 64 + q];
            const unsigned long long bw = (esz == 4 ? 512 : 128) >> (d * esz == 256 ? 1 : 0);     // bucket width in cycles per tile
            for (int q = 0; q < 64; ++q) { tot += b[q]; wsum += b[q] * (q * bw + bw / 2); }
            fprintf(stderr, "[crh dma timing] wave %d: %llu tiles, mean %.0f cycles; histogram (%llu-cycle buckets from 0):", wv, tot,
                    tot ? (double)wsum / (double)tot : 0.0, bw);
            for (int q = 0; q < 64; ++q) fprintf(stderr, " %llu", b[q]);
            fprintf(stderr, "\n");
        }
    }
    if (timing && !use_wg) {
        const int trc = print_wave_timing(a.wave_clock, n_waves, timing, st);
        if (trc != CRH_OK) return trc;
    }
#endif
    if (a.n_splits > 1)
        return crh_merge_topk(a.out_score, a.out_idx, a.n_splits, n_users, k, k, out_score, out_idx, stream);
    return CRH_OK;
}
}  // namespace

// Partial lists of the item-range splits + the fragment-ordered copy of the item shard.  A caller that can
// only afford the first part (crh_score_topk_min_workspace_bytes) still gets the same results from the
// row-major kernel, at ~0.9x the speed.
namespace {
// everything a call can use: partial lists / dense block, packed copy, lockstep counters -- and for the seeded route the
// seed lists in front plus the prefix's own dense block
size_t full_workspace_bytes(int64_t n_users, int64_t n_items, int d, int k, int esz, bool dim_ok) {
    if (n_users <= 0 || k <= 0) return 0;
    const size_t tail = dim_ok && n_items > 0 ? packed_bytes(n_items, d, esz) + sync_bytes(n_items) + tbits_bytes(n_items) : 0;
    size_t need = std::max(lists_bytes(n_users, k), dense_block_bytes(n_users, n_items)) + tail;
    const int64_t P = dim_ok ? seed_route(esz, n_users, n_items, d) : 0;      // only shapes that DO seed pay for the prefix
    if (P > 0)
        need = std::max(need, std::max(dense_block_bytes(n_users, P) + packed_bytes(P, d, esz), lists_bytes(n_users, k) + tail)) +
               seed_bytes(n_users, k);
    return need;
}
}  // namespace

extern "C" size_t crh_score_topk_workspace_bytes(int64_t n_users, int64_t n_items, int d, int k) {
    return full_workspace_bytes(n_users, n_items, d, k, 4, crh_score_topk_supports_dim(d) != 0);
}

extern "C" size_t crh_score_topk_f16_workspace_bytes(int64_t n_users, int64_t n_items, int d, int k) {
    return full_workspace_bytes(n_users, n_items, d, k, 2, crh_score_topk_f16_supports_dim(d) != 0);
}

extern "C" size_t crh_score_topk_min_workspace_bytes(int64_t n_users, int k) {
    if (n_users <= 0 || k <= 0) return 0;
    return lists_bytes(n_users, k);
}

// Which kernels a crh_score_topk_{f32,f16}[_ex] call of this shape takes: the dispatcher's own predicates (seed_route,
// plan_route, plan_splits) under the same environment switches, evaluated without touching the GPU.
extern "C" int crh_score_topk_route(int elem_bytes, int64_t n_users, int64_t n_items, int d, int k, size_t workspace_bytes,
                                    int has_bitmap, int n_splits, int64_t* prefix_items, int* picked_splits) {
    CRH_CHECK_ARG(elem_bytes == 4 || elem_bytes == 2, "crh_score_topk_route: elem_bytes=%d (4 = fp32 tables, 2 = fp16)", elem_bytes);
    CRH_CHECK_ARG(n_users > 0 && n_items > 0 && k >= 1 && k <= CRH_MAX_K, "crh_score_topk_route: empty block or k outside 1..%d",
                  CRH_MAX_K);
    CRH_CHECK_ARG(elem_bytes == 4 ? crh_score_topk_supports_dim(d) : crh_score_topk_f16_supports_dim(d),
                  "crh_score_topk_route: d=%d unsupported", d);
    CRH_CHECK_ARG(n_splits >= 0 && n_splits <= 64, "crh_score_topk_route: n_splits=%d outside 0..64", n_splits);
    const bool has_ws = workspace_bytes > 0;
    int64_t P = n_splits == 0 && has_ws ? seed_route(elem_bytes, n_users, n_items, d) : 0;
    size_t ws = workspace_bytes;
    if (P > 0) {    // the seeded route needs room for the seeds in front of either stage (score_topk_any)
        const size_t sb = seed_bytes(n_users, k);
        const size_t stage1 = dense_block_bytes(n_users, P) + packed_bytes(P, d, elem_bytes);
        const size_t stage2 = lists_bytes(n_users, k) + packed_bytes(n_items - P, d, elem_bytes);
        if (workspace_bytes >= sb + std::max(stage1, stage2)) ws = workspace_bytes - sb;
        else P = 0;
    }
    const int64_t n_main = n_items - P;
    const RoutePlan r = plan_route(elem_bytes, n_users, n_main, d, k, has_ws, ws, has_bitmap != 0, n_splits, P > 0);
    int route = r.dense ? CRH_ROUTE_DENSE : (r.use_dma ? CRH_ROUTE_FUSED_DMA : (r.use_wg ? CRH_ROUTE_FUSED_WG : CRH_ROUTE_FUSED_WAVE));
    if (P > 0) route |= CRH_ROUTE_SEEDED;
    if (r.use_dma && !r.dense && r.dma_mode == 2) route |= CRH_ROUTE_DMA_FLAGS;
    if (prefix_items) *prefix_items = P;
    if (picked_splits) *picked_splits = r.dense ? 1 : plan_splits(r, elem_bytes, n_users, n_main, k, n_splits, P > 0);
    return route;
}

// the scoring kernel of a route as rocprofv3 prints it (prefix of the demangled name)
extern "C" const char* crh_score_topk_route_kernel(int route) {
    switch (route & 15) {
        case CRH_ROUTE_DENSE: return "score_topk_kernel";          // + mask_topk_kernel over the score block
        case CRH_ROUTE_FUSED_WAVE: return "score_topk_kernel";
        case CRH_ROUTE_FUSED_WG: return "score_topk_wg_kernel";
        case CRH_ROUTE_FUSED_DMA: return "score_topk_dma_kernel";
        default: return "";
    }
}

extern "C" int crh_score_topk_f32_ex(const float* user_emb, const int32_t* users, int64_t n_users,
                                     const float* item_emb, int64_t n_items, int d,
                                     const int64_t* rated_rowptr, const int32_t* rated_col,
                                     const uint32_t* cand_bitmap, int k, int64_t item_base,
                                     float* out_score, int32_t* out_idx, void* workspace,
                                     size_t workspace_bytes, void* stream, int n_splits,
                                     void* ev_kernel_start, void* ev_kernel_stop) {
    return score_topk_any(4, user_emb, users, n_users, item_emb, n_items, d, rated_rowptr, rated_col, cand_bitmap, k,
                          item_base, out_score, out_idx, workspace, workspace_bytes, stream, n_splits,
                          ev_kernel_start, ev_kernel_stop, "crh_score_topk_f32");
}

extern "C" int crh_score_topk_f32(const float* user_emb, const int32_t* users, int64_t n_users,
                                  const float* item_emb, int64_t n_items, int d,
                                  const int64_t* rated_rowptr, const int32_t* rated_col,
                                  const uint32_t* cand_bitmap, int k, int64_t item_base,
                                  float* out_score, int32_t* out_idx, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return crh_score_topk_f32_ex(user_emb, users, n_users, item_emb, n_items, d, rated_rowptr, rated_col,
                                 cand_bitmap, k, item_base, out_score, out_idx, workspace, workspace_bytes,
                                 stream, 0, nullptr, nullptr);
}

// fp16 tables (IEEE half, row-major), fp32 accumulation on v_mfma_f32_32x32x16_f16; everything else --
// masks, canonical order of the returned lists, splits, workspace protocol -- as crh_score_topk_f32_ex.
extern "C" int crh_score_topk_f16_ex(const void* user_emb, const int32_t* users, int64_t n_users,
                                     const void* item_emb, int64_t n_items, int d,
                                     const int64_t* rated_rowptr, const int32_t* rated_col,
                                     const uint32_t* cand_bitmap, int k, int64_t item_base,
                                     float* out_score, int32_t* out_idx, void* workspace,
                                     size_t workspace_bytes, void* stream, int n_splits,
                                     void* ev_kernel_start, void* ev_kernel_stop) {
    return score_topk_any(2, user_emb, users, n_users, item_emb, n_items, d, rated_rowptr, rated_col, cand_bitmap, k,
                          item_base, out_score, out_idx, workspace, workspace_bytes, stream, n_splits,
                          ev_kernel_start, ev_kernel_stop, "crh_score_topk_f16");
}
