// Canonical top-k merge of partial lists, and mask + top-k over a dense score block.
//
// merge_topk  : item-range splits inside one GPU and item shards after the RCCL all-gather
//               (SURVEY.md 8(e)); the key (score desc, global index asc) makes the result
//               independent of how the catalogue was cut.
// mask_topk   : model/BaseRecommender.py:175-183 for score blocks produced by an arbitrary
//               batch_predict (VBPR/AMR/ALDI/CGRC or user models).  HBM-bound: one streaming
//               read of the block, 16 B per lane.
#include <stdlib.h>

#include "crh_common.h"
#include "topk_list.h"

namespace {

// ------------------------------------------------------------------ merge
// One wave per user.  The n_lists (<= 64) sorted lists are staged in LDS; lane l walks list l.
// Each of the k_out steps is a 64-lane butterfly arg-best on (score, idx, lane).
__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ in_score,
                                                         const int32_t* __restrict__ in_idx, int n_lists,
                                                         int64_t n_users, int k_in, int k_out,
                                                         float* __restrict__ out_score,
                                                         int32_t* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int per_wave = n_lists * k_in;
    float* ss = reinterpret_cast<float*>(smem) + (size_t)wave * per_wave * 2;
    int* si = reinterpret_cast<int*>(ss + per_wave);

    const int wpb = (int)(blockDim.x >> 6);      // waves per block: 4, fewer when the staged lists are large (k > 64)
    for (int64_t user = (int64_t)blockIdx.x * wpb + wave; user < n_users; user += (int64_t)gridDim.x * wpb) {
        for (int e = lane; e < per_wave; e += 64) {
            const int l = e / k_in, t = e - l * k_in;
            const int64_t g = ((int64_t)l * n_users + user) * k_in + t;
            ss[e] = in_score[g];
            si[e] = in_idx[g];
        }
        int hp = 0;
        for (int step = 0; step < k_out; ++step) {
            float s = CRH_NEG_INF;
            int ix = CRH_PAD_IDX;
            if (lane < n_lists && hp < k_in) {
                s = ss[lane * k_in + hp];
                ix = si[lane * k_in + hp];
            }
            if (ix == CRH_PAD_IDX) s = CRH_NEG_INF;
            int who = lane;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float os = __shfl_xor(s, off);
                const int oi = __shfl_xor(ix, off);
                const int ow = __shfl_xor(who, off);
                const bool take = crh_better(os, oi, s, ix) || (os == s && oi == ix && ow < who);
                s = take ? os : s;
                ix = take ? oi : ix;
                who = take ? ow : who;
            }
            if (lane == who) ++hp;
            if (lane == 0) {
                out_score[user * k_out + step] = s;
                out_idx[user * k_out + step] = ix;
            }
        }
    }
}

// ------------------------------------------------------------------ dense mask + top-k
// LDS per wave: one k-entry list.  ONE streaming pass: the candidate bitmap is applied in registers while the
// row streams by (a masked score is -1e9 before the threshold test, so it rarely reaches the slow path), and
// with write_back the modified 16-byte vectors go straight back (the reference mutates the block,
// model/BaseRecommender.py:175-180); the user's rated items are tested on the slow path only and written back
// after the row.  Four 16-byte loads per lane are in flight per iteration (HBM-bound: 4 B per pair).
// WPR = waves per row: 1 -> four rows per block; 4 -> the four waves of a block split ONE row (few-row
// blocks would otherwise leave most of the chip idle) and merge their lists in LDS.
template <int WPR, int NL>
__global__ __launch_bounds__(256) void mask_topk_kernel(float* __restrict__ S, int64_t n_users,
                                                        int64_t n_items, int64_t stride,
                                                        const int64_t* __restrict__ rated_rowptr,
                                                        const int32_t* __restrict__ rated_col,
                                                        const uint32_t* __restrict__ bitmap, int K,
                                                        int64_t item_base, int write_back,
                                                        float* __restrict__ out_score,
                                                        int32_t* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* ls = reinterpret_cast<float*>(smem) + (size_t)wave * (2 * K + 4);
    int* li = reinterpret_cast<int*>(ls + K);
    int* cnt = li + K;
    const int64_t row_step = WPR == 1 ? (int64_t)gridDim.x * 4 : (int64_t)gridDim.x;

    for (int64_t row = WPR == 1 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x; row < n_users; row += row_step) {
        if (lane == 0) *cnt = 0;
        // the user's rated list: its bounds and its first 64 entries (ascending) stay in registers for the whole row, so
        // the membership test of a candidate -- ~k ln(N / k) of them per row, each one formerly three dependent memory
        // round trips (rowptr, rowptr, list probe) in the middle of the stream -- is a compare and a ballot for the usual
        // list; only lists above 64 entries go back to memory, and only for ids beyond their 64th entry
        int64_t rlo = 0, rhi = 0;
        if (rated_rowptr) {
            rlo = rated_rowptr[row];
            rhi = rated_rowptr[row + 1];
        }
        const int rv = (rlo + lane < rhi) ? rated_col[rlo + lane] : CRH_PAD_IDX;
        const int r63 = __builtin_amdgcn_readlane(rv, 63);
        float* srow = S + row * stride;
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(srow) & 15) == 0);
        float tau = CRH_NEG_INF;
        // this wave's item range, a multiple of the (256 * NL)-item iteration
        constexpr int STEP = 256 * NL;
        static_assert(8 * NL + 1 <= 64, "one bitmap word per lane covers the window");
        int64_t i0 = 0, i1 = n_items;
        if (WPR > 1) {
            const int64_t q = (((n_items + WPR - 1) / WPR) + STEP - 1) / STEP * STEP;
            i0 = q * wave < n_items ? q * wave : n_items;
            i1 = i0 + q < n_items ? i0 + q : n_items;
        }
        for (int64_t base = i0; base < i1; base += STEP) {
            f32x4 vq[NL];
            bool fullq[NL];
            // candidate bitmap of the iteration's window of 256 * NL items: ONE coalesced load (lane l takes word
            // wbase + l; NL <= 4: 32 words + the one a misaligned window spills into) instead of one or two loads per
            // lane and 256 items -- measured on the 4096 x 1 M block: 3.45 ms with the per-lane loads, 2.74 without masks
            unsigned bword = 0;
            int off = 0;
            if (bitmap) {
                const int64_t gb = item_base + base;
                off = (int)(gb & 31);
                const int64_t wi = (gb >> 5) + lane, last = (item_base + n_items - 1) >> 5;
                if (lane <= 8 * NL && wi <= last) bword = bitmap[wi];
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int64_t e0 = base + u * 256 + lane * 4;
                fullq[u] = vec_ok && e0 + 3 < i1;
                if (fullq[u]) {
                    // the block is read once: nontemporal (streaming) loads keep it out of the way of the bitmap words
                    vq[u] = write_back ? *reinterpret_cast<const f32x4*>(srow + e0)
                                       : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(srow + e0));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) vq[u][c] = e0 + c < i1 ? srow[e0 + c] : CRH_NEG_INF;
                }
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int64_t e0 = base + u * 256 + lane * 4;
                float v[4] = {vq[u][0], vq[u][1], vq[u][2], vq[u][3]};
                if (!write_back) {
                    // Lazy masking: a masked item scores -1e9 <= tau once tau is finite, so a vector whose raw
                    // maximum cannot beat tau cannot contribute whatever its bits say.  (tau = -inf until the list
                    // is full: every vector takes the masking path then.)
                    const float mr = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    if (__ballot(mr > tau) == 0ull) continue;
                }
                unsigned bits = 0;
                if (bitmap) {
                    // the 4 candidate bits of this lane's 4 items: bit p = off + u * 256 + lane * 4 of the window, i.e.
                    // word (p >> 5) and possibly the next one, fetched from the lanes that hold them
                    const int p = off + u * 256 + lane * 4;
                    const unsigned lo = __shfl(bword, p >> 5), hi = __shfl(bword, (p >> 5) + 1);
                    bits = (unsigned)((((uint64_t)hi << 32) | lo) >> (p & 31)) & 0xfu;
                }
                if (bits) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (((bits >> c) & 1u) && e0 + c < i1) v[c] = CRH_MASKED_SCORE;
                    if (write_back) {
                        if (fullq[u]) {
                            *reinterpret_cast<f32x4*>(srow + e0) = f32x4{v[0], v[1], v[2], v[3]};
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (((bits >> c) & 1u) && e0 + c < i1) srow[e0 + c] = CRH_MASKED_SCORE;
                        }
                    }
                }
                const float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                if (__ballot(m > tau) == 0ull) continue;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned long long cand = __ballot(v[c] > tau && e0 + c < i1);
                    while (cand) {
                        const int L = __builtin_ctzll(cand);
                        cand &= cand - 1;
                        float sc = __builtin_bit_cast(
                            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[c]), L));
                        const int gi = (int)(item_base + base + u * 256 + L * 4 + c);
                        const int n = __builtin_amdgcn_readfirstlane(*cnt);
                        if (wave_list_rejects(ls, li, n, K, sc, gi)) continue;
                        // (the candidate bitmap was applied before the threshold test: only the rated list is left)
                        bool masked = __ballot(rv == gi) != 0ull;
                        if (!masked && rhi - rlo > 64 && gi > r63)
                            masked = wave_is_masked_at(gi, rlo + 64, rhi, rated_col, nullptr, lane);
                        if (masked) sc = CRH_MASKED_SCORE;
                        wave_list_insert(ls, li, cnt, K, sc, gi, lane);
                    }
                }
                tau = wave_list_tau(ls, *cnt, K);
            }
        }
        if (WPR > 1) {                           // waves 1..3 hand their lists to wave 0 (final scores: no mask test)
            __syncthreads();
            if (wave == 0) {
                for (int ow = 1; ow < WPR; ++ow) {
                    const float* os = reinterpret_cast<float*>(smem) + (size_t)ow * (2 * K + 4);
                    const int* oi = reinterpret_cast<const int*>(os + K);
                    const int on = __builtin_amdgcn_readfirstlane(oi[K]);
                    for (int e = 0; e < on; ++e) {
                        const float sc = os[e];
                        const int gi = oi[e];
                        const int n = __builtin_amdgcn_readfirstlane(*cnt);
                        if (wave_list_rejects(ls, li, n, K, sc, gi) && sc >= CRH_MASKED_SCORE) continue;
                        wave_list_insert(ls, li, cnt, K, sc, gi, lane);
                    }
                }
            }
        }
        if (WPR == 1 || wave == 0) {
            const int n = __builtin_amdgcn_readfirstlane(*cnt);
            wave_list_store(ls, li, n, K, out_score + row * K, out_idx + row * K, lane);
            if (write_back && rated_rowptr) {       // the row has been read: now it may be scribbled on
                for (int64_t e = rated_rowptr[row] + lane; e < rated_rowptr[row + 1]; e += 64) {
                    const int64_t il = (int64_t)rated_col[e] - item_base;
                    if (il >= 0 && il < n_items) srow[il] = CRH_MASKED_SCORE;
                }
            }
        }
        if (WPR > 1) __syncthreads();            // lists are reused by the next row of this block
    }
}

// ------------------------------------------------------------------ dense mask + top-k, register-resident chunks
// The streaming kernel above pays ~k (1 + ln(N / k)) wave-serial list insertions per row whatever N is: on the rows
// the trainers rank every epoch (validation: 3 706 - 16 980 items; the seed prefix of the mid-size route: 4 096 -
// 16 384) the insertions, not the bytes, are the cost.  Here a wave takes its row in chunks of 256 * NV items held in
// registers (NV 16-byte loads per lane in flight), applies BOTH masks first -- the candidate bitmap's words of the
// chunk and the user's rated ids are OR-ed into a per-wave bitmap in LDS (the rated list is ascending: one walk per
// row) -- and only then looks for candidates: the k-th largest of the 64 lane maxima (a 32-step radix select on
// ballots) is a lower bound of the chunk's k-th best score, because the k lanes at or above it hold k different items.
// Items below it cannot be in the row's top k; what is left is ~k items per chunk instead of ~k ln(chunk / k); they
// and the list are merged by a rank sort across the lanes (or, when they outnumber the lanes, go through the same list
// insertion as above) -- canonical order, exact ties either way.  Chunks whose raw maximum cannot beat the
// list are skipped before any mask work, so a long row streams as before.  Read-only (write_back = 0), k <= 64.
__device__ __forceinline__ unsigned order_key(float f) {
    const unsigned b = __builtin_bit_cast(unsigned, f);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float order_key_inv(unsigned k) {
    return __builtin_bit_cast(float, k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

// Register allocation held to three waves per SIMD (167 VGPRs, no scratch): left alone the compiler takes 248 for the unrolled
// chunk (two waves per SIMD); at four it spills.  Same-box A/B on the validation shapes, 2 / 3 / 4 waves: 0.115 / 0.109 / 0.115 ms
// (6 040 x 3 706) and 0.402 / 0.386 / 0.403 ms (5 551 x 16 980).
#ifndef CRH_CHUNK_OCC
#define CRH_CHUNK_OCC 3
#endif
template <int NV>
__global__ __launch_bounds__(256, CRH_CHUNK_OCC) void mask_topk_chunk_kernel(const float* __restrict__ S, int64_t n_users,
                                                              int64_t n_items, int64_t stride,
                                                              const int64_t* __restrict__ rated_rowptr,
                                                              const int32_t* __restrict__ rated_col,
                                                              const uint32_t* __restrict__ bitmap, int K,
                                                              int64_t item_base, float* __restrict__ out_score,
                                                              int32_t* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = 256 * NV, WORDS = 8 * NV;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* ls = reinterpret_cast<float*>(smem) + (size_t)wave * (2 * K + 4 + WORDS + 128);
    int* li = reinterpret_cast<int*>(ls + K);
    int* cnt = li + K;
    unsigned* bm = reinterpret_cast<unsigned*>(cnt + 4);
    float* cs = reinterpret_cast<float*>(bm + WORDS);          // candidates of a chunk: 64 scores, 64 ids
    int* ci = reinterpret_cast<int*>(cs + 64);
    const bool have_masks = bitmap != nullptr || rated_rowptr != nullptr;
    const int64_t last_word = bitmap ? (item_base + n_items - 1) >> 5 : 0;

    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n_users; row += (int64_t)gridDim.x * 4) {
        if (lane == 0) *cnt = 0;
        int64_t rp = 0, rhi = 0;                 // the part of the rated list not yet behind the stream
        if (rated_rowptr) {
            rp = rated_rowptr[row];
            rhi = rated_rowptr[row + 1];
        }
        const float* srow = S + row * stride;
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(srow) & 15) == 0);
        float tau = CRH_NEG_INF;                 // the list's k-th score once it is full (strict: later ids lose ties)
        for (int64_t base = 0; base < n_items; base += CH) {
            f32x4 v[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int64_t e0 = base + u * 256 + lane * 4;
                if (vec_ok && e0 + 3 < n_items) {
                    v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(srow + e0));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[u][c] = e0 + c < n_items ? srow[e0 + c] : CRH_NEG_INF;
                }
            }
            // What the mask pass of this chunk will want is requested NOW, beside the row's scores: the chunk's bitmap words and
            // the next 64 ids of the rated list.  A row is one latency chain per wave (a validation block is ONE chunk per row,
            // one row per wave): asked for one after the other -- list bounds, scores, bitmap words, rated ids -- the chain was
            // four memory round trips long, now two.  (A chunk that turns out to need no masks has fetched 0.8 KB beside its 16.)
            const int64_t g0 = item_base + base;              // global id of the chunk's first item
            constexpr int WJ = (WORDS + 63) / 64;
            unsigned blo[WJ], bhi[WJ];
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int64_t wi = (g0 >> 5) + lane + 64 * j;
                blo[j] = bitmap && lane + 64 * j < WORDS && wi <= last_word ? bitmap[wi] : 0u;
                bhi[j] = bitmap && lane + 64 * j < WORDS && wi + 1 <= last_word ? bitmap[wi + 1] : 0u;
            }
            const int64_t rp_pre = rp;
            const int cid_pre = rp + lane < rhi ? rated_col[rp + lane] : 0;
            float m = CRH_NEG_INF;
#pragma unroll
            for (int u = 0; u < NV; ++u) m = fmaxf(m, fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
            // masking only lowers a score (to -1e9 <= a finite tau): a chunk whose raw maximum cannot beat the list is done
            if (__ballot(m > tau) == 0ull) continue;
            if (have_masks) {
#pragma unroll
                for (int j = 0; j < WJ; ++j) {
                    const int w = lane + 64 * j;
                    if (w >= WORDS) break;
                    const int sh = (int)(g0 & 31);
                    bm[w] = sh ? (blo[j] >> sh) | (bhi[j] << (32 - sh)) : blo[j];
                }
                __builtin_amdgcn_wave_barrier();
                if (rp < rhi) {
                    const int64_t g1 = g0 + CH;
                    for (;;) {
                        const int64_t e = rp + lane;
                        const int64_t cid = e >= rhi ? ((int64_t)1 << 62)                            // past the end: never behind
                                                     : (int64_t)(rp == rp_pre ? cid_pre : rated_col[e]);
                        if (cid >= g0 && cid < g1) atomicOr(&bm[(cid - g0) >> 5], 1u << ((cid - g0) & 31));
                        const int behind = __popcll(__ballot(cid < g1));
                        rp += behind;
                        if (behind < 64) break;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                m = CRH_NEG_INF;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const unsigned bits = (bm[u * 8 + (lane >> 3)] >> ((lane & 7) * 4)) & 0xfu;
                    const int64_t e0 = base + u * 256 + lane * 4;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (((bits >> c) & 1u) && e0 + c < n_items) v[u][c] = CRH_MASKED_SCORE;
                    m = fmaxf(m, fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
                }
                __builtin_amdgcn_wave_barrier();
                if (__ballot(m > tau) == 0ull) continue;
            }
            // k-th largest lane maximum: most significant bit first, keep a bit if k lanes still reach the value
            const unsigned mk = order_key(m);
            unsigned cur = 0;
#pragma unroll
            for (int b = 31; b >= 0; --b) {
                const unsigned t = cur | (1u << b);
                if (__popcll(__ballot(mk >= t)) >= K) cur = t;
            }
            // one threshold for both tests of a candidate, v >= tau0 (the chunk's bound) and v > tau (strict: the list's k-th
            // score): v > tau <=> v >= the next float above tau (-FLT_MAX for tau = -inf), and both are wave-uniform.  Items past
            // the row's end were loaded as -inf and stay below it.  The edges of "next float" (ADVICE r4): a zero threshold is
            // taken as +0.0 whatever its sign (the successor of -0.0 is +0.0, and v >= +0.0 would admit every zero of a row of
            // cold items -- more candidates than lanes, one by one -- where v > 0 admits none); fp32 denormals are kept by this
            // build, so the successor of +0.0 (the smallest denormal) compares as it should; nothing is above +inf (its
            // successor is a NaN that fmaxf would drop).  Ties that slip through are re-filtered by crh_better either way.
            if (tau == __builtin_inff()) continue;
            const float tau_c = tau == 0.0f ? 0.0f : tau;
            const float thr = fmaxf(order_key_inv(cur), order_key_inv(order_key(tau_c) + 1u));
            // The chunk's candidates (>= tau0, and > the list's k-th score: list entries come from earlier chunks, i.e. lower
            // ids) are compacted into LDS; if they and the list fit one entry per lane, the new list is a RANK SORT of the
            // union -- every lane counts the entries that beat its own (canonical order, ids are distinct) and stores it at
            // that rank -- instead of one wave-serial insertion per candidate.
            {
                // (the launch is instruction-bound: a candidate's slot = running count + v_mbcnt of the ballot, clamped to the
                // last slot instead of tested -- beyond 64 candidates the slots are not used, see below)
                int nc = 0;
                const int id0 = (int)(item_base + base) + lane * 4;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const float mu = fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3]));
                    if (__ballot(mu >= thr) == 0ull) continue;
                    const float vc[4] = {v[u][0], v[u][1], v[u][2], v[u][3]};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool pr = vc[c] >= thr;
                        const unsigned long long bal = __ballot(pr);
                        if (pr) {
                            const int pos = min(nc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32),
                                                                                    __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)), 63);
                            cs[pos] = vc[c];
                            ci[pos] = id0 + u * 256 + c;
                        }
                        nc += __popcll(bal);
                    }
                }
                if (nc == 0) continue;
                const int n = __builtin_amdgcn_readfirstlane(*cnt);
                if (n + nc <= 64) {
                    __builtin_amdgcn_wave_barrier();
                    const int tot_n = n + nc;
                    float ms = CRH_NEG_INF;
                    int mi = CRH_PAD_IDX;
                    if (lane < n) {
                        ms = ls[lane];
                        mi = li[lane];
                    } else if (lane < tot_n) {
                        ms = cs[lane - n];
                        mi = ci[lane - n];
                    }
                    // crh_better as ONE unsigned 64-bit compare: (order key of the score, zeros of either sign alike) above the
                    // complemented id -- half the instructions of the three-compare form in the launch's longest loop
                    const unsigned long long mykey =
                        ((unsigned long long)order_key(ms == 0.0f ? 0.0f : ms) << 32) | (unsigned)(0x7fffffff - mi);
                    const unsigned klo = (unsigned)mykey, khi = (unsigned)(mykey >> 32);
                    int rank = 0;
                    for (int j = 0; j < tot_n; ++j) {
                        const unsigned long long kj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi, j) << 32) |
                                                      (unsigned)__builtin_amdgcn_readlane((int)klo, j);
                        rank += kj > mykey ? 1 : 0;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < tot_n && rank < K) {
                        ls[rank] = ms;
                        li[rank] = mi;
                    }
                    if (lane == 0) *cnt = tot_n < K ? tot_n : K;
                    __builtin_amdgcn_wave_barrier();
                    tau = wave_list_tau(ls, *cnt, K);
                    continue;
                }
            }
            // (more candidates than lanes: one by one)
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const float mu = fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3]));
                if (__ballot(mu >= thr) == 0ull) continue;
                const float vc[4] = {v[u][0], v[u][1], v[u][2], v[u][3]};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned long long cand = __ballot(vc[c] >= thr);
                    while (cand) {
                        const int L = __builtin_ctzll(cand);
                        cand &= cand - 1;
                        const float sc = __builtin_bit_cast(
                            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vc[c]), L));
                        const int gi = (int)(item_base + base + u * 256 + L * 4 + c);
                        const int n = __builtin_amdgcn_readfirstlane(*cnt);
                        if (wave_list_rejects(ls, li, n, K, sc, gi)) continue;
                        wave_list_insert(ls, li, cnt, K, sc, gi, lane);
                    }
                }
                tau = wave_list_tau(ls, *cnt, K);
            }
        }
        const int n = __builtin_amdgcn_readfirstlane(*cnt);
        wave_list_store(ls, li, n, K, out_score + row * K, out_idx + row * K, lane);
    }
}

}  // namespace

extern "C" int crh_merge_topk(const float* in_score, const int32_t* in_idx, int n_lists, int64_t n_users,
                              int k_in, int k_out, float* out_score, int32_t* out_idx, void* stream) {
    CRH_CHECK_ARG(in_score && in_idx && out_score && out_idx, "crh_merge_topk: NULL pointer");
    CRH_CHECK_ARG(n_lists >= 1 && n_lists <= 64, "crh_merge_topk: n_lists=%d outside 1..64", n_lists);
    CRH_CHECK_ARG(k_in >= 1 && k_in <= CRH_MAX_K && k_out >= 1 && k_out <= CRH_MAX_K,
                  "crh_merge_topk: k_in=%d / k_out=%d outside 1..%d", k_in, k_out, CRH_MAX_K);
    CRH_CHECK_ARG(n_users > 0, "crh_merge_topk: n_users=%lld", (long long)n_users);
    const size_t per_wave = (size_t)n_lists * k_in * 8;          // <= 64 lists x 128 entries x 8 B = 64 KiB
    int wpb = 4;
    while (wpb > 1 && per_wave * wpb > 152 * 1024) wpb >>= 1;
    const size_t lds = per_wave * wpb;
    if (lds > 64 * 1024)
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(merge_topk_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (n_users + wpb - 1) / wpb;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)blocks), dim3(64 * wpb), lds,
                       reinterpret_cast<hipStream_t>(stream), in_score, in_idx, n_lists, n_users, k_in, k_out,
                       out_score, out_idx);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

extern "C" int crh_mask_topk_f32(float* scores, int64_t n_users, int64_t n_items, int64_t row_stride,
                                 const int64_t* rated_rowptr, const int32_t* rated_col,
                                 const uint32_t* cand_bitmap, int k, int64_t item_base, int write_back,
                                 float* out_score, int32_t* out_idx, void* stream) {
    CRH_CHECK_ARG(scores && out_score && out_idx, "crh_mask_topk_f32: NULL pointer");
    CRH_CHECK_ARG(n_users > 0 && n_items > 0 && row_stride >= n_items, "crh_mask_topk_f32: bad shape");
    CRH_CHECK_ARG(k >= 1 && k <= CRH_MAX_K, "crh_mask_topk_f32: k=%d outside 1..%d", k, CRH_MAX_K);
    CRH_CHECK_ARG(item_base >= 0 && item_base + n_items < (int64_t)CRH_PAD_IDX, "crh_mask_topk_f32: item ids exceed int32");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)4 * (2 * k + 4) * 4;
    // 16-byte loads in flight per lane: NL = 4 (94 VGPRs, all 16 waves of a CU resident; 8 needs 134 VGPRs, loses a
    // quarter of the waves and runs the 4096 x 1 M block in 5.99 ms instead of 3.36)
    static const int wpr_rows = CRH_TUNE_ENV("CRH_MASK_WPR_ROWS") ? atoi(CRH_TUNE_ENV("CRH_MASK_WPR_ROWS")) : 4096;
#define CRH_MASK_LAUNCH(W, B)                                                                                              \
    hipLaunchKernelGGL((mask_topk_kernel<W, 4>), dim3((unsigned)(B)), dim3(256), lds, st, scores, n_users, n_items, row_stride, \
                       rated_rowptr, rated_col, cand_bitmap, k, item_base, write_back, out_score, out_idx)
    // rows short enough for the insertions to be the cost (and any row of a read-only call up to the limit): chunks in
    // registers, masks first, lane-maximum threshold.  CRH_MASK_CHUNK_ITEMS = largest row that takes this kernel (0: off).
    static const int64_t chunk_items = CRH_TUNE_ENV("CRH_MASK_CHUNK_ITEMS") ? atoll(CRH_TUNE_ENV("CRH_MASK_CHUNK_ITEMS")) : 65536;
    if (!write_back && k <= 64 && n_items <= chunk_items && !(n_users < wpr_rows && n_items >= 8192)) {
#ifndef CRH_CHUNK_NV
#define CRH_CHUNK_NV 16          // 16-byte loads per lane and chunk; 8 (chunks of 2 048 items, 128 VGPRs at four waves) measured slower:
#endif                           // 0.113 vs 0.109 ms at 6 040 x 3 706, 0.405 vs 0.388 ms at 5 551 x 16 980
        constexpr int NV = CRH_CHUNK_NV;
        const size_t lds_c = (size_t)4 * (2 * k + 4 + 8 * NV + 128) * 4;
        int64_t blocks = (n_users + 3) / 4;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL((mask_topk_chunk_kernel<NV>), dim3((unsigned)blocks), dim3(256), lds_c, st, scores, n_users, n_items,
                           row_stride, rated_rowptr, rated_col, cand_bitmap, k, item_base, out_score, out_idx);
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    }
    if (n_users < wpr_rows && n_items >= 8192) {    // few rows: four waves per row
        CRH_MASK_LAUNCH(4, n_users > 16384 ? 16384 : n_users);
    } else {
        int64_t blocks = (n_users + 3) / 4;
        if (blocks > 8192) blocks = 8192;
        CRH_MASK_LAUNCH(1, blocks);
    }
#undef CRH_MASK_LAUNCH
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
