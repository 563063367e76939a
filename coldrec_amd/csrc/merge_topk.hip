// Canonical top-k merge of partial lists, and mask + top-k over a dense score block.
//
// merge_topk  : item-range splits inside one GPU and item shards after the RCCL all-gather
//               (SURVEY.md 8(e)); the key (score desc, global index asc) makes the result
//               independent of how the catalogue was cut.
// mask_topk   : model/BaseRecommender.py:175-183 for score blocks produced by an arbitrary
//               batch_predict (VBPR/AMR/ALDI/CGRC or user models).  HBM-bound: one streaming
//               read of the block, 16 B per lane.
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

    for (int64_t user = (int64_t)blockIdx.x * 4 + wave; user < n_users; user += (int64_t)gridDim.x * 4) {
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
__global__ __launch_bounds__(256) void mask_writeback_kernel(float* __restrict__ S, int64_t n_users,
                                                             int64_t n_items, int64_t stride,
                                                             const int64_t* __restrict__ rated_rowptr,
                                                             const int32_t* __restrict__ rated_col,
                                                             const uint32_t* __restrict__ bitmap,
                                                             int64_t item_base) {
    // one block per row: rated scatter, then the column bitmap
    const int64_t row = blockIdx.x;
    float* srow = S + row * stride;
    if (rated_rowptr) {
        for (int64_t e = rated_rowptr[row] + threadIdx.x; e < rated_rowptr[row + 1]; e += blockDim.x) {
            const int64_t il = (int64_t)rated_col[e] - item_base;
            if (il >= 0 && il < n_items) srow[il] = CRH_MASKED_SCORE;
        }
    }
    if (bitmap) {
        for (int64_t il = threadIdx.x; il < n_items; il += blockDim.x) {
            const int64_t gi = item_base + il;
            if ((bitmap[gi >> 5] >> (gi & 31)) & 1u) srow[il] = CRH_MASKED_SCORE;
        }
    }
}

// One wave per row, 4 rows per block.  LDS per wave: one k-entry list.
__global__ __launch_bounds__(256) void mask_topk_kernel(const float* __restrict__ S, int64_t n_users,
                                                        int64_t n_items, int64_t stride,
                                                        const int64_t* __restrict__ rated_rowptr,
                                                        const int32_t* __restrict__ rated_col,
                                                        const uint32_t* __restrict__ bitmap, int K,
                                                        int64_t item_base, float* __restrict__ out_score,
                                                        int32_t* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* ls = reinterpret_cast<float*>(smem) + (size_t)wave * (2 * K + 4);
    int* li = reinterpret_cast<int*>(ls + K);
    int* cnt = li + K;

    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n_users; row += (int64_t)gridDim.x * 4) {
        if (lane == 0) *cnt = 0;
        const float* srow = S + row * stride;
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(srow) & 15) == 0);
        float tau = CRH_NEG_INF;
        for (int64_t base = 0; base < n_items; base += 256) {
            const int64_t e0 = base + lane * 4;
            float v[4];
            if (vec_ok && e0 + 3 < n_items) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(srow + e0);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = e0 + c < n_items ? srow[e0 + c] : CRH_NEG_INF;
            }
            const float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            if (__ballot(m > tau) == 0ull) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned long long cand = __ballot(v[c] > tau && e0 + c < n_items);
                while (cand) {
                    const int L = __builtin_ctzll(cand);
                    cand &= cand - 1;
                    float sc = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[c]), L));
                    const int gi = (int)(item_base + base + L * 4 + c);
                    const int n = __builtin_amdgcn_readfirstlane(*cnt);
                    if (wave_list_rejects(ls, li, n, K, sc, gi)) continue;
                    if (wave_is_masked(gi, row, rated_rowptr, rated_col, bitmap, lane)) sc = CRH_MASKED_SCORE;
                    wave_list_insert(ls, li, cnt, K, sc, gi, lane);
                }
            }
            tau = wave_list_tau(ls, *cnt, K);
        }
        const int n = __builtin_amdgcn_readfirstlane(*cnt);
        if (lane < K) {
            out_score[row * K + lane] = lane < n ? ls[lane] : CRH_NEG_INF;
            out_idx[row * K + lane] = lane < n ? li[lane] : CRH_PAD_IDX;
        }
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
    const size_t lds = (size_t)4 * n_lists * k_in * 8;
    if (lds > 64 * 1024)
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(merge_topk_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = (n_users + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)blocks), dim3(256), lds,
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
    if (write_back && (rated_rowptr || cand_bitmap)) {
        hipLaunchKernelGGL(mask_writeback_kernel, dim3((unsigned)n_users), dim3(256), 0, st, scores, n_users,
                           n_items, row_stride, rated_rowptr, rated_col, cand_bitmap, item_base);
        CRH_HIP(hipGetLastError());
    }
    int64_t blocks = (n_users + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    const size_t lds = (size_t)4 * (2 * k + 4) * 4;
    hipLaunchKernelGGL(mask_topk_kernel, dim3((unsigned)blocks), dim3(256), lds, st, scores, n_users, n_items,
                       row_stride, rated_rowptr, rated_col, cand_bitmap, k, item_base, out_score, out_idx);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
