/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product
 * (coldrec_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and only as the checker.
 *
 * CPU restatement of ColdRec's full-catalogue scoring + masking + top-k:
 *   scoring   model/MF.py:58-63 (and 21 identical batch_predict copies)
 *               score = user_emb[users] @ item_emb.T
 *   masking   model/BaseRecommender.py:175-180
 *               S[j, rated_j] = -10e8 ; S[:, candidate_mask] = -10e8   (replaced, not removed)
 *   top-k     model/BaseRecommender.py:182   torch.topk(S, max_N, largest, sorted)
 *
 * The arithmetic itself lives in PyTorch (mm -> MKL sgemm, topk), whose summation order and
 * tie order are unspecified.  This file fixes the CANONICAL order the HIP
 * kernels are required to reproduce bit for bit:
 *   score(u,i) = fmaf chain over k = 0..d-1 ascending, starting from +0.0f, one rounding per
 *                product (what v_mfma_f32_32x32x2_f32 computes for a k-ordered accumulator);
 *   masked entries take the value -1e9f exactly (-10e8) and stay candidates;
 *   ranking key = (score descending, global item index ascending);
 *   lists shorter than k are padded with (-inf, INT32_MAX).
 * Pinned against the reference's own outputs through tests/golden/g6_eval_*.npz
 * (tests/test_oracle_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MASKED (-1.0e9f)
#define ORC_PAD_IDX 0x7fffffff

static inline int better(float sa, int32_t ia, float sb, int32_t ib) {
    return sa > sb || (sa == sb && ia < ib);
}

/* insert (s,i) into a list of n<=k entries kept sorted best-first; returns new n */
static int list_insert(float* ls, int32_t* li, int n, int k, float s, int32_t i) {
    int p = n;
    while (p > 0 && better(s, i, ls[p - 1], li[p - 1])) --p;
    if (p >= k) return n;
    int last = n < k ? n : k - 1;
    for (int t = last; t > p; --t) { ls[t] = ls[t - 1]; li[t] = li[t - 1]; }
    ls[p] = s; li[p] = i;
    return n < k ? n + 1 : k;
}

static void list_pad(float* ls, int32_t* li, int n, int k) {
    for (int t = n; t < k; ++t) { ls[t] = -INFINITY; li[t] = ORC_PAD_IDX; }
}

float orc_dot_chain(const float* a, const float* b, int d) {
    float s = 0.0f;
    for (int k = 0; k < d; ++k) s = fmaf(a[k], b[k], s);
    return s;
}

/* dense canonical scores S[b][i] (model/MF.py:62 restated with the fmaf chain) */
int orc_scores_dense(const float* Uemb, const int64_t* users, int64_t n_users, const float* V,
                     int64_t n_items, int d, float* S) {
    for (int64_t b = 0; b < n_users; ++b) {
        const float* u = Uemb + (users ? users[b] : b) * (int64_t)d;
        for (int64_t i = 0; i < n_items; ++i) S[b * n_items + i] = orc_dot_chain(u, V + i * (int64_t)d, d);
    }
    return 0;
}

/*
 * Fused restatement.  Uemb (n_user_rows, d) row-major; users[b] = row of Uemb for block slot b
 * (NULL = identity); Vshard = rows [item_base, item_base+n_items) of the item table;
 * rated_rowptr (n_users+1) / rated_col: per block slot, GLOBAL item ids, ascending (may be NULL);
 * cand_bitmap: bit gi set => item gi masked (may be NULL).
 * out_score/out_idx (n_users, k) best first, idx global.
 */
int orc_score_topk(const float* Uemb, const int64_t* users, int64_t n_users, const float* Vshard,
                   int64_t n_items, int d, const int64_t* rated_rowptr, const int32_t* rated_col,
                   const uint32_t* cand_bitmap, int k, int64_t item_base, float* out_score,
                   int32_t* out_idx) {
    for (int64_t b = 0; b < n_users; ++b) {
        const float* u = Uemb + (users ? users[b] : b) * (int64_t)d;
        float* ls = out_score + b * k;
        int32_t* li = out_idx + b * k;
        int n = 0;
        int64_t rp = rated_rowptr ? rated_rowptr[b] : 0;
        int64_t re = rated_rowptr ? rated_rowptr[b + 1] : 0;
        while (rp < re && rated_col[rp] < item_base) ++rp;
        for (int64_t i = 0; i < n_items; ++i) {
            int64_t gi = item_base + i;
            float s = orc_dot_chain(u, Vshard + i * (int64_t)d, d);
            int masked = 0;
            while (rp < re && rated_col[rp] < gi) ++rp;
            if (rp < re && rated_col[rp] == gi) masked = 1;
            if (cand_bitmap && ((cand_bitmap[gi >> 5] >> (gi & 31)) & 1u)) masked = 1;
            if (masked) s = ORC_MASKED;
            n = list_insert(ls, li, n, k, s, (int32_t)gi);
        }
        list_pad(ls, li, n, k);
    }
    return 0;
}

/* Dense-score variant: S (n_users, n_items) row-major is MODIFIED like the reference does
 * (masked entries overwritten with -1e9) when write_back != 0. */
int orc_mask_topk(float* S, int64_t n_users, int64_t n_items, const int64_t* rated_rowptr,
                  const int32_t* rated_col, const uint32_t* cand_bitmap, int k, int64_t item_base,
                  int write_back, float* out_score, int32_t* out_idx) {
    for (int64_t b = 0; b < n_users; ++b) {
        float* row = S + b * n_items;
        float* ls = out_score + b * k;
        int32_t* li = out_idx + b * k;
        int n = 0;
        int64_t rp = rated_rowptr ? rated_rowptr[b] : 0;
        int64_t re = rated_rowptr ? rated_rowptr[b + 1] : 0;
        while (rp < re && rated_col[rp] < item_base) ++rp;
        for (int64_t i = 0; i < n_items; ++i) {
            int64_t gi = item_base + i;
            float s = row[i];
            int masked = 0;
            while (rp < re && rated_col[rp] < gi) ++rp;
            if (rp < re && rated_col[rp] == gi) masked = 1;
            if (cand_bitmap && ((cand_bitmap[gi >> 5] >> (gi & 31)) & 1u)) masked = 1;
            if (masked) { s = ORC_MASKED; if (write_back) row[i] = s; }
            n = list_insert(ls, li, n, k, s, (int32_t)gi);
        }
        list_pad(ls, li, n, k);
    }
    return 0;
}

/* Merge n_lists partial lists per user (layout [list][user][k]) with the canonical key.
 * Pad entries (idx == INT32_MAX) are skipped.  Used for item-range splits and for the
 * cross-GPU merge after the all-gather. */
int orc_merge_topk(const float* in_score, const int32_t* in_idx, int n_lists, int64_t n_users,
                   int k_in, int k_out, float* out_score, int32_t* out_idx) {
    for (int64_t b = 0; b < n_users; ++b) {
        float* ls = out_score + b * k_out;
        int32_t* li = out_idx + b * k_out;
        int n = 0;
        for (int l = 0; l < n_lists; ++l) {
            const float* s = in_score + ((int64_t)l * n_users + b) * k_in;
            const int32_t* ix = in_idx + ((int64_t)l * n_users + b) * k_in;
            for (int t = 0; t < k_in; ++t) {
                if (ix[t] == ORC_PAD_IDX) continue;
                n = list_insert(ls, li, n, k_out, s[t], ix[t]);
            }
        }
        list_pad(ls, li, n, k_out);
    }
    return 0;
}

/* CSR SpMM restatement of torch.sparse.mm(A_hat, E) (model/LightGCN.py:90): sequential
 * ascending-column multiply-add per output element in fp32 (tolerance compare only:
 * the reference's accumulation order is unspecified). Y = alpha*A*X + beta*Z. */
int orc_spmm_csr(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                 const float* X, int d, float alpha, float beta, const float* Z, float* Y) {
    for (int64_t r = 0; r < n_rows; ++r) {
        for (int c = 0; c < d; ++c) {
            float acc = 0.0f;
            for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e)
                acc = fmaf(val[e], X[(int64_t)col[e] * d + c], acc);
            float y = alpha * acc;
            if (Z) y += beta * Z[r * (int64_t)d + c];
            Y[r * (int64_t)d + c] = y;
        }
    }
    return 0;
}
