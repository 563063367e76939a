/*
 * coldrec_hip.h -- C ABI of libcoldrec_hip.so (MI355X / gfx950).
 *
 * The upstream project (YuanchenBei/ColdRec) is pure Python and has NO FFI of its own; the
 * hot path is a handful of PyTorch calls.  Each entry point below names the reference call
 * site it replaces (file:line relative to the upstream repository); INTEGRATION.md shows the
 * ctypes binding a ColdRec maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - nothing is allocated or freed here: the caller owns every buffer (PyTorch's caching
 *     allocator in our host layer); scratch is passed in after a *_workspace_bytes() query;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - return 0 on success, <0 on error; crh_last_error() gives the thread-local message;
 *   - item / user ids are int32 (catalogues up to 2^31-2 rows); row offsets are int64.
 *   - top-k lists are ordered by the CANONICAL key (score descending, global item index
 *     ascending); lists with fewer than k candidates are padded with (-inf, CRH_PAD_IDX).
 */
#ifndef COLDREC_HIP_H
#define COLDREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRH_OK 0
#define CRH_ERR_ARG (-1)     /* bad argument (NULL, unsupported d / k, misaligned pointer) */
#define CRH_ERR_HIP (-2)     /* a HIP runtime call failed */
#define CRH_ERR_WS (-3)      /* workspace too small */
#define CRH_PAD_IDX 0x7fffffff
#define CRH_MASKED_SCORE (-1.0e9f) /* -10e8, model/BaseRecommender.py:177,180 */
#define CRH_MAX_K 128

const char* crh_last_error(void);
int crh_version(void);
/* 1 when the embedding width is handled by the MFMA kernel without padding (8,16,32,64,128,256) */
int crh_score_topk_supports_dim(int d);

/*
 * Fused full-catalogue scoring + masking + top-k for one block of users against one item shard.
 * Replaces, per user block of BaseColdStartTrainer._evaluate (model/BaseRecommender.py:172-183):
 *     batch_predict:  user_emb[users] @ item_emb.T            model/MF.py:58-63 (+21 copies)
 *     S[j, rated_j] = -10e8 ; S[:, candidate_mask] = -10e8    model/BaseRecommender.py:175-180
 *     torch.topk(S, max_N, dim=1, largest=True, sorted=True)  model/BaseRecommender.py:182
 * without materialising S.  score(u,i) is the fp32 k-ascending fma chain (exact fp32 MFMA).
 *
 *   user_emb   (n_user_rows, d) fp32 row-major, 16-byte aligned
 *   users      (n_users) int32 rows of user_emb for the block, or NULL for rows 0..n_users-1
 *   item_emb   (n_items, d) fp32: rows [item_base, item_base+n_items) of the item table
 *   rated_rowptr (n_users+1) int64 / rated_col int32: per block slot, the GLOBAL ids of the
 *              user's training items, ascending within a row; both NULL = nothing rated
 *   cand_bitmap  bit (gi & 31) of word gi>>5 set => global item gi is masked; NULL = none
 *   k          1..CRH_MAX_K;  out_score/out_idx (n_users, k), idx are GLOBAL item ids
 *   workspace  crh_score_topk_workspace_bytes(...) bytes: partial lists of the item-range splits plus a
 *              copy of the shard in MFMA-fragment order (n_items*d*4 bytes, written once per call by a
 *              streaming kernel; removes every cross-lane shuffle from the scoring loop).  With only
 *              crh_score_topk_min_workspace_bytes(...) the row-major kernel runs: identical results.
 *              Calls of up to 5e9 (user, item) pairs and 262144 items -- and any user count up to 32768 items: the
 *              trainers' per-epoch validation -- take another route when the
 *              full workspace is given and n_splits is 0: the MFMA kernel writes the score block (n_users x
 *              ceil32(n_items) fp32, at most 8 GiB at a time) into the workspace and the wave-per-user ranking of
 *              crh_mask_topk_f32 selects from it -- same scores, masks and order, ~10-30x lower latency at these
 *              sizes, where the fused selection is bound by its per-candidate slow path.
 */
size_t crh_score_topk_workspace_bytes(int64_t n_users, int64_t n_items, int d, int k);
size_t crh_score_topk_min_workspace_bytes(int64_t n_users, int k);

/* Which kernels a crh_score_topk_{f32,f16}[_ex] call of this shape takes, answered by the dispatcher's own code (the same
 * predicates and environment switches, no GPU work): the library reports its route, callers do not re-derive it.
 *   elem_bytes 4 (fp32 tables) | 2 (fp16);  workspace_bytes as it will be passed (0 = no workspace);  has_bitmap: a
 *   candidate bitmap will be passed;  n_splits as in the _ex calls (0 = library's choice).
 * Returns the route of the stage that walks the catalogue -- CRH_ROUTE_DENSE (score block + crh_mask_topk_f32),
 * CRH_ROUTE_FUSED_WAVE (one wave per user group), CRH_ROUTE_FUSED_WG (8-wave workgroups, register-staged LDS ring),
 * CRH_ROUTE_FUSED_DMA (4-wave workgroups fed by LDS-DMA) -- OR-ed with CRH_ROUTE_SEEDED when a catalogue prefix is ranked
 * first and seeds the lists, and with CRH_ROUTE_DMA_FLAGS when the DMA kernel runs in its flag form (ring slots guarded by
 * LDS counters instead of a barrier per tile: fp32 d=128 below 6 M items); < 0 on bad arguments.  prefix_items / picked_splits (either may be NULL) receive the prefix
 * length (0: not seeded) and the item-range cut count of the main stage.
 * There is no counterpart in the reference (model/BaseRecommender.py:172-183 is one matmul + topk whatever the shape). */
#define CRH_ROUTE_DENSE 1
#define CRH_ROUTE_FUSED_WAVE 2
#define CRH_ROUTE_FUSED_WG 3
#define CRH_ROUTE_FUSED_DMA 4
#define CRH_ROUTE_SEEDED 16
#define CRH_ROUTE_DMA_FLAGS 32
int crh_score_topk_route(int elem_bytes, int64_t n_users, int64_t n_items, int d, int k, size_t workspace_bytes,
                         int has_bitmap, int n_splits, int64_t* prefix_items, int* picked_splits);
/* kernel-name prefix of a route's scoring kernel as rocprofv3 prints it ("score_topk_dma_kernel", ...) */
const char* crh_score_topk_route_kernel(int route);
int crh_score_topk_f32(const float* user_emb, const int32_t* users, int64_t n_users,
                       const float* item_emb, int64_t n_items, int d,
                       const int64_t* rated_rowptr, const int32_t* rated_col,
                       const uint32_t* cand_bitmap, int k, int64_t item_base,
                       float* out_score, int32_t* out_idx,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Tuning / test / measurement hook: same as above with the item-range split count forced
 * (0 = automatic; results are identical for every split count, canonical merge) and two
 * optional hipEvent_t (as void*, may be NULL) recorded on `stream` immediately before and
 * after the scoring kernel itself, so a caller can time that kernel apart from the merge. */
int crh_score_topk_f32_ex(const float* user_emb, const int32_t* users, int64_t n_users,
                          const float* item_emb, int64_t n_items, int d,
                          const int64_t* rated_rowptr, const int32_t* rated_col,
                          const uint32_t* cand_bitmap, int k, int64_t item_base,
                          float* out_score, int32_t* out_idx,
                          void* workspace, size_t workspace_bytes, void* stream, int n_splits,
                          void* ev_kernel_start, void* ev_kernel_stop);

/*
 * fp16 variant for generated embeddings (SURVEY.md 8(f)3 / BASELINE.json configs[4]: model/DropoutNet.py:126-135
 * produces the tables, :67-72 scores them with the same user_emb[users] @ item_emb.T).  Tables are IEEE half,
 * row-major, d in {16,32,64,128,256}; products are exact in fp32 and accumulated in fp32 by
 * v_mfma_f32_32x32x16_f16 (the accumulation order inside one 16-wide MFMA step is the hardware's, so
 * scores are compared with a tolerance, not bit for bit; the ORDER of the returned list is canonical for
 * the scores the kernel computed).  Masks, splits, workspace protocol, events: as crh_score_topk_f32_ex.
 */
int crh_score_topk_f16_supports_dim(int d);
size_t crh_score_topk_f16_workspace_bytes(int64_t n_users, int64_t n_items, int d, int k);
int crh_score_topk_f16_ex(const void* user_emb, const int32_t* users, int64_t n_users,
                          const void* item_emb, int64_t n_items, int d,
                          const int64_t* rated_rowptr, const int32_t* rated_col,
                          const uint32_t* cand_bitmap, int k, int64_t item_base,
                          float* out_score, int32_t* out_idx,
                          void* workspace, size_t workspace_bytes, void* stream, int n_splits,
                          void* ev_kernel_start, void* ev_kernel_stop);

/*
 * Mask + top-k over an already materialised dense score block (any batch_predict, e.g.
 * model/VBPR.py:68-75, model/ALDI.py:149-160): model/BaseRecommender.py:175-183.
 * scores (n_users, row_stride) fp32; masked entries are also written back as -1e9 when
 * write_back != 0 (the reference mutates the block in place).
 */
int crh_mask_topk_f32(float* scores, int64_t n_users, int64_t n_items, int64_t row_stride,
                      const int64_t* rated_rowptr, const int32_t* rated_col,
                      const uint32_t* cand_bitmap, int k, int64_t item_base, int write_back,
                      float* out_score, int32_t* out_idx, void* stream);

/*
 * Canonical merge of n_lists partial top-k lists per user, layout [list][user][k_in]
 * (item-range splits inside one GPU; shards after the all-gather).
 * n_lists <= 64, k_in, k_out <= CRH_MAX_K (the reference takes any --topN, model/BaseRecommender.py:27-29; lists of
 * up to 128 entries are kept two per lane).
 */
int crh_merge_topk(const float* in_score, const int32_t* in_idx, int n_lists, int64_t n_users,
                   int k_in, int k_out, float* out_score, int32_t* out_idx, void* stream);

/*
 * One BPR training step's loss + gradients (everything but the optimiser).
 * Replaces model/MF.py:21-26 / model/LightGCN.py:23-27:
 *     u, p, n = user_tab[user_idx], pos_tab[pos_idx], neg_tab[neg_idx]        (list-index gather)
 *     loss    = bpr_loss(u,p,n) + l2_reg_loss(reg,u,p,n)      util/utils.py:25-29,44-48
 *     loss.backward()      (index backward = index_put_(accumulate=True) into dense gradients)
 * tables (rows, d) fp32, d % 4 == 0, 16-byte aligned; *_idx (batch) int32 or NULL (= rows 0..batch-1,
 * i.e. the tensors are already gathered: this is then bpr_loss/l2_reg_loss on (B,d) tensors).
 * grad_* are dense tables the gradients are ACCUMULATED into (zero them first; pos and neg may be the
 * same table); pass all three NULL for forward only.  loss_out[0] = bpr, loss_out[1] = l2 (device).
 * plan (device int32, or NULL): the batch's reverse index -- [nu, ni, L, user rows[L], offsets[L+1],
 * triple ids[L], item rows[2L], offsets[2L+1], entries[2L] = b | role<<30, n_heavy, heavy slots[3L/T+2]], T = crh_bpr_heavy_threshold(),
 * L >= batch the layout size; a heavy slot names a row with more than T entries (r < nu: user row r, else item
 * row r - nu), ascending -- built by crh_bpr_plan_build_host or on the device (crh_bpr_plan_build).  With a
 * plan every touched gradient row is summed in a fixed order (by one lane group; a heavy row by one workgroup, in
 * the same launch) and STORED (deterministic, no atomics; rows not touched are left as they are, i.e. zero);
 * it requires pos_table == neg_table and grad_pos == grad_neg.  Without a plan rows are accumulated
 * with fp32 atomics.
 */
size_t crh_bpr_workspace_bytes(int64_t batch);
int64_t crh_bpr_plan_ints(int64_t batch);
int crh_bpr_heavy_threshold(void);   /* a row with more entries than this is on the plan's heavy list */
int crh_bpr_plan_build_host(const int32_t* user_idx_host, const int32_t* pos_idx_host,
                            const int32_t* neg_idx_host, int64_t batch, int64_t layout_batch,
                            int32_t* plan_out_host);   /* crh_bpr_plan_ints(layout_batch) ints */
/* Device version: the plans of all ceil(n_records / batch_size) batches of an epoch in one launch
 * (LDS bitonic sort per batch; batch_size <= 8192).  plans_out: n_batches * crh_bpr_plan_ints(batch_size). */
int crh_bpr_plan_build(const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx,
                       int64_t n_records, int64_t batch_size, int32_t* plans_out, void* stream);
/* Same plans for ANY batch size up to 524 288 (S-TRAIN-XL: 65 536): several workgroups per batch -- chunk sorts in LDS,
 * merge-path passes in global memory, segment emission, heavy lists -- a handful of launches for all batches of the
 * call.  Bit-identical to crh_bpr_plan_build where both apply.  Workspace: crh_bpr_plan_build_large_workspace_bytes. */
size_t crh_bpr_plan_build_large_workspace_bytes(int64_t n_records, int64_t batch_size);
int crh_bpr_plan_build_large(const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx,
                             int64_t n_records, int64_t batch_size, int32_t* plans_out, void* workspace,
                             size_t workspace_bytes, void* stream);
int crh_bpr_fwd_bwd_f32(const float* user_table, const float* pos_table, const float* neg_table, int d,
                        const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx,
                        int64_t batch, float reg, float* grad_user, float* grad_pos, float* grad_neg,
                        float* loss_out, const int32_t* plan, void* workspace, size_t workspace_bytes,
                        void* stream);

/*
 * The same step split for data-parallel training (the batch is sharded over the ranks,
 * tables replicated).  l2_reg_loss (util/utils.py:44-48) uses the Frobenius norm of the WHOLE gathered
 * batch and bpr_loss (util/utils.py:25-29) its mean, so the backward pass of a slice needs four sums over
 * all slices.  crh_bpr_fwd_f32 runs the forward over this rank's `batch` triples and leaves
 * sums_out[4] = {sum u^2, sum p^2, sum n^2, sum -log(1e-5+sigmoid(x))} (device); the caller all-reduces
 * them (RCCL, 16 bytes); crh_bpr_bwd_f32 then accumulates / stores this slice's gradient rows using the
 * global sums and `global_batch`, and writes loss_out = the GLOBAL [bpr, l2].  `workspace` must be the
 * buffer the forward of the same slice used (it carries the per-triple score differences).
 */
int crh_bpr_fwd_f32(const float* user_table, const float* pos_table, const float* neg_table, int d,
                    const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx, int64_t batch,
                    float* sums_out, void* workspace, size_t workspace_bytes, void* stream);
int crh_bpr_bwd_f32(const float* user_table, const float* pos_table, const float* neg_table, int d,
                    const int32_t* user_idx, const int32_t* pos_idx, const int32_t* neg_idx, int64_t batch,
                    int64_t global_batch, float reg, const float* sums, float* grad_user, float* grad_pos,
                    float* grad_neg, float* loss_out, const int32_t* plan, void* workspace,
                    size_t workspace_bytes, void* stream);

/*
 * Row-ownership split of the same backward, for the data-parallel TOUCHED-ROWS step (SURVEY.md 8(e), S-TRAIN-XL: the
 * dense gradient all-reduce of the split above is 5.6 GB per step there).  Every rank holds the whole batch, its plan,
 * the forward of the whole batch (crh_bpr_fwd_f32 over `workspace`) and its `sums`; rank r computes only the gradient rows
 * of the plan's row slots w with w % own_mod == own_rem -- each row summed by ONE rank in the plan's entry order, i.e. the
 * bits of the single-GPU launch of model/MF.py:22-26's backward.  The ranks then exchange whole rows: crh_rows_pack_f32
 * gathers the owned rows of the gradient table into crh_rows_pack_cap(batch, own_mod) slots of (row id, d floats) (ids past
 * the plan's rows: -1), one all-gather moves them, crh_rows_unpack_f32 STORES every rank's rows into the gradient table
 * (each row has exactly one owner: nothing is added), and crh_adam_rows_f32 runs over the whole plan on every replica.
 * item_table is the shared positive / negative table (a plan requires it); d <= 256.
 */
int crh_bpr_bwd_owned_f32(const float* user_table, const float* item_table, int d, const int32_t* user_idx,
                          const int32_t* pos_idx, const int32_t* neg_idx, int64_t batch, float reg, const float* sums,
                          float* grad_user, float* grad_item, float* loss_out, const int32_t* plan, int own_mod,
                          int own_rem, void* workspace, size_t workspace_bytes, void* stream);
int64_t crh_rows_pack_cap(int64_t batch, int own_mod);
int crh_rows_pack_f32(const float* table, const int32_t* plan, int64_t batch, int64_t user_rows, int d, int own_mod,
                      int own_rem, int32_t* out_ids, float* out_rows, void* stream);
int crh_rows_unpack_f32(float* table, const int32_t* ids, const float* rows, int64_t n, int d, void* stream);

/*
 * torch.optim.Adam(lr) defaults, dense, for up to two tensors in one launch (model/MF.py:14,27:
 * user table then item table, equal step counters).  Mirrors torch/optim/adam.py
 * _single_tensor_adam op for op; scalar factors are evaluated in double.  step starts at 1.
 * zero_grad != 0 also clears g (the next optimizer.zero_grad()).  n0, n1 multiples of 4; n1 may be 0.
 * step_scalars (device, 2 floats, or NULL): when given, the step-dependent factors
 * {sqrt(1-beta2^step), -lr/(1-beta1^step)} are read from memory instead of being derived from
 * `step`, so that a captured hipGraph of an epoch can be replayed with fresh values
 * (crh_adam_step_scalars_host computes them on the host).
 */
int crh_adam_dense_f32(float* p0, float* g0, float* m0, float* v0, int64_t n0,
                       float* p1, float* g1, float* m1, float* v1, int64_t n1,
                       double lr, double beta1, double beta2, double eps, int64_t step, int zero_grad,
                       const float* step_scalars, void* stream);
void crh_adam_step_scalars_host(double lr, double beta1, double beta2, int64_t step, float* out2_host);
void crh_adam_step_scalars_range_host(double lr, double beta1, double beta2, int64_t first_step, int64_t n,
                                      float* out_host);   /* n pairs, steps first_step .. first_step+n-1 */

/*
 * Touched-rows replay of the same dense Adam, for tables whose gradient is zero in almost every row (BPR-MF at
 * catalogue scale: SURVEY.md 8(d) S-TRAIN-XL touches 196 K of 11 M rows per step, and the dense pass is 99.5 % of
 * the step's HBM traffic).  Rows are independent and the update is elementwise, so a row that was not touched
 * between steps s and t is brought up to date later by replaying its zero-gradient steps s+1..t-1 in registers:
 * the same fp32 operations in the same order as crh_adam_dense_f32, hence the same bits.
 *   p, g, m, v   (n_rows, d) fp32 tables (users first); last_step (n_rows) int32, zero-initialised
 *   plan         the batch's reverse index (crh_bpr_plan_build*); its item rows are offset by user_rows
 *   scalar_table device floats [2 * (max_step + 1)]: entry s = crh_adam_step_scalars_host(lr, b1, b2, s)
 *   mode 0  catch-up: rows of the plan become valid for step-1   (before the forward pass of `step`)
 *   mode 1  step:     rows of the plan take step `step` with g = their gradient rows, which are cleared
 *   mode 2  flush:    ALL rows become valid for `step`            (before the tables are read or saved)
 * catch-up(t) -> crh_bpr_fwd_bwd_f32(plan) -> step(t) for t = 1, 2, ..., then flush(T), equals T dense steps.
 */
int crh_adam_rows_f32(float* p, float* g, float* m, float* v, int32_t* last_step, int64_t n_rows, int d,
                      const int32_t* plan, int64_t batch, int64_t user_rows, int64_t step,
                      const float* scalar_table, double beta1, double beta2, double eps, int mode, void* stream);

/*
 * CSR SpMM with the LightGCN layer sum fused (model/LightGCN.py:88-93 and its autograd):
 *     P = A * x ;  y = P (if y) ;  acc_out = (acc_in * s_in + P) * s_out (if acc_out; acc_in NULL = 0)
 * rowptr (n_rows+1) int64, col int32 ascending per row, val fp32 (util/databuilder.py:220-254,953-962
 * produce exactly this matrix, as COO); x, y, acc_* are (n_rows, d) fp32, d % 4 == 0.
 * acc_out may alias acc_in; outputs must not alias x.
 * sched (HOST struct of DEVICE pointers, or NULL): optional load-balancing schedule built once per
 * graph -- a list of n_seg work items, each a ROW, in any order (the host mirror orders them by descending
 * length so that the lane groups of a wave walk rows of similar length):
 *     seg_row[w] = the row; seg_slot[w] = -1: finished by one lane group, bit-identical to the edge-order fma
 *     chain; >= 0: the row is "heavy" (more than crh_spmm_segment_edges() edges) and skipped here;
 *     seg_ptr: not read (kept for layout compatibility, must be non-NULL);
 *     multi_row[m], m < n_multi: the heavy WORKGROUPS, which lead the grid (list the longest rows first) -- a heavy
 *     row is given to a whole workgroup whose lane groups split its edge list and combine their partial sums in a
 *     fixed order (deterministic; the association differs from the single chain);
 *     multi_count[m] = n_sub | sub << 8 (NULL = 1 everywhere): n_sub in {1, 2, 4} cuts the row's column slice into
 *     n_sub ranges, workgroup `sub` taking one of them with 1 / n_sub of the lanes per lane group (n_sub times the
 *     lane groups per row: for the few rows of thousands of edges that set a launch's critical path); a row with
 *     n_sub > 1 is listed n_sub times, sub = 0 .. n_sub - 1;
 *     multi_first / n_partial: not read; nnz = number of stored edges.
 * With a schedule, and when the dense operand is larger than one XCD's L2 but a half / quarter of its columns
 * is not (and the edge list is small against it), the feature columns are processed in 2 / 4 slices pinned
 * to XCDs (block b -> slice (b % 8) % slices) so the random row gathers stay in that XCD's L2.
 * workspace: unused since the heavy rows are combined on chip (crh_spmm_workspace_bytes returns 0).
 * Without a schedule one lane group walks each row (bit-identical to the edge-order fma chain).
 */
#define CRH_SPMM_SLAB_BUCKETS 12
typedef struct {
    const int32_t* seg_row;
    const int64_t* seg_ptr;
    const int32_t* seg_slot;
    int64_t n_seg;
    const int32_t* multi_row;
    const int32_t* multi_first;
    const int32_t* multi_count;
    int32_t n_multi;
    int64_t n_partial;
    int64_t nnz;
    /* optional (NULL = absent): seg_desc[w] = {row, first edge, edges, slot} as four int32 -- the work item, its edge
     * range and its class in ONE 16-byte load instead of three dependent ones (needs nnz < 2^31) */
    const int32_t* seg_desc;
    /* optional (NULL = absent), round 3: the light rows as a STREAM of 8-byte pairs in work-item order.  Record of a row
     * with cnt edges: pair 0 = {row, cnt}, pairs 1.. = {col, fp32 bits of val} in edge order, padded with {0, 0} to
     * `units` units of slab_lanes pairs (units = 1 for cnt <= slab_lanes - 1, else 1 + ceil((cnt - (slab_lanes - 1)) /
     * slab_lanes)).  Work items are ordered by descending length, so the records of one unit count are contiguous
     * ("bucket" b: work items slab_first[b] .. slab_first[b + 1] - 1, units slab_units[b], first pair slab_base[b]) and
     * the address of a record is arithmetic: the dependent chain of a row is {record} -> {gathers} -> {store} instead of
     * {descriptor} -> {edge list} -> {gathers} -> ...  Used when slab_lanes equals the launch's lanes per lane group
     * (crh_spmm_lane_group); the stream must be followed by 2 * slab_lanes readable pairs.  n_slab = light work items. */
    const void* slab;
    int32_t slab_lanes;
    int32_t slab_buckets;
    int64_t n_slab;
    int32_t slab_first[CRH_SPMM_SLAB_BUCKETS];
    int32_t slab_units[CRH_SPMM_SLAB_BUCKETS];
    int64_t slab_base[CRH_SPMM_SLAB_BUCKETS];
    /* layout version of THIS struct, CRH_SPMM_SCHED_VERSION: the meaning of multi_count changed in round 3 (segments per
     * heavy row -> n_sub | sub << 8 per heavy workgroup) and the struct grew; a schedule built for another layout is
     * refused (CRH_ERR_ARG) instead of being decoded as something else.  multi_count entries must have n_sub in {1, 2, 4}
     * and sub < n_sub (checked by the builder, coldrec_amd/ops.py SpmmSchedule). */
    int32_t version;
} crh_spmm_sched;
#define CRH_SPMM_SCHED_VERSION 4
int crh_spmm_segment_edges(void);
/* lanes per lane group (16 bytes of a row each) the SpMM launches use for an n_rows x d operand and nnz stored edges under
 * a schedule: d / 4 / (column slices), a power of two -- what a caller needs to lay out crh_spmm_sched::slab */
int crh_spmm_lane_group(int64_t n_rows, int d, int64_t nnz);
size_t crh_spmm_workspace_bytes(const crh_spmm_sched* sched, int d);
int crh_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                     const float* x, int d, float* y, const float* acc_in, float s_in,
                     float* acc_out, float s_out, const crh_spmm_sched* sched, void* workspace,
                     size_t workspace_bytes, void* stream);

/*
 * The last SpMM of LightGCN's backward pass with torch.optim.Adam fused into its epilogue (model/LightGCN.py:26-28):
 * g = (acc_in * s_in + A x) * s_out is the gradient of the embedding table p (also stored to acc_out if not NULL);
 * one Adam step with crh_adam_dense_f32's arithmetic (same bits) on (p, m, v) in place; zero_acc_in != 0 clears
 * acc_in's row after it was consumed (ready for the next step's gradient scatter; acc_in must then differ from x).
 */
int crh_spmm_csr_adam_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                          const float* x, int d, float* acc_in, float s_in, float* acc_out, float s_out,
                          const crh_spmm_sched* sched, float* p, float* m, float* v, double lr,
                          double beta1, double beta2, double eps, int64_t step,
                          const float* step_scalars, int zero_acc_in, void* stream);

/*
 * One optimiser step of model/MF.py:19-27 (gather, bpr_loss + l2_reg_loss, backward, torch.optim.Adam) in ONE
 * launch, for tables that live in cache (MovieLens / CiteULike scale) where a step is launch and memory latency.
 * Organised by table row: each row's gradient is summed in plan-list order (score differences recomputed from the
 * rows, no forward pass, no gradient table), Adam is applied in registers and the new row goes to the OTHER
 * parameter buffer (table_in != table_out; m, v in place).  The Frobenius norms a step needs are produced by
 * the PREVIOUS call from the rows it has just updated (multiplicity in the next batch x |row|^2), as partial sums:
 *   part_in   [n_parts_in][4]: (sum u^2, sum p^2, sum n^2) of THIS batch and the loss sum of the PREVIOUS one;
 *             from the previous call's part_out (n_parts_in = crh_mf_step_parts(rows, d)) or, for the first step
 *             of an epoch, the first crh_bpr_fwd_parts(batch, d) x 4 floats of crh_bpr_fwd_f32's workspace
 *   part_out  [crh_mf_step_parts][4]
 *   crh_mf_step_tables flattens an epoch's plans once: range (n_batches, rows) int2 = the row's slice of the
 *             batch's entries, (0,0) if untouched; entries (n_batches, 3*batch_size) int2 = the two other rows of
 *             the triple (table rows, users first; role << 30 in .x on the item side); mult (n_batches, rows) =
 *             multiplicities of the row in the batch (user rows: count; item rows: positives | negatives << 16)
 *   a step takes its batch's plan (heavy-row list), range and entries rows, and the NEXT batch's mult row
 *             (NULL for the last step of the epoch)
 *   loss_out[1] (l2) is written by this call; loss_out[0] (bpr) by the NEXT call through loss_prev_out
 *   (batch_prev = this call's batch) or by crh_mf_step_finish after the last step.
 *   step_scalars: device {sqrt(1-beta2^step), -lr/(1-beta1^step)} (crh_adam_step_scalars_range_host).
 * Deterministic (no atomics); equal to crh_bpr_fwd_bwd_f32(plan) + crh_adam_dense_f32 up to fp32 summation
 * order of the three norms.  d % 4 == 0, d <= 256, batch_size < 32768.
 */
int crh_bpr_fwd_parts(int64_t batch, int d);
int crh_mf_step_parts(int64_t n_rows, int d);
int crh_mf_step_tables(const int32_t* plans, const int32_t* user_idx, const int32_t* pos_idx,
                       const int32_t* neg_idx, int64_t n_records, int64_t batch_size, int64_t user_rows,
                       int64_t item_rows, int32_t* range_out, int32_t* mult_out, int32_t* entries_out,
                       void* stream);
int crh_mf_step_f32(const float* table_in, float* table_out, float* m, float* v, int64_t user_rows,
                    int64_t item_rows, int d, int64_t batch, float reg, const int32_t* plan,
                    const int32_t* range, const int32_t* entries, const int32_t* mult_next,
                    const float* part_in, int n_parts_in, float* part_out, float* loss_prev_out,
                    int64_t batch_prev, float* loss_out, double beta1, double beta2, double eps,
                    const float* step_scalars, void* stream);
int crh_mf_step_finish(const float* part_in, int n_parts_in, int64_t batch, float* loss_out, void* stream);
/*
 * north_star's "BPR loss + SGD update": torch.optim.SGD(lr) defaults (no momentum, no weight decay) in place of
 * torch.optim.Adam.  The reference itself trains with Adam (model/MF.py:14, model/LightGCN.py:16), so
 * these are the extra mode; canonical arithmetic p <- fma(-lr, g, p) (oracle/oracle_np.py sgd_dense, pinned to
 * torch.optim.SGD within fp32 rounding).  A zero gradient moves nothing, so the dense pass (crh_sgd_dense_f32, n % 4
 * == 0, zero_grad clears g), the pass over the rows of a batch's plan (crh_sgd_rows_f32: tables (rows, d), users
 * first, item rows of the plan offset by user_rows; clears the gradient rows it consumed) and the fused forms give
 * the same tables:
 *   crh_mf_step_sgd_f32   crh_mf_step_f32 without m / v / step_scalars: a row is read once and written once
 *   crh_spmm_csr_sgd_f32  crh_spmm_csr_adam_f32 with the SGD update in the epilogue
 */
int crh_sgd_dense_f32(float* p, float* g, int64_t n, double lr, int zero_grad, void* stream);
int crh_sgd_rows_f32(float* p, float* g, int d, const int32_t* plan, int64_t batch, int64_t user_rows, double lr,
                     void* stream);
int crh_mf_step_sgd_f32(const float* table_in, float* table_out, int64_t user_rows, int64_t item_rows, int d,
                        int64_t batch, float reg, const int32_t* plan, const int32_t* range, const int32_t* entries,
                        const int32_t* mult_next, const float* part_in, int n_parts_in, float* part_out,
                        float* loss_prev_out, int64_t batch_prev, float* loss_out, double lr, void* stream);
int crh_spmm_csr_sgd_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, const float* x,
                         int d, float* acc_in, float s_in, float* acc_out, float s_out, const crh_spmm_sched* sched,
                         float* p, double lr, int zero_acc_in, void* stream);

/*
 * l2_reg_loss(reg, *embeddings) of util/utils.py:44-48 = reg * sum_e |e|_F / rows(e), for plugins that call it with
 * their own tensors (2..6 of them, any shapes): crh_l2_norm_f32 leaves |x|_F of n contiguous floats in norm_out
 * (device scalar; deterministic two-stage reduction through `workspace`, crh_l2_workspace_bytes() bytes);
 * crh_l2_reg_bwd_f32 is the backward of one term: gx (+)= x * reg * grad_out / (rows * norm), 0 where norm == 0
 * (grad_out: device scalar or NULL = 1; accumulate != 0 adds to gx).
 */
size_t crh_l2_workspace_bytes(void);
int crh_l2_norm_f32(const float* x, int64_t n, float* norm_out, void* workspace, size_t workspace_bytes, void* stream);
int crh_l2_reg_bwd_f32(const float* x, int64_t n, int64_t rows, float reg, const float* norm, const float* grad_out,
                       float* gx, int accumulate, void* stream);

/*
 * Multi-GPU exchange steps over RCCL (the reference is single-process, nothing to cite): one
 * communicator per process / GPU.  crh_comm_unique_id (rank 0; 128 bytes, hand them to the other ranks out of band)
 * -> crh_comm_init on every rank (collective) -> the two collectives below, asynchronous on `stream` -> destroy.
 *   crh_comm_allgather_topk  every rank's (n_users, k) shard lists -> gathered_* laid out [rank][user][k], which is
 *                            crh_merge_topk's input with n_lists = world (eval over a row-sharded item table).  ONE
 *                            collective: scores and ids travel packed as (n_users, 2k) 32-bit words (the layout
 *                            coldrec_amd/eval.py sends through torch.distributed) through the caller's device
 *                            `workspace` of crh_comm_allgather_topk_workspace_bytes(world, n_users, k) bytes
 *   crh_comm_allreduce_f32   in-place sum of n floats: the 4 batch sums of crh_bpr_fwd_f32, the dense gradient table
 *   crh_comm_allgather_rows  every rank's crh_rows_pack_cap slots of (row id, d floats) -> gathered_* laid out [rank][slot],
 *                            crh_rows_unpack_f32's input with n = world * cap (the touched-rows step over the ranks)
 * librccl.so is dlopen'ed on first use (an already loaded copy is reused); single-GPU callers never load it.
 * coldrec_amd's Python host layer issues the same collectives through torch.distributed ("nccl" = RCCL).
 */
typedef struct crh_comm crh_comm;
int crh_comm_unique_id(void* id128_host);
crh_comm* crh_comm_init(int rank, int world, const void* id128_host);
int crh_comm_destroy(crh_comm* c);
int crh_comm_rank(const crh_comm* c);
int crh_comm_world(const crh_comm* c);
int crh_comm_allreduce_f32(crh_comm* c, float* buf, int64_t n, void* stream);
int crh_comm_allgather_rows(crh_comm* c, const int32_t* ids, const float* rows, int64_t cap, int d, int32_t* gathered_ids,
                            float* gathered_rows, void* stream);
size_t crh_comm_allgather_topk_workspace_bytes(int world, int64_t n_users, int k);
int crh_comm_allgather_topk(crh_comm* c, const float* score, const int32_t* idx, int64_t n_users, int k,
                            float* gathered_score, int32_t* gathered_idx, void* workspace, size_t workspace_bytes,
                            void* stream);

/*
 * HOST-side negative sampler reproducing util/utils.py:123-157 (next_batch_pairwise) and NumPy's
 * legacy MT19937 stream bit for bit (np.random.seed / shuffle / choice), on internal ids.
 * All pointers are HOST pointers.  rec_* are the training records in file order; n_items_seen =
 * len(data.item) (negatives are drawn from every id in the item table, cold ones included).
 * crh_sampler_epoch fills one epoch (n_records triples, batches concatenated, last one short) and
 * keeps the cumulative in-place shuffle of the reference across epochs.
 * set/get_state exchange the 624-word key + position with np.random.get_state()/set_state().
 */
typedef struct crh_sampler crh_sampler;
crh_sampler* crh_sampler_create(const int32_t* rec_user_host, const int32_t* rec_item_host,
                                int64_t n_records, int32_t n_users, int32_t n_items_seen);
void crh_sampler_destroy(crh_sampler* s);
int crh_sampler_seed(crh_sampler* s, uint32_t seed);
int crh_sampler_set_state(crh_sampler* s, const uint32_t* key624_host, int pos);
int crh_sampler_get_state(const crh_sampler* s, uint32_t* key624_host, int* pos_host);
/* One epoch in the background: _async queues it for the sampler's persistent worker thread (kept spinning between
 * epochs so that its core stays at speed) and returns; _wait blocks until the epoch is complete and returns its code.
 * Between the two calls the output arrays and the sampler belong to the worker. */
int crh_sampler_epoch_async(crh_sampler* s, int64_t batch_size, int32_t* user_out_host, int32_t* pos_out_host,
                            int32_t* neg_out_host, int snapshot_first);   /* != 0: crh_sampler_snapshot on the worker first */
int crh_sampler_epoch_wait(crh_sampler* s);
/* save / bring back everything an epoch call advances (generators + cumulative permutation): speculative sampling.
 * _snapshot copies no table: the crh_sampler_epoch that follows logs the targets of its swaps and _restore replays them
 * backwards; anything else that moves the permutation under a snapshot turns it into a plain copy first.  _restore may be
 * called again after further epochs: the snapshot stays armed until the next _snapshot. */
int crh_sampler_snapshot(crh_sampler* s);
int crh_sampler_restore(crh_sampler* s);
int crh_sampler_epoch(crh_sampler* s, int64_t batch_size, int32_t* user_out_host,
                      int32_t* pos_out_host, int32_t* neg_out_host);

/*
 * The other samplers of util/utils.py, same conventions (host pointers, internal ids, one call
 * per epoch, records concatenated in shuffled order, the cumulative shuffle shared with crh_sampler_epoch).
 * They draw from CPython's `random` module stream (set/get_py_state exchange random.getstate()[1] = 624 key
 * words + position) and, where the reference does, from NumPy's (set/get_state above).
 * crh_sampler_set_catalogue must be called once first: n_users_seen = len(data.user) (pool of negative users),
 * item_is_cold[n_items_seen] = 1 for data.mapped_cold_item_idx (excluded from the CLCRec / CCFCRec candidate
 * pools), NULL = no cold item.
 *   crh_sampler_epoch_lara     util/utils.py:160-188  neg_user/neg_item: (n_records, n_negs)
 *   crh_sampler_epoch_clcrec   util/utils.py:191-233  item_out: (n_records, 1 + n_negs) = positive, then
 *                              random.sample(candidates, n_negs); sample_setsize = 21 (+ 4**ceil(log(3*n_negs, 4))
 *                              when n_negs > 5), the pool / selected-set switch of random.sample
 *   crh_sampler_epoch_ccfcrec  util/utils.py:237-300  pos_items (n, P) [NumPy stream], neg_items (n, P*N),
 *                              self_neg (n, S), neg_user (n)
 *   crh_sampler_epoch_cgrc     util/utils.py:303-336  NumPy stream only; per batch the item set in CPython's
 *                              list(set) order: bset_out[bset_ptr_out[b] .. bset_ptr_out[b+1]), n_batches + 1 offsets
 * crh_sampler_min_candidates = the smallest candidate pool over the training users (-1 before set_catalogue).
 */
int crh_sampler_set_catalogue(crh_sampler* s, int32_t n_users_seen, const uint8_t* item_is_cold_host);
int crh_sampler_set_py_state(crh_sampler* s, const uint32_t* key624_host, int pos);
int crh_sampler_get_py_state(const crh_sampler* s, uint32_t* key624_host, int* pos_host);
int64_t crh_sampler_min_candidates(const crh_sampler* s);
int crh_sampler_epoch_lara(crh_sampler* s, int32_t n_negs, int32_t* user_out_host, int32_t* item_out_host,
                           int32_t* neg_user_out_host, int32_t* neg_item_out_host);
int crh_sampler_epoch_clcrec(crh_sampler* s, int32_t n_negs, int64_t sample_setsize, int32_t* user_out_host,
                             int32_t* item_out_host);
int crh_sampler_epoch_ccfcrec(crh_sampler* s, int32_t positive_number, int32_t negative_number,
                              int32_t self_neg_number, int32_t* user_out_host, int32_t* item_out_host,
                              int32_t* neg_user_out_host, int32_t* pos_items_out_host,
                              int32_t* neg_items_out_host, int32_t* self_neg_out_host);
int crh_sampler_epoch_cgrc(crh_sampler* s, int64_t batch_size, int32_t ranking_neg_per_user,
                           int32_t* user_out_host, int32_t* item_out_host, int64_t* bset_ptr_out_host,
                           int32_t* bset_out_host, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* COLDREC_HIP_H */
