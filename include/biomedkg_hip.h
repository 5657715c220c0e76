/* C-ABI of libbiomedkg_hip.so -- the drop-in boundary of the MI355X-native BioMedKG hot path.
 *
 * The reference (HySonLab/BioMedKG) is pure Python: it has no FFI of its own.  The boundary a
 * maintainer binds is therefore the set of third-party / in-repo operator calls its hot path makes;
 * every entry point below names the reference call site it replaces (paths relative to the reference
 * root) and INTEGRATION.md shows the ctypes stub.  All pointers are DEVICE pointers unless marked
 * (host); matrices are row-major with an explicit leading dimension in ELEMENTS; `stream` is a
 * hipStream_t.  Every function returns 0 on success, a hipError_t (> 0) or -1 (bad argument)
 * otherwise, with text available from bkg_last_error().  fp32 = IEEE binary32 throughout.
 *
 * Every entry point only enqueues on `stream` (no synchronisation, no allocation) and may be recorded into a hipGraph:
 * fills are kernels of the library, never runtime memsets (a hipMemsetAsync node re-executes wrongly on ROCm 7.2), and
 * bkg_sort_edges uses the merge sort while its stream is being captured (the radix sort's one-sweep form zeroes its
 * histograms with runtime memsets).
 */
#ifndef BIOMEDKG_HIP_H
#define BIOMEDKG_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int bkg_abi_version(void);
const char* bkg_last_error(void); /* (host) thread-local text of the last failure */

/* Deterministic mode (Lightning's Trainer(deterministic=True): reference train_gcl.py:66, train_kge.py:46).  With a
 * workspace set, the split-K weight-gradient products (bkg_gemm_f32 deep-K form, bkg_gemm_grouped_k_f32, bkg_gemm_tn_bf16,
 * bkg_gemm_grouped_k_bf16) stop adding their K chunks with float atomics: every chunk writes its [M, N] tile set into a
 * slab of the workspace and a second launch sums the slabs in chunk order -- bitwise repeatable.  The workspace (device,
 * caller-owned, >= chunks * M * N floats for the largest product: 64 Mi floats cover the bench step) is used by launches
 * on ONE stream at a time; NULL switches back to atomics.  Everything else on the path is already order-fixed. */
int bkg_set_deterministic_workspace(float* workspace, int64_t floats);

/* hipGraph replay of a captured training step (biomedkg_amd/graphs.py): kernel arguments are frozen at capture, so
 * the in-kernel Philox streams (layer dropout of bkg_spmm_csr_f32, bkg_mask_cast_bf16) and ReDAF's keep hash (bkg_redaf_combine_f32, bkg_redaf_fused_xw) add
 * `*epoch << 16` to their stream id -- a DEVICE counter the captured step increments once per replay.  NULL = off. */
int bkg_set_rng_epoch(const uint32_t* epoch);

/* ---- graph preparation (replaces PyG's COO bookkeeping inside GCNConv / RGCNConv /
 *      negative_sampling: biomedkg/model/encoder.py:52,57,155,160; biomedkg/kge_module.py:93) */
int64_t bkg_sort_workspace_bytes(int64_t n);
/* stable radix sort of keys lo[i] + hi[i]*n_lo (hi may be NULL); outputs sorted keys + source positions */
int bkg_sort_edges(const int64_t* lo, const int64_t* hi, int64_t n_lo, int64_t n_hi, int64_t n, uint32_t* keys_tmp,
                   uint32_t* iota_tmp, uint32_t* sorted_keys, uint32_t* perm, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* bkg_sort_edges followed by bkg_lower_bound_u32(sorted_keys, n, n_queries, stride, rowptr) -- the row pointers of the CSR the
 * sort is for -- in one call; key sets of <= 8192 entries: ONE single-workgroup launch (sort and search in LDS). */
int bkg_sort_edges_rowptr(const int64_t* lo, const int64_t* hi, int64_t n_lo, int64_t n_hi, int64_t n, uint32_t* keys_tmp,
                          uint32_t* iota_tmp, uint32_t* sorted_keys, uint32_t* perm, int64_t n_queries, uint32_t stride,
                          int32_t* rowptr, void* workspace, int64_t workspace_bytes, void* stream);
/* The outputs of bkg_sort_edges_rowptr for keys that are ALREADY ascending (a NeighborLoader batch lists its edges target by
 * target in local-id order): sorted_keys = the keys, perm = identity, rowptr by binary search -- no sort.  The caller vouches
 * for the order. */
int bkg_keys_rowptr_presorted(const int64_t* lo, const int64_t* hi, int64_t n_lo, int64_t n_hi, int64_t n, uint32_t* sorted_keys,
                              uint32_t* perm, int64_t n_queries, uint32_t stride, int32_t* rowptr, void* stream);
/* out[s] = lower_bound(sorted, s*stride), s in [0, n_queries) -- CSR row pointers from sorted keys */
int bkg_lower_bound_u32(const uint32_t* sorted, int64_t n, int64_t n_queries, uint32_t stride, int32_t* out,
                        void* stream);
int bkg_gather_i64_to_i32(const int64_t* src, const uint32_t* idx, int32_t* out, int64_t n, void* stream);
int bkg_gather_i32(const int32_t* src, const int32_t* idx, int32_t* out, int64_t n, void* stream);
int bkg_gather_f32(const float* src, const int32_t* idx, float* out, int64_t n, void* stream);
/* A dropout_edge view as CSRs of its own (biomedkg/model/gcl.py:42-43,76: PyG dropout_edge hands GCNConv the KEPT edges
 * only): both CSRs of the graph compacted to the view's entries (kept, not a self loop) with their gcn_norm weights
 * deg^-1/2[row] deg^-1/2[col] (degrees of the view: kept in-edges + the unit loop; self_w = 1 / deg) -- the propagates of
 * an augmented step then gather 60 % of the rows at p = 0.4 instead of all of them.  ..._d: the CSR by target (col =
 * sources, perm = original edge of each entry, row = its target), ..._s: by source.  rowptr_out_* [n + 1], col_out_* /
 * w_out_* [nnz] (the first counts[0] / counts[1] entries are meaningful; no host read); dis [n], flag_tmp / offs_tmp
 * [2 nnz]: scratch. */
int64_t bkg_gcn_view_workspace_bytes(int64_t nnz);
/* The same cut for ONE CSR that already carries the view's weights (a node-partitioned graph's CSRs: columns are table
 * slots, the self loops are entries, weights from the global degrees -- biomedkg_amd/partition.py): the entries whose
 * weight is not zero as a CSR of its own over the same rows. */
int64_t bkg_csr_compact_workspace_bytes(int64_t nnz);
int bkg_csr_compact(const int32_t* rowptr, const int32_t* col, const float* w, int64_t n_rows, int64_t nnz, int32_t* flag_tmp,
                    int32_t* offs_tmp, int32_t* rowptr_out, int32_t* col_out, float* w_out, int32_t* count, void* workspace,
                    int64_t workspace_bytes, void* stream);
int bkg_gcn_view(const int32_t* rowptr_d, const int32_t* col_d, const int32_t* perm_d, const int32_t* row_d,
                 const int32_t* rowptr_s, const int32_t* col_s, const int32_t* perm_s, const int32_t* row_s, const uint8_t* keep,
                 int64_t n_nodes, int64_t nnz, int32_t* flag_tmp, int32_t* offs_tmp, float* dis, float* self_w,
                 int32_t* rowptr_out_d, int32_t* col_out_d, float* w_out_d, int32_t* rowptr_out_s, int32_t* col_out_s,
                 float* w_out_s, int32_t* counts, void* workspace, int64_t workspace_bytes, void* stream);
/* PyG gcn_norm (add_remaining_self_loops + deg^-1/2 by target) over the dst-sorted CSR; keep = optional
 * dropout_edge mask [E] in edge order (biomedkg/model/gcl.py:42-43,76).  Outputs dis[N], self_w[N], w_edge[E]. */
int bkg_gcn_norm(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int64_t* src,
                 const int64_t* dst, const uint8_t* keep, int64_t n_nodes, int64_t n_edges, float* dis, float* self_w,
                 float* w_edge, void* stream);
/* RGCNConv(aggr="mean") (relation, target) segments over edges sorted by rel*N + dst */
int bkg_rgcn_segments(const uint32_t* sorted, int64_t n, int64_t n_nodes, int32_t* flag_tmp, int32_t* scan_tmp,
                      int32_t* seg_of, int32_t* seg_ptr, int32_t* seg_dst, int32_t* seg_rel, float* inv_cnt,
                      int32_t* n_seg, void* workspace, int64_t workspace_bytes, void* stream);
/* per-sorted-edge segment id and 1 / |segment| back in original edge order (perm from bkg_sort_edges) */
/* bkg_rgcn_segments with a FIXED segment count n (padded mini-batches of a captured step: no host read of n_seg):
 * slots past the real count are empty segments of relation last_rel whose targets are spread over the nodes */
int bkg_rgcn_segments_static(const uint32_t* sorted, int64_t n, int64_t n_nodes, int32_t last_rel, int32_t* flag_tmp,
                             int32_t* scan_tmp, int32_t* seg_of, int32_t* seg_ptr, int32_t* seg_dst, int32_t* seg_rel,
                             float* inv_cnt, int32_t* n_seg, void* workspace, int64_t workspace_bytes, void* stream);
int bkg_rgcn_edge_order(const uint32_t* perm, const int32_t* seg_of, const float* inv_cnt, int64_t n,
                        int64_t* seg_of_edge, float* w_edge, void* stream);

/* Long-row splitting of a CSR for the segment-reduce kernels: rows longer than `chunk` entries are cut
 * into chunks summed by separate waves (partials added in chunk order by a second pass).  All arrays are
 * sized by the upper bounds n_items_max = n_rows + nnz/chunk, n_slots_max = 2*(nnz/chunk) + 2,
 * n_long_max = nnz/chunk + 1, n_short_max = n_rows; totals[4] = {n_items, n_slots, n_long, n_short} stays on the
 * device (no host sync). */
typedef struct bkg_rowsplit {
  const int32_t* item_row;   /* [n_items_max] */
  const int32_t* item_beg;   /* [n_items_max] entry range of the item */
  const int32_t* item_end;
  const int32_t* item_slot;  /* [n_items_max] partial slot, or -1: the item is a whole row */
  const int32_t* long_row;   /* [n_long_max] */
  const int32_t* long_slot0; /* [n_long_max] first slot of the row */
  const int32_t* long_cnt;   /* [n_long_max] number of slots */
  const int32_t* short_desc; /* [n_short_max][4] {row, first entry, length, 0} of the rows with <= 8 entries when the table
                                was built with pack != 0 (16-byte aligned; they are then NOT in the item list), else NULL */
  const int32_t* short_col;  /* [n_short_max][8] their column indices, padded with the row's last one (16-byte aligned) */
  const int32_t* totals;     /* [4] device: n_items, n_slots, n_long, n_short */
  int64_t n_items_max, n_slots_max, n_long_max, n_short_max;
} bkg_rowsplit;
/* pack != 0 (the tables bkg_spmm_csr_f32 walks): rows of 0..8 entries go to short_desc instead of the item list and are
 * finished several per wave (one row per group of 4..16 lanes).  0 for the tables the KGE kernels read.
 * cnt_tmp / scan_tmp: [4 * (n_rows + 1)]. */
int bkg_rowsplit_build(const int32_t* rowptr, int64_t n_rows, int64_t nnz, int32_t chunk, int32_t pack, int32_t* cnt_tmp,
                       int32_t* scan_tmp, int32_t* item_row, int32_t* item_beg, int32_t* item_end,
                       int32_t* item_slot, int32_t* long_row, int32_t* long_slot0, int32_t* long_cnt,
                       int32_t* short_desc /* [n_rows][4], nullable when pack == 0 */,
                       const int32_t* col /* [nnz] the CSR's column indices (read when pack != 0) */,
                       int32_t* short_col /* [n_rows][8], nullable when pack == 0 */, int32_t* totals /* [4] */,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* ---- message passing: out[i,:] = epi(sum_k w[k] x[col[k],:] (+ self_w[i] x[i,:]) (+ addend[i,:]))
 *      epi = dropout(relu(. + bias)).  Replaces MessagePassing.propagate + scatter of GCNConv / RGCNConv and
 *      the ReLU + F.dropout(0.2) that follow (biomedkg/model/encoder.py:50-59,153-162).  Dropout keep decisions:
 *      explicit uint8 mask `keep` [n_rows,d], or in-kernel Philox(seed, stream_id) when keep == NULL and drop_p > 0.
 *      split (host, nullable) + partials [n_slots_max, d] enable long-row splitting.
 *      x is the gathered table: fp32 [*, ldx] (x_is_bf16 = 0) or bf16 [*, ldx] (x_is_bf16 = 1, "bf16 storage / fp32
 *      accumulate" of BASELINE configs[2]); out, the partial sums and every epilogue operand are fp32.
 *      bkg_cast_f32_bf16 makes such a table: dst = bf16(gate ? (gate > 0 ? src * gate_scale : 0) : src), n % 8 == 0. */
int bkg_cast_f32_bf16(const float* src, const float* gate, float gate_scale, void* dst, int64_t n, void* stream);
/*      typed form: src and gate may each be fp32 or bf16 (bf16 hidden activations / gradients of the bf16 mode) */
/*      feature masking of the GCL augmentations (biomedkg/model/gcl.py:42,76: x * (rand >= p), no rescale) fused with
 *      the cast: dst = bf16(Philox(seed, stream_id, i) >= p ? src[i] : 0) */
int bkg_mask_cast_bf16(const void* src, int32_t src_is_bf16, float p, uint64_t seed, uint32_t stream_id, void* dst, int64_t n,
                       void* stream);   /* src: fp32, or the bf16 copy of the same features (the same result, half the bytes) */
int bkg_gate_cast_bf16(const void* src, int32_t src_is_bf16, const void* gate, int32_t gate_is_bf16, float gate_scale,
                       void* dst, int64_t n, void* stream);
int bkg_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* w, const void* x, int32_t x_is_bf16,
                     int64_t ldx, void* out /* fp32, or bf16 when out_is_bf16: a result only GEMMs / gathers read */,
                     int32_t out_is_bf16, int64_t ldo, int64_t n_rows, int32_t d, const float* self_w,
                     const void* addend /* fp32, or bf16 when addend_is_bf16: RGCNConv's root term */, int32_t addend_is_bf16,
                     int64_t ld_add, const float* bias, int32_t relu, const uint8_t* keep, float drop_p, uint64_t seed,
                     uint32_t stream_id,
                     const void* out_gate /* nullable bf16 [n_rows, ld_gate], with a bf16 out only: out = bf16(out_gate > 0 ?
                                             bf16(epi) * out_gate_scale : 0) -- the result is the gradient of ANOTHER layer's
                                             dropout(relu(.)) output `out_gate`, and leaves through that layer's backward gate
                                             (what bkg_gate_cast_bf16 over the finished table would give) */,
                     int64_t ld_gate, float out_gate_scale, const bkg_rowsplit* split, float* partials,
                     int32_t col_halves /* 1; 2 (bf16 table, split given, d % 16 == 0): the launch runs the two d / 2-column
                                           halves as two problems, each on four of the eight XCDs (an L2 then caches pieces
                                           of half the table's bytes); same sums per column, same dropout mask */,
                     void* stream);

/* ---- exact-fp32 MFMA GEMM: C = epi(alpha op(A) op(B)), epi = (+bias[col]) -> act (0 none, 1 relu, 2 elu) ->
 *      (* gate>0 ? gate_scale : 0) -> (+ beta C).  Replaces torch Linear inside GCNConv.lin, the GRACE/GGD/DGI heads
 *      (biomedkg/model/gcl.py:19-21,49-51,86-90), fusion projections (biomedkg/utils/fusion.py:17-20,70).
 *      Large-K plain products (weight gradients) run split-K with f32 atomics into a zeroed C.  Two shapes of the exact
 *      mode take pipelined (LDS-DMA ring) kernels of their own: A^T B over row-major [K, M] / [K, N] operands with
 *      128-aligned M, N (gemm_tn32.hip) and tall A[M, 256] x W with K = N = 256, no gate / beta (gemm_xw32.hip). */
/*      precision: 0 = exact fp32 (v_mfma_f32_32x32x2_f32, the parity default); 1 = operands rounded to bf16 while
 *      staged into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulate and fp32 C (BASELINE configs[2] "bf16" runs). */
int bkg_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                 int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* bias,
                 int32_t act, const float* gate, int64_t ldg, float gate_scale, int32_t precision, void* stream);
/* bf16-mode weight gradients with fp32 or bf16 operands (gradients / activations kept as bf16 tables by the producing
 * kernel): C[M,N] = A[K,M]^T B[K,N] split over K with f32 atomics (C is zeroed inside), and its grouped form
 * C[g] += A[rows g]^T B[rows g] (C zeroed by the caller).  bf16 MFMA operands, fp32 accumulate / result. */
int bkg_gemm_tn_bf16(int64_t M, int64_t N, int64_t K, const void* A, int32_t a_is_bf16, int64_t lda, const void* B,
                     int32_t b_is_bf16, int64_t ldb, float* C, int64_t ldc, void* stream);
int bkg_gemm_grouped_k_bf16(int64_t rows_total, int64_t M, int64_t N, const void* A, int32_t a_is_bf16, int64_t lda,
                            const void* B, int32_t b_is_bf16, int64_t ldb,
                            const int32_t* b_rows /* nullable: row k of B is B[b_rows[k]]; bf16 x bf16, M, N % 128 == 0 */,
                            float* C, int64_t ldc, int64_t strideC,
                            const int32_t* group_ptr, int32_t G, void* stream);
/* bf16-mode streaming form of the tall "activation x small weight" products (GCNConv.lin, heads, RGCNConv root and
 * per-relation weights; forward and input gradient): C[M,N] = epi(alpha * A[M,K] @ Wt[N,K]^T).  A is fp32 or bf16
 * (a_is_bf16), Wt is the bf16 [N, K] image made by bkg_weight_to_bf16 (out[b] = bf16(W[b]) or bf16(W[b]^T), W[b] =
 * W + b*stride_w of shape [rows, cols]), C is fp32 or bf16 (c_is_bf16; beta needs fp32).  K % 256 == 0, N % 32 == 0,
 * 16-byte aligned rows.  group_ptr != NULL: rows of group g use Wt + g*stride_w (device group_ptr[G+1]).
 * bf16 rows with K = N = 256 and no gate / beta -- every hidden layer of the bench step -- run the 8-wave form that keeps
 * the weight in registers and streams A through LDS in full lines (plain; grouped with G <= 63; gathered).  bf16 rows with
 * K = 512 | 768 and N a multiple of 256 (<= 2048), no groups / gate / beta -- ReDAF, AttentionFusion's two products, the
 * first GCN layer: reference biomedkg/utils/fusion.py:17-31,49-90, model/encoder.py:153-162 -- run the weight-slab form
 * (round 5: a 256-column slab of the weight in registers for the block's life, 32-row A tiles through an LDS-DMA ring,
 * the column groups of a row tile on one XCD; env BKG_XW_SLAB=0 keeps the streaming kernel).  Same results either way
 * (bit-equal at K = 768: one accumulator, k ascending). */
int bkg_weight_to_bf16(const float* W, int64_t ldw, int64_t stride_w, int64_t rows, int64_t cols, int32_t batch,
                       void* out /* [batch, rows, cols] bf16, nullable */, int64_t ldo, int64_t stride_o,
                       void* out_t /* [batch, cols, rows] bf16 (transposed), nullable */, int64_t ldt, int64_t stride_t,
                       void* stream);
int bkg_gemm_xw(int64_t M, int64_t N, int64_t K, float alpha, const void* A, int32_t a_is_bf16, int64_t lda,
                const int32_t* a_rows /* nullable: output row m reads A[a_rows[m]] (a gathered operand, never stored) */,
                const void* Wt, int64_t ldw, int64_t stride_w, const int32_t* group_ptr, int32_t G, float beta, void* C,
                int32_t c_is_bf16, int64_t ldc, const float* bias, int32_t act, const float* gate, int64_t ldg,
                float gate_scale, void* stream);
/* C = bf16(gate > 0 ? bf16(A Wt^T) * gate_scale : 0) for a 256 -> 256 product over bf16 tables, and colsum[c] = sum_m C[m, c]:
 * the input gradient of a GCNConv layer (reference biomedkg/model/encoder.py:153-162, autograd of `x = conv(x)`) already
 * through the ReLU / dropout backward of the layer below, with that layer's bias gradient -- no separate gate pass.
 * workspace: bkg_gemm_xw8_gated_workspace_floats(M) floats. */
int64_t bkg_gemm_xw8_gated_workspace_floats(int64_t M);
int bkg_gemm_xw8_gated(int64_t M, const void* A /* bf16 [M, 256] */, int64_t lda, const void* Wt /* bf16 [256, 256] */,
                       int64_t ldw, void* C /* bf16 [M, 256] */, int64_t ldc, const void* gate /* bf16 [M, 256] */,
                       int64_t ldg, float gate_scale, float* workspace, float* colsum /* [256] */, void* stream);
/* RGCNConv per-relation products: rows grouped by device-side group_ptr[G+1]; group g uses B + g*strideB */
int bkg_gemm_grouped_rows_f32(int32_t trans_b, int64_t M_total, int64_t N, int64_t K, const float* A, int64_t lda,
                              const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                              const int32_t* group_ptr, int32_t G, const float* gate, int64_t ldg, float gate_scale,
                              int32_t precision, void* stream);
/* C[g] += A[rows g]^T B[rows g] (C zeroed by the caller; split over row chunks of bkg_gemm_grouped_k_chunk rows, f32
 * atomics -- or, in deterministic mode, one workspace slab per chunk: rows_total / chunk + G + 1 slabs at most) */
int32_t bkg_gemm_grouped_k_chunk(int64_t rows_total, int64_t M, int64_t N, int32_t precision);
int bkg_gemm_grouped_k_f32(int64_t rows_total, int64_t M, int64_t N, const float* A, int64_t lda, const float* B,
                           int64_t ldb, float* C, int64_t ldc, int64_t strideC, const int32_t* group_ptr, int32_t G,
                           int32_t precision, void* stream);

/* ---- KGE triplet scorer: kind 0 = DistMult (biomedkg/model/decoder.py:45-49), 1 = TransE (:31-38).
 *      head/tail/rel are int64 [T] (the rows of edge_index / edge_type as the reference hands them over). */
int bkg_kge_score_f32(int32_t kind, const float* z, int64_t ldz, int32_t d, const int64_t* head, const int64_t* tail,
                      const int64_t* rel, int64_t T, const float* rel_emb, int64_t ldr, float* score, void* stream);
int bkg_row_l1norm_f32(const float* z, int64_t ldz, int64_t n, int32_t d, float* l1, void* stream);
/* dz[i] = sum over the incidences of node i (inc_rowptr/inc_perm: the 2T entries cat(head,tail) sorted by node;
 * entry e < T: i is the head of triplet e) -- segment reduce, plain stores, no atomics */
int bkg_kge_bwd_nodes_f32(int32_t kind, const void* z /* fp32 or bf16 table */, int32_t z_is_bf16, int64_t ldz,
                          int64_t n_nodes, int32_t d, const int64_t* head,
                          const int64_t* tail, const int64_t* rel, int64_t T, const float* rel_emb, int64_t ldr,
                          int64_t n_rel /* rows of rel_emb (relation ids < n_rel): staged in LDS when n_rel * d * 4 <= 64 KB;
                                           0: read from the caches */,
                          const float* gscore,
                          const void* rec /* nullable int32[T][4] {head, tail, rel, bits(g)} from the fused pass: replaces
                                             head / tail / rel / gscore (one 16-byte gather per incidence) */,
                          const float* l1, const int32_t* inc_rowptr, const int32_t* inc_perm,
                          float* dz, int64_t lddz, const bkg_rowsplit* split, float* partials, int32_t accumulate,
                          int64_t self_row0 /* incidence row i is node self_row0 + i of z / l1 (0: a whole graph; a
                                               node-partitioned rank reduces its own nodes' lists over the gathered table) */,
                          void* stream);
/* The same DistMult backward over per-incidence STREAMS in list order (what makes it a weighted segment reduce like
 * bkg_spmm_csr_f32: no entry id -> record -> row chain): bkg_kge_incidence_streams writes other[e] (the other endpoint),
 * relid[e] and g[e] = d score for the 2T entries in inc_perm order (other / relid nullable: a cached structure whose g alone
 * changes; rec: the fused pass's records instead of head / tail / rel / gscore); bkg_kge_bwd_stream_f32 then computes
 * dz[i] (+)= sum_e g[e] rel_emb[relid[e]] (*) z[other[e]] over node i's list (split: the row table of the lists, required;
 * the relation table, n_rel rows + a zero row for id n_rel, must fit 64 KB). */
int bkg_kge_incidence_streams(const int32_t* perm, int64_t n_inc, int64_t T, const int64_t* head, const int64_t* tail,
                              const int64_t* rel, const void* rec, const float* gscore, int32_t* other, int32_t* relid,
                              float* g, void* stream);
int bkg_kge_bwd_stream_f32(const void* z, int32_t z_is_bf16, int64_t ldz, int64_t n_nodes, int32_t d, const float* rel_emb,
                           int64_t ldr, int64_t n_rel, const int32_t* other, const int32_t* relid, const float* g, float* dz,
                           int64_t lddz, const bkg_rowsplit* split, float* partials, int32_t accumulate,
                           const float* tail_gloss /* nullable device scalar: the LAST launch over dz leaves
                                                      dz * tail_gloss[0] + tail_z * (tail_gloss[0] * tail_coef) -- the loss's upstream
                                                      gradient and the L2 term of kge_module.py:128-129 without another pass */,
                           const float* tail_z /* nullable [n_nodes, ld_tail_z] fp32 */, int64_t ld_tail_z, float tail_coef,
                           const float* acc_src /* nullable: an accumulating launch adds to THESE rows (dz is only written) */,
                           int64_t ld_acc_src, void* stream);
/* The whole DistMult TRAINING pass of a graph's POSITIVE triplets -- scores, BCE-with-logits terms, dz and d rel_emb -- in ONE
 * walk over incidence lists sorted by (node, role, relation) (reference biomedkg/kge_module.py:121-128 +
 * biomedkg/model/decoder.py:45-49 and their autograd): z_n stays in registers, every run of equal (role, relation) adds
 * rel (*) sum g z_other to dz_n, and each (head, relation) run leaves ONE record row z_n (*) sum g z_other that the caller adds
 * per relation into d rel_emb (fixed order: no atomics).  2 T row reads instead of the 4 T of scorer + node-side gradient.
 * items [n_items] int32x4 {node, first batch, end batch, partial slot or -1}; batches int32x4 {first entry, count | first_of_run
 * << 8 | last_of_run << 9, relation + n_rel * role (0: the node is the head), record row}; a batch holds at most
 * bkg_kge_onepass_batch(d, z_is_bf16) entries of ONE run.  loss_part: 4 * bkg_kge_onepass_blocks(n_items) unscaled partial sums;
 * dz is written (not added to); split: the row table the items were cut by.
 * bkg_kge_onepass_batch returns 0 for a width the walk has no instance for; bkg_kge_onepass_smem_bytes = the dynamic LDS one
 * workgroup asks for (the relation table + every wave's scratch rows; 0: unsupported width) -- the walk is taken only when it
 * fits 64 KB, else the caller keeps the two-pass form (scorer + node-side gradient). */
int32_t bkg_kge_onepass_batch(int32_t d, int32_t z_is_bf16);
int32_t bkg_kge_onepass_blocks(int64_t n_items);
int64_t bkg_kge_onepass_smem_bytes(int32_t d, int32_t z_is_bf16, int64_t n_rel);
int bkg_kge_onepass_f32(const void* z, int32_t z_is_bf16, int64_t ldz, int64_t n_nodes, int32_t d, const float* rel_emb,
                        int64_t ldr, int64_t n_rel, const int32_t* other, const int32_t* tid, const void* batches,
                        const void* items, int64_t n_items, float label, float inv_total, float* score /* nullable */,
                        float* loss_part, float* records, float* dz, int64_t lddz, const bkg_rowsplit* split, float* partials,
                        void* stream);
/* The DistMult TRAINING pass of ONE label's triplets -- the step's fresh NEGATIVES -- in one walk over their 2 T incidences
 * sorted by (node, role): scores, BCE-with-logits terms, dz and d rel_emb (reference biomedkg/kge_module.py:121-128 +
 * biomedkg/model/decoder.py:45-49 and their autograd; csrc/kge_onepass_mx.hip).  The node's row stays in registers, every
 * entry's OTHER row is read once (2 T row reads where scorer + node-side gradient read 4 T); d rel_emb -- a segmented sum
 * over a random relation id per entry -- goes to the matrix cores as a product with a one-hot matrix (the 16 entries of a
 * chunk staged in LDS as bf16 rows g z_n (*) o, read back k-major by the hardware transpose read): no atomics, no
 * relation order, no records.  bf16 tables of width 256 and <= 32 relations only (bkg_kge_onepass_mx_ok); everything else
 * keeps the two-pass form.  rowptr2 [2 n_nodes + 1]: first entry of key 2 * node + role (role 0: the node is the head);
 * packed / perm [n_inc]: the entries' (other endpoint | relation id << 24) and position in cat(head, tail) in list order
 * (perm only with score; bkg_kge_role_incidence builds rowptr2 and packed).  loss_part [4 * blocks] unscaled BCE sums; dz rows = the lists' sums (written);
 * drel_part [blocks][32][256] unscaled (the caller sums over the workgroups and multiplies by inv_total);
 * blocks = bkg_kge_onepass_mx_blocks(n_nodes). */
/* The lists that walk takes (csrc/graph.hip): the 2 T incidences cat(head, tail) sorted by key 2 * node + role (one key-making
 * launch, one radix sort of (key, payload) pairs, one lower bound; no host read).  keys_tmp / vals_tmp / sorted_keys: uint32
 * [2 T] scratch; workspace of bkg_sort_workspace_bytes(2 T).  Node ids < 2^24, relation ids < 256. */
int bkg_kge_role_incidence(const int64_t* head, const int64_t* tail, const int64_t* rel, int64_t T, int64_t n_nodes,
                           uint32_t* keys_tmp, uint32_t* vals_tmp, uint32_t* sorted_keys, int32_t* packed, int32_t* rowptr2,
                           void* workspace, int64_t workspace_bytes, void* stream);
int32_t bkg_kge_onepass_mx_ok(int32_t d, int32_t z_is_bf16, int64_t n_rel);
int32_t bkg_kge_onepass_mx_blocks(int64_t n_nodes);
int bkg_kge_onepass_mx_f32(const void* z, int64_t ldz, int64_t n_nodes, const float* rel_emb, int64_t ldr, int64_t n_rel,
                           const int32_t* rowptr2, const int32_t* packed, const int32_t* perm, int64_t n_inc, float label,
                           float inv_total, float* score /* nullable */, float* loss_part,
                           float* dz, int64_t lddz, float* drel_part, int32_t blocks, void* stream);
/* Fused training pass for one label (1 = positives, 0 = negatives): score[T], g[T] = (sigmoid(s) - label) *
 * inv_total, per-wave BCE-with-logits partial sums loss_part[W] and per-wave partial rows drel[W, lddr] of the
 * relation-side gradient (W = bkg_kge_fused_bce_waves(T, n_rel); relation r owns ceil(cnt_r / 64) consecutive waves;
 * the caller segment-reduces them), in one sweep over the triplets' rows visited in relation-sorted order.
 * Replaces decode + cat + binary_cross_entropy_with_logits of biomedkg/kge_module.py:121-127 and their autograd. */
int32_t bkg_kge_fused_chunk(void);   /* triplets per wave of the fused pass (the wave tables of its callers are cut by it) */
int64_t bkg_kge_fused_bce_waves(int64_t T, int32_t n_rel);
/* The scalar tail of a mini-batch step, one launch each (a captured step is bound by its launch count):
 * loss[0] = inv_total * sum(part[0..n_part)) + reg * (mean over rows < n_real of z^2 + mean of rel^2), the reference's
 * BCE mean + 1e-2 L2 (biomedkg/kge_module.py:127-129); inv_total_dev / n_real_dev (nullable) read the count / the real
 * row count from device memory; mean_rows > 0: the rows the mean of z^2 is taken over when z is a shard of them (a
 * node-partitioned step: N / world -- not an integer when world does not divide N, hence a double: the backward
 * normalises by the same value), 0 = the rows summed.  bkg_scale_add_rows_f32: dst = dst * g + src * (g * coef / rows) on the first `rows`
 * rows (n_real_dev[0], or n), dst * g beyond; g = gloss[0] -- the backward of that L2 term folded into the scaling of
 * the scorer's gradient by the upstream gradient. */
int64_t bkg_kge_loss_finish_scratch_bytes(void);   /* zero it once; the kernel leaves its ticket zero */
int bkg_kge_loss_finish_f32(const float* part, int32_t n_part, float inv_total, const float* inv_total_dev, const float* z,
                            int64_t ldz, int32_t n_cap, int32_t d, const int64_t* n_real_dev, double mean_rows,
                            const float* rel, int64_t ldr, int32_t n_rel, float reg, void* scratch, float* loss,
                            void* stream);
int bkg_scale_add_rows_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t n, int32_t d,
                           const float* gloss, float coef, const int64_t* n_real_dev, void* stream);
/* out = base * g + src * (g * coef / n), out of place (base: a tensor the autograd graph saved) */
int bkg_scale_add_rows_out_f32(const float* base, int64_t ldb, const float* src, int64_t lds, int32_t n, int32_t d,
                               const float* gloss, float coef, float* out, int64_t ldo, void* stream);
/* The node-side gradient of a step whose two one-pass walks ran in the forward (reference kge_module.py:127-129 through
 * autograd): out = (a + b) * gloss[0] + z * (gloss[0] * coef / n) -- the positives' sums, the negatives' sums (b nullable),
 * the upstream gradient and the L2 term in ONE pass (16 bytes per lane; d % 4 == 0, 16-byte aligned rows). */
int bkg_kge_dz_finish_f32(const float* a, int64_t lda, const float* b /* nullable */, int64_t ldb, const float* z /* nullable */,
                          int64_t ldz, int64_t n, int32_t d, const float* gloss, float coef, float* out, int64_t ldo,
                          void* stream);
/* wave_ptr[n_rel + 1]: first wave of each relation in that pass (chunk = 64 triplets per wave) */
int bkg_relation_wave_ptr(const int32_t* rel_ptr, int32_t n_rel, int32_t chunk, int32_t* wave_ptr, void* stream);
int bkg_kge_fused_bce_f32(int32_t kind, const void* z /* fp32 or bf16 table */, int32_t z_is_bf16, int64_t ldz,
                          int32_t d, const int64_t* head,
                          const int64_t* tail, int64_t T, float label, float inv_total, const float* rel_emb,
                          int64_t ldr, int32_t n_rel, const int32_t* order, const int32_t* rel_ptr, float* score,
                          float* g /* nullable when rec is given */,
                          void* rec /* nullable int32[T][4]: {head, tail, rel, bits(g)} for the backward */,
                          float* loss_part, float* drel, int64_t lddr, void* stream);
/* The same pass for the padded mini-batches of a captured step: 1 / total is read from device memory when
 * inv_total_dev is given, and triplets whose relation group lies past n_rel in rel_ptr (the pads' sentinel id) are
 * never visited. */
int bkg_kge_fused_bce_dev_f32(int32_t kind, const void* z, int32_t z_is_bf16, int64_t ldz, int32_t d,
                              const int64_t* head, const int64_t* tail, int64_t T, float label, float inv_total,
                              const float* inv_total_dev /* nullable */, const float* rel_emb, int64_t ldr, int32_t n_rel,
                              const int32_t* order, const int32_t* rel_ptr, float* score, float* g, void* rec,
                              float* loss_part, float* drel, int64_t lddr, void* stream);
/* drel[r] += sum_t ... (drel zeroed by the caller); order/rel_ptr (nullable together) = triplet ids sorted by
 * relation and the [n_rel+1] offsets of each relation in that order */
int bkg_kge_bwd_rel_f32(int32_t kind, const float* z, int64_t ldz, int32_t d, const int64_t* head, const int64_t* tail,
                        const int64_t* rel, int64_t T, const float* rel_emb, int64_t ldr, const float* gscore,
                        int32_t n_rel, float* drel, int64_t lddr, const int32_t* order, const int32_t* rel_ptr,
                        void* stream);

/* ---- NT-Xent (PyGCL DualBranchContrast(InfoNCE(tau), "L2L", intraview_negs=True): biomedkg/gcl_module.py:171-173,
 *      186-190) over Z = [normalize(h1); normalize(h2)] (M = 2N rows, h <= 256):
 *        L = 1/M sum_i ( log D_i - z_i.z_p(i)/tau ),  D_i = sum_{j != i} exp(z_i.z_j / tau),  p(i) = (i + N) mod M.
 *      bkg_ntxent_hpad: padded row width of the normalised images (fp32 operands: next multiple of 32; bf16: 128 | 256).
 *      bkg_ntxent_prepare: F.normalize of both views into zero-padded fp32 [2N, hpad] (and, if zn_bf16 != NULL, bf16)
 *        images, the row norms [2N] and the positive-pair dots [N].
 *      bkg_ntxent_forward: D over the Gram tiles on the matrix cores (bf16 = 0: exact fp32 MFMA on the fp32 image;
 *        1: bf16 operands from the bf16 image, fp32 exp / sums), using the symmetry of the Gram (upper-triangular
 *        tiles, row + column sums, fixed-order reduction: bitwise repeatable).  Outputs den[M], rden[M] = 1/D,
 *        term[M] (double, nullable) = log D_i - (i < N) 2 z_i.z_(i+N)/tau and loss[1] (nullable) = sum(term) / M, the
 *        terms added in double in a fixed order inside the last launch (no reduction left to the caller).
 *      bkg_ntxent_backward: dL/dh1, dL/dh2 [N, h] (times the upstream scalar *gout, a DEVICE float), tiles recomputed.
 *      Workspaces (floats) by the *_workspace_floats queries; every function is stream-ordered, no host sync. */
int32_t bkg_ntxent_hpad(int32_t h, int32_t bf16);
/* the bf16 image holds zn * bkg_ntxent_bf16_scale(tau) = zn * sqrt(log2(e) / tau): the Gram of two such rows is already
 * the exp2 argument, and scaling before the bf16 rounding costs no accuracy (the rounding is relative) */
float bkg_ntxent_bf16_scale(float tau);
int bkg_ntxent_prepare(const float* h1, int64_t ld1, const float* h2, int64_t ld2, int64_t N, int32_t h, int32_t hpad,
                       float tau, float* zn, void* zn_bf16, float* norm, float* posdot, void* stream);
int64_t bkg_ntxent_forward_workspace_floats(int64_t M, int32_t h, int32_t bf16);
int bkg_ntxent_forward(const void* zn, int32_t bf16, int64_t M, int32_t h, float tau, const float* posdot,
                       float* workspace, float* den, float* rden, double* term, float* loss, void* stream);
int64_t bkg_ntxent_backward_workspace_floats(int64_t M, int32_t h, int32_t bf16);
int bkg_ntxent_backward(const void* zn, int32_t bf16, const float* zn_f32, const float* norm, const float* rden,
                        const float* gout, int64_t M, int32_t h, float tau, float* workspace, float* dh1, int64_t ld1,
                        float* dh2, int64_t ld2, void* stream);
/*      Row-partitioned form (multi-GPU: the 2N rows are partitioned like the nodes): the rank's n_anchor anchor
 *      rows zn[anchors[a]] against ALL M (all-gathered) rows.  rows_forward: den[n_anchor] = D of the anchors.
 *      rows_backward: rden[M] = 1/D of all rows (all-gathered), norm[n_anchor]; anchor a < n_half -> dh1[a], the rest ->
 *      dh2[a - n_half]; gradient scaled by *gout * grad_scale.  W_ij = E_ij (1/D_i + 1/D_j) is symmetric, so a rank's
 *      rows need nothing back from the other ranks. */
int64_t bkg_ntxent_rows_forward_workspace_floats(int64_t M, int64_t n_anchor, int32_t h, int32_t bf16);
int bkg_ntxent_rows_forward(const void* zn, int32_t bf16, int64_t M, int32_t h, float tau, const int32_t* anchors,
                            int64_t n_anchor, float* workspace, float* den, void* stream);
int64_t bkg_ntxent_rows_backward_workspace_floats(int64_t M, int64_t n_anchor, int32_t h, int32_t bf16);
int bkg_ntxent_rows_backward(const void* zn, int32_t bf16, const float* zn_f32, const float* norm, const float* rden,
                             const float* gout, int64_t M, int32_t h, float tau, const int32_t* anchors, int64_t n_anchor,
                             int64_t n_half, float grad_scale, float* workspace, float* dh1, int64_t ld1, float* dh2,
                             int64_t ld2, void* stream);

/* ---- device mini-batch samplers (the reference's PyG loaders: biomedkg/data_module.py:89-107).
 *      bkg_random_walk: walks[w, 0..L] over the source-sorted CSR, uniform out-neighbour per step (sinks stay put).
 *      bkg_sample_neighbors: out_pos[f, 0..fanout) = CSR positions (destination-sorted CSR) of up to `fanout`
 *      distinct uniformly chosen in-edges of frontier[f], -1 padded; fanout <= 64.  Philox keyed by (seed, draw). */
/*      tape (nullable, every sampler entry point): raw 32-bit words that REPLACE the Philox stream -- the draw of an index
 *      below n is (word * n) >> 32 either way -- so that the CPU restatement of the two loaders (oracle/samplers.py) and
 *      these kernels can be fed the same draws: walks read tape[w * walk_length + step]; neighbour picks read
 *      tape[node * 64 + Floyd step] (keyed by the node, which is expanded once, not by its place in the frontier). */
int bkg_random_walk(const int32_t* rowptr, const int32_t* col, const int64_t* start, int64_t n_walks,
                    int32_t walk_length, uint64_t seed, const uint32_t* tape, int64_t* walks, void* stream);
int bkg_sample_neighbors(const int32_t* rowptr, const int64_t* frontier, int64_t n_frontier, int32_t fanout,
                         uint64_t seed, uint32_t hop, const uint32_t* tape, int32_t* out_pos, void* stream);
/* NeighborLoader(num_neighbors=[..]) batches (biomedkg/data_module.py:89-99) with ONE host read at the end instead of
 * three per hop: begin (seeds -> n_id[0..n_seeds), seen flags, ctr = {frontier start, nodes, edges}), one hop call per
 * fan-out (up to `fanout` distinct uniformly chosen in-edges per frontier node; new sources appended to n_id in
 * ascending id order and becoming the next frontier; the picked edges appended as (global source, global target,
 * original edge id) in (frontier row, pick) order), finish (global -> local ids).  Same draws and orders as
 * bkg_sample_neighbors driven hop by hop from the host.  Capacities: seen / n_id / lut / new_flag / node_offs: n_nodes;
 * pos / eflag / edge_offs: n_nodes * fanout; e_src / e_dst / e_id: edge_cap (n_nodes * the largest fan-out always
 * suffices: every node is expanded once); workspace of bkg_neighbor_workspace_bytes.  When eflag = new_flag + n_nodes and
 * edge_offs = node_offs + n_nodes (one array each, the slots' part behind the nodes'), a hop runs ONE scan over both (the
 * edge offsets as written then start at the hop's number of new nodes). */
int64_t bkg_neighbor_workspace_bytes(int64_t n_nodes, int32_t max_fanout);
int bkg_neighbor_begin(const int64_t* seeds, int64_t n_seeds, int64_t n_nodes, int32_t* seen, int64_t* n_id, int64_t* ctr,
                       void* stream);
int bkg_neighbor_hop(const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t n_nodes, int32_t fanout,
                     uint64_t seed, uint32_t hop, const uint32_t* tape, int32_t* seen, int64_t* n_id, int64_t* ctr, int32_t* pos,
                     int32_t* new_flag, int32_t* node_offs, int32_t* eflag, int32_t* edge_offs, int64_t edge_cap,
                     int64_t* e_src, int64_t* e_dst, int64_t* e_id, void* workspace, int64_t workspace_bytes, void* stream);
int bkg_neighbor_finish(int64_t n_nodes, const int64_t* n_id, const int64_t* ctr, int32_t* lut, int64_t edge_cap,
                        int64_t* e_src, int64_t* e_dst, void* stream);
/* One whole GraphSAINT random-walk batch (GraphSAINTRandomWalkSampler, biomedkg/data_module.py:100-107) with no host
 * round trip: walks from start[] (NULL: uniform roots drawn from the keyed stream), node set = distinct visited nodes in ascending order, edge set = ALL edges among them
 * in edge order, relabelled.  Sizes stay on the device: sizes[0] = nodes, sizes[1] = edges, sizes[2] = edges with
 * head != tail.  Outputs are padded to fixed capacities so that a captured (hipGraph) step can replay over them:
 * n_id[node_cap] (pads = n_nodes, the zero row the caller appends to the feature table), out_src / out_dst / out_type /
 * out_eid[edge_cap] (pads = edges into `pad_node`, the batch's dummy local node, from sources i % (pad_node + 1) with
 * types i % pad_types and eid -1 -- spread so that no row of a structure built over them grows long; edges past
 * edge_cap are dropped -- the caller compares sizes[1] with edge_cap).  epoch (nullable): device replay counter mixed into the walk seed.
 * Scratch: node_flag / node_offs / lut int32[n_nodes] (lut = local id or -1 survives), edge_flag / edge_offs
 * int32[n_edges], workspace of bkg_saint_batch_workspace_bytes. */
int64_t bkg_saint_batch_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int bkg_saint_batch(const int32_t* rowptr, const int32_t* col, const int64_t* start, int64_t n_walks,
                    int32_t walk_length, uint64_t seed, const uint32_t* epoch, const uint32_t* tape, int64_t* walks, const int64_t* src,
                    const int64_t* dst, const int64_t* type /* nullable */, int64_t n_nodes, int64_t n_edges,
                    int32_t* node_flag, int32_t* node_offs, int32_t* lut, int32_t* edge_flag, int32_t* edge_offs,
                    int64_t node_cap, int64_t edge_cap, int64_t pad_node, int64_t pad_types, int64_t* n_id, int64_t* out_src,
                    int64_t* out_dst, int64_t* out_type, int64_t* out_eid, int64_t* sizes, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* ---- ranking evaluation (filtered MRR / Hits@k for DistMult; build-defined: the reference has no ranking metric,
 *      BASELINE.json's north_star asks for one).  q [n_q, h] = z[anchor] * rel[r]; z [n_nodes, h] candidates, h <= 256.
 *      true[i] = q_i . z[target_i] as the MFMA's k-ordered fma chain; gt/eq_partial [splits, n_q] count candidates
 *      j != target_i with score > / == true[i]; bkg_rank_filter_f32 subtracts the known true candidates
 *      cand[qptr[i] .. qptr[i+1]) (filtered setting) from summed counts. */
int bkg_rank_true_score_f32(const float* q, int64_t ldq, const float* z, int64_t ldz, const int64_t* target,
                            int64_t n_q, int32_t h, float* out, void* stream);
int bkg_rank_count_f32(const float* q, int64_t ldq, int64_t n_q, const float* z, int64_t ldz, int64_t n_nodes,
                       int32_t h, const float* true_score, const int64_t* target, int32_t splits, int32_t* gt_partial,
                       int32_t* eq_partial, void* stream);
int bkg_rank_filter_f32(const float* q, int64_t ldq, const float* z, int64_t ldz, int32_t h, const float* true_score,
                        const int64_t* target, const int64_t* qptr, const int64_t* cand, int64_t n_q, int32_t* gt,
                        int32_t* eq, void* stream);

/* TransE variant (score = -||q - zn_j||_1 over the L1-normalised table zn; q = zn[h] + r for tails, zn[t] - r for
 * heads; d <= 768): same three steps as above, VALU kernels (csrc/rank_l1.hip). */
int bkg_rank_l1_true_f32(const float* q, int64_t ldq, const float* zn, int64_t ldz, const int64_t* target, int64_t n_q,
                         int32_t d, float* out, void* stream);
int bkg_rank_l1_count_f32(const float* q, int64_t ldq, int64_t n_q, const float* zn, int64_t ldz, int64_t n_nodes,
                          int32_t d, const float* true_score, const int64_t* target, int32_t splits,
                          int32_t* gt_partial, int32_t* eq_partial, void* stream);
int bkg_rank_l1_filter_f32(const float* q, int64_t ldq, const float* zn, int64_t ldz, int32_t d,
                           const float* true_score, const int64_t* target, const int64_t* qptr, const int64_t* cand,
                           int64_t n_q, int32_t* gt, int32_t* eq, void* stream);

/* ---- two views side by side (GGD / DGI corruption = a row permutation, biomedkg/model/gcl.py:69-71,85):
 *      out[i] = [x[i] | x[perm[i]]] over bf16 rows of width h (h % 8 == 0), and the gradient of that map:
 *      out[i] = g[i, :h] + g[inv[i], h:] with inv the inverse permutation. */
int bkg_pair_rows_bf16(const void* x, const int64_t* perm, void* out, int64_t n, int32_t h, void* stream);
int bkg_unpair_rows_bf16(const void* g, const int64_t* inv, void* out, int64_t n, int32_t h, void* stream);

/* ---- bias gradients: out[c] = sum_i x[i, c] over an fp32 or bf16 table (autograd of the `+ bias` of every Linear /
 *      GCNConv / RGCNConv on the path).  workspace: bkg_colsum_workspace_floats(n, d) floats. */
int64_t bkg_colsum_workspace_floats(int64_t n, int32_t d);
int bkg_colsum_f32(const void* x, int32_t x_is_bf16, int64_t ldx, int64_t n, int32_t d, float* workspace, float* out,
                   void* stream);
/* the gated bf16 table of bkg_gate_cast_bf16 over a contiguous [n, d] table (n > 4096) together with its column sums
 * (the bias gradient of the layer whose output gradient it is): one pass, same sums as bkg_colsum_f32 over the result */
int bkg_gate_cast_colsum_bf16(const void* src, int32_t src_is_bf16, const void* gate, int32_t gate_is_bf16, float gate_scale,
                              void* dst, int64_t n, int32_t d, float* workspace, float* out, void* stream);
/* The gradient of a row dot product through the gate of the layer below: dst[r, c] = bf16(gate[r, c] > 0 ? bf16(g[r] w[c]) *
 * gate_scale : 0) and out[c] = sum_r dst[r, c] -- bkg_scaled_rows (bf16) + bkg_gate_cast_colsum_bf16 in one pass (GGD's linear
 * head behind its last hidden layer, reference biomedkg/model/gcl.py:86-90 through autograd).  Workspace as bkg_colsum_f32. */
int bkg_outer_gate_colsum_bf16(const float* g, const float* w, const void* gate /* bf16 [n, d] */, float gate_scale,
                               void* dst /* bf16 [n, d] */, int64_t n, int32_t d, float* workspace, float* out, void* stream);
/* ---- a Linear whose outputs are summed over the features (GGD's head, biomedkg/model/gcl.py:86-90:
 *      `mlp[-1](h).sum(1)` = h . (sum_j W[j, :]) + sum_j b_j): out[i] = sum_c x[i, c] * w[c]; its backward pieces
 *      d h = g (x) w (bkg_scaled_rows, fp32 or bf16 table out) and d w = sum_i g[i] * x[i, :] (column sums with row
 *      weights; workspace as for bkg_colsum_f32). */
int bkg_rowdot_f32(const void* x, int32_t x_is_bf16, int64_t ldx, int64_t n, int32_t d, const float* w, float* out,
                   void* stream);
int bkg_scaled_rows(const float* g, const float* w, int64_t n, int32_t d, void* out, int32_t out_is_bf16, int64_t ldo,
                    void* stream);
int bkg_colsum_weighted_f32(const void* x, int32_t x_is_bf16, int64_t ldx, int64_t n, int32_t d, const float* row_w,
                            float* workspace, float* out, void* stream);
/* out = dout * scale where gate > 0, 0 elsewhere (the backward of the encoder's dropout(relu(.)), read off the saved
 * output: biomedkg/model/encoder.py:50-59); contiguous fp32 tables */
int bkg_gate_f32(const float* dout, const float* gate, float scale, int64_t numel, float* out, void* stream);

/* ---- negative sampling of the KGE step (PyG 2.5.3 negative_sampling + the permuted relation ids of
 *      biomedkg/kge_module.py:86-102).  The population is PyG's: the N*(N-1) off-diagonal cells, cell (r, c) coded
 *      r*(N-1) + c - (r < c) (edge_index_to_vector, non-bipartite directed form); self-loop positives are not part of it
 *      and no negative is a self pair.  bkg_hashset_insert_offdiag adds the codes of the (head, tail) pairs with
 *      head != tail to an open-addressing table of uint64 (empty = ~0, capacity a power of two >= 2x the pairs).
 *      bkg_negative_sample: candidate c is pi(c), pi a bijection of [0, N*(N-1)) keyed by seed (candidates are distinct by
 *      construction = random.sample); candidates [first, first + n) that are not in pos_table are written, in candidate
 *      order, as (head, tail) pairs (vector_to_edge_index: tail += (head <= tail)) at out_offset + rank while < max_out;
 *      *count (device) = survivors of this call; the slots behind the survivors (as far as n_candidates reaches, below
 *      max_out) are set to (0, 0).  Bitwise repeatable, no atomics.
 *      keys_tmp / flag_tmp / offs_tmp: [n_candidates] scratch.  bkg_permute_gather_i64: out[i] = src[pi(i)], pi a keyed
 *      bijection of [0, n) (x[randperm(n)] without the sort). */
int bkg_hashset_insert_offdiag(const int64_t* head, const int64_t* tail, int64_t n_nodes, int64_t n, uint64_t* table,
                               int64_t capacity, void* stream);
int64_t bkg_negative_sample_workspace_bytes(int64_t n_candidates);
int bkg_negative_sample(int64_t n_nodes, int64_t first_candidate, int64_t n_candidates, uint64_t seed,
                        const uint64_t* pos_table, int64_t pos_capacity, uint64_t* keys_tmp, int32_t* flag_tmp,
                        int32_t* offs_tmp, int64_t* out_head, int64_t* out_tail, int64_t out_offset, int64_t max_out,
                        int32_t* count, void* workspace, int64_t workspace_bytes, void* stream);
/* Block-structured negative sampling for node-partitioned training (biomedkg_amd/partition.py; replaces the per-rank
 * call of PyG negative_sampling, reference biomedkg/kge_module.py:93, when a graph's nodes are partitioned): the cell grid
 * is cut into (head owner, tail owner) blocks, each with its own keyed permutation and its share of the K negatives, so the
 * two owners of a block draw identical triplets and nothing crosses the wire.  blocks: device array of n_blocks
 * descriptors of bkg_negative_blocks_struct_bytes() bytes (13 int64: row_lo, n_rows, col_lo, n_cols, population, cand0,
 * n_cand, out0, k, gpos0, id, row_shift, col_shift -- the shifts are added to head / tail on the way out: node id -> slot
 * of the rank's gathered table).  out_type (nullable) [p] = pos_type[sigma(gpos0 + rank) % n_pos]: edge_type.repeat(k)
 * [randperm] (kge_module.py:97-100) evaluated position by position; callers that allot the ids to the blocks themselves
 * (sampling.BlockNegativeSampler: relation-grouped inside every block) pass NULL. */
int64_t bkg_negative_blocks_struct_bytes(void);
int bkg_negative_sample_blocks(const void* blocks, int32_t n_blocks, int64_t n_cand_total, int64_t n_nodes, uint64_t seed,
                               int64_t k_global, const uint64_t* pos_table, int64_t pos_capacity, const int64_t* pos_type,
                               int64_t n_pos, uint64_t* keys_tmp, int32_t* flag_tmp, int32_t* offs_tmp, int64_t* out_head,
                               int64_t* out_tail, int64_t* out_type, int32_t* count, void* workspace,
                               int64_t workspace_bytes, void* stream);
int bkg_permute_gather_i64(const int64_t* src, int64_t n, uint64_t seed, int64_t* out, void* stream);
/* out[i] = src[pi(i) % n_src], i < n_src * repeat, pi a keyed bijection of that range: edge_type.repeat(k)[randperm(k E)] of
 * reference biomedkg/kge_module.py:97-100 in one launch, without the repeated copy. */
int bkg_permute_gather_repeat_i64(const int64_t* src, int64_t n_src, int64_t repeat, uint64_t seed, int64_t* out, void* stream);
/* ... from a one-byte image of the ids (values < 256; bkg_narrow_i64_u8 makes it once per id tensor): the same outputs for
 * the same seed, the random reads stay in the L2. */
int bkg_narrow_i64_u8(const int64_t* src, int64_t n, uint8_t* out, void* stream);
int bkg_permute_gather_repeat_u8(const uint8_t* src, int64_t n_src, int64_t repeat, uint64_t seed, int64_t* out, void* stream);
/* torch.randperm over the REAL rows of a padded mini-batch (GGD's corruption, reference biomedkg/model/gcl.py:70-71, inside
 * a captured step): out[i] = pi(i) for i < n_dev[0] (device scalar), pi a bijection of [0, n_dev[0]) keyed by seed and the
 * device replay counter `epoch` (NULL: the library-wide counter of bkg_set_rng_epoch, if one is set); out[i] = i for
 * n_dev[0] <= i < cap.  With n_dev[0] = cap: x[torch.randperm(n)] of GGD / DGI's corruption without the sort behind randperm. */
int bkg_randperm_dev(const int64_t* n_dev, int64_t cap, uint64_t seed, const uint32_t* epoch, int64_t* out, void* stream);
/* The sampler of biomedkg/kge_module.py:86-102 for one PADDED mini-batch with every extent in device memory (sizes as
 * written by bkg_saint_batch; head / tail / type hold `rows` entries, the first sizes[1] are edges): hash set of the
 * positives, ONE round of int(1.1 k / prob) candidates (k = neg_ratio * sizes[1]; the caller has checked prob > 0.9, so
 * PyG's later rounds never run), the first k survivors in candidate order -> out_head / out_tail[out_cap], *count =
 * how many (slots past it hold pairs spread over the batch's nodes); out_type (nullable) = type[pi(i) % E] for i < count (edge_type.repeat(neg_ratio)
 * [randperm]) and type_pad beyond.  epoch (nullable): device replay counter mixed into the keys.  Scratch: table
 * uint64[capacity >= 2 rows, power of two], keys / flag / offs of bkg_negative_sample_dev_candidates(out_cap) entries,
 * workspace of bkg_negative_sample_workspace_bytes(that many). */
int64_t bkg_negative_sample_dev_candidates(int64_t max_negatives);
int bkg_negative_sample_dev(const int64_t* sizes, int64_t neg_ratio, const int64_t* head, const int64_t* tail,
                            const int64_t* type, int64_t rows, uint64_t seed, const uint32_t* epoch, uint64_t* table,
                            int64_t capacity, uint64_t* keys_tmp, int32_t* flag_tmp, int32_t* offs_tmp, int64_t out_cap,
                            int64_t type_pad, int64_t* out_head, int64_t* out_tail, int64_t* out_type /* nullable */,
                            int64_t* count, void* workspace, int64_t workspace_bytes, void* stream);
/* The loss-side view of a padded batch (sizes / count as above): rows past sizes[1] get the relation id pad_rel (one
 * past the real ones: never scored) and endpoints spread over [0, node_cap); inv_total[0] = 1 / (sizes[1] + count[0]). */
int bkg_padded_loss_inputs(const int64_t* sizes, const int64_t* count, const int64_t* src, const int64_t* dst,
                           const int64_t* type, int64_t rows, int64_t node_cap, int64_t pad_rel, int64_t* out_src,
                           int64_t* out_dst, int64_t* out_type, float* inv_total, void* stream);
/* The padded view of a device NeighborLoader batch (the reference's GCL loader, biomedkg/data_module.py:93-99, as a
 * sync-free captured step): sizes[1] / sizes[2] = the batch's node / edge counts (device); out_n_id[n_cap] = n_id or
 * pad_row past the count, out_src / out_dst[e_b] = the edges or self loops of last_node past the count, real[n_cap] = 1
 * for the batch's own nodes. */
int bkg_padded_gcl_batch(const int64_t* sizes, const int64_t* n_id, const int64_t* src, const int64_t* dst, int64_t n_cap,
                         int64_t e_b, int64_t pad_row, int64_t last_node, int64_t* out_n_id, int64_t* out_src,
                         int64_t* out_dst, uint8_t* real, void* stream);

/* ---- modality fusion epilogues (biomedkg/utils/fusion.py:17-31 AttentionFusion, :49-90 ReDAF) */
int bkg_attn_fuse_f32(const float* qkv, int64_t ld, int64_t n, int32_t M, int32_t D, float* out, int64_t ldo,
                      void* stream);
/* AttentionFusion in projected form (bf16 mode; reference biomedkg/utils/fusion.py:17-31 evaluated through two exact
 * identities of its arithmetic -- q_a . k_b = x_a^T (Wq^T Wk) x_b + (Wq^T bk) . x_a + (Wk^T bq) . x_b + bq . bk, and
 * sum_b c_b (Wv x_b + bv) = Wv (sum_b c_b x_b) + bv since the attention weights of a row sum to one): t = X (Wq^T Wk) and
 * x as [2 n, D] bf16 tables -> the four scores per node, the 2 x 2 softmax, the modality-mean coefficients c_b and the
 * combined row out[i] = c_0 x_{i,0} + c_1 x_{i,1} (bf16), over which the V projection then runs. */
int bkg_attn_scores_combine_bf16(const void* t, int64_t ldt, const void* x, int64_t ldx, int64_t n, int32_t D,
                                 const float* u, const float* w, float c0, void* out, int64_t ldo, void* stream);
/* ReDAF dropout: `keep` (explicit mask, [n*M, D] bytes) or, when NULL and drop_p > 0, an in-kernel counter-based keep hash
 * (three rounds of a 32-bit mixer per four elements, 16-bit thresholds: drop probability ceil(drop_p * 2^16) / 2^16) keyed by
 * (seed, stream_id) -- the caller advances stream_id on every call, as nn.Dropout draws a fresh mask per forward. */
int bkg_redaf_combine_f32(const float* t, int64_t ld, int64_t n, int32_t M, int32_t D, const float* scale,
                          const uint8_t* keep, float drop_p, uint64_t seed, uint32_t stream_id,
                          int64_t row0 /* index of t's first row in the whole [n_total * M, D] input: the keep hash is
                                          keyed by the global element, so a chunked call draws the same mask */,
                          float* out, int64_t ldo, void* stream);
/* ReDAF.forward (biomedkg/utils/fusion.py:49-90, two modalities) as ONE launch: the scale / dropout / ReLU / modality mean run
 * in the epilogue of the product x W^T (the two token rows of a node are adjacent lanes of an accumulator tile), so the
 * [2n, D] transform is never written.  A [rows, K] fp32 or bf16 (rows = 2 n, node-major), Wt the [N, K] bf16 weight image of
 * bkg_weight_to_bf16, bias [N], scale [2, N]; dropout as in bkg_redaf_combine_f32 (keep mask [rows, N], or drop_p + the keep
 * hash of (seed, stream_id) indexed by row * N + column: the same masks, bit-equal results); out fp32 [n, N]. */
int bkg_redaf_fused_xw(int64_t rows, int64_t N, int64_t K, const void* A, int32_t a_is_bf16, int64_t lda, const void* Wt,
                       int64_t ldw, const float* bias, const float* scale, const uint8_t* keep, float drop_p, uint64_t seed,
                       uint32_t stream_id, float* out, int64_t ldo, void* stream);

/* ---- the step tail: global-norm gradient clipping + Adam over a list of parameter tensors, two launches (csrc/optim.hip).
 *      Replaces torch.nn.utils.clip_grad_norm_(params, max_norm) -- Lightning's gradient_clip_val = 1.0 of
 *      train_gcl.py:63-68 / train_kge.py:44-50 -- followed by torch.optim.Adam.step() (gcl_module.py:78-87,
 *      kge_module.py:208-217; no amsgrad, no weight decay): c = min(1, max_norm / (||g|| + 1e-6)), g' = c g,
 *      m = b1 m + (1 - b1) g', v = b2 v + (1 - b2) g'^2, p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps).
 *      params / grads / exp_avg / exp_avg_sq: HOST arrays of `count` device pointers (<= bkg_clip_adam_max_tensors());
 *      max_norm <= 0: no clipping.  lr_dev / step_dev (nullable device scalars, captured steps): the learning rate is
 *      read from lr_dev, the step count t = step_dev[0] + 1 and step_dev is incremented; else the host values lr / step
 *      (t = step, already incremented by the caller).  partial: bkg_clip_adam_chunks(numel, count) floats of scratch;
 *      norm_out (nullable): the total gradient norm before clipping.  The gradients themselves are left unscaled. */
int64_t bkg_clip_adam_max_tensors(void);
int64_t bkg_clip_adam_chunks(const int64_t* numel, int32_t count);
int bkg_clip_adam_f32(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const int64_t* numel, float max_norm, float lr, const float* lr_dev,
                      float beta1, float beta2, float eps, float step, float* step_dev, float* partial, int64_t n_partial,
                      float* norm_out, void* stream);
/* The same step, which also refreshes the bf16 images the bf16-mode products read of the weights it moves (the two
 * orientations bkg_weight_to_bf16 writes): w16[i] / w16t[i] (host arrays of device pointers, an entry nullable) = the
 * [batch][rows[i]][cols[i]] image of tensor i as stored and its transpose.  A weight only changes here, so the conversion
 * launch per weight at the top of the next step goes away. */
int bkg_clip_adam_images_f32(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numel, void* const* w16, void* const* w16t,
                             const int32_t* rows, const int32_t* cols, float max_norm, float lr, const float* lr_dev,
                             float beta1, float beta2, float eps, float step, float* step_dev, float* partial,
                             int64_t n_partial, float* norm_out, void* stream);

/* GGD's loss (biomedkg/gcl_module.py:229-234): binary_cross_entropy_with_logits(cat(pos, neg), cat(ones, zeros)), mean over
 * the 2 n logits, forward (partial sums in a fixed order + a one-block finish) and gradient, one launch each; pos / neg /
 * dpos / dneg: fp32 vectors with an element stride (the two columns of an [n, 2] score table).  n_dev (nullable DEVICE
 * scalar): only rows below n_dev[0] count and the mean divides by 2 n_dev[0] (padded mini-batches); gradients of the other
 * rows are 0.  partial: bkg_ggd_bce_partials() floats of scratch; gout: the upstream scalar on the device. */
int64_t bkg_ggd_bce_partials(void);
int bkg_ggd_bce_f32(const float* pos, int64_t stride_pos, const float* neg, int64_t stride_neg, int64_t n,
                    const int64_t* n_dev, float* partial, float* out, void* stream);
int bkg_ggd_bce_bwd_f32(const float* pos, int64_t stride_pos, const float* neg, int64_t stride_neg, int64_t n,
                        const int64_t* n_dev, const float* gout, float* dpos, int64_t stride_dpos, float* dneg,
                        int64_t stride_dneg, void* stream);

#ifdef __cplusplus
}
#endif
#endif
