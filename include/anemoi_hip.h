/*
 * anemoi_hip.h — C ABI of libanemoi_hip.so: the MI355X (gfx950) kernels behind the
 * anemoi-models encoder-processor-decoder hot path.
 *
 * The reference (ecmwf/anemoi-core) is pure Python: it has no FFI of its own.  Its kernel boundary
 * is the registered PyTorch op  anemoi::graph_transformer_attention(q,k,v,e,row,colptr,...)
 * (models/src/anemoi/models/triton/gt.py:390-428) plus the torch.nn layers chosen through
 * ``layer_kernels`` (models/src/anemoi/models/layers/utils.py:87-142).  Each entry point below names
 * the reference interface it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *  - row-major matrices with an explicit leading dimension ``ld*`` counted in ELEMENTS;
 *  - ``dtype`` selects the storage type of every floating tensor of the call (accumulation is
 *    always fp32); indices are int32 (the reference uses int64 at its op boundary, gt.py:413-414,
 *    the host wrapper converts once because the graph is static; N, M < 2^31);
 *  - ``stream`` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *  - return value 0 = success, otherwise a negative ANEMOI_E_* code; anemoi_hip_last_error()
 *    returns a thread-local message.  Nothing is allocated or freed by the library, except the
 *    peer-exchange arenas of anemoi_peer_alloc / anemoi_peer_free (memory other PROCESSES map
 *    through hipIpc cannot come from the caller's caching allocator).
 */
#ifndef ANEMOI_HIP_H
#define ANEMOI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANEMOI_HIP_ABI_VERSION 13

typedef enum { ANEMOI_F32 = 0, ANEMOI_BF16 = 1, ANEMOI_F16 = 2 } anemoi_dtype_t;
typedef enum { ANEMOI_ACT_NONE = 0, ANEMOI_ACT_GELU = 1 } anemoi_act_t;

#define ANEMOI_OK 0
#define ANEMOI_E_INVALID (-1)   /* bad argument (shape, dtype, alignment, null pointer) */
#define ANEMOI_E_UNSUPPORTED (-2)
#define ANEMOI_E_LAUNCH (-3)    /* hipGetLastError() != hipSuccess after the launch */

int anemoi_hip_abi_version(void);
const char* anemoi_hip_last_error(void);

/* Fused graph-transformer edge attention, forward.
 * Replaces: anemoi::graph_transformer_attention / _gt_fwd (triton/gt.py:81-179, 390-428) and
 * GraphTransformerConv (layers/conv.py:84-147).
 *   for every destination node d, head h, over the in-edges e=(s->d) in CSC order:
 *     score = <q[d,h,:], k[s,h,:] + E[e,h,:]> / sqrt(C);  a = softmax_e(score)
 *     out[d,h,:] = sum_e a * (v[s,h,:] + E[e,h,:])  (+ addend[d,h,:] if addend != NULL)
 *     lse[d,h]   = max + log(sum exp)               (if lse != NULL; 0 for empty d)
 * q,out,addend: [n_dst, H*C]; k,v: [n_src, H*C]; e: [M, H*C] in CSC (dst-sorted) order or NULL
 * (no edge term); row[M] = source id per edge; colptr[n_dst+1].  Zero-in-degree d -> out = addend or 0. */
int anemoi_gt_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                            const void* e, int64_t lde, const int32_t* row, const int32_t* colptr,
                            const void* addend, int64_t ldadd, void* out, int64_t ldo, float* lse,
                            int32_t n_dst, int32_t n_src, int32_t H, int32_t C, anemoi_dtype_t dtype, void* stream);

/* Backward of the op above (materialised E).
 * Replaces: anemoi::graph_transformer_attention_backward = _gt_bwd_dst_pass + _gt_bwd_src_pass (triton/gt.py:182-376,
 * 447-492) and its autograd registration (triton/gt.py:526-556).
 *   out, lse: the forward's results; d_out: gradient of out [n_dst, H*C];
 *   row/colptr: CSC as in the forward; rowptr[n_src+1], edge_ids[M] (CSC edge ids grouped by source), edge_dst[M]
 *   (destination of every CSC edge): the reverse CSR of the same graph (triton/utils.py:25-70);
 *   dq [n_dst, H*C], dk, dv [n_src, H*C], de [M, H*C] (CSC order): outputs, every row written (zeros where a node has no
 *   edges);  p_ws, ds_ws: fp32 workspace [M, H] each (per-edge softmax weight and score gradient, handed from the
 *   destination pass to the source pass).  Deterministic: no atomics. */
int anemoi_gt_attention_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                            const void* e, int64_t lde, const void* out, int64_t ldo, const float* lse,
                            const void* d_out, int64_t lddo, const int32_t* row, const int32_t* colptr,
                            const int32_t* rowptr, const int32_t* edge_ids, const int32_t* edge_dst,
                            void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* de, int64_t ldde,
                            float* p_ws, float* ds_ws, int32_t n_dst, int32_t n_src, int32_t n_edges, int32_t H, int32_t C,
                            anemoi_dtype_t dtype, void* stream);

/* The two ops above with DROPOUT on the softmax weights (training mode of the reference's conv: layers/conv.py:145,
 * `alpha = dropout(alpha, p, training)` after the segment softmax): out[d] = sum_e c_e alpha_e (v_s + E_e) with c_e = 0 (dropped,
 * probability drop_p) or 1 / (1 - drop_p); the softmax itself (and lse) sums every edge.  The keep decision of (CSC edge, head) is
 * a pure function of (drop_seed, edge, head) - a counter-based generator, csrc/common.h: attn_dropout_scale - so the backward takes
 * the SAME (drop_p, drop_seed) and re-derives the mask instead of reading a stored one.  drop_p = 0 is the plain op, bit for bit
 * (anemoi_gt_attention_fwd / _bwd are these entry points with drop_p = 0).  anemoi_attention_dropout_mask writes c_e as fp32
 * [n_edges, H]: what a caller needs to restate the op with an explicit mask (tests/test_attention_dropout_gpu.py). */
int anemoi_gt_attention_dropout_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                    const void* e, int64_t lde, const int32_t* row, const int32_t* colptr,
                                    const void* addend, int64_t ldadd, void* out, int64_t ldo, float* lse,
                                    int32_t n_dst, int32_t n_src, int32_t H, int32_t C, float drop_p, uint64_t drop_seed,
                                    anemoi_dtype_t dtype, void* stream);
int anemoi_gt_attention_dropout_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                    const void* e, int64_t lde, const void* out, int64_t ldo, const float* lse,
                                    const void* d_out, int64_t lddo, const int32_t* row, const int32_t* colptr,
                                    const int32_t* rowptr, const int32_t* edge_ids, const int32_t* edge_dst,
                                    void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* de, int64_t ldde,
                                    float* p_ws, float* ds_ws, int32_t n_dst, int32_t n_src, int32_t n_edges, int32_t H, int32_t C,
                                    float drop_p, uint64_t drop_seed, anemoi_dtype_t dtype, void* stream);
int anemoi_attention_dropout_mask(float* out, int32_t n_edges, int32_t H, float drop_p, uint64_t drop_seed, void* stream);

/* Same op with ``lin_edge`` fused: E[e] = edge_attr[e] @ w_edge^T + b_edge is never materialised.
 * Replaces: lin_edge(...) + the op above (layers/block.py:623-635 + triton/gt.py:81-179).
 * edge_feat: fp32 [M, fe_pad] with fe_pad = 4*ceil((Fe+1)/4): columns [0,Fe) = edge_attr, column Fe = 1.0
 * (carries the bias), the rest 0 (anemoi_pack_edge_features);  w_packed: fp32 [H*C, fe_pad] = [w_edge | b_edge | 0]
 * (anemoi_pack_edge_weights).  Both are built once per static graph / parameter version.
 * dst_order (NULL = 0, 1, 2, ...): a permutation of the destinations giving the order in which the kernel WORKS on them (the
 * reference's kernel takes program id = destination, triton/gt.py:100); every XCD processes a contiguous eighth of it, so an
 * order that keeps mesh neighbours together keeps the gathered K|V rows in that XCD's L2.  The result does not depend on it. */
int anemoi_gt_attention_fused_edge_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                       const float* edge_feat, int32_t fe_pad, const float* w_packed,
                                       const int32_t* row, const int32_t* colptr, const int32_t* dst_order, const void* addend, int64_t ldadd,
                                       void* out, int64_t ldo, float* lse, int32_t n_dst, int32_t n_src, int32_t H,
                                       int32_t C, anemoi_dtype_t dtype, void* stream);

/* Backward of the fused-edge op (scope row f1): dq / dk / dv as anemoi_gt_attention_bwd, plus d_w_packed fp32 [H*C, fe_pad]
 * (= [d w_edge | d b_edge | .]) and, when d_edge_feat != NULL, d_edge_feat fp32 [M, fe_pad] (gradient of the edge attributes,
 * CSC order); E and dE are never materialised.  `out` / `lse`: the forward's results; when the forward ran with an addend pass the
 * same rows as `addend` (o = out - addend is formed in the kernel); `d_addend` (nullable) receives the addend's gradient (= d_out)
 * from the same pass.  Workspaces (fp32):
 * p_ws, ds_ws [M, H]; sf_ws, qg_ws [n_dst, H, 2, fe_pad] (qg_ws only with d_edge_feat); part_ws
 * [anemoi_gt_attention_fused_edge_bwd_partial_floats].  Shapes outside the fused fast path return ANEMOI_E_UNSUPPORTED (train
 * through anemoi_gt_attention_bwd with a materialised E then).  Replaces the autograd of lin_edge + the op
 * (layers/block.py:623-635, triton/gt.py:182-376). */
int64_t anemoi_gt_attention_fused_edge_bwd_partial_floats(int32_t H, int32_t C, int32_t fe_pad);
int anemoi_gt_attention_fused_edge_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                       const float* edge_feat, int32_t fe_pad, const float* w_packed, const void* out, int64_t ldo,
                                       const float* lse, const void* d_out, int64_t lddo, const int32_t* row, const int32_t* colptr,
                                       const int32_t* rowptr, const int32_t* edge_ids, const int32_t* edge_dst, void* dq,
                                       int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, float* d_w_packed,
                                       float* d_edge_feat, float* p_ws, float* ds_ws, float* sf_ws, float* qg_ws, float* part_ws,
                                       const void* addend, int64_t ldadd, void* d_addend, int64_t lddadd, int32_t n_dst,
                                       int32_t n_src, int32_t n_edges, int32_t H, int32_t C, anemoi_dtype_t dtype, void* stream);

/* Pack lin_edge's parameters for the fused op: out fp32 [D, fe_pad] = [w_edge [D,Fe] | b_edge [D] (or 0) | 0...]. */
int anemoi_pack_edge_weights(const void* w_edge, const void* b_edge, float* out, int32_t D, int32_t fe, int32_t fe_pad,
                             anemoi_dtype_t dtype, void* stream);

/* Pack edge attributes for the fused op: out fp32 [M, fe_pad] = [edge_attr | 1 | 0...]. */
int anemoi_pack_edge_features(const void* edge_attr, int64_t ld, float* out, int32_t M, int32_t fe, int32_t fe_pad,
                              anemoi_dtype_t dtype, void* stream);

/* LayerNorm over the last dimension (eps inside the sqrt, affine; beta may be NULL), optional fused residual:
 *   y = LayerNorm(x) * gamma + beta (+ residual).
 * Replaces: torch.nn.LayerNorm / AutocastLayerNorm via layer_kernels.LayerNorm
 * (layers/utils.py:107-121, layers/normalization.py:19-31) and the "MLP(...LayerNorm) + x" tail of the GraphConv
 * blocks (layers/block.py:391, 469).  x, y, residual: [n_rows, D]. */
int anemoi_layernorm_fwd(const void* x, int64_t ldx, const void* gamma, const void* beta, const void* residual,
                         int64_t ldr, void* y, int64_t ldy, int32_t n_rows, int32_t D, float eps, anemoi_dtype_t dtype,
                         void* stream);

/* ConditionalLayerNorm forward (layers/normalization.py:34-94): y = LN(x) * (scale[row] + 1) + shift[row]; scale / shift are
 * per-row modulation tensors [n_rows, D] (leading dimension 0 = one row for all), e.g. the two halves of one fused
 * Linear(cond).  No affine parameters of its own. */
int anemoi_cond_layernorm_fwd(const void* x, int64_t ldx, const void* scale, int64_t lds, const void* shift, int64_t ldsh,
                              void* y, int64_t ldy, int32_t n_rows, int32_t D, float eps, anemoi_dtype_t dtype, void* stream);

/* Backward of the above: d_x [n_rows, D] and d_scale [n_rows, D] = d_y * x^ (per row; d_shift = d_y needs no kernel); the
 * gradients of the conditioning's two Linear maps follow from d_scale / d_shift through anemoi_linear_fwd. */
int anemoi_cond_layernorm_bwd(const void* x, int64_t ldx, const void* scale, int64_t lds, const void* d_y, int64_t lddy,
                              void* d_x, int64_t lddx, void* d_scale, int64_t ldds, int32_t n_rows, int32_t D, float eps,
                              anemoi_dtype_t dtype, void* stream);

/* Output boundings at the model edge, in place and in configuration order (layers/bounding.py:81-307;
 * models/encoder_processor_decoder.py:160-162).  ops: int32 [n_ops][4] = (kind, column, total column, 0); params: fp32
 * [n_ops][2].  kind 1 ReluBounding, 2 LeakyReluBounding, 3 / 4 Normalized(Leaky)ReluBounding (params[0] = the normalised
 * minimum), 5 / 6 (Leaky)HardtanhBounding (min, max), 7 / 8 (Leaky)FractionBounding (min, max; times column ``total``),
 * 9 de-normalisation (x - params[0]) / params[1] = InputNormalizer.inverse_transform of that column
 * (preprocessing/normalizer.py:217-252), appended after the boundings when the output normaliser is fused. */
int anemoi_bound_columns(void* x, int64_t ldx, int32_t n_rows, int32_t n_cols, const int32_t* ops, const float* params,
                         int32_t n_ops, anemoi_dtype_t dtype, void* stream);

/* Input assembly at the model edge for batch = ensemble = 1 (models/encoder_processor_decoder.py:98-143): out[n] = [x[0, n, :] | ... |
 * x[T-1, n, :] | attrs[n, :] | zeros up to W], x time slices ld_t elements apart, rows ldx apart.  Replaces the permute-copy, the cat
 * with the node attributes and the zero padding of the embedding GEMMs' K dimension. */
int anemoi_assemble_input(const void* x, int64_t ld_t, int64_t ldx, int32_t T_steps, int32_t V, const void* attrs, int64_t lda, int32_t A,
                          void* out, int64_t ldo, int32_t W, int32_t n_rows, anemoi_dtype_t dtype, void* stream);

/* Output assembly at the model edge for batch = ensemble = output steps = 1 (models/encoder_processor_decoder.py:145-163):
 * out[n, v] = x_out[n, v] + x_skip[n, col_map[v]] where col_map[v] >= 0 (the SkipConnection residual on the prognostic
 * columns), else x_out[n, v].  Replaces clone + index_select + index_add_. */
int anemoi_assemble_output(const void* x_out, int64_t ldx, const void* x_skip, int64_t lds, const int32_t* col_map, void* out,
                           int64_t ldo, int32_t n_rows, int32_t n_cols, anemoi_dtype_t dtype, void* stream);

/* LayerNorm backward.  Replaces: autograd of layer_kernels.LayerNorm (layers/utils.py:107-121).
 *   d_x [n_rows, D] (same dtype), d_gamma / d_beta fp32 [D] (either may be NULL; both NULL: no column sums);
 *   workspace: anemoi_reduce_workspace_bytes(D) bytes of fp32 scratch (per-wave partial column sums, added in a fixed
 *   order by a second kernel: deterministic, no atomics).  Statistics are recomputed from x (nothing saved by the forward). */
int64_t anemoi_reduce_workspace_bytes(int32_t D);
int anemoi_layernorm_bwd(const void* x, int64_t ldx, const void* gamma, const void* d_y, int64_t lddy, void* d_x,
                         int64_t lddx, float* d_gamma, float* d_beta, float* workspace, int32_t n_rows, int32_t D,
                         float eps, anemoi_dtype_t dtype, void* stream);

/* out[c] = sum_r x[r, c] in fp32 (bias gradient of torch.nn.Linear); same workspace and determinism as above. */
int anemoi_colsum(const void* x, int64_t ldx, float* out, float* workspace, int32_t n_rows, int32_t D,
                  anemoi_dtype_t dtype, void* stream);

/* d_pre = d_y * gelu'(pre), gelu'(x) = Phi(x) + x phi(x) (exact erf form; autograd of torch.nn.GELU, layers/utils.py:111). */
int anemoi_gelu_bwd(const void* pre, int64_t ldp, const void* d_y, int64_t lddy, void* d_pre, int64_t lddp,
                    int32_t n_rows, int32_t D, anemoi_dtype_t dtype, void* stream);

/* y = gelu(x), exact erf form: torch.nn.GELU as selected by `layer_kernels.Activation` (layers/utils.py:111) when it runs
 * as a stand-alone module, e.g. inside the reference's own MLP (layers/mlp.py:158-169).  The fused blocks of this package
 * apply GELU in the epilogue of the preceding GEMM instead. */
int anemoi_gelu_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t n_rows, int32_t D, anemoi_dtype_t dtype, void* stream);

/* LayerNorm folded into the GEMMs around it (inference).  For y = LN(x; gamma, beta) W^T + b:
 *     y = rstd (x (W diag gamma)^T - mean c) + d,   c = row sums of W diag(gamma),  d = W beta + b
 * so the normalisation needs only the per-row mean / rstd of x, and those come for free from the GEMM that PRODUCED x:
 *  - anemoi_linear_stats_fwd: y = x W^T + bias + residual (as anemoi_linear_fwd; residual may be NULL: the embedding in
 *    front of a mapper's LayerNorm, layers/mapper.py:556-570) and stats_out[n_rows][O/64][2] (fp32) =
 *    (sum, sum of squares) of every 64-column strip of the stored (rounded) output row.  Plain stores, one per (row, strip):
 *    no atomics, no zeroing, deterministic.  O and K multiples of 64.
 *  - anemoi_linear_lnfold_fwd: y = act(LN(x) W^T + b) from raw x [n_rows, K], w_scaled = W diag(gamma) [O, K], fp32 c, d [O]
 *    and stats_in = the producer's statistics of x (strips * 64 == K).  Partials are added in a fixed order.
 *  The pair is a two-kernel protocol with ONE tail rule on both sides: when n_rows exceeds a multiple of 320 by at most 32
 *  (the "+ 2" of an icosphere's 10 * 4^r + 2 nodes) the producer may compute those trailing rows outside its tiles and
 *  leaves their stats_out entries UNWRITTEN; the consumer never reads them - it takes the statistics of exactly those rows
 *  (n_rows % 320 <= 32) from the rows themselves.  Every other row has its strip sums written (ragged last tiles included).
 * Replaces the two LayerNorm launches of a GraphTransformerProcessorBlock (layer_norm_attention / layer_norm_mlp_dst,
 * layers/block.py:1237, 1271) and layer_norm_attention_src / _dest of a GraphTransformerMapperBlock (layers/block.py:979-984)
 * in the unsharded inference path.  Returns ANEMOI_E_UNSUPPORTED for shapes / alignments the
 * ring kernels do not take (the caller falls back to LayerNorm + anemoi_linear_fwd). */
int anemoi_linear_stats_fwd(const void* x, int64_t ldx, int32_t K, const void* w, int64_t ldw, const void* bias,
                            const void* residual, int64_t ldr, void* y, int64_t ldy, float* stats_out, int32_t n_rows,
                            int32_t O, anemoi_dtype_t dtype, void* stream);
int anemoi_linear_lnfold_fwd(const void* x, int64_t ldx, int32_t K, const void* w_scaled, int64_t ldw, const float* ln_c,
                             const float* ln_d, const float* stats_in, int32_t strips, float eps, anemoi_act_t act, void* y,
                             int64_t ldy, int32_t n_rows, int32_t O, anemoi_dtype_t dtype, void* stream);

/* y[n_rows, O] (fp32, ZEROED BY THE CALLER) += x[n_rows, K] @ w[O, K]^T with the reduction split into ``splits`` chunks that
 * run on different CUs and accumulate with fp32 atomics: the weight-gradient GEMM dW = dZ^T X of torch.nn.Linear's autograd
 * (small output, reduction over 10^4..10^5 rows).  16-bit operands, K a multiple of 64 * splits. */
int anemoi_linear_splitk_f32(const void* x, int64_t ldx, const void* w, int64_t ldw, float* y, int64_t ldy, int32_t n_rows,
                             int32_t O, int32_t K, int32_t splits, anemoi_dtype_t dtype, void* stream);

/* Weight gradient of torch.nn.Linear without HBM transposes: dw[o, i] = sum_r dz[r, o] * x[r, i] (16-bit operands, fp32
 * accumulation; O, I, lddz, ldx multiples of 8).  The row-major tiles are read through the LDS transpose read; the reduction is
 * split over workgroups into fp32 partial tiles in `workspace` (anemoi_linear_wgrad_workspace_bytes bytes) summed in fixed order:
 * deterministic, no atomics.  `db` (nullable, [O]) receives the bias gradient = column sums of dz, accumulated by the same kernel.
 * Autograd of block.py:623-635 / mlp.py:158-169 (scope row f1). */
int64_t anemoi_linear_wgrad_workspace_bytes(int32_t n_rows, int32_t O, int32_t I);
int anemoi_linear_wgrad(const void* dz, int64_t lddz, const void* x, int64_t ldx, void* dw, int64_t lddw, void* db, void* workspace,
                        int32_t n_rows, int32_t O, int32_t I, anemoi_dtype_t dtype, void* stream);

/* Linear layer with fused epilogue.  Replaces torch.nn.Linear (+ GELU + residual add) as used by
 * get_qkve / projection / MLP (layers/block.py:623-635,1268-1271; layers/mlp.py:158-169).
 *   y[n, o] = act( sum_k A[n,k] * w[o,k] + bias[o] + g1[idx1[n], o] + g2[idx2[n], o] ) + residual[n, o]
 * A = [x | x2] concatenated along K (x: [n_rows,K1], x2: [n_rows,K2] or NULL with K2=0);  w: [O, K1+K2]
 * row-major (torch layout);  bias, g1/idx1, g2/idx2, residual optional (NULL).  The gather-add terms
 * serve GraphConv's first edge-MLP layer (layers/conv.py:73-76: cat[x_i, x_j, e] @ W^T is split into
 * two node-level products gathered per edge and one edge-level product). */
int anemoi_linear_fwd(const void* x, int64_t ldx, int32_t K1, const void* x2, int64_t ldx2, int32_t K2, const void* w,
                      int64_t ldw, const void* bias, const void* g1, int64_t ldg1, const int32_t* idx1, const void* g2,
                      int64_t ldg2, const int32_t* idx2, const void* residual, int64_t ldr, void* y, int64_t ldy,
                      int32_t n_rows, int32_t O, anemoi_act_t act, anemoi_dtype_t dtype, void* stream);

/* Same, and the pre-activation z (the argument of GELU) is stored to y_pre as well: training keeps it for GELU's backward
 * instead of recomputing it with a second GEMM (scope row f1).  act must be GELU and the shape one of the DMA-ring GEMM shapes
 * (16-bit, K multiple of 64); otherwise ANEMOI_E_UNSUPPORTED and nothing is launched. */
int anemoi_linear_fwd_pre(const void* x, int64_t ldx, int32_t K1, const void* x2, int64_t ldx2, int32_t K2, const void* w,
                          int64_t ldw, const void* bias, const void* g1, int64_t ldg1, const int32_t* idx1, const void* g2,
                          int64_t ldg2, const int32_t* idx2, const void* residual, int64_t ldr, void* y, int64_t ldy, void* y_pre,
                          int64_t ldy_pre, int32_t n_rows, int32_t O, anemoi_act_t act, anemoi_dtype_t dtype, void* stream);

/* GraphConv edge epilogue + aggregation, one pass over the dst-sorted edges (no atomics).
 * Replaces: MLP.layer_norm + "+ edge_attr" + scatter(sum) (layers/conv.py:73-81, layers/mlp.py:176-178).
 *   e_new[m,:] = LayerNorm(z[m,:]) + e_old[m,:];   agg[d,:] = sum_{m in in(d)} e_new[m,:]
 * z, e_old, e_new: [M, D]; agg: [n_dst, D]; gamma/beta: [D] (gamma NULL -> no LayerNorm). */
int anemoi_edge_ln_residual_segment_sum_fwd(const void* z, int64_t ldz, const void* e_old, int64_t lde,
                                            const void* gamma, const void* beta, float eps, const int32_t* colptr,
                                            void* e_new, int64_t ldn, void* agg, int64_t ldagg, int32_t n_dst,
                                            int32_t D, anemoi_dtype_t dtype, void* stream);

/* Row gather: out[i,:] = x[idx[i],:].  Packs the halo send buffer (distributed/primitives.py:455). */
int anemoi_gather_rows(const void* x, int64_t ldx, const int32_t* idx, void* out, int64_t ldo, int32_t n_out, int32_t D,
                       anemoi_dtype_t dtype, void* stream);

/* out[r] = sum over i in [ptr[r], ptr[r+1]) of x[ids ? ids[i] : i]  (fp32 accumulation; deterministic).
 * Adjoint of the row gathers: with ids = NULL and ptr = colptr it sums the contiguous in-edge rows of every destination,
 * with (ptr, ids) = the reverse CSR (rowptr, edge_ids) the out-edge rows of every source.  Replaces: autograd of the
 * x_i / x_j index_select in GraphConv (layers/conv.py:66-81) and of scatter(sum). */
int anemoi_segment_sum_rows(const void* x, int64_t ldx, const int32_t* ptr, const int32_t* ids, void* out, int64_t ldo,
                            int32_t n_out, int32_t D, anemoi_dtype_t dtype, void* stream);

/* out[i] = a[i] + b[idx[i]]: adjoint of scatter(sum) + the carried edge gradient in GraphConv's backward. */
int anemoi_gather_add_rows(const void* a, int64_t lda, const void* b, int64_t ldb, const int32_t* idx, void* out, int64_t ldo,
                           int32_t n_out, int32_t D, anemoi_dtype_t dtype, void* stream);

/* out[c][r] = x[r][c] (r < n_rows), 0 for n_rows <= r < n_pad.  Builds the K-contiguous operands of the weight-gradient GEMM
 * dW = dZ^T X (autograd of torch.nn.Linear), whose reduction runs over the rows. */
int anemoi_transpose_pad(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t n_rows, int32_t n_cols, int32_t n_pad,
                         anemoi_dtype_t dtype, void* stream);

/* Gated feed-forward layers (GatedMLPLayer, layers/mlp.py:25-59): out[r, c] = act(gv[r, c]) * gv[r, D + c], with
 * gv = [gate_proj(x) | value_proj(x)] from ONE fused projection.  kind: 0 sigmoid ("glu"), 1 SiLU ("swiglu"), 2 GELU erf
 * ("geglu"), 3 ReLU ("reglu").  anemoi_glu_bwd returns d_gv [n_rows, 2D] = [d_out * value * act'(gate) | d_out * act(gate)]. */
int anemoi_glu_fwd(const void* gate_value, int64_t ldgv, void* out, int64_t ldo, int32_t n_rows, int32_t D, int32_t kind,
                   anemoi_dtype_t dtype, void* stream);
int anemoi_glu_bwd(const void* gate_value, int64_t ldgv, const void* d_out, int64_t lddo, void* d_gate_value, int64_t lddgv,
                   int32_t n_rows, int32_t D, int32_t kind, anemoi_dtype_t dtype, void* stream);

/* ---- input normaliser at the model edges (scope row f4) ---------------------------------------------------------------
 * The reference normalises the batch before the model and de-normalises its output after it
 * (`pre_processors` / `post_processors` around `forward` in models/base.py:303-391; `InputNormalizer.transform`:
 * x * _norm_mul + _norm_add per variable, `inverse_transform`: (x - _norm_add) / _norm_mul, preprocessing/normalizer.py:154-252).
 * Here the transform rides in the kernels that touch every input / output element anyway. */

/* anemoi_assemble_input with the normaliser as a column program: the T * V time/variable columns become
 * x * col_mul[v] + col_add[v] (fp32, two roundings like torch's mul_ / add_; NULL/NULL = plain copy).  `x` may be fp32
 * (x_dtype) while `attrs` / `out` are in the model `dtype`. */
int anemoi_assemble_input_norm(const void* x, anemoi_dtype_t x_dtype, int64_t ld_t, int64_t ldx, int32_t T_steps, int32_t V,
                               const float* col_mul, const float* col_add, const void* attrs, int64_t lda, int32_t A, void* out,
                               int64_t ldo, int32_t W, int32_t n_rows, anemoi_dtype_t dtype, void* stream);

/* anemoi_assemble_output with the skip connection read from the RAW input and normalised on the fly:
 * out[n, v] = x_out[n, v] + (col_map[v] >= 0 ? x_skip[n, m] * col_mul[m] + col_add[m] : 0), m = col_map[v].
 * x_out in `model_dtype`; x_skip / out in `dtype` (= model_dtype or fp32: the reference adds the residual in the input's
 * dtype, models/encoder_processor_decoder.py:145-158).  The de-normalisation of the result is op 9 of anemoi_bound_columns. */
int anemoi_assemble_output_norm(const void* x_out, int64_t ldx, anemoi_dtype_t model_dtype, const void* x_skip, int64_t lds,
                                const int32_t* col_map, const float* col_mul, const float* col_add, void* out, int64_t ldo,
                                int32_t n_rows, int32_t n_cols, anemoi_dtype_t dtype, void* stream);

/* Stand-alone InputNormalizer.transform (inverse = 0: y = x * mul[c] + add[c]) / inverse_transform (inverse = 1:
 * y = (x - add[c]) / mul[c]) over the last dimension of a [n_rows, V] view; y may alias x (in_place). */
int anemoi_affine_columns(const void* x, int64_t ldx, void* y, int64_t ldy, const float* col_mul, const float* col_add,
                          int32_t inverse, int32_t n_rows, int32_t V, anemoi_dtype_t dtype, void* stream);

/* ---- device-initiated row exchange between the ranks of a model-parallel group (scope row e) ----------------------------
 * The reference exchanges node rows with host-issued NCCL collectives: the per-layer halo all-to-all
 * (`_halo_exchange` = dist.all_to_all_single with per-peer splits, distributed/primitives.py:422-460, called from
 * GraphTransformerProcessorBlock.forward, layers/block.py:1159-1172), the all-gather of shards (`_gather`,
 * primitives.py:60-183) and the source-row sync of the mappers (distributed/khop_edges.py:386-392).  On one MI355X node
 * the same data movement is ONE kernel of the rank's own stream per exchange: it stores the rows straight into the peers'
 * receive buffers (peer memory mapped with hipIpc; xGMI is point to point, every link carries only its own rows), publishes
 * them with a system-scope release + an epoch flag, and waits for the flags of the peers it receives from.  Being a plain
 * kernel node it is captured with the rest of the forward: one hipGraph per rank, no host in the loop
 * (anemoi_core_amd/distributed/peer.py holds the host side; RCCL stays available as the fallback wire).
 *
 * Memory.  anemoi_peer_alloc returns zeroed device memory of one of three kinds; anemoi_peer_export writes the
 * ANEMOI_PEER_HANDLE_BYTES-byte hipIpc handle other processes pass to anemoi_peer_open (never the exporting process itself);
 * anemoi_peer_close unmaps.  Handles travel over the caller's control plane (torch.distributed object collectives). */
#define ANEMOI_PEER_MEM_DEFAULT 0     /* hipMalloc: receive buffers, read by later kernels of the receiving stream */
#define ANEMOI_PEER_MEM_FINEGRAINED 1 /* hipExtMallocWithFlags(hipDeviceMallocFinegrained) */
#define ANEMOI_PEER_MEM_UNCACHED 2    /* hipExtMallocWithFlags(hipDeviceMallocUncached): flag words polled while a peer writes */
#define ANEMOI_PEER_HANDLE_BYTES 64
int anemoi_peer_alloc(void** ptr, int64_t bytes, int32_t kind);
int anemoi_peer_free(void* ptr);
int anemoi_peer_export(void* ptr, void* handle_out /* host, 64 bytes */);
int anemoi_peer_open(const void* handle /* host, 64 bytes */, void** ptr_out);
int anemoi_peer_close(void* ptr);

/* One exchange of one channel.  Packed row i of the send order is src[(send_index ? send_index[i] : i)] (row_bytes bytes, a
 * multiple of 16, ld_src_bytes apart); `table` is int64 [6][n_peers] in device memory: remote_base (address of this rank's
 * first row in peer p's receive buffer), remote_flag (address of the word peer p polls for this rank), send_begin,
 * send_count (rows of the packed order for peer p), signal (1: publish the epoch to p), expect (1: wait for p's epoch).
 * local_flags = this rank's words of the channel: [n_peers flags | seq | ticket] (uint32, ANEMOI_PEER_MEM_UNCACHED).  total_rows
 * may be 0 (signal / expect only: the forward-level barrier).  A peer that does not show up within timeout_ticks (100 MHz
 * wall clock) sets *status = 0x80000000 | peer instead of hanging the device.  `status` points at a block of >= 5 words:
 * [0] the time-out word, [2] exchanges run, [3] ticks summed over them between "my rows released" and "every expected flag
 * seen", [4] the longest such wait (diagnostics; the caller may zero them between forwards). */
int anemoi_peer_exchange_rows(const void* src, int64_t ld_src_bytes, const int32_t* send_index, const int64_t* table,
                              int32_t n_peers, int32_t row_bytes, int32_t total_rows, uint32_t* local_flags, uint32_t* status,
                              int64_t timeout_ticks, void* stream);

/* ---- row-resident layer chain (csrc/gt_chain2.hip) -----------------------------------------------------------------------------
 * Everything of a GraphTransformer block that is local to a node row, after the edge attention, as ONE launch:
 *     x1   = attn W_p^T + b_p + x_res                       projection + skip            (layers/block.py:1263-1266)
 *     h    = GELU(LayerNorm(x1; ln1) W_1^T + b_1)           node_dst_mlp, first Linear   (layers/block.py:1268-1271, layers/mlp.py:158-169)
 *     x2   = h W_2^T + b_2 + x1;  x_out = x2 [+ extra]      second Linear + skip [+ the model's latent skip, added to the ROUNDED
 *                                                           block output as `x_latent_proc + x_latent` does,
 *                                                           models/encoder_processor_decoder.py:295-296]
 *     q_out = LayerNorm(x_out; lnq) W_q^T + b_q             optional: the NEXT block's layer_norm_attention + fused
 *                                                           [lin_query; lin_key; lin_value; lin_self] (layers/block.py:1237-1245)
 * replacing four anemoi_linear_* launches per block (and the LayerNorm fold's statistics hand-off between them).  A workgroup keeps a
 * panel of <= 48 rows in LDS through the whole chain and streams the weights from L2 straight into MFMA operand registers; the
 * hidden activations [n_rows, hidden] are never written.  The workgroup's eight waves form two groups of four that work DIFFERENT
 * GEMM segments (group A: projection, MLP first Linear + GELU, even chunks of the trailing projection; group B: MLP second Linear, odd
 * chunks), so that one group's epilogue runs beside the other group's MFMA and weight stream.
 *
 * Weights are FRAGMENT-MAJOR images made once per parameter version (ops.pack_weight_frag): for W [O, K] row-major (O % 64 == 0,
 * K % 32 == 0) the image is [O/64 slabs][K/32 k-steps][4 column blocks][4 k-slots][16 rows][8 elements], i.e. element
 * (slab, ks, ni, kslot, row, e) = W[slab*64 + ni*16 + row][ks*32 + kslot*8 + e] - one contiguous KiB per MFMA B fragment.
 *
 * The two LayerNorms are computed as fp32 statistics of the rounded rows and applied WITHOUT their affine part, the normalised row
 * rounded to the model dtype; the affine part is folded by the caller into the Linear that follows (ops.gt_layer_chain2 does it once
 * per parameter version):
 *     w1 = fragment-major image of W_1 diag(gamma_1) (rounded to the model dtype),   d1 = W_1 beta_1 + b_1
 *     wq = fragment-major image of W_q diag(gamma_q),                                 dq = W_q beta_q + b_q
 * and every bias enters as the START value of its GEMM's accumulators: vec = [b_p (512) | d1 (hidden) | b_2 (512) | dq (q_out_features)]
 * in the model dtype, kept in LDS (2*512 + hidden + q_out_features <= 6144, else ANEMOI_E_UNSUPPORTED).  With BOTH extra and a trailing
 * projection the projection reads LayerNorm(x2 + extra): the decoder's layer_norm_attention_src + [lin_key; lin_value] behind the last processor
 * block and the latent skip (encoder_processor_decoder.py:295-296, layers/block.py:981-984).  q_out_features: a multiple of 512, or 128 / 256 /
 * 384 (a NARROW trailing projection: the decoder's node_data_extractor, layers/mapper.py:688-704, zero-padded to a multiple of 128 rows); x_out
 * may be NULL when there is a trailing projection and no extra (x2 is then not written).  channels must be 512; hidden a multiple of 512;
 * 16-bit dtypes; all row pointers 16-byte aligned, all leading dimensions multiples of 8 elements.
 * rows_per_tile = 0 lets the
 * library choose (anemoi_gt_chain_rows_per_tile: 48). */
typedef struct anemoi_gt_chain2_args {
  const void* attn;   int64_t ld_attn;    /* [n_rows, channels]   attention output + self term */
  const void* x_res;  int64_t ld_x;       /* [n_rows, channels]   the block's input */
  const void* wp;                         /* projection, fragment-major [channels, channels] */
  const void* w1;     int32_t hidden;     /* fragment-major [hidden, channels], layer_norm_mlp_dst's gamma folded in */
  const void* w2;                         /* fragment-major [channels, hidden] */
  const void* wq;     int32_t q_out_features; /* fragment-major [q_out_features, channels], the next block's gamma folded in; 0: none */
  const void* vec;                        /* [b_p | d1 | b_2 | dq], model dtype, 16-byte aligned */
  float ln1_eps;      float lnq_eps;
  const void* extra;  int64_t ld_extra;   /* optional [n_rows, channels] or NULL */
  void* x_out;        int64_t ld_out;     /* [n_rows, channels] */
  void* q_out;        int64_t ld_q;       /* [n_rows, q_out_features] */
  int32_t n_rows;     int32_t channels;   int32_t rows_per_tile;
  void* timeline;     /* must be NULL (ANEMOI_E_UNSUPPORTED otherwise); the experiments build of the library takes a uint64
                         [min(256, panels)][8 waves][48] buffer for its instrumented instantiation (tools/chain2_timeline.py) */
} anemoi_gt_chain2_args_t;
int anemoi_gt_chain2_fwd(const anemoi_gt_chain2_args_t* args, anemoi_dtype_t dtype, void* stream);
int anemoi_gt_chain_rows_per_tile(int32_t n_rows);

/* ---- row-resident embedding chain of a GraphTransformer mapper side (round 6; csrc/gt_rowchain.hip) ---------------------------------
 * One side of a GraphTransformer mapper in ONE launch:
 *     y     = x W_e^T + b_e                       emb_nodes_src / emb_nodes_dst = Linear(in, 512)      (layers/mapper.py:556-566, 688-694)
 *     q_out = LayerNorm(y) [W_a; W_b]^T + b       layer_norm_attention_src + [lin_key; lin_value] or
 *                                                 layer_norm_attention_dest + [lin_query; lin_self]      (layers/block.py:981-984)
 * replacing anemoi_linear_stats_fwd (the embedding GEMM + row statistics) and anemoi_linear_lnfold_fwd (the projection with the folded
 * LayerNorm).  x_out = NULL: y is never written (the encoder's source side: the block returns the source rows untouched).
 *   x  [n_rows, in_features], in_features a multiple of 8 up to 512;
 *   we fragment-major image (see anemoi_gt_chain2_fwd) of W_e zero-padded to [512, 128 ceil(in_features / 128)];
 *   wq fragment-major image of [W_a; W_b] diag(gamma) [q_out_features, 512] (the LayerNorm's affine part folded in by the caller,
 *      ops.gt_row_chain), vec = [b_e (512) | W beta + b (q_out_features)] in the model dtype;
 *   channels must be 512, q_out_features a multiple of 512 up to 2048, 16-bit dtypes, row pointers 16-byte aligned. */
typedef struct anemoi_gt_rowchain_args {
  const void* x;      int64_t ld_x;       int32_t in_features;
  const void* we;                         /* fragment-major [channels, 128 ceil(in_features / 128)] */
  const void* wq;     int32_t q_out_features; /* fragment-major [q_out_features, channels] */
  const void* vec;                        /* [b_e | dq], model dtype, 16-byte aligned */
  float ln_eps;
  void* x_out;        int64_t ld_out;     /* optional [n_rows, channels] or NULL */
  void* q_out;        int64_t ld_q;       /* [n_rows, q_out_features] */
  int32_t n_rows;     int32_t channels;   int32_t rows_per_tile;  /* rows_per_tile = 0: 48 */
} anemoi_gt_rowchain_args_t;
int anemoi_gt_rowchain_fwd(const anemoi_gt_rowchain_args_t* args, anemoi_dtype_t dtype, void* stream);

/* ---- cluster chain: the block tail for FEW rows (round 6; csrc/gt_cluster_chain.hip) ----------------------------------------------
 * What anemoi_gt_chain2_fwd computes (same operands, same folded LayerNorms, same vec layout with hidden = 2048:
 * [b_p 512 | d1 2048 | b_2 512 | dq q_out_features]), organised for block tails of a few thousand rows - a rank's share of a sharded
 * mesh, small hidden meshes: FOUR CUs of one XCD own a 48-row panel as a tensor-parallel group over the MLP's hidden width (each streams
 * the projection, one 512-column chunk of the first Linear, the matching K-slice of the second Linear and one chunk of the trailing
 * projection: 2 MiB instead of 6.5 MiB per layer), exchange the second Linear's fp32 partial sums ONCE per panel through `workspace`
 * (agent-scope stores / loads + an atomic counter per cluster) and add them in member order, so that every member holds the same x2.
 * Replaces layers/block.py:1263-1273 (+ :1237-1245 of the next block) like the chain; hidden must be 2048, q_out_features <= 2048.
 *   ln_out (nullable): LayerNorm_attn'(x2) WITHOUT its affine part [n_rows, 512] - what a sharded block sends to its halo peers;
 *   workspace: anemoi_gt_cluster_chain_workspace_bytes() bytes, 128-byte aligned, ZERO at allocation and then owned by the library (the
 *   cluster counters in it are monotonic across launches); one workspace per device and stream of launches.
 * A cluster's four workgroups must be resident together (the launch uses at most one workgroup per CU); a member that waits for a
 * partner for more than ~1 s traps: the launch fails, it never returns partial sums. */
typedef struct anemoi_gt_cluster_chain_args {
  const void* attn;   int64_t ld_attn;    /* [n_rows, channels]   attention output + self term */
  const void* x_res;  int64_t ld_x;       /* [n_rows, channels]   the block's input */
  const void* wp;                         /* projection, fragment-major [channels, channels] */
  const void* w1;     int32_t hidden;     /* fragment-major [hidden, channels], layer_norm_mlp_dst's gamma folded in; hidden = 2048 */
  const void* w2;                         /* fragment-major [channels, hidden] */
  const void* wq;     int32_t q_out_features; /* fragment-major [q_out_features, channels], the next block's gamma folded in; 0: none */
  const void* vec;                        /* [b_p | d1 | b_2 | dq], model dtype, 16-byte aligned */
  float ln1_eps;      float lnq_eps;
  const void* extra;  int64_t ld_extra;   /* optional [n_rows, channels] or NULL (not together with q_out / ln_out) */
  void* x_out;        int64_t ld_out;     /* [n_rows, channels] */
  void* q_out;        int64_t ld_q;       /* [n_rows, q_out_features] */
  void* ln_out;       int64_t ld_ln;      /* optional [n_rows, channels] */
  void* q_out2;       int64_t ld_q2;      int32_t q_split;  /* optional second destination of the trailing projection: its 512-column chunks
                                             q_split, q_split + 1, ... go to q_out2 [n_rows, q_out_features - 512 q_split] instead of q_out (a
                                             sharded block: q | self for its own rows, k | v into the head of the buffer its halo rows arrive in);
                                             q_out2 = NULL: everything to q_out */
  void* workspace;    int64_t workspace_bytes;
  int32_t n_rows;     int32_t channels;
} anemoi_gt_cluster_chain_args_t;
int anemoi_gt_cluster_chain_fwd(const anemoi_gt_cluster_chain_args_t* args, anemoi_dtype_t dtype, void* stream);
int64_t anemoi_gt_cluster_chain_workspace_bytes(void);

/* ---- row-resident chains of the GraphConv (GNN) processor block (round 4; csrc/gnn_chain.hip) ------------------------------------
 * GraphConv (layers/conv.py:29-81) in its gather-add form, with an edge MLP of three Linears (mlp_extra_layers = 0):
 *     e_new = LayerNorm(W_2 gelu(W_1 gelu(W_e e + g1[idx1] + g2[idx2] + b_0) + b_1) + b_2; ln) + e
 * where g1 = x_dst W_i^T and g2 = x_src W_j^T are node-level rows gathered by the edge's destination / source, W_e = the edge
 * columns of the first Linear's weight.  ONE launch instead of three edge-level GEMMs + the LayerNorm / residual half of
 * anemoi_edge_ln_residual_segment_sum_fwd (whose arithmetic and rounding points it keeps: the GEMM outputs rounded to the model
 * dtype, one rounding of LayerNorm(z) + e); the scatter-sum over e_new is anemoi_segment_sum_rows.  w0 / w1 / w2 are fragment-major
 * images (see anemoi_gt_chain2_fwd) of [512, 512] weights. */
int anemoi_gnn_edge_chain_fwd(const void* e, int64_t ld_e, const void* g1, int64_t ld_g1, const int32_t* idx1, const void* g2, int64_t ld_g2,
                              const int32_t* idx2, const void* w0, const void* b0, const void* w1, const void* b1, const void* w2, const void* b2,
                              const void* ln_w, const void* ln_b, float eps, void* e_new, int64_t ld_o, int32_t n_rows, int32_t channels,
                              anemoi_dtype_t dtype, void* stream);
/* An embedding MLP of the GNN mappers / processor as ONE launch (the edge chain without gathered rows):
 *   out = LayerNorm(W_2 gelu(W_1 gelu(W_0 x + b_0) + b_1) + b_2) [+ res]
 * Replaces: MLP.forward for `emb_edges` / `emb_nodes_src` / `emb_nodes_dst` (layers/mlp.py:29-100 as built at
 * layers/mapper.py:640-700 and layers/processor.py GNNProcessor: Linear -> GELU -> Linear -> GELU -> Linear -> LayerNorm).
 *   x [n_rows, in_features] with in_features in {128, 256, 384, 512} (the caller zero-pads the raw width; ld_x >= in_features);
 *   w0: fragment-major image of the [512, in_features] weight (zero columns for the padding), w1 / w2: of [512, 512];
 *   res (nullable): [n_rows, 512] rows added after the LayerNorm. */
int anemoi_gnn_mlp_chain_fwd(const void* x, int64_t ld_x, int32_t in_features, const void* w0, const void* b0, const void* w1, const void* b1,
                             const void* w2, const void* b2, const void* ln_w, const void* ln_b, float eps, const void* res, int64_t ld_res,
                             void* out, int64_t ld_o, int32_t n_rows, int32_t channels, anemoi_dtype_t dtype, void* stream);

/* The node MLP of a GraphConv block (layers/block.py:392-394; MLP of three Linears + LayerNorm, layers/mlp.py:97-179) with its skip:
 *     x_out = LayerNorm(W_c gelu(W_b gelu(W_a [x | agg] + b_a) + b_b) + b_c; ln) + x
 * and optionally t_out = x_out W_t^T [+ b_t] - the NEXT block's stacked node-level terms [x W_i^T | x W_j^T] its edge chain gathers.
 * wa [512, 1024], wb, wc [512, 512], wt [t_out_features, 512] fragment-major. */
int anemoi_gnn_node_chain_fwd(const void* x, int64_t ld_x, const void* agg, int64_t ld_a, const void* wa, const void* ba, const void* wb,
                              const void* bb, const void* wc, const void* bc, const void* ln_w, const void* ln_b, float eps, void* x_out,
                              int64_t ld_o, const void* wt, const void* bt, int32_t t_out_features, void* t_out, int64_t ld_t, int32_t n_rows,
                              int32_t channels, anemoi_dtype_t dtype, void* stream);
/* The same launch with GraphConv's scatter-sum inside it (layers/conv.py:81: `out = scatter(edge_attr_new, edge_index[1], reduce="sum")`):
 * instead of an aggregated table the kernel takes the EDGE rows edge_rows [M, 512] (sorted by destination) and seg_ptr [n_rows + 1] (the CSC
 * column pointer) and forms agg[n] = sum of the rows [seg_ptr[n], seg_ptr[n+1]) in fp32, in edge order, rounded once - the arithmetic of
 * anemoi_segment_sum_rows - while it loads the panel.  Saves that launch, the [N, 512] table it writes and its read-back. */
int anemoi_gnn_node_chain_segsum_fwd(const void* x, int64_t ld_x, const void* edge_rows, int64_t ld_e, const int32_t* seg_ptr, const void* wa,
                                     const void* ba, const void* wb, const void* bb, const void* wc, const void* bc, const void* ln_w,
                                     const void* ln_b, float eps, void* x_out, int64_t ld_o, const void* wt, const void* bt,
                                     int32_t t_out_features, void* t_out, int64_t ld_t, int32_t n_rows, int32_t channels,
                                     anemoi_dtype_t dtype, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* ANEMOI_HIP_H */
