/* Entry points of the EXPERIMENTS build of the library only (python -m anemoi_core_amd.build --experiments ->
 * lib/libanemoi_hip_exp.so, compiled with -DANEMOI_EXPERIMENTS): measured-and-superseded kernels kept buildable for same-box A/Bs, and
 * the instrumented (in-kernel timeline) instantiations.  NOT part of the product ABI (include/anemoi_hip.h). */
#pragma once
#include "anemoi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- row-resident layer chain (round 4) ---------------------------------------------------------------------------------
 * Everything of a GraphTransformerProcessorBlock that is local to a node row, after the edge attention, as ONE launch:
 *     x1   = attn W_p^T + b_p + x_res                       projection + skip            (layers/block.py:1263-1266)
 *     h    = GELU(LayerNorm(x1; ln1) W_1^T + b_1)           node_dst_mlp, first Linear   (layers/block.py:1268-1271, layers/mlp.py:158-169)
 *     x2   = h W_2^T + b_2 + x1;  x_out = x2 [+ extra]      second Linear + skip [+ the model's latent skip, added to the ROUNDED
 *                                                           block output as `x_latent_proc + x_latent` does,
 *                                                           models/encoder_processor_decoder.py:295-296]
 *     q_out = LayerNorm(x_out; lnq) W_q^T + b_q             optional: the NEXT block's layer_norm_attention + fused
 *                                                           [lin_query; lin_key; lin_value; lin_self] (layers/block.py:1237-1245)
 * replacing four anemoi_linear_* launches per block (and the LayerNorm fold's statistics hand-off between them).  A workgroup keeps a
 * panel of <= 48 rows in LDS through the whole chain and streams the weights from L2 straight into MFMA operand registers; the
 * hidden activations [n_rows, hidden] are never written (csrc/gt_chain.hip).  LayerNorm here is the plain fp32 LayerNorm of the
 * rounded 16-bit rows with its output rounded to the model dtype - the arithmetic of the reference under autocast.
 *
 * Weights are FRAGMENT-MAJOR images made once per parameter version (ops.pack_weight_frag): for W [O, K] row-major (O % 64 == 0,
 * K % 32 == 0) the image is [O/64 slabs][K/32 k-steps][4 column blocks][4 k-slots][16 rows][8 elements], i.e. element
 * (slab, ks, ni, kslot, row, e) = W[slab*64 + ni*16 + row][ks*32 + kslot*8 + e] - one contiguous KiB per MFMA B fragment.
 * channels must be 512 (8 waves x 64 columns); hidden and q_out_features multiples of 512; 16-bit dtypes; all row pointers
 * 16-byte aligned.  rows_per_tile = 0 lets the library choose (anemoi_gt_chain_rows_per_tile). */
typedef struct anemoi_gt_chain_args {
  const void* attn;   int64_t ld_attn;    /* [n_rows, channels]   attention output + self term */
  const void* x_res;  int64_t ld_x;       /* [n_rows, channels]   the block's input */
  const void* wp;     const void* bp;     /* projection: fragment-major [channels, channels], bias [channels] */
  const void* ln1_w;  const void* ln1_b;  float ln1_eps;  /* layer_norm_mlp_dst (ln1_b may be NULL) */
  const void* w1;     const void* b1;     int32_t hidden; /* fragment-major [hidden, channels], bias [hidden] */
  const void* w2;     const void* b2;     /* fragment-major [channels, hidden], bias [channels] */
  const void* extra;  int64_t ld_extra;   /* optional [n_rows, channels] or NULL */
  void* x_out;        int64_t ld_out;     /* [n_rows, channels] */
  const void* lnq_w;  const void* lnq_b;  float lnq_eps;  /* the next block's layer_norm_attention (q_out_features > 0) */
  const void* wq;     const void* bq;     int32_t q_out_features; /* fragment-major [q_out_features, channels], bias; 0: no trailing projection */
  void* q_out;        int64_t ld_q;       /* [n_rows, q_out_features] */
  int32_t n_rows;     int32_t channels;   int32_t rows_per_tile;
  void* timeline;     /* NULL; developer aid: uint64 [min(256, panels)][8 waves][48] device buffer - the instrumented instantiation stamps the
                         shader clock at every phase boundary of each workgroup's first panel (tools/chain_timeline.py) */
} anemoi_gt_chain_args_t;
int anemoi_gt_chain_fwd(const anemoi_gt_chain_args_t* args, anemoi_dtype_t dtype, void* stream);


/* Developer aid (tools/edge_chain_timeline.py): anemoi_gnn_edge_chain_fwd (bf16) through an instrumented instantiation that stamps the
 * shader clock at the phase boundaries of every panel - all 8 waves, a workgroup's first five panels.
 * timeline: uint64 [min(256, ceil(n_rows / 64))][8][48], slot 0 = kernel entry, then 8 slots per panel. */
int anemoi_gnn_edge_chain_timeline(const void* e, int64_t ld_e, const void* g1, int64_t ld_g1, const int32_t* idx1, const void* g2,
                                   int64_t ld_g2, const int32_t* idx2, const void* w0, const void* b0, const void* w1, const void* b1,
                                   const void* w2, const void* b2, const void* ln_w, const void* ln_b, float eps, void* e_new,
                                   int64_t ld_o, int32_t n_rows, unsigned long long* timeline, void* stream);


#ifdef __cplusplus
}
#endif
