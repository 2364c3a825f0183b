// Argument block of the GraphConv edge-MLP chain kernels (gnn_chain.hip: eight symmetric waves; experiments/gnn_chain2.hip: two wave groups
// with roles, experiments build only).
#pragma once
#include <cstdint>

namespace anemoi {

struct EdgeChainArgs {
  const void* e;   int64_t ld_e;            // [M, 512] edge features (A operand of the first GEMM and the residual)
  const void* g1;  int64_t ld_g1; const int32_t* idx1;  // rows added in the first epilogue: g1[idx1[m]] (x W_i^T by destination)
  const void* g2;  int64_t ld_g2; const int32_t* idx2;  //                                  g2[idx2[m]] (x W_j^T by source)
  const char* w0;  const void* b0;          // fragment-major [512, 512] each
  const char* w1;  const void* b1;
  const char* w2;  const void* b2;
  const void* ln_g; const void* ln_b; float ln_eps;
  void* e_new;     int64_t ld_o;
  int n_rows, rows_per_tile, n_tiles;
  int dbg;  // experiments build only (the product kernels ignore it): bit 0 no GELU, bit 2 (value 4) no gathered rows, bit 3 (value 8) no global
            // stores - timing only, WRONG results -; bit 4 (value 16, results valid) alternating wave priorities.  gnn_chain2: 2 = no gather,
            // 4 = no residual, 8 = no store
  // the MLP instantiation (no gathered rows): y = LayerNorm(W_2 gelu(W_1 gelu(W_0 x + b_0) + b_1) + b_2) [+ res]
  const void* res = nullptr; int64_t ld_res = 0;  // optional residual rows (the edge chain's residual is e itself)
  int k0_groups = 4;                              // width of x / K of the first GEMM in units of 128 columns (w0: fragment-major [512, 128 k0_groups])
  unsigned long long* timeline = nullptr;         // developer aid (TL instantiation): [workgroups][8 waves][kETlSlots] shader-clock stamps
};

// experiments/gnn_chain2.hip: the role-split launch of the same computation (returns an ANEMOI_* code); mlp: the embedding-MLP instantiation
int launch_edge_chain2(const EdgeChainArgs& a, int dtype, void* stream, bool mlp);

}  // namespace anemoi
