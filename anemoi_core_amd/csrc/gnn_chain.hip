// Row-resident chains of the GraphConv (GNN) processor block for gfx950 - the two MLPs of reference layers/conv.py:29-81 and
// layers/block.py:361-395, each as ONE launch on the machinery of chain_core.h (a 48-row panel in LDS, fragment-major weights
// streamed into MFMA operand registers):
//
//  edge chain   e' = LayerNorm(W_2 gelu(W_1 gelu(W_e e + (x W_i^T)[dst] + (x W_j^T)[src] + b_0) + b_1) + b_2) + e
//      = GraphConv's edge MLP on cat[x_i, x_j, e] in its gather-add form (the two node-level terms are precomputed rows, gathered in
//      the first epilogue) + the trailing LayerNorm and residual.  Replaces three edge-level [M x 512] -> 512 GEMM launches and
//      the LayerNorm half of edge_ln_res_segsum: 251 us per processor layer at M = 81 840, of which the GEMMs' intermediates are
//      4 x 84 MB of HBM round trips.  Here only e is read and e' written; the segment sum over e' is anemoi_segment_sum_rows.
//  node chain   x' = LayerNorm(W_c gelu(W_b gelu(W_a [x | agg] + b_a) + b_b) + b_c) + x, and optionally the NEXT block's stacked
//      node-level projection [x' W_i'^T | x' W_j'^T] (the gather operands of its edge chain).  agg is either a table or - segsum entry
//      point - formed in the launch from the dst-sorted edge rows e' (GraphConv's scatter-sum, layers/conv.py:81).
//  MLP chain    the edge chain's MLP instantiation: an embedding MLP (Linear-GELU-Linear-GELU-Linear-LayerNorm into 512 channels) of
//      the mappers / first processor block on a zero-padded input of 128..512 columns (layers/mlp.py:29-100).
//
// Why the row-resident form pays HERE when it did not for the GraphTransformer block: a panel needs 1.5 MB
// (edge chain) / 2.5-3.5 MB (node chain) of weights through its CU's L1 instead of 6.5 MB, against launches whose K = N = 512
// GEMMs sit on their fixed costs and on HBM round trips of the [M, 512] intermediates.
#include "chain_core.h"
#include "gnn_chain_args.h"

namespace anemoi {

constexpr int kETlSlots = 48;  // entry + 8 stamps per panel of a workgroup's first 5 panels (tools/edge_chain_timeline.py)

// Edge panels are 64 rows (4 MFMA row bands): the same weight stream serves a third more rows than a 48-row panel would (81 840
// edges = 256 CUs x 5 panels of 64: five passes over the 1.5 MB of weights per CU instead of seven), and the two hidden layers
// live IN PLACE in one LDS buffer next to the e panel (2 x 64 KiB): a barrier between a GEMM's last fragment read and the
// epilogue's write-back replaces the third buffer.
constexpr int kENB = 4, kERows = 16 * kENB, kEBuf = kERows * kRowBytes;
constexpr int kERedOff = 2 * kEBuf;
constexpr int kEdgeSmem = kERedOff + kERows * 8 * 2 * 4;

// MLP = true: the same chain without the gathered rows, the residual optional and the first GEMM's K a multiple of 128 up to 512 -
// the embedding MLPs of the GNN mappers / processor (reference layers/mlp.py:29-100 as built at layers/mapper.py:640-700:
// Linear -> GELU -> Linear -> GELU -> Linear -> LayerNorm over [rows, in] -> 512).
template <typename T, bool MLP = false, bool TL = false>
__global__ __launch_bounds__(512, 1) void gnn_edge_chain_kernel(EdgeChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const bufE = smem;          // e panel: operand of GEMM 1, residual of the LayerNorm
  unsigned char* const bufH = smem + kEBuf;  // h1, then h2 in place, then the staging strips of the stores
  float* const red = reinterpret_cast<float*>(smem + kERedOff);
  constexpr int NB = kENB, kLd = 2 * NB;     // panel loads per thread
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const uint32_t loff = lane * 16;
  const int dbg = kExperiments ? a.dbg : 0;  // (timing experiments: compiled out of the product library)
  const int flip = (dbg & 16) ? (wave >> 2) : -1;  // (experiment: alternating issue priority of the two waves of a SIMD, chain_core.h)
  const int nq0 = MLP ? a.k0_groups : 4;  // K of the first GEMM / 128
  const int nslots = nq0 * 16;            // 16-byte slots per panel row
  const char* const w0 = a.w0 + (int64_t)wave * (nq0 * 16384);
  const char* const w1 = a.w1 + (int64_t)wave * kSlab;
  const char* const w2 = a.w2 + (int64_t)wave * kSlab;
  const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
  [[maybe_unused]] int tl_n = 0;
  auto stamp = [&]() {  // (to LDS: a global store per stamp would sit in the vmcnt queue of the weight ring)
    if constexpr (TL) {
      if (tl_n < kETlSlots) {  // the low 32 bits (wrap: seconds) - one register, or the instrumented build spills around the GEMM segments
        const unsigned now = (unsigned)__builtin_readcyclecounter();
        if (lane == 0) reinterpret_cast<unsigned*>(smem + kEdgeSmem)[wave * kETlSlots + tl_n] = now;
      }
      ++tl_n;
    }
  };

  int tile = blockIdx.x;
  if (tile >= a.n_tiles) return;
  stamp();  // kernel entry
  // The e panel is only the first GEMM's operand (the LayerNorm's residual is re-read from global, L2-hot), so the NEXT panel
  // moves into the free e buffer under the second GEMM - the phase with registers to spare (no gathered rows) - a quarter per
  // group of 4 K-steps: requested at the top of a group, written to LDS at the top of the next (8 registers in flight; the whole
  // panel at once is 32 and spills).  (The same with LDS-DMA - no registers at all - was built first: with a global_load_lds in
  // flight hipcc's waitcnt pass puts vmcnt(0) in front of every MFMA that consumes a ds_read fragment, draining the weight ring
  // at every K-step.)
  constexpr int kPiece = kLd / 4;
  u32x4 va[kPiece];
  auto request_piece = [&](int t, int pc) {
    const int r0 = t * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    int td = tid;
    asm volatile("" : "+v"(td));
#pragma unroll
    for (int i = 0; i < kPiece; ++i) {
      const int idx = td + 512 * (pc * kPiece + i);
      const int rr = min(idx >> 6, nr - 1), slot = idx & 63;
      if (!MLP || slot < nslots) va[i] = *reinterpret_cast<const u32x4*>((const T*)a.e + (int64_t)(r0 + rr) * a.ld_e + slot * 8);
    }
  };
  auto write_piece = [&](int t, int pc) {
    const int nr = min(a.rows_per_tile, a.n_rows - t * a.rows_per_tile);
    int td = tid;
    asm volatile("" : "+v"(td));
#pragma unroll
    for (int i = 0; i < kPiece; ++i) {
      const int idx = td + 512 * (pc * kPiece + i);
      const int row = idx >> 6, slot = idx & 63;
      if (!MLP || slot < nslots)  // (the swizzle permutes the 16 slots of a 128-column group among themselves)
        *reinterpret_cast<u32x4*>(bufE + row * kRowBytes + ((slot ^ (row & 15)) << 4)) = row < nr ? va[i] : zero4;
    }
  };
#pragma unroll 1
  for (int pc = 0; pc < 4; ++pc) {  // the first panel: nothing to hide behind
    request_piece(tile, pc);
    write_piece(tile, pc);
  }
  __builtin_amdgcn_sched_barrier(0);
  frag8 bq[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      bq[j][ni] = *reinterpret_cast<gfrag_t>(uniform_ptr(w0 + j * 4096) + loff + ni * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  lds_barrier();

  for (;;) {
    const int r0 = tile * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    const int next_tile = tile + (int)gridDim.x;
    const bool more = next_tile < a.n_tiles;
    stamp();  // panel p + 0: the e panel is in LDS (behind the barrier)
    // The gathered node-level rows of this lane's panel rows (2 x 4 x 8 bytes per row).  Loads retire in order, so every weight-ring wait
    // behind a gather waits for the gather too: with indices and rows requested in front of the GEMM, its steps 4.. stood behind two
    // dependent round trips of scattered 8-byte loads (in-kernel timeline: the first GEMM 10.4 us, the other two 6.6).  Now: the indices
    // at the panel's start, the rows behind the GEMM's 12th K-step - the ring loads issued after them are the NEXT segment's, first
    // consumed behind the epilogue that consumes the rows anyway.  What is left of their cost is their share of the CU's L1 path, which the
    // weight stream saturates: 3 us per panel (ANEMOI_EDGE_CHAIN_DBG=4 against 0, same box: 177 against 192 us per launch).  Tried on top,
    // same-box A/Bs, no gain, not kept: the next panel's indices prefetched an epilogue ahead (two registers per lane, __shfl at the
    // use: 176.3 against 176.2 us); sixteen 16-byte loads per lane instead of thirty-two 8-byte ones, the two lanes of a row 16 lanes
    // apart fetching what both need and trading halves with v_permlane16_swap (176.0-177.7 against 174.0-176.6 us; the tail segment
    // then has registers for three of the four bands only).  Neither are the e' stores in front of the next panel what delays it
    // (DBG=8: the first GEMM stays at 10.9 us).
    u32x2 ga[MLP ? 1 : NB][4], gb[MLP ? 1 : NB][4], pb[4];
    [[maybe_unused]] int i1[NB], i2[NB];
    if constexpr (!MLP) {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < NB; ++mi) {
        const int m = r0 + min(mi * 16 + lc.x, nr - 1);
        i1[mi] = a.idx1[m];
        i2[mi] = a.idx2[m];
      }
    }
    load_cols<T>((const T*)a.b0, wave, g, pb);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[NB][4];
    // ---- h1 = gelu(e W_e^T + g1[dst] + g2[src] + b0) -> bufH
    zero_acc<T, NB>(acc);
    if constexpr (MLP) {
      gemm_seg<T, NB>(bufE, lane, bq, w0, w1, loff, acc, NoHook(), nq0, flip);
    } else {
      gemm_seg<T, NB>(bufE, lane, bq, w0, w0 + 3 * 16384, loff, acc, NoHook(), 3, flip);  // K = 0 .. 383
      {
        const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
        for (int mi = 0; mi < NB; ++mi) {
          const T* r1 = (const T*)a.g1 + (int64_t)i1[mi] * a.ld_g1 + wave * 64 + lc.g * 4;
          const T* r2 = (const T*)a.g2 + (int64_t)i2[mi] * a.ld_g2 + wave * 64 + lc.g * 4;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            if (dbg & 4) {  // (timing experiment: no gathered rows)
              ga[mi][ni] = gb[mi][ni] = u32x2{0u, 0u};
            } else {
              ga[mi][ni] = *reinterpret_cast<const u32x2*>(r1 + ni * 16);
              gb[mi][ni] = *reinterpret_cast<const u32x2*>(r2 + ni * 16);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // K = 384 .. 511: slots 48 + s of a panel row; the swizzle only touches the low four bits of a slot index, so the offset is a pointer offset
      gemm_seg<T, NB>(bufE + 48 * 16, lane, bq, w0 + 3 * 16384, w1, loff, acc, NoHook(), 1, flip < 0 ? -1 : flip ^ 1);
    }
    stamp();  // + 1: first GEMM done
    {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < NB; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float bias[4], t1[4], t2[4], t[4];
          unpack4<T>(pb[ni], bias);
          if constexpr (MLP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = acc[mi][ni][r] + bias[r];
          } else {
            unpack4<T>(ga[mi][ni], t1);
            unpack4<T>(gb[mi][ni], t2);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = (acc[mi][ni][r] + bias[r]) + (t1[r] + t2[r]);  // the association of linear.hip's gather-add epilogue
          }
          if (!(dbg & 1)) {
            gelu_fast2(t[0], t[1]);
            gelu_fast2(t[2], t[3]);
          }
          *reinterpret_cast<u32x2*>(bufH + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]) = pack4<T>(t);
        }
    }
    load_cols<T>((const T*)a.b1, wave, g, pb);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    stamp();  // + 2: gather-add + GELU epilogue written, barrier passed
    // ---- h2 = gelu(h1 W_1^T + b1), written over h1 once every wave has read its last fragment of it
    zero_acc<T, NB>(acc);
    if (more) {  // (every wave is behind its last fragment read of the e panel: the E1 barrier)
      auto hook = [&](int q) {
        if (q > 0) write_piece(next_tile, q - 1);
        request_piece(next_tile, q);
      };
      gemm_seg<T, NB>(bufH, lane, bq, w1, w2, loff, acc, hook, 4, flip);
      write_piece(next_tile, 3);  // (published by the barrier at the end of this panel)
    } else {
      gemm_seg<T, NB>(bufH, lane, bq, w1, w2, loff, acc, NoHook(), 4, flip);
    }
    stamp();  // + 3: second GEMM done (and the next panel moved into LDS)
    u32x2 hp[NB][4];
#pragma unroll
    for (int mi = 0; mi < NB; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        float bias[4], t[4];
        unpack4<T>(pb[ni], bias);
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = acc[mi][ni][r] + bias[r];
        if (!(dbg & 1)) {
          gelu_fast2(t[0], t[1]);
          gelu_fast2(t[2], t[3]);
        }
        hp[mi][ni] = pack4<T>(t);
      }
    lds_barrier();
    {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < NB; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<u32x2*>(bufH + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]) = hp[mi][ni];
    }
    u32x2 pg[4], pt[4];
    load_cols<T>((const T*)a.b2, wave, g, pb);
    load_cols<T>((const T*)a.ln_g, wave, g, pg);
    if (a.ln_b != nullptr) {
      load_cols<T>((const T*)a.ln_b, wave, g, pt);
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) pt[ni] = u32x2{0u, 0u};
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    stamp();  // + 4: GELU epilogue written in place (two barriers)
    // ---- z = h2 W_2^T + b2 (rounded, as the Linear's output is); e' = LayerNorm(z) + e -> global
    zero_acc<T, NB>(acc);
    gemm_seg<T, NB>(bufH, lane, bq, w2, w0, loff, acc, NoHook(), 4, flip);  // (behind it: the next panel's first segment; after the last panel a harmless re-read)
    stamp();  // + 5: third GEMM done
    {
      const LaneCtx lc = lane_ctx(lane, wave);
      u32x2 er[NB][4];  // this lane's values of the e rows (the residual): in flight under the bias add and the row statistics
      const T* const resp = MLP ? (const T*)a.res : (const T*)a.e;
      const int64_t ld_res = MLP ? a.ld_res : a.ld_e;
      if (!MLP || resp != nullptr) {
#pragma unroll
        for (int mi = 0; mi < NB; ++mi) {
          const T* erow = resp + (int64_t)(r0 + min(mi * 16 + lc.x, nr - 1)) * ld_res + wave * 64 + lc.g * 4;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) er[mi][ni] = *reinterpret_cast<const u32x2*>(erow + ni * 16);
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < NB; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) er[mi][ni] = u32x2{0u, 0u};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < NB; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float bias[4], t[4];
          unpack4<T>(pb[ni], bias);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = acc[mi][ni][r] + bias[r];
          unpack4<T>(pack4<T>(t), t);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mi][ni][r] = t[r];
        }
      float mean[NB], rstd[NB];
      panel_row_stats<T, NB>(acc, a.ln_eps, red, wave, lc.x, lc.g, mean, rstd);  // (one barrier: behind it no wave reads bufH)
      stamp();  // + 6: row statistics merged (barrier passed)
      u32x2 pk[NB][4];
#pragma unroll
      for (int mi = 0; mi < NB; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float gv[4], bv[4], ev[4], o[4];
          unpack4<T>(pg[ni], gv);
          unpack4<T>(pt[ni], bv);
          unpack4<T>(er[mi][ni], ev);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaf((acc[mi][ni][r] - mean[mi]) * rstd[mi], gv[r], bv[r]) + ev[r];  // edge_ln_res_segsum's arithmetic
          pk[mi][ni] = pack4<T>(o);
        }
      store_block_via_strip<T, NB>(pk, bufH + wave * (kERows * 128), (T*)a.e_new + (int64_t)r0 * a.ld_o + wave * 64, a.ld_o, (dbg & 8) ? 0 : nr, lane, wave);
    }
    stamp();  // + 7: LayerNorm + residual applied, rows stored through the strips
    if (!more) break;
    tile = next_tile;
    lds_barrier();  // every wave has emptied its strip (bufH is the next panel's h1) and written its part of the next e panel
  }
  if constexpr (TL) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < kETlSlots)
      a.timeline[((size_t)blockIdx.x * 8 + wave) * kETlSlots + lane] =
          lane < tl_n ? (unsigned long long)reinterpret_cast<const unsigned*>(smem + kEdgeSmem)[wave * kETlSlots + lane] : 0ull;
  }
}

struct NodeChainArgs {
  const void* x;    int64_t ld_x;           // [N, 512] node rows (first K half of the first GEMM; the residual)
  const void* agg;  int64_t ld_a;           // [N, 512] aggregated messages (second K half)
  const char* wa;   const void* ba;         // fragment-major [512, 1024]
  const char* wb;   const void* bb;         // fragment-major [512, 512]
  const char* wc;   const void* bc;
  const void* ln_g; const void* ln_b; float ln_eps;
  void* x_out;      int64_t ld_o;
  const char* wt;   const void* bt; int tc;  // optional trailing projection [512 tc, 512] of x_out (bt may be null: no bias); tc = 0: none
  void* t_out;      int64_t ld_t;
  int n_rows, rows_per_tile, n_tiles;
  // seg_ptr != null: `agg` holds the EDGE rows [M, 512] (sorted by destination) and the panel's agg rows are their segmented sums,
  // agg[n] = sum_{i in [seg_ptr[n], seg_ptr[n+1])} edge_rows[i] - fp32, in edge order, rounded once: anemoi_segment_sum_rows' arithmetic
  const int32_t* seg_ptr = nullptr;
};

template <typename T>
__global__ __launch_bounds__(512, 1) void gnn_node_chain_kernel(NodeChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const bufA = smem;                  // x panel: operand of GEMM 1a, residual of the LayerNorm
  unsigned char* const bufB = smem + kBufBytes;      // h1; later x' as the trailing projection's operand
  unsigned char* const bufC = smem + 2 * kBufBytes;  // agg panel (GEMM 1b), then h2, then the staging strips
  float* const red = reinterpret_cast<float*>(smem + kRedOff);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const uint32_t loff = lane * 16;
  const char* const wa0 = a.wa + (int64_t)wave * 2 * kSlab;  // K = 1024: two segments per slab
  const char* const wb0 = a.wb + (int64_t)wave * kSlab;
  const char* const wc0 = a.wc + (int64_t)wave * kSlab;
  const char* const wt0 = a.tc > 0 ? a.wt + (int64_t)wave * kSlab : wa0;  // + c * 8 slabs
  const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};

  int tile = blockIdx.x;
  if (tile >= a.n_tiles) return;
  u32x4 va[6], vb[6];
  auto request_panel = [&](int t) {
    const int r0 = t * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + 512 * i;
      const int rr = min(idx >> 6, nr - 1), slot = idx & 63;
      va[i] = *reinterpret_cast<const u32x4*>((const T*)a.x + (int64_t)(r0 + rr) * a.ld_x + slot * 8);
      if (a.seg_ptr == nullptr) vb[i] = *reinterpret_cast<const u32x4*>((const T*)a.agg + (int64_t)(r0 + rr) * a.ld_a + slot * 8);
    }
  };
  // The aggregation inside the launch (GraphConv's scatter-sum, layers/conv.py:81): a wave sums the in-edge rows of panel rows wave,
  // wave + 8, ... (1 KiB per edge row, one 16-byte slot per lane, 4 rows in flight) - the panel's ~8 x 48 edge rows come through this
  // CU's L1 once, instead of a launch that reads every edge row and writes an [N, 512] table this kernel then reads back.
  auto aggregate_panel = [&](int t) {
    const int r0 = t * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    const T* const base = (const T*)a.agg + lane * 8;
    // the segment bounds of this wave's six rows in ONE vector load (lane 2 i + w holds seg_ptr[row_i + w]), broadcast with v_readlane;
    // the first 8 edge rows of row i + 1 are requested before row i is summed (the loop is a chain of memory round trips otherwise:
    // measured 11 us per panel with one row at a time, bounds by scalar loads)
    int pv = 0;
    if (lane < 12) pv = a.seg_ptr[r0 + min(wave + 8 * (lane >> 1), nr - 1) + (lane & 1)];
    u32x4 buf[2][8];
    auto issue = [&](int beg, int end, u32x4 (&b)[8]) {
      if (end > beg) {
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const u32x4*>(base + (int64_t)min(beg + u, end - 1) * a.ld_a);
      }
    };
    auto add_row = [&](const u32x4& r, float (&s8)[8]) {
      float lo[4], hi[4];
      unpack4<T>(u32x2{r[0], r[1]}, lo);
      unpack4<T>(u32x2{r[2], r[3]}, hi);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s8[c] += lo[c];
        s8[4 + c] += hi[c];
      }
    };
    int beg[6], end[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      beg[i] = __builtin_amdgcn_readlane(pv, 2 * i);
      end[i] = (wave + 8 * i < nr) ? __builtin_amdgcn_readlane(pv, 2 * i + 1) : beg[i];
    }
    issue(beg[0], end[0], buf[0]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i + 1 < 6) issue(beg[i + 1], end[i + 1], buf[(i + 1) & 1]);
      const int row = wave + 8 * i;
      float s8[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) s8[c] = 0.f;
      const int n = end[i] - beg[i];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < n) add_row(buf[i & 1][u], s8);
#pragma unroll 1
      for (int e = beg[i] + 8; e < end[i]; e += 4) {  // (degrees beyond 8: the mappers' hubs)
        u32x4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const u32x4*>(base + (int64_t)min(e + u, end[i] - 1) * a.ld_a);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e + u < end[i]) add_row(r[u], s8);
      }
      float lo[4] = {s8[0], s8[1], s8[2], s8[3]}, hi[4] = {s8[4], s8[5], s8[6], s8[7]};
      const u32x2 pl = pack4<T>(lo), ph = pack4<T>(hi);
      *reinterpret_cast<u32x4*>(bufC + row * kRowBytes + ((lane ^ (row & 15)) << 4)) = u32x4{pl[0], pl[1], ph[0], ph[1]};
    }
  };
  request_panel(tile);
  __builtin_amdgcn_sched_barrier(0);
  frag8 bq[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      bq[j][ni] = *reinterpret_cast<gfrag_t>(uniform_ptr(wa0 + j * 4096) + loff + ni * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }

  for (;;) {
    const int r0 = tile * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + 512 * i;
      const int row = idx >> 6, slot = idx & 63;
      const int off = row * kRowBytes + ((slot ^ (row & 15)) << 4);
      *reinterpret_cast<u32x4*>(bufA + off) = row < nr ? va[i] : zero4;
      if (a.seg_ptr == nullptr) *reinterpret_cast<u32x4*>(bufC + off) = row < nr ? vb[i] : zero4;
    }
    if (a.seg_ptr != nullptr) aggregate_panel(tile);
    u32x2 pb[4];
    load_cols<T>((const T*)a.ba, wave, g, pb);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();

    f32x4 acc[3][4];
    // ---- h1 = gelu([x | agg] W_a^T + b_a) -> bufB
    zero_acc<T>(acc);
    gemm_seg<T>(bufA, lane, bq, wa0, wa0 + kSlab, loff, acc);
    gemm_seg<T>(bufC, lane, bq, wa0 + kSlab, wb0, loff, acc);
    {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float bias[4], t[4];
          unpack4<T>(pb[ni], bias);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = acc[mi][ni][r] + bias[r];
          gelu_fast2(t[0], t[1]);
          gelu_fast2(t[2], t[3]);
          *reinterpret_cast<u32x2*>(bufB + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]) = pack4<T>(t);
        }
    }
    load_cols<T>((const T*)a.bb, wave, g, pb);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();  // h1 complete; every wave is done with the agg panel
    // ---- h2 = gelu(h1 W_b^T + b_b) -> bufC
    zero_acc<T>(acc);
    gemm_seg<T>(bufB, lane, bq, wb0, wc0, loff, acc);
    {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float bias[4], t[4];
          unpack4<T>(pb[ni], bias);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = acc[mi][ni][r] + bias[r];
          gelu_fast2(t[0], t[1]);
          gelu_fast2(t[2], t[3]);
          *reinterpret_cast<u32x2*>(bufC + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]) = pack4<T>(t);
        }
    }
    u32x2 pg[4], pt[4];
    load_cols<T>((const T*)a.bc, wave, g, pb);
    load_cols<T>((const T*)a.ln_g, wave, g, pg);
    if (a.ln_b != nullptr) {
      load_cols<T>((const T*)a.ln_b, wave, g, pt);
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) pt[ni] = u32x2{0u, 0u};
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    // ---- y = h2 W_c^T + b_c (rounded); x' = LayerNorm(y) + x -> global (and bufB for the trailing projection)
    zero_acc<T>(acc);
    gemm_seg<T>(bufC, lane, bq, wc0, a.tc > 0 ? wt0 : wa0, loff, acc);
    const int next_tile = tile + (int)gridDim.x;
    const bool more = next_tile < a.n_tiles;
    unsigned char* const strip = bufC + wave * (kPanel * 128);  // behind the statistics' barrier no wave reads bufC any more
    {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float bias[4], t[4];
          unpack4<T>(pb[ni], bias);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = acc[mi][ni][r] + bias[r];
          unpack4<T>(pack4<T>(t), t);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mi][ni][r] = t[r];
        }
      float mean[3], rstd[3];
      panel_row_stats<T>(acc, a.ln_eps, red, wave, lc.x, lc.g, mean, rstd);
      u32x2 pk[3][4];
#pragma unroll
      for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float gv[4], bv[4], xv[4], o[4];
          unpack4<T>(pg[ni], gv);
          unpack4<T>(pt[ni], bv);
          unpack4<T>(*reinterpret_cast<const u32x2*>(bufA + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]), xv);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaf((acc[mi][ni][r] - mean[mi]) * rstd[mi], gv[r], bv[r]) + xv[r];  // layernorm_fwd's arithmetic (+ residual)
          pk[mi][ni] = pack4<T>(o);
          if (a.tc > 0) *reinterpret_cast<u32x2*>(bufB + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]) = pk[mi][ni];
        }
      store_block_via_strip<T>(pk, strip, (T*)a.x_out + (int64_t)r0 * a.ld_o + wave * 64, a.ld_o, nr, lane, wave);
    }
    if (a.tc > 0) {
      lds_barrier();  // x' panel complete
      for (int c = 0; c < a.tc; ++c) {
        if (a.bt != nullptr) {
          load_cols<T>((const T*)a.bt + c * kCh, wave, g, pb);
        } else {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) pb[ni] = u32x2{0u, 0u};
        }
        __builtin_amdgcn_sched_barrier(0);
        zero_acc<T>(acc);
        gemm_seg<T>(bufB, lane, bq, wt0 + (int64_t)c * 8 * kSlab, c + 1 < a.tc ? wt0 + (int64_t)(c + 1) * 8 * kSlab : wa0, loff, acc);
        u32x2 pk[3][4];
#pragma unroll
        for (int mi = 0; mi < 3; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            float bias[4], o[4];
            unpack4<T>(pb[ni], bias);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[mi][ni][r] + bias[r];
            pk[mi][ni] = pack4<T>(o);
          }
        store_block_via_strip<T>(pk, strip, (T*)a.t_out + (int64_t)r0 * a.ld_t + c * kCh + wave * 64, a.ld_t, nr, lane, wave);
      }
    }
    if (!more) break;
    tile = next_tile;
    request_panel(tile);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();  // every wave is done with this panel's buffers
  }
}

static int chain_rows_per_tile(int n_rows, int cap = kPanel) {
  static const int forced = env_int(getenv("ANEMOI_CHAIN_ROWS"), 0, 0, 64);
  if (forced > 0) return forced < cap ? forced : cap;
  const int64_t rounds = ((int64_t)n_rows + 256 * cap - 1) / (256 * cap);
  const int64_t r = ((int64_t)n_rows + 256 * rounds - 1) / (256 * rounds);
  return (int)(r < 1 ? 1 : (r > cap ? cap : r));
}

}  // namespace anemoi

using namespace anemoi;

static bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

extern "C" int anemoi_gnn_edge_chain_fwd(const void* e, int64_t ld_e, const void* g1, int64_t ld_g1, const int32_t* idx1, const void* g2,
                                         int64_t ld_g2, const int32_t* idx2, const void* w0, const void* b0, const void* w1, const void* b1,
                                         const void* w2, const void* b2, const void* ln_w, const void* ln_b, float eps, void* e_new,
                                         int64_t ld_o, int32_t n_rows, int32_t channels, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "gnn_edge_chain_fwd: 16-bit model dtypes only");
  ANEMOI_REQUIRE(n_rows >= 0 && channels == kCh, "gnn_edge_chain_fwd: channels=%d (this kernel is built for %d)", channels, kCh);
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(e && g1 && idx1 && g2 && idx2 && w0 && b0 && w1 && b1 && w2 && b2 && ln_w && e_new, "gnn_edge_chain_fwd: null operand");
  ANEMOI_REQUIRE(al(e, 16) && al(e_new, 16) && al(w0, 16) && al(w1, 16) && al(w2, 16) && al(g1, 8) && al(g2, 8) && al(b0, 8) && al(b1, 8) && al(b2, 8) &&
                     al(ln_w, 8) && al(ln_b, 8) && ld_e % 8 == 0 && ld_o % 8 == 0 && ld_g1 % 4 == 0 && ld_g2 % 4 == 0 && ld_e >= kCh && ld_o >= kCh &&
                     ld_g1 >= kCh && ld_g2 >= kCh,
                 "gnn_edge_chain_fwd: operand alignment / leading dimensions");
  EdgeChainArgs a{e, ld_e, g1, ld_g1, idx1, g2, ld_g2, idx2, (const char*)w0, b0, (const char*)w1, b1, (const char*)w2, b2, ln_w, ln_b, eps, e_new, ld_o,
                  n_rows, chain_rows_per_tile(n_rows, kERows), 0, 0};
  // round 5: ANEMOI_GNN_CHAIN_V2=1 selects the two-group launch (gnn_chain2.hip): built, parity-green, and slower than the symmetric
  // kernel below (212 against 195 us at 81 840 rows: its 40 / 48-row panels need 7-8 passes over the weights instead of 5,
  // profiles/r05_gnn_edge_chain_role_split.txt) - kept for the A/B
#ifdef ANEMOI_EXPERIMENTS
  static const int v2 = env_int(getenv("ANEMOI_GNN_CHAIN_V2"), 0, 0, 1);
  if (v2) return launch_edge_chain2(a, dtype, stream, false);
#endif
  static const int dbg = ANEMOI_EXPERIMENT_ENV("ANEMOI_EDGE_CHAIN_DBG", 0, 0, 31);
  a.dbg = dbg;
  a.n_tiles = (n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  hipStream_t st = as_stream(stream);
  if (dtype == ANEMOI_BF16) {
    static PerDeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gnn_edge_chain_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kEdgeSmem); });
    hipLaunchKernelGGL((gnn_edge_chain_kernel<bf16_t>), dim3(grid), dim3(512), kEdgeSmem, st, a);
  } else {
    static PerDeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gnn_edge_chain_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kEdgeSmem); });
    hipLaunchKernelGGL((gnn_edge_chain_kernel<f16_t>), dim3(grid), dim3(512), kEdgeSmem, st, a);
  }
  return check_launch("gnn_edge_chain_kernel");
}

#ifdef ANEMOI_EXPERIMENTS
// Developer aid: the same launch through the instrumented instantiation (shader-clock stamps at the phase boundaries of every panel,
// all waves, a workgroup's first five panels) - tools/edge_chain_timeline.py.  timeline: [min(256, panels)][8][48] uint64.
extern "C" int anemoi_gnn_edge_chain_timeline(const void* e, int64_t ld_e, const void* g1, int64_t ld_g1, const int32_t* idx1, const void* g2,
                                              int64_t ld_g2, const int32_t* idx2, const void* w0, const void* b0, const void* w1, const void* b1,
                                              const void* w2, const void* b2, const void* ln_w, const void* ln_b, float eps, void* e_new,
                                              int64_t ld_o, int32_t n_rows, unsigned long long* timeline, void* stream) {
  ANEMOI_REQUIRE(n_rows > 0 && e && g1 && idx1 && g2 && idx2 && w0 && b0 && w1 && b1 && w2 && b2 && ln_w && e_new && timeline, "gnn_edge_chain_timeline: null operand");
  // (the argument checks of anemoi_gnn_edge_chain_fwd; this entry point is bf16 only)
  ANEMOI_REQUIRE(al(e, 16) && al(e_new, 16) && al(w0, 16) && al(w1, 16) && al(w2, 16) && al(g1, 8) && al(g2, 8) && al(b0, 8) && al(b1, 8) && al(b2, 8) &&
                     al(ln_w, 8) && al(ln_b, 8) && ld_e % 8 == 0 && ld_o % 8 == 0 && ld_g1 % 4 == 0 && ld_g2 % 4 == 0 && ld_e >= kCh && ld_o >= kCh &&
                     ld_g1 >= kCh && ld_g2 >= kCh,
                 "gnn_edge_chain_timeline: operand alignment / leading dimensions");
  EdgeChainArgs a{e, ld_e, g1, ld_g1, idx1, g2, ld_g2, idx2, (const char*)w0, b0, (const char*)w1, b1, (const char*)w2, b2, ln_w, ln_b, eps, e_new, ld_o,
                  n_rows, chain_rows_per_tile(n_rows, kERows), 0, 0};
  a.timeline = timeline;
  static const int dbg = ANEMOI_EXPERIMENT_ENV("ANEMOI_EDGE_CHAIN_DBG", 0, 0, 31);
  a.dbg = dbg;
  a.n_tiles = (n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  constexpr int smem = kEdgeSmem + 8 * kETlSlots * 8;
  static PerDeviceOnce once;
  once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gnn_edge_chain_kernel<bf16_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
  hipLaunchKernelGGL((gnn_edge_chain_kernel<bf16_t, false, true>), dim3(grid), dim3(512), smem, as_stream(stream), a);
  return check_launch("gnn_edge_chain_kernel<timeline>");
}
#endif

extern "C" int anemoi_gnn_mlp_chain_fwd(const void* x, int64_t ld_x, int32_t in_features, const void* w0, const void* b0, const void* w1, const void* b1,
                                        const void* w2, const void* b2, const void* ln_w, const void* ln_b, float eps, const void* res, int64_t ld_res,
                                        void* out, int64_t ld_o, int32_t n_rows, int32_t channels, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "gnn_mlp_chain_fwd: 16-bit model dtypes only");
  ANEMOI_REQUIRE(n_rows >= 0 && channels == kCh, "gnn_mlp_chain_fwd: channels=%d (this kernel is built for %d)", channels, kCh);
  ANEMOI_REQUIRE(in_features >= 128 && in_features <= kCh && in_features % 128 == 0, "gnn_mlp_chain_fwd: in_features=%d (zero-padded width: 128, 256, 384 or 512)", in_features);
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && w0 && b0 && w1 && b1 && w2 && b2 && ln_w && out, "gnn_mlp_chain_fwd: null operand");
  ANEMOI_REQUIRE(al(x, 16) && al(out, 16) && al(w0, 16) && al(w1, 16) && al(w2, 16) && al(res, 8) && al(b0, 8) && al(b1, 8) && al(b2, 8) && al(ln_w, 8) &&
                     al(ln_b, 8) && ld_x % 8 == 0 && ld_o % 8 == 0 && ld_x >= in_features && ld_o >= kCh && (!res || (ld_res % 4 == 0 && ld_res >= kCh)),
                 "gnn_mlp_chain_fwd: operand alignment / leading dimensions");
  EdgeChainArgs a{x, ld_x, nullptr, 0, nullptr, nullptr, 0, nullptr, (const char*)w0, b0, (const char*)w1, b1, (const char*)w2, b2, ln_w, ln_b, eps, out, ld_o,
                  n_rows, chain_rows_per_tile(n_rows, kERows), 0, 0};
  a.res = res;
  a.ld_res = ld_res;
  a.k0_groups = in_features / 128;
#ifdef ANEMOI_EXPERIMENTS
  static const int v2 = env_int(getenv("ANEMOI_GNN_CHAIN_V2"), 0, 0, 1);
  if (v2) return launch_edge_chain2(a, dtype, stream, true);
#endif
  a.n_tiles = (n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  hipStream_t st = as_stream(stream);
  if (dtype == ANEMOI_BF16) {
    static PerDeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gnn_edge_chain_kernel<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kEdgeSmem); });
    hipLaunchKernelGGL((gnn_edge_chain_kernel<bf16_t, true>), dim3(grid), dim3(512), kEdgeSmem, st, a);
  } else {
    static PerDeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gnn_edge_chain_kernel<f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kEdgeSmem); });
    hipLaunchKernelGGL((gnn_edge_chain_kernel<f16_t, true>), dim3(grid), dim3(512), kEdgeSmem, st, a);
  }
  return check_launch("gnn_mlp_chain_kernel");
}

static int node_chain_launch(const void* x, int64_t ld_x, const void* agg, int64_t ld_a, const int32_t* seg_ptr, const void* wa, const void* ba, const void* wb,
                             const void* bb, const void* wc, const void* bc, const void* ln_w, const void* ln_b, float eps, void* x_out,
                             int64_t ld_o, const void* wt, const void* bt, int32_t t_out_features, void* t_out, int64_t ld_t,
                             int32_t n_rows, int32_t channels, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "gnn_node_chain_fwd: 16-bit model dtypes only");
  ANEMOI_REQUIRE(n_rows >= 0 && channels == kCh, "gnn_node_chain_fwd: channels=%d (this kernel is built for %d)", channels, kCh);
  if (n_rows == 0) return ANEMOI_OK;
  // (agg may be NULL when the segment pointer says every destination has no in-edge at all: an empty edge table has no rows to point at)
  ANEMOI_REQUIRE(x && (agg || seg_ptr) && wa && ba && wb && bb && wc && bc && ln_w && x_out, "gnn_node_chain_fwd: null operand");
  ANEMOI_REQUIRE(t_out_features >= 0 && t_out_features % kCh == 0 && (t_out_features == 0 || (wt && t_out && ld_t >= t_out_features && ld_t % 8 == 0 && al(t_out, 16) && al(wt, 16))),
                 "gnn_node_chain_fwd: the trailing projection needs wt, t_out and t_out_features %% %d == 0", kCh);
  ANEMOI_REQUIRE(al(x, 16) && al(agg, 16) && al(x_out, 16) && al(wa, 16) && al(wb, 16) && al(wc, 16) && al(ba, 8) && al(bb, 8) && al(bc, 8) && al(bt, 8) &&
                     al(ln_w, 8) && al(ln_b, 8) && ld_x % 8 == 0 && ld_a % 8 == 0 && ld_o % 8 == 0 && ld_x >= kCh && ld_a >= kCh && ld_o >= kCh,
                 "gnn_node_chain_fwd: operand alignment / leading dimensions");
  NodeChainArgs a{x, ld_x, agg, ld_a, (const char*)wa, ba, (const char*)wb, bb, (const char*)wc, bc, ln_w, ln_b, eps, x_out, ld_o,
                  (const char*)wt, bt, t_out_features / kCh, t_out, ld_t, n_rows, chain_rows_per_tile(n_rows), 0};
  // ANEMOI_GNN_NODE_ROWS: rows per panel of this launch alone (A/B of the even-spread rule against full 48-row panels on fewer CUs,
  // the rule that won for gt_chain2_kernel; profiles/r05_gnn_node_rows_ab.txt)
  static const int node_rows = ANEMOI_EXPERIMENT_ENV("ANEMOI_GNN_NODE_ROWS", 0, 0, kPanel);
  if (node_rows > 0) a.rows_per_tile = node_rows;
  a.seg_ptr = seg_ptr;
  a.n_tiles = (n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  hipStream_t st = as_stream(stream);
  if (dtype == ANEMOI_BF16) {
    static PerDeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gnn_node_chain_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainSmem); });
    hipLaunchKernelGGL((gnn_node_chain_kernel<bf16_t>), dim3(grid), dim3(512), kChainSmem, st, a);
  } else {
    static PerDeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gnn_node_chain_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainSmem); });
    hipLaunchKernelGGL((gnn_node_chain_kernel<f16_t>), dim3(grid), dim3(512), kChainSmem, st, a);
  }
  return check_launch("gnn_node_chain_kernel");
}

extern "C" int anemoi_gnn_node_chain_fwd(const void* x, int64_t ld_x, const void* agg, int64_t ld_a, const void* wa, const void* ba, const void* wb,
                                         const void* bb, const void* wc, const void* bc, const void* ln_w, const void* ln_b, float eps, void* x_out,
                                         int64_t ld_o, const void* wt, const void* bt, int32_t t_out_features, void* t_out, int64_t ld_t,
                                         int32_t n_rows, int32_t channels, anemoi_dtype_t dtype, void* stream) {
  return node_chain_launch(x, ld_x, agg, ld_a, nullptr, wa, ba, wb, bb, wc, bc, ln_w, ln_b, eps, x_out, ld_o, wt, bt, t_out_features, t_out, ld_t, n_rows,
                           channels, dtype, stream);
}

extern "C" int anemoi_gnn_node_chain_segsum_fwd(const void* x, int64_t ld_x, const void* edge_rows, int64_t ld_e, const int32_t* seg_ptr, const void* wa,
                                                const void* ba, const void* wb, const void* bb, const void* wc, const void* bc, const void* ln_w,
                                                const void* ln_b, float eps, void* x_out, int64_t ld_o, const void* wt, const void* bt,
                                                int32_t t_out_features, void* t_out, int64_t ld_t, int32_t n_rows, int32_t channels,
                                                anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(seg_ptr != nullptr || n_rows == 0, "gnn_node_chain_segsum_fwd: null segment pointer");
  return node_chain_launch(x, ld_x, edge_rows, ld_e, seg_ptr, wa, ba, wb, bb, wc, bc, ln_w, ln_b, eps, x_out, ld_o, wt, bt, t_out_features, t_out, ld_t,
                           n_rows, channels, dtype, stream);
}
