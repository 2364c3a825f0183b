// Linear layers with fused epilogues for gfx950:  y = act(A W^T + bias + gather terms) + residual.
//
// Replaces torch.nn.Linear (+GELU, + residual) as used by the reference's blocks
// (models/src/anemoi/models/layers/block.py:623-635, 1268-1271; layers/mlp.py:158-169).
//
//  * linear_mfma_kernel — the hot path: bf16/f16 operands, fp32 accumulation on the matrix cores
//    (v_mfma_f32_16x16x32_{bf16,f16}).  128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per
//    wave = 4x4 MFMA tiles), BK = 64, double-buffered LDS (64 KiB -> 2 workgroups per CU), register-staged
//    global->LDS copies issued one K-tile ahead, XOR-swizzled 16-byte LDS slots so both ds_write_b128 and the
//    fragment ds_read_b128 are bank-conflict free, XCD-aware workgroup->tile mapping (tiles that share an A
//    row-panel land on the same XCD/L2).  The MFMA is issued with operands swapped (D^T = W_tile . A_tile^T) so
//    that every lane ends up holding 4 CONSECUTIVE output columns of one row: bias / residual / gather terms and
//    the store are 8-byte vector accesses instead of 2-byte scatters.
//  * linear_generic_kernel — any dtype / any K, O (embeddings with K = 11..20, extractor with O = n_vars,
//    the fp32 parity path): 64x64 tile, fp32 FMA on the vector ALU.
//
// A = [x | x2] is a K-concatenation (GraphConv's cat[x, agg] / cat[x_i, x_j, e] never materialised).
#include <type_traits>

#include "common.h"

namespace anemoi {

struct LinArgs {
  const void* x;
  int64_t ldx;
  int K1;
  const void* x2;
  int64_t ldx2;
  int K2;
  const void* w;
  int64_t ldw;
  const void* bias;
  const void* g1;
  int64_t ldg1;
  const int32_t* idx1;
  const void* g2;
  int64_t ldg2;
  const int32_t* idx2;
  const void* residual;
  int64_t ldr;
  void* y;
  int64_t ldy;
  int n_rows;
  int O;
  int act;
  int splits = 1;     // persistent kernel only: split-K (see tile_origin)
  // LayerNorm fold (inference): a producer GEMM leaves per-row partial sums of its OUTPUT, the consumer GEMM applies the
  // normalisation of its INPUT rows in the epilogue:  LN(x) W^T = rstd (x (W gamma)^T - mean c) + d
  float* stats_out = nullptr;       // [n_rows][O / 64][2] fp32: sum and sum of squares of every 64-column strip (EPI_STATS)
  const float* stats_in = nullptr;  // [n_rows][ln_strips][2] of the input rows (EPI_LNFOLD)
  int ln_tail_begin = 0x7fffffff;   // rows >= this carry NO strip sums (the one tail rule: n_rows % 320 <= 32): statistics from the row
  const float* ln_c = nullptr;      // [O] row sums of the gamma-scaled weight
  const float* ln_d = nullptr;      // [O] W beta + bias
  int ln_strips = 0, ln_D = 0;
  float ln_eps = 0.f;
  int f32_atomic = 0; // persistent kernel only: y is fp32 and accumulated with atomics (caller zeroes it)
  void* y_pre = nullptr;  // training, with act = GELU on the DMA-ring kernels: the pre-activation is stored here as well (saves
  int64_t ldy_pre = 0;    // the backward a recomputing GEMM)
  int tail_rows = 0;  // big-tile kernel only: rows [n_rows, n_rows + tail_rows) are computed on the VALU, a column per wave
  int fast_epi = 1;   // interior tiles take mfma_epilogue_fast (ANEMOI_GEMM_FAST_EPI=0: the generic epilogue everywhere, for A/B runs)
};

// ---------------------------------------------------------------------------------------------- generic (VALU)
constexpr int GT = 64;   // generic tile (rows and cols)
constexpr int GK = 16;   // generic K step

template <typename T>
__global__ __launch_bounds__(256) void linear_generic_kernel(LinArgs a) {
  __shared__ float As[GK][GT + 1];
  __shared__ float Ws[GK][GT + 1];
  const T* __restrict__ x = (const T*)a.x;
  const T* __restrict__ x2 = (const T*)a.x2;
  const T* __restrict__ w = (const T*)a.w;
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int K = a.K1 + a.K2;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < K; k0 += GK) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = tid + 256 * r;
      const int rr = i >> 4, kk = i & 15;
      const int kg = k0 + kk;
      float av = 0.f, wv = 0.f;
      const int m = m0 + rr, n = n0 + rr;
      if (kg < K) {
        if (m < a.n_rows) av = kg < a.K1 ? to_float(x[(int64_t)m * a.ldx + kg]) : to_float(x2[(int64_t)m * a.ldx2 + (kg - a.K1)]);
        if (n < a.O) wv = to_float(w[(int64_t)n * a.ldw + kg]);
      }
      As[kk][rr] = av;
      Ws[kk][rr] = wv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; ++kk) {
      float av[4], wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = Ws[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
    }
    __syncthreads();
  }

  const T* __restrict__ bias = (const T*)a.bias;
  const T* __restrict__ g1 = (const T*)a.g1;
  const T* __restrict__ g2 = (const T*)a.g2;
  const T* __restrict__ res = (const T*)a.residual;
  T* __restrict__ y = (T*)a.y;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= a.n_rows) continue;
    const int64_t r1 = g1 ? (int64_t)a.idx1[m] * a.ldg1 : 0;
    const int64_t r2 = g2 ? (int64_t)a.idx2[m] * a.ldg2 : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= a.O) continue;
      float vv = acc[i][j];
      if (bias) vv += to_float(bias[n]);
      if (g1) vv += to_float(g1[r1 + n]);
      if (g2) vv += to_float(g2[r2 + n]);
      if (a.act == ANEMOI_ACT_GELU) vv = gelu_erf(vv);
      if (res) vv += to_float(res[(int64_t)m * a.ldr + n]);
      y[(int64_t)m * a.ldy + n] = from_float<T>(vv);
    }
  }
}

// ---------------------------------------------------------------------------------------------- MFMA path
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK * 2;  // one operand tile (16 KiB)

using frag8 = __attribute__((ext_vector_type(8))) short;  // 8 x 16-bit operand elements (4 VGPRs)
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <typename T>
__device__ __forceinline__ f32x4 mfma16(frag8 a, frag8 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<bf16_t>(frag8 a, frag8 b, f32x4 c) {
  using bf8 = __attribute__((ext_vector_type(8))) __bf16;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<f16_t>(frag8 a, frag8 b, f32x4 c) {
  using h8 = __attribute__((ext_vector_type(8))) _Float16;
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// byte offset of (row, 16-byte slot) inside a [128][64] 16-bit tile with XOR-swizzled slots
__device__ __forceinline__ int lds_off(int row, int slot) { return row * (BK * 2) + ((slot ^ ((row >> 1) & 7)) << 4); }

// Epilogue shared by the MFMA kernels.
// acc[mi][ni][r] = out[m = m0 + wr*64 + mi*16 + (lane & 15)][n = n0 + wc*64 + ni*16 + (lane>>4)*4 + r].
// The accumulators of a wave (64 x 64 fp32 = 16 KiB) are first transposed through the wave's private slice of LDS
// (XOR-swizzled 16-byte slots, conflict-free writes) so that the read-back hands every lane 4 consecutive columns of
// a row with 16 lanes covering the row's 64 columns: bias / gather / residual loads and the final store then touch
// whole 128-byte lines (4 rows per instruction) instead of 32-byte fragments of 16 different rows.
// `epi` = this wave's 16 KiB LDS slice; the caller has already synchronised the workgroup (the slice overlaps the
// operand stages).
template <typename T>
__device__ __forceinline__ void mfma_epilogue(const LinArgs& a, f32x4 (&acc)[4][4], int m0, int n0, int wr, int wc, int lane,
                                              unsigned char* epi) {
  const T* __restrict__ bias = (const T*)a.bias;
  const T* __restrict__ g1 = (const T*)a.g1;
  const T* __restrict__ g2 = (const T*)a.g2;
  const T* __restrict__ res = (const T*)a.residual;
  T* __restrict__ y = (T*)a.y;
  using V4 = Vec<T, 4>;
  // write: row = mi*16 + (lane&15), logical 16-byte slot = ni*4 + (lane>>4); physical slot = logical ^ (row & 15)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int row = mi * 16 + (lane & 15);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int slot = (ni * 4 + (lane >> 4)) ^ (row & 15);
      *reinterpret_cast<f32x4*>(epi + row * 256 + slot * 16) = acc[mi][ni];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: no workgroup barrier needed
  // read-back: instruction `it` covers rows it*4 .. it*4+3; lane -> row it*4 + (lane>>4), columns 4*(lane&15) ..+3
  const int nc = n0 + wc * 64 + (lane & 15) * 4;
  const bool n_ok = nc < a.O;  // O % 4 == 0: the 4-column group is entirely inside or outside
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr && n_ok) load_vec<T, 4>(bias + nc, bv);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4);
    const int m = m0 + wr * 64 + row;
    const bool ok = n_ok && m < a.n_rows;
    const int slot = (lane & 15) ^ (row & 15);
    const f32x4 c = *reinterpret_cast<const f32x4*>(epi + row * 256 + slot * 16);
    float vv[4] = {c[0] + bv[0], c[1] + bv[1], c[2] + bv[2], c[3] + bv[3]};
    V4 t1{}, t2{}, rv{};
    if (g1 != nullptr && ok) t1 = *reinterpret_cast<const V4*>(g1 + (int64_t)a.idx1[m] * a.ldg1 + nc);
    if (g2 != nullptr && ok) t2 = *reinterpret_cast<const V4*>(g2 + (int64_t)a.idx2[m] * a.ldg2 + nc);
    if (res != nullptr && ok) rv = *reinterpret_cast<const V4*>(res + (int64_t)m * a.ldr + nc);
    if (g1 != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] += to_float(t1.v[r]) + to_float(t2.v[r]);
    }
    if (a.act == ANEMOI_ACT_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] = gelu_erf(vv[r]);
    }
    if (res != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] += to_float(rv.v[r]);
    }
    if (ok) store_vec<T, 4>(y + (int64_t)m * a.ldy + nc, vv);
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void linear_mfma_kernel(LinArgs a, int tiles_n, int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 buffers][A tile | W tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware bijective remap: workgroups dispatched to the same XCD (id % 8) get consecutive tile ids.
  int id = blockIdx.x;
  {
    const int q = num_tiles >> 3, r = num_tiles & 7, xcd = id & 7, pos = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;

  const T* __restrict__ x = (const T*)a.x;
  const T* __restrict__ x2 = (const T*)a.x2;
  const T* __restrict__ w = (const T*)a.w;
  const int K = a.K1 + a.K2;
  const int nk = (K + BK - 1) / BK;

  // staging assignment: 1024 16-byte chunks per operand tile, 4 per thread; chunk c -> row c/8, slot c%8.
  // K need not be a multiple of BK: 8-element slots past the end are zero-filled (K1, K2 are multiples of 8).
  int st_row[4], st_slot[4];
  int64_t a_off[4], a2_off[4], w_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = tid + 256 * r;
    st_row[r] = c >> 3;
    st_slot[r] = c & 7;
    const int m = min(m0 + st_row[r], a.n_rows - 1);  // clamp: rows past the end are computed but never stored
    const int n = min(n0 + st_row[r], a.O - 1);
    a_off[r] = (int64_t)m * a.ldx;
    a2_off[r] = (int64_t)m * a.ldx2 - a.K1;
    w_off[r] = (int64_t)n * a.ldw;
  }

  u32x4 ra[4], rw[4];
  auto g_load = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kg = k0 + st_slot[r] * 8;
      u32x4 va = u32x4{0u, 0u, 0u, 0u}, vw = va;
      if (kg < K) {
        const T* ap = (kg < a.K1) ? (x + a_off[r] + kg) : (x2 + a2_off[r] + kg);
        va = *reinterpret_cast<const u32x4*>(ap);
        vw = *reinterpret_cast<const u32x4*>(w + w_off[r] + kg);
      }
      ra[r] = va;
      rw[r] = vw;
    }
  };
  auto s_store = [&](int buf) {
    unsigned char* base = smem + buf * 2 * kTileBytes;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = lds_off(st_row[r], st_slot[r]);
      *reinterpret_cast<u32x4*>(base + off) = ra[r];
      *reinterpret_cast<u32x4*>(base + kTileBytes + off) = rw[r];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  g_load(0);
  s_store(0);
  __syncthreads();

  const int frow = lane & 15, fslot = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) g_load(kt + 1);
    const unsigned char* As = smem + buf * 2 * kTileBytes;
    const unsigned char* Ws = As + kTileBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag8 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const frag8*>(As + lds_off(wr * 64 + i * 16 + frow, fslot + 4 * ks));
        fw[i] = *reinterpret_cast<const frag8*>(Ws + lds_off(wc * 64 + i * 16 + frow, fslot + 4 * ks));
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<T>(fw[ni], fa[mi], acc[mi][ni]);  // D^T tile: rows = n, cols = m
    }
    if (kt + 1 < nk) s_store(buf ^ 1);
    __syncthreads();
  }

  mfma_epilogue<T>(a, acc, m0, n0, wr, wc, lane, smem + wave * 16384);  // 4 x 16 KiB = the 64 KiB of operand stages
}

// ---------------------------------------------------------------------------------------------- persistent ring kernel
// Main MFMA path (K1, K2 multiples of 64).  Design notes, from measurements on MI355X at [10242 x 512] x [512 -> 2048]:
//  * operands travel HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPR round trip / ds_write pass); the DMA writes
//    lane-linear, so the XOR slot swizzle is applied to the per-lane SOURCE address and again on the fragment reads;
//  * 256 x 128 output tile per 8-wave workgroup (4 x 2 waves of 64 x 64): half the operand bytes per flop of a 128^2
//    tile (a CU's vector-memory path moves ~64 B/clk, a 128^2 x 64 step needs 32 KiB for 2 MFLOP);
//  * one workgroup per CU walks its tiles in a loop and the 3-stage operand ring never drains: the first K-tiles of
//    tile t+1 are already in flight during the epilogue of tile t.  Waits are COUNTED (vmcnt), the barrier is a raw
//    s_barrier (a __syncthreads() would drain the DMA queue);
//  * gfx950 counts stores in vmcnt in issue order: an interior tile leaves exactly 16 stores per wave in the queue, so
//    the next tile's first K-steps wait with vmcnt(ring + 16) and the stores drain under the MFMAs;
//  * the epilogue is specialised at compile time (EPI flags): residual / gather rows are fetched up front, the
//    accumulators go through a wave-private LDS band (swizzled, conflict-free) so that every global access of the
//    epilogue is a whole 128-byte line.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

enum : int { EPI_RES = 1, EPI_GATHER = 2, EPI_GELU = 4, EPI_STATS = 8, EPI_LNFOLD = 16, EPI_PRE = 32 };  // PRE: also store GELU's argument (training)

// acc[mi][ni][r] = out[m0 + wr*64 + mi*16 + (lane & 15)][n0 + wc*64 + ni*16 + (lane>>4)*4 + r]
// epi: this wave's 4 KiB LDS slice (16 rows x 256 B); one 16-row band (mi) at a time.  The band is written in the MFMA
// layout (16-byte slots, 32-byte pairs XOR-swizzled by the row: conflict-free ds_write_b128) and read back as 8
// CONSECUTIVE columns per lane, 8 lanes per row: every global access of the epilogue (bias, residual, gather rows,
// output) is a 16-byte-per-lane access covering whole 128-byte lines - the 8-byte form measured store-issue-bound
// (guide T21).  Interior tiles issue exactly kEpiStores = 8 stores per wave.
constexpr int kEpiStores = 8;

// LayerNorm fold, small-tile kernels: mean / rstd of one row straight from the producer's strip sums (the big-tile kernel
// stages them in LDS once per tile; with 64-row tiles on a few thousand rows the 8 lanes that share a row re-read its 64
// bytes from L1 instead).  Same summation order as the LDS variant: bit-identical statistics.
template <typename T>
__device__ __forceinline__ void ln_row_stats(const LinArgs& a, int m, int cp, float& mu, float& rs) {
  if (m >= a.n_rows + a.tail_rows) {  // a row of a ragged last tile that does not exist: never stored, nothing to fetch for it
    mu = 0.f;
    rs = 1.f;
    return;
  }
  const float* st = a.stats_in + (int64_t)m * a.ln_strips * 2;
  float s1 = 0.f, s2 = 0.f;
  if (m >= a.ln_tail_begin) {
    // a producer's peeled tail row: sums of the stored row itself, shared out over the 8 lanes that hold the row in the epilogue
    // (cp = lane & 7; all eight are here together: same m) - eight loads in flight per lane and one butterfly, where one lane walking
    // the row load by load took 60 us of a 9 us kernel (the reason the fold looked slow on small meshes)
    const T* xr = (const T*)a.x + (int64_t)m * a.ldx;
    for (int k0 = 0; k0 < a.ln_D; k0 += 512) {
      float xv[8][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j * 64 + cp * 8;
        if (k < a.ln_D) {
          load_vec<T, 8>(xr + k, xv[j]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) xv[j][i] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          s1 += xv[j][i];
          s2 = fmaf(xv[j][i], xv[j][i], s2);
        }
    }
    s1 = group_sum<8>(s1);
    s2 = group_sum<8>(s2);
  } else if (a.ln_strips == 8) {
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const f32x4*>(st)[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s1 += v[q][0];
      s2 += v[q][1];
      s1 += v[q][2];
      s2 += v[q][3];
    }
  } else {
    for (int q = 0; q < a.ln_strips; ++q) {
      s1 += st[2 * q];
      s2 += st[2 * q + 1];
    }
  }
  const float inv = 1.0f / (float)a.ln_D;
  mu = s1 * inv;
  rs = rsqrtf(fmaxf(s2 * inv - mu * mu, 0.f) + a.ln_eps);
}

// The gather-add epilogue's row indices for one lane (rows mi * 16 + it * 8 of its wave's slab), loaded by the big-tile kernel during
// the first K-step of the tile: otherwise every band of the epilogue pays two DEPENDENT global latencies - index, then row - with
// the matrix cores idle (the GNN's edge GEMM: 98 us with the gather-add against 57 us without; GNN forward 8.03 -> 7.86 ms).
template <int MI>
struct GatherIdx {
  int i1[MI][2], i2[MI][2];
};

// Fast path of the epilogue below for INTERIOR tiles (every row and column of the tile exists) with bias / residual / gather /
// GELU only - the hot case.  The generic epilogue carries, per 16-byte store, the predicates of ragged tiles, the 4-column tail
// of O % 8 == 4, the fp32-atomic split-K branch and 64-bit address products: measured on MI355X its on-chip work (stores
// removed) was 11-13 us of a 35 us [10242 x 512] -> 2048 GEMM and 17 us with GELU, against 22 us for the whole kernel without
// an epilogue.  Here: no predicates, one 64-bit base per lane + uniform row offsets, and NO waits between a band's LDS writes
// and its read-back (a wave's LDS instructions execute in order, so the read-back sees the writes; only the compiler has
// to be kept from reordering them) - the transposition of band mi+1 overlaps the arithmetic and stores of band mi.
template <typename T, int EPI, int MI, bool LDS_STATS = false>
__device__ __forceinline__ void mfma_epilogue_fast(const LinArgs& a, f32x4 (&acc)[MI][4], int m0, int n0, int wr, int wc, int lane,
                                                   unsigned char* epi, const float* ln_rows = nullptr, const GatherIdx<MI>* gidx = nullptr) {
  using V8 = Vec<T, 8>;
  const int cp = lane & 7;
  // LayerNorm fold without the per-tile LDS table (the 64-row kernels): at D = 512 the 8 lanes that hold a row in this epilogue
  // each fetch ONE of its 8 strip sums, for all 2 * MI rows of the lane up front (one latency instead of one per 16-row band: the
  // band loop below used to wait ~1 us eight times), and a butterfly over the 8 lanes completes them where they are used.
  [[maybe_unused]] f32x2 ln_part[MI][2];
  [[maybe_unused]] const bool ln_strip_per_lane = a.ln_strips == 8;
  if constexpr ((EPI & EPI_LNFOLD) != 0 && !LDS_STATS) {
    if (ln_strip_per_lane) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int m = min(m0 + wr * (16 * MI) + (lane >> 3) + mi * 16 + it * 8, a.n_rows + a.tail_rows - 1);
          ln_part[mi][it] = *reinterpret_cast<const f32x2*>(a.stats_in + ((int64_t)m * 8 + cp) * 2);
        }
    }
  }
  const int nc = n0 + wc * 64 + cp * 8;
  const int mrow0 = m0 + wr * (16 * MI) + (lane >> 3);
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float lc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr ((EPI & EPI_LNFOLD) != 0) {  // LayerNorm fold: c = row sums of the gamma-scaled weight, d = W beta + bias (fp32)
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.ln_c + nc), c1 = *reinterpret_cast<const f32x4*>(a.ln_c + nc + 4);
    const f32x4 d0 = *reinterpret_cast<const f32x4*>(a.ln_d + nc), d1 = *reinterpret_cast<const f32x4*>(a.ln_d + nc + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lc[r] = c0[r];
      lc[4 + r] = c1[r];
      bv[r] = d0[r];
      bv[4 + r] = d1[r];
    }
  } else if (a.bias != nullptr) {
    const V8 braw = *reinterpret_cast<const V8*>((const T*)a.bias + nc);
#pragma unroll
    for (int r = 0; r < 8; ++r) bv[r] = to_float(braw.v[r]);
  }
  T* __restrict__ ylane = (T*)a.y + (int64_t)mrow0 * a.ldy + nc;
  const T* __restrict__ rlane = (EPI & EPI_RES) ? (const T*)a.residual + (int64_t)mrow0 * a.ldr + nc : nullptr;
  // LDS addresses: write = MFMA layout (row lane & 15, 16-byte slot ni * 4 + lane >> 4, 32-byte pairs XOR-swizzled by the row),
  // read-back = 8 consecutive columns per lane, 8 lanes per row
  const int wrow = lane & 15;
  unsigned char* wbase = epi + wrow * 256;
  int wphys[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int slot = ni * 4 + (lane >> 4);
    wphys[ni] = ((((slot >> 1) ^ (wrow & 7)) << 1) | (slot & 1)) * 16;
  }
  const unsigned char* rsrc[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = it * 8 + (lane >> 3);
    rsrc[it] = epi + row * 256 + ((cp ^ (row & 7)) << 5);
  }
  // residual / gather rows of band mi+1 are requested BEFORE the stores of band mi: vmcnt retires in issue order (stores
  // included), so a load issued after a store cannot be waited for without waiting for that store's write acknowledgement too
  // (two bands of gathered rows in flight instead of one: measured slower, 8.07 against 7.86 ms for the GNN forward)
  V8 rv[2], t1[2], t2[2], rv_n[2], t1_n[2], t2_n[2];
  auto fetch = [&](int mi, V8 (&r)[2], V8 (&g1)[2], V8 (&g2)[2]) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      r[it] = V8{};
      g1[it] = V8{};
      g2[it] = V8{};
      if constexpr ((EPI & EPI_RES) != 0) r[it] = *reinterpret_cast<const V8*>(rlane + (int64_t)(mi * 16 + it * 8) * a.ldr);
      if constexpr ((EPI & EPI_GATHER) != 0) {
        const int m = mrow0 + mi * 16 + it * 8;
        const int r1 = gidx != nullptr ? gidx->i1[mi][it] : a.idx1[m];
        g1[it] = *reinterpret_cast<const V8*>((const T*)a.g1 + (int64_t)r1 * a.ldg1 + nc);
        if (a.g2 != nullptr) {
          const int r2 = gidx != nullptr ? gidx->i2[mi][it] : a.idx2[m];
          g2[it] = *reinterpret_cast<const V8*>((const T*)a.g2 + (int64_t)r2 * a.ldg2 + nc);
        }
      }
    }
  };
  if constexpr ((EPI & (EPI_RES | EPI_GATHER)) != 0) fetch(0, rv_n, t1_n, t2_n);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      rv[it] = rv_n[it];
      t1[it] = t1_n[it];
      t2[it] = t2_n[it];
    }
    if constexpr ((EPI & (EPI_RES | EPI_GATHER)) != 0) {
      if (mi + 1 < MI) fetch(mi + 1, rv_n, t1_n, t2_n);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(wbase + wphys[ni]) = acc[mi][ni];
    asm volatile("" ::: "memory");  // compiler barrier only: the LDS executes a wave's instructions in order
    f32x4 c[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      c[it][0] = *reinterpret_cast<const f32x4*>(rsrc[it]);
      c[it][1] = *reinterpret_cast<const f32x4*>(rsrc[it] + 16);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float vv[8];
      if constexpr ((EPI & EPI_LNFOLD) != 0) {  // mean / rstd of the tile's rows sit in LDS (written at the start of the tile)
        f32x2 mr;
        if constexpr (LDS_STATS) {
          mr = *reinterpret_cast<const f32x2*>(ln_rows + 2 * (wr * (16 * MI) + (lane >> 3) + mi * 16 + it * 8));
        } else {
          const int m = m0 + wr * (16 * MI) + (lane >> 3) + mi * 16 + it * 8;
          float mu_, rs_;
          if (ln_strip_per_lane && m < a.ln_tail_begin) {  // (the 8 lanes of a row agree on both conditions)
            const float s1 = group_sum<8>(ln_part[mi][it][0]), s2 = group_sum<8>(ln_part[mi][it][1]);
            const float inv = 1.0f / (float)a.ln_D;
            mu_ = s1 * inv;
            rs_ = rsqrtf(fmaxf(s2 * inv - mu_ * mu_, 0.f) + a.ln_eps);
          } else {
            ln_row_stats<T>(a, m, cp, mu_, rs_);
          }
          mr = f32x2{mu_, rs_};
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = fmaf(mr[1], fmaf(-mr[0], lc[r], c[it][r >> 2][r & 3]), bv[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = c[it][r >> 2][r & 3] + bv[r];
      }
      if constexpr ((EPI & EPI_GATHER) != 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] += to_float(t1[it].v[r]) + to_float(t2[it].v[r]);
      }
      if constexpr ((EPI & EPI_GELU) != 0) {
#pragma unroll
        for (int r = 0; r < 8; r += 2) gelu_fast2(vv[r], vv[r + 1]);
      }
      if constexpr ((EPI & EPI_RES) != 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] += to_float(rv[it].v[r]);
      }
      V8 o8;
#pragma unroll
      for (int r = 0; r < 8; ++r) o8.v[r] = from_float<T>(vv[r]);
      if constexpr ((EPI & EPI_STATS) != 0) {
        // sums of what is actually stored (rounded), over this wave's 64-column strip of the row: DPP butterfly over the 8
        // lanes of the row, one plain store per (row, strip) - no atomics, every slot written, deterministic
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float t = to_float(o8.v[r]);
          s1 += t;
          s2 = fmaf(t, t, s2);
        }
        s1 = group_sum<8>(s1);
        s2 = group_sum<8>(s2);
        if (cp == 0) {
          const int m = mrow0 + mi * 16 + it * 8;
          *reinterpret_cast<f32x2*>(a.stats_out + ((int64_t)m * (a.O >> 6) + ((n0 + wc * 64) >> 6)) * 2) = f32x2{s1, s2};
        }
      }
      *reinterpret_cast<V8*>(ylane + (int64_t)(mi * 16 + it * 8) * a.ldy) = o8;
    }
  }
}

template <typename T, int EPI, int MI = 4, bool LDS_STATS = false>
__device__ __forceinline__ void mfma_epilogue_band(const LinArgs& a, f32x4 (&acc)[MI][4], int m0, int n0, int wr, int wc,
                                                   int lane, unsigned char* epi, bool interior, const float* ln_rows = nullptr,
                                                   const GatherIdx<MI>* gidx = nullptr) {
  if constexpr ((EPI & ~(EPI_RES | EPI_GATHER | EPI_GELU | EPI_STATS | EPI_LNFOLD)) == 0) {
    if (interior && !a.f32_atomic && a.fast_epi) {  // wave-uniform: one branch per tile
      mfma_epilogue_fast<T, EPI, MI, LDS_STATS>(a, acc, m0, n0, wr, wc, lane, epi, ln_rows, gidx);
      return;
    }
  }
  const T* __restrict__ bias = (const T*)a.bias;
  T* __restrict__ y = (T*)a.y;
  using V8 = Vec<T, 8>;
  const int cp = lane & 7;                           // which 8-column group (32-byte fp32 pair) of the 64 columns
  const int nc = n0 + wc * 64 + cp * 8;
  // O % 4 == 0 only: the second half of an 8-column group may fall outside; handled by narrowing to 4 columns
  const int ncols = interior ? 8 : max(0, min(8, a.O - nc));
  const int mrow0 = m0 + wr * (16 * MI) + (lane >> 3);  // + mi*16 + it*8
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  using V4 = Vec<T, 4>;
  auto load8 = [&](const T* p, V8& dst) {
    if (ncols == 8) {
      dst = *reinterpret_cast<const V8*>(p);
    } else if (ncols >= 4) {  // O % 8 == 4: only the first 4 columns of the last group exist
      const V4 h = *reinterpret_cast<const V4*>(p);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst.v[r] = h.v[r];
    }
  };
  if (bias != nullptr) {
    V8 braw{};
    load8(bias + nc, braw);
#pragma unroll
    for (int r = 0; r < 8; ++r) bv[r] = to_float(braw.v[r]);
  }
  float lc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr ((EPI & EPI_LNFOLD) != 0) {  // O % 8 == 0 on this path: c and d (fp32) for this lane's 8 columns
    if (ncols == 8) {
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.ln_c + nc), c1 = *reinterpret_cast<const f32x4*>(a.ln_c + nc + 4);
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(a.ln_d + nc), d1 = *reinterpret_cast<const f32x4*>(a.ln_d + nc + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lc[r] = c0[r];
        lc[4 + r] = c1[r];
        bv[r] = d0[r];  // d already contains the bias
        bv[4 + r] = d1[r];
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    V8 rv[2], t1[2], t2[2];
    float ln_mu[2] = {0.f, 0.f}, ln_rs[2] = {1.f, 1.f};
    if constexpr ((EPI & EPI_LNFOLD) != 0) {  // mean / rstd of this tile's rows were put into LDS at the start of the tile
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int rl = wr * (16 * MI) + (lane >> 3) + mi * 16 + it * 8;
        if (ln_rows != nullptr) {
          ln_mu[it] = ln_rows[2 * rl];
          ln_rs[it] = ln_rows[2 * rl + 1];
        } else {
          ln_row_stats<T>(a, m0 + rl, cp, ln_mu[it], ln_rs[it]);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int m = mrow0 + mi * 16 + it * 8;
      const bool ok = interior || m < a.n_rows;
      rv[it] = V8{};
      t1[it] = V8{};
      t2[it] = V8{};
      if constexpr ((EPI & EPI_RES) != 0) {
        if (ok) load8((const T*)a.residual + (int64_t)m * a.ldr + nc, rv[it]);
      }
      if constexpr ((EPI & EPI_GATHER) != 0) {
        if (ok) {
          load8((const T*)a.g1 + (int64_t)a.idx1[m] * a.ldg1 + nc, t1[it]);
          if (a.g2 != nullptr) load8((const T*)a.g2 + (int64_t)a.idx2[m] * a.ldg2 + nc, t2[it]);
        }
      }
    }
    {
      const int row = lane & 15;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int slot = ni * 4 + (lane >> 4);                       // logical 16-byte slot (4 columns)
        const int phys = ((((slot >> 1) ^ (row & 7)) << 1) | (slot & 1));
        *reinterpret_cast<f32x4*>(epi + row * 256 + phys * 16) = acc[mi][ni];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 c[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 8 + (lane >> 3);
      const unsigned char* src = epi + row * 256 + ((cp ^ (row & 7)) << 5);
      c[it][0] = *reinterpret_cast<const f32x4*>(src);
      c[it][1] = *reinterpret_cast<const f32x4*>(src + 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // band consumed: the next band may overwrite the slice
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int m = mrow0 + mi * 16 + it * 8;
      const bool ok = (interior || m < a.n_rows) && ncols > 0;
      float vv[8];
      if constexpr ((EPI & EPI_LNFOLD) != 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = fmaf(ln_rs[it], fmaf(-ln_mu[it], lc[r], c[it][r >> 2][r & 3]), bv[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = c[it][r >> 2][r & 3] + bv[r];
      }
      if constexpr ((EPI & EPI_GATHER) != 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] += to_float(t1[it].v[r]) + to_float(t2[it].v[r]);
      }
      if constexpr ((EPI & EPI_GELU) != 0) {
        if constexpr ((EPI & EPI_PRE) != 0) if (ok) {  // training variants only: the inference kernels are untouched
          V8 p8;
#pragma unroll
          for (int r = 0; r < 8; ++r) p8.v[r] = from_float<T>(vv[r]);
          T* dp = (T*)a.y_pre + (int64_t)m * a.ldy_pre + nc;
          if (ncols == 8) {
            *reinterpret_cast<V8*>(dp) = p8;
          } else {
            V4 p4;
#pragma unroll
            for (int r = 0; r < 4; ++r) p4.v[r] = p8.v[r];
            *reinterpret_cast<V4*>(dp) = p4;
          }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = gelu_fast(vv[r]);
      }
      if constexpr ((EPI & EPI_RES) != 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] += to_float(rv[it].v[r]);
      }
      if (ok && a.f32_atomic) {  // split-K partial: fp32 accumulation in place (no bias / activation / residual here)
        float* dst32 = (float*)a.y + (int64_t)m * a.ldy + nc;
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (r < ncols) unsafeAtomicAdd(dst32 + r, vv[r]);
      } else if (ok) {
        T* dst = y + (int64_t)m * a.ldy + nc;
        V8 o8;
#pragma unroll
        for (int r = 0; r < 8; ++r) o8.v[r] = from_float<T>(vv[r]);
        if constexpr ((EPI & EPI_STATS) != 0) {
          // sums of what is actually stored (rounded), over this wave's 64-column strip of the row: the 8 lanes that
          // share the row are adjacent -> DPP butterfly, one plain store per (row, strip): no atomics, every slot written
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float t = (r < ncols) ? to_float(o8.v[r]) : 0.f;
            s1 += t;
            s2 = fmaf(t, t, s2);
          }
          s1 = group_sum<8>(s1);
          s2 = group_sum<8>(s2);
          if (cp == 0) {
            float* so = a.stats_out + ((int64_t)m * (a.O >> 6) + ((n0 + wc * 64) >> 6)) * 2;
            so[0] = s1;
            so[1] = s2;
          }
        }
        if (ncols == 8) {
          *reinterpret_cast<V8*>(dst) = o8;
        } else {
          V4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o4.v[r] = o8.v[r];
          *reinterpret_cast<V4*>(dst) = o4;
        }
      }
    }
  }
}

template <typename T, int WM, int WN, int STAGES, int EPI, bool PP>
__global__ __launch_bounds__(64 * WM * WN, 1) void linear_mfma_persistent_kernel(LinArgs a, int tiles_n, int num_tiles) {
  constexpr int NW = WM * WN;
  constexpr int TBM = 64 * WM, TBN = 64 * WN;
  // 1-KiB pieces (8 rows x 64 elements) per wave per K-tile.  When the piece count does not divide evenly over the
  // waves (3 x 2 waves: 16 W pieces over 6 waves) every wave still issues the same number of DMAs - the surplus ones
  // land in a 1-KiB dummy area after the ring - so that the counted vmcnt waits are identical for all waves.
  constexpr int kAPieces = TBM / 8, kWPieces = TBN / 8;
  constexpr int kAPW = (kAPieces + NW - 1) / NW, kWPW = (kWPieces + NW - 1) / NW;
  constexpr int kPPW = kAPW + kWPW;
  constexpr int kStageBytes = (TBM + TBN) * BK * 2;
  static_assert(NW * 4096 <= kStageBytes, "epilogue band staging must fit in one stage");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [STAGES][A tile | W tile] [1 KiB dummy]
  unsigned char* const dummy = smem + STAGES * kStageBytes;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int G = gridDim.x;  // multiple of 8 (or == num_tiles): all tiles of a workgroup map to its XCD's id range

  const char* __restrict__ xb = (const char*)a.x;
  const char* __restrict__ x2b = (const char*)a.x2;
  const char* __restrict__ wb = (const char*)a.w;
  const int K = a.K1 + a.K2;
  // split-K: ``num_tiles`` counts (tile, split) pairs; a pair is an ordinary tile of a GEMM over K / splits whose operand
  // origins are shifted by the split's K offset (it rides on the per-tile byte offsets of the DMA addressing)
  const int nk = K / BK / a.splits;
  const int my_tiles = (num_tiles - (int)blockIdx.x + G - 1) / G;
  const int total_g = my_tiles * nk;

  auto tile_origin = [&](int j, int& m0, int& n0) -> int {
    int id = blockIdx.x + j * G;
    const int q = num_tiles >> 3, r = num_tiles & 7, xcd = id & 7, pos = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    const int split = id % a.splits;
    id /= a.splits;
    m0 = (id / tiles_n) * TBM;
    n0 = (id % tiles_n) * TBN;
    return split;
  };

  // ---- DMA issue side (runs STAGES-1 K-tiles ahead of the MFMA side, across tile boundaries).  Addresses are a
  // wave-uniform 64-bit base (operand + K offset) plus a per-lane 32-bit byte offset fixed for the tile.
  uint32_t a_voff[kAPW], a2_voff[kAPW], w_voff[kWPW];
  int64_t x_toff = 0, x2_toff = 0, w_toff = 0;  // byte offset of the tile being issued (wave-uniform, 64-bit)
  auto setup_issue_tile = [&](int j) {
    int m0, n0;
    const int64_t k_off = (int64_t)tile_origin(j, m0, n0) * nk * BK * 2;
    x_toff = (int64_t)m0 * a.ldx * 2 + k_off;
    x2_toff = (int64_t)m0 * a.ldx2 * 2;
    w_toff = (int64_t)n0 * a.ldw * 2 + k_off;
#pragma unroll
    for (int i = 0; i < kAPW; ++i) {
      const int row = min(wave * kAPW + i, kAPieces - 1) * 8 + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      const int m = min(m0 + row, a.n_rows - 1);  // rows past the end are computed but never stored
      a_voff[i] = (uint32_t)(((int64_t)(m - m0) * a.ldx + slot * 8) * 2);  // relative to the tile's first row: fits 32 bits
      a2_voff[i] = (uint32_t)(((int64_t)(m - m0) * a.ldx2 + slot * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < kWPW; ++i) {
      const int row = min(wave * kWPW + i, kWPieces - 1) * 8 + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      const int n = min(n0 + row, a.O - 1);
      w_voff[i] = (uint32_t)(((int64_t)(n - n0) * a.ldw + slot * 8) * 2);
    }
  };
  int ig = 0, ikt = 0, ij = 0;  // next K-tile to issue: global index, index within its tile, tile
  auto issue_next = [&]() {
    if (ig >= total_g) return;
    if (ikt == 0) setup_issue_tile(ij);
    const int k0 = ikt * BK;
    unsigned char* stage = smem + (ig % STAGES) * kStageBytes;
    const bool first = k0 < a.K1;  // uniform
    const char* abase = first ? xb + x_toff + (int64_t)k0 * 2 : x2b + x2_toff + (int64_t)(k0 - a.K1) * 2;
    const char* wbase = wb + w_toff + (int64_t)k0 * 2;
#pragma unroll
    for (int i = 0; i < kAPW; ++i) {
      const int pc = wave * kAPW + i;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(abase + (first ? a_voff[i] : a2_voff[i])),
                                       (lds_void_t*)(pc < kAPieces ? stage + pc * 1024 : dummy), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < kWPW; ++i) {
      const int pc = wave * kWPW + i;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(wbase + w_voff[i]),
                                       (lds_void_t*)(pc < kWPieces ? stage + TBM * BK * 2 + pc * 1024 : dummy), 16, 0, 0);
    }
    ++ig;
    if (++ikt == nk) {
      ikt = 0;
      ++ij;
    }
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue_next();

  // fragment read offsets inside a stage: row = w*64 + i*16 + (lane&15) -> the swizzle term does not depend on i
  const int frow = lane & 15, fslot = lane >> 4;
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_rd[ks] = lds_off(wr * 64 + frow, fslot + 4 * ks);
    w_rd[ks] = TBM * BK * 2 + lds_off(wc * 64 + frow, fslot + 4 * ks);
  }

  int g = 0;
  bool counted_stores = false, drain_all = false;
  frag8 fa[2][4], fw[2][4];
  f32x4 acc[4][4];
  auto read_frags = [&](const unsigned char* st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[ks][i] = *reinterpret_cast<const frag8*>(st + a_rd[ks] + i * 16 * BK * 2);
        fw[ks][i] = *reinterpret_cast<const frag8*>(st + w_rd[ks] + i * 16 * BK * 2);
      }
  };
  auto mfma_half = [&](int ks) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<T>(fw[ks][ni], fa[ks][mi], acc[mi][ni]);  // D^T tile: rows n, cols m
  };
  // MODE 0: all waves in lock-step (read, multiply).  MODE 1 / 2: the two halves of a ping-pong pair - the two waves of
  // a SIMD run half a K-step apart: while the MODE-1 wave reads the fragments of K-tile g from LDS, the MODE-2 wave
  // multiplies K-tile g-1, then they swap, so the matrix pipe of every SIMD always has a wave feeding it instead of
  // idling while all 8 waves read fragments.  Each mode is a separate instantiation (clean loops, no per-step branch).
  auto run_tiles = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    for (int j = 0; j < my_tiles; ++j) {
      int m0, n0;
      tile_origin(j, m0, n0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

      for (int kt = 0; kt < nk; ++kt, ++g) {
        // K-tile g must have landed; the STAGES-2 newer K-tiles (and, right after an interior epilogue, its stores)
        // may stay in flight
        if (drain_all || g + STAGES - 2 >= total_g) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          drain_all = false;
          counted_stores = false;
        } else if (counted_stores && kt < STAGES - 1) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * kPPW + kEpiStores) : "memory");
        } else {
          counted_stores = false;
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * kPPW) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char* st = smem + (g % STAGES) * kStageBytes;
        if constexpr (MODE == 0) {
          // both K-halves' fragments are requested up front (32 VGPRs); the DMA refill of the stage read in the
          // previous K-step is issued between the two MFMA groups so that its address arithmetic overlaps matrix work
          read_frags(st);
          mfma_half(0);
          issue_next();
          mfma_half(1);
        } else if constexpr (MODE == 1) {
          read_frags(st);
          issue_next();
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          mfma_half(0);
          mfma_half(1);
        } else {
          if (kt > 0) {
            mfma_half(0);
            mfma_half(1);
          }
          issue_next();
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" ::: "memory");
          read_frags(st);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this stage is refilled after the next barrier
        }
      }
      if constexpr (MODE == 2) {
        mfma_half(0);
        mfma_half(1);
      }
      // epilogue in the stage that was just read (stage (g-1) % STAGES): every wave must be done reading it
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool interior = (m0 + TBM <= a.n_rows) && (n0 + TBN <= a.O);
      mfma_epilogue_band<T, EPI>(a, acc, m0, n0, wr, wc, lane, smem + ((g - 1) % STAGES) * kStageBytes + wave * 4096, interior);
      // An interior tile issues exactly kEpiStores stores per wave, and every load of its epilogue has been consumed
      // (waited for, with everything older) before the last store was issued.  Edge tiles predicate their stores.
      if (interior && nk >= STAGES && !a.f32_atomic && (EPI & (EPI_STATS | EPI_PRE)) == 0)  // statistics / pre-activation add stores: drain instead of counting
        counted_stores = true;
      else
        drain_all = true;
    }
  };
  if constexpr (!PP) {
    run_tiles(std::integral_constant<int, 0>{});
  } else {
    if (wave < NW / 2)  // waves w and w + NW/2 share a SIMD
      run_tiles(std::integral_constant<int, 1>{});
    else
      run_tiles(std::integral_constant<int, 2>{});
  }
}

// ---------------------------------------------------------------------------------------------- tail rows
// A handful of rows (the 2 rows by which an icosphere's 10 * 4^r + 2 nodes exceed a multiple of the big tile): one wave
// per output column, lanes split K in 16-byte chunks, fp32 dot product + butterfly, lane 0 applies the epilogue.
// Called at the end of the big-tile kernel by every wave of the grid (column = global wave index, stride = waves in the
// grid): 2 rows x 2048 columns are one column per wave, a few hundred cycles hidden behind the draining stores.
template <typename T, bool PRE = false>
__device__ __forceinline__ void tail_rows_valu(const LinArgs& a, int m_begin, int m_end, int n_first, int n_stride, int lane) {
  for (int n = n_first; n < a.O; n += n_stride) {
    const T* __restrict__ w = (const T*)a.w + (int64_t)n * a.ldw;
    for (int m = m_begin; m < m_end; ++m) {
      float acc = 0.f;
      const T* __restrict__ xr = (const T*)a.x + (int64_t)m * a.ldx;
      for (int k = lane * 8; k < a.K1; k += 512) {
        float xv[8], wv[8];
        load_vec<T, 8>(xr + k, xv);
        load_vec<T, 8>(w + k, wv);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = fmaf(xv[i], wv[i], acc);
      }
      if (a.K2 > 0) {
        const T* __restrict__ x2r = (const T*)a.x2 + (int64_t)m * a.ldx2;
        for (int k = lane * 8; k < a.K2; k += 512) {
          float xv[8], wv[8];
          load_vec<T, 8>(x2r + k, xv);
          load_vec<T, 8>(w + a.K1 + k, wv);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc = fmaf(xv[i], wv[i], acc);
        }
      }
      acc = wave_sum(acc);
      float ln_mu = 0.f, ln_rs = 1.f;
      if (a.ln_c) {  // LayerNorm fold: the statistics of a tail row are taken from the row itself (the producer's tail rows,
        // computed a column per wave as here, leave no strip sums)
        float s1 = 0.f, s2 = 0.f;
        for (int k = lane * 8; k < a.K1; k += 512) {
          float xv[8];
          load_vec<T, 8>(xr + k, xv);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            s1 += xv[i];
            s2 = fmaf(xv[i], xv[i], s2);
          }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        ln_mu = s1 / (float)a.ln_D;
        ln_rs = rsqrtf(fmaxf(s2 / (float)a.ln_D - ln_mu * ln_mu, 0.f) + a.ln_eps);
      }
      if (lane == 0) {
        float vv = acc;
        if (a.ln_c) vv = ln_rs * (vv - ln_mu * a.ln_c[n]) + a.ln_d[n];  // LayerNorm fold, as in the MFMA epilogue
        if (a.bias) vv += to_float(((const T*)a.bias)[n]);
        if (a.g1) vv += to_float(((const T*)a.g1)[(int64_t)a.idx1[m] * a.ldg1 + n]);
        if (a.g2) vv += to_float(((const T*)a.g2)[(int64_t)a.idx2[m] * a.ldg2 + n]);
        if constexpr (PRE) ((T*)a.y_pre)[(int64_t)m * a.ldy_pre + n] = from_float<T>(vv);  // pre-activation for the backward
        if (a.act == ANEMOI_ACT_GELU) vv = gelu_fast(vv);  // same formula as the rows of the MFMA epilogue
        if (a.residual) vv += to_float(((const T*)a.residual)[(int64_t)m * a.ldr + n]);
        ((T*)a.y)[(int64_t)m * a.ldy + n] = from_float<T>(vv);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- small M: K split over waves
// Few output tiles (one rank's rows of a sharded mesh, small meshes): a lone 64 x 128 tile worked by two waves walks its
// K-loop at ~0.6 us per 64-wide step (DMA issue, fragment reads and 32 MFMAs serialised in each wave), so a K = 2048 GEMM on
// 1.3 k rows takes 23 us on a mostly idle chip.  Here the tile gets 8 waves = 4 K-groups x 2 column halves: a stage holds 128
// K elements, group kg multiplies the 32-wide block kg of every stage (16 MFMAs per wave and stage, 6 DMA pieces per wave),
// and the four partial accumulators are added through LDS in group order (deterministic) before the usual epilogue.
//   LDS stage = [64 + 128 rows][256 B], 16-byte slots XOR-swizzled with (row & 15): conflict-free fragment ds_read_b128 (16
//   lanes = 16 rows of one slot column) while a DMA piece stays lane-linear (4 rows x 16 slots, the lane fetches the slot that
//   belongs at its position).
constexpr int SK = 128, SM = 64, SN = 128, kSStages = 3;


// Generalised: (16 MI WR) x 128 output tile, 8 waves = KG K-groups x WR row groups x 2 column halves, every wave an (16 MI) x 64
// sub-tile over its 128 / KG slice of each 128-wide K stage; STAGES-deep DMA ring of [(16 MI WR) + 128 rows][256 B].
//   <MI 4, WR 1, KG 4, 3 stages>  64 x 128: few tiles (small M), see above.
//   <MI 5, WR 2, KG 2, 2 stages> 160 x 128: the NARROW outputs of the hot path (projection and MLP-2: O = 512).  On the 256 x 128
//       / 192 x 128 ring kernels these walk a 64-wide K-step in ~1.0 us (40-48 KiB of operands, 2 barriers) - a quarter of the
//       rate the LDS-DMA path reaches (tools/dma_rate_probe.hip: 72 KiB in 0.7 us) - because every step pays the fixed price of
//       its waits and barriers.  128-wide stages halve the number of steps, and [10240 x K] -> 512 is exactly 64 x 4 = 256 such
//       tiles = one per CU (the 2 rows beyond 64 x 160 of an icosphere mesh ride on the VALU, as in the big-tile kernel).
template <typename T, int EPI, int MI, int WR, int KG, int STAGES, int SKW = SK>
__global__ __launch_bounds__(512, 1) void linear_mfma_splitwave_kernel(LinArgs a, int tiles_n, int num_tiles) {
  static_assert(KG * WR * 2 == 8, "8 waves");
  static_assert(SKW == 128 || SKW == 64, "stage width");
  constexpr int TM = 16 * MI * WR;                 // tile rows
  constexpr int kRow = SKW * 2;                    // bytes of a stage row: 256 (16 slots of 16 B) or 128 (8 slots)
  constexpr int kRPP = 1024 / kRow, kLPR = kRow / 16;  // rows per 1-KiB DMA piece, lanes per row
  constexpr int kA = TM * kRow, kStage = (TM + SN) * kRow;
  constexpr int kAP = TM / kRPP, kPieces = (TM + SN) / kRPP;
  constexpr int kPPW = (kPieces + 7) / 8;          // per wave; surplus pieces (uneven division) go to the dummy KiB
  constexpr int kKB = (SKW / 32) / KG;             // 32-wide k-blocks of a stage per K-group
  static_assert(kKB >= 1, "K-groups");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wr = (wave >> 1) % WR, kg = (wave >> 1) / WR;
  int id = blockIdx.x;
  {  // XCD-aware bijective remap: the column tiles of a row panel run on one XCD (the A panel is fetched into one L2)
    const int q = num_tiles >> 3, r = num_tiles & 7, xcd = id & 7, pos = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int m0 = (id / tiles_n) * TM, n0 = (id % tiles_n) * SN;
  const int nk = a.K1 / SKW;
  const uint32_t smem_l = (uint32_t)(size_t)(lds_void_t*)smem;
  // 16-byte slots of a row are XOR-swizzled so that the fragment ds_read_b128 (16 rows, one logical slot) are conflict-free:
  // 16 slots per row: slot ^ (row & 15); 8 slots per row (two rows per 256-byte bank line): slot ^ ((row >> 1) & 7)
  auto swz = [](int row) { return SKW == 128 ? (row & 15) : ((row >> 1) & 7); };

  // DMA: TM / kRPP A pieces + 128 / kRPP W pieces (kRPP rows x kRow bytes) per stage, kPPW per wave
  const int pr = lane / kLPR, pos = lane % kLPR;
  const char* src_row[kPPW];
  uint32_t dst_off[kPPW];
  bool real[kPPW];
#pragma unroll
  for (int j = 0; j < kPPW; ++j) {
    const int p0 = wave * kPPW + j;
    real[j] = p0 < kPieces;
    const int p = real[j] ? p0 : kPieces - 1;
    const bool is_a = p < kAP;
    const int row_t = (is_a ? p : p - kAP) * kRPP + pr;
    const int slot = pos ^ swz(row_t);
    const int64_t row_g = is_a ? min(m0 + row_t, a.n_rows - 1) : min(n0 + row_t, a.O - 1);  // clamped rows are never stored
    src_row[j] = (is_a ? (const char*)a.x + row_g * a.ldx * 2 : (const char*)a.w + row_g * a.ldw * 2) + slot * 16;
    dst_off[j] = (is_a ? 0 : kA) + (is_a ? p : p - kAP) * 1024;
  }
  // The pieces of a stage are issued ONE BY ONE between the MFMA groups of the step that runs STAGES-1 steps earlier: a
  // burst of kPPW global_load_lds right behind the barrier costs the wave ~100 cycles per piece during which it feeds no
  // MFMA - as long as the step's whole matrix work (measured: 1.7 us per 128-wide step with the burst).  Past the last
  // stage the pieces re-fetch into a dummy KiB so that every step issues the same number (uniform counted waits).
  const uint32_t dummy_l = smem_l + STAGES * kStage;
  auto piece = [&](int j, int kt_src, uint32_t base, bool valid) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(src_row[j] + (int64_t)kt_src * kRow), (lds_void_t*)(size_t)((valid && real[j]) ? base + dst_off[j] : dummy_l), 16, 0, 0);
  };
#pragma unroll
  for (int p = 0; p < STAGES - 1; ++p) {
    const bool valid = p < nk;
#pragma unroll
    for (int j = 0; j < kPPW; ++j) piece(j, valid ? p : 0, smem_l + (p % STAGES) * kStage, valid);
  }
  // the tail rows (a few rows beyond a multiple of the tile height) ride in the shadow of the first stages' flight time
  if (a.tail_rows > 0) tail_rows_valu<T, false>(a, a.n_rows, a.n_rows + a.tail_rows, (int)blockIdx.x * 8 + wave, (int)gridDim.x * 8, lane);

  // fragments: row (lane & 15) of every 16-row block (the swizzle term is the same for all of them), logical slot
  // (kg kKB + kb) 4 + (lane >> 4)
  const int frow = lane & 15;
  int a_rd[kKB], w_rd[kKB];
#pragma unroll
  for (int kb = 0; kb < kKB; ++kb) {
    const int fphys = (((kg * kKB + kb) * 4 + (lane >> 4)) ^ swz(frow)) << 4;
    a_rd[kb] = (wr * 16 * MI + frow) * kRow + fphys;
    w_rd[kb] = kA + (wc * 64 + frow) * kRow + fphys;
  }

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int kGroups = kKB * MI;                          // MFMA groups (4 MFMAs each) per wave and step
  constexpr int kPerGroup = (kPPW + kGroups - 1) / kGroups;  // pieces issued behind each group
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * kPPW) : "memory");  // stage kt landed; STAGES-2 newer ones may fly
    __builtin_amdgcn_s_barrier();  // stage kt complete for every wave; stage kt-1 no longer read by anybody
    asm volatile("" ::: "memory");
    const int ktn = kt + STAGES - 1;
    const bool valid = ktn < nk;
    const uint32_t nbase = smem_l + (ktn % STAGES) * kStage;
    const int kt_src = valid ? ktn : kt;
    const unsigned char* st = smem + (kt % STAGES) * kStage;
#pragma unroll
    for (int kb = 0; kb < kKB; ++kb) {
      frag8 fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const frag8*>(st + w_rd[kb] + i * 16 * kRow);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const frag8 fa = *reinterpret_cast<const frag8*>(st + a_rd[kb] + mi * 16 * kRow);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<T>(fw[ni], fa, acc[mi][ni]);
#pragma unroll
        for (int q = 0; q < kPerGroup; ++q) {
          const int j = (kb * MI + mi) * kPerGroup + q;
          if (j < kPPW) piece(j, kt_src, nbase, valid);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing dummy pieces

  // add the K-groups in group order (deterministic): MI KiB x 4 of LDS per non-leading wave, 4 KiB epilogue bands behind them
  constexpr int kAccBytes = MI * 4 * 1024;
  __syncthreads();
  if (kg > 0) {
    f32x4* dst = reinterpret_cast<f32x4*>(smem + (wave - 2 * WR) * kAccBytes) + lane;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[(i * 4 + j) * 64] = acc[i][j];
  }
  __syncthreads();
  if (kg == 0) {
#pragma unroll
    for (int gsrc = 1; gsrc < KG; ++gsrc) {
      const f32x4* src = reinterpret_cast<const f32x4*>(smem + (((gsrc - 1) * WR + wr) * 2 + wc) * kAccBytes) + lane;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += src[(i * 4 + j) * 64];
    }
    const bool interior = (m0 + TM <= a.n_rows) && (n0 + SN <= a.O);
    mfma_epilogue_band<T, EPI, MI>(a, acc, m0, n0, wr, wc, lane, smem + (KG - 1) * WR * 2 * kAccBytes + (wr * 2 + wc) * 4096, interior);
  }
}


// ---------------------------------------------------------------------------------------------- big-tile kernel
// (32*MI) x 256 output tile per 8-wave workgroup, 2 (M) x 4 (N) waves of (16*MI) x 64.  Why: with every CU streaming,
// the LDS-DMA path sustains only ~20-25 B/clk/CU next to running MFMAs (12-13 TB/s over the chip; the 256 x 128 kernel
// above and the loader/consumer variant that was tried both sit on that rate), so the K-loop is bound by operand BYTES
// per flop: a 320 x 256 tile moves 72 KiB per 10.5 MFLOP K-step where 256 x 128 moves 48 KiB per 4.2 MFLOP (0.6x).
// [10240 x K] x [K -> 2048] is exactly 32 x 8 = 256 such tiles: one per CU, one round.  The ring is 2 stages of 72 KiB;
// the refill of the other stage is issued piece by piece between the MFMAs of the first K-half, so the DMA queue never
// waits for the step to end and the issue stalls hide behind matrix work.  Fragments are fetched per 16-row band right
// before use (160 accumulator VGPRs leave no room for a whole K-half of fragments).
template <typename T, int MI, int EPI>
__global__ __launch_bounds__(512, 1) void linear_mfma_bigtile_kernel(LinArgs a, int tiles_n, int num_tiles) {
  constexpr int WN = 4, NW = 8, STAGES = 2;
  constexpr int TBM = 32 * MI, TBN = 64 * WN;
  constexpr int kAPieces = TBM / 8, kWPieces = TBN / 8;
  constexpr int kAPW = (kAPieces + NW - 1) / NW, kWPW = kWPieces / NW;
  constexpr int kPPW = kAPW + kWPW;
  constexpr int kStageBytes = (TBM + TBN) * BK * 2;
  constexpr int kStores = 2 * MI;  // stores per wave of an interior epilogue
  constexpr bool kPreIdx = (EPI & EPI_GATHER) != 0 && MI <= 5;  // 4 * MI more live registers: the 320-row tile has none to spare
  static_assert(kPPW <= 2 * MI, "one DMA piece per 16-row band of a K-step");
  static_assert(kPPW + kStores <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][A tile | W tile] [1 KiB dummy]
  // LDS destinations of the DMA as 32-bit LDS addresses (no flat -> local pointer casts on the issue path)
  const uint32_t smem_l = (uint32_t)(size_t)(lds_void_t*)smem;
  const uint32_t dummy_l = smem_l + STAGES * kStageBytes;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int G = gridDim.x;
  const char* __restrict__ xb = (const char*)a.x;
  const char* __restrict__ x2b = (const char*)a.x2;
  const char* __restrict__ wb = (const char*)a.w;
  const int nk = (a.K1 + a.K2) / BK;
  const int my_tiles = (num_tiles - (int)blockIdx.x + G - 1) / G;
  const int total_g = my_tiles * nk;

  auto tile_origin = [&](int j, int& m0, int& n0) {
    int id = blockIdx.x + j * G;
    const int q = num_tiles >> 3, r = num_tiles & 7, xcd = id & 7, pos = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    m0 = (id / tiles_n) * TBM;
    n0 = (id % tiles_n) * TBN;
  };

  // ---- DMA issue side: K-tile ig (one ahead of the MFMA side), one 1-KiB piece per call
  uint32_t a_voff[kAPW], a2_voff[kAPW], w_voff[kWPW];
  int64_t x_toff = 0, x2_toff = 0, w_toff = 0;  // byte offset of the tile being issued (wave-uniform, 64-bit)
  auto setup_issue_tile = [&](int j) {
    int m0, n0;
    tile_origin(j, m0, n0);
    x_toff = (int64_t)m0 * a.ldx * 2;
    x2_toff = (int64_t)m0 * a.ldx2 * 2;
    w_toff = (int64_t)n0 * a.ldw * 2;
#pragma unroll
    for (int i = 0; i < kAPW; ++i) {
      const int row = min(wave * kAPW + i, kAPieces - 1) * 8 + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      const int m = min(m0 + row, a.n_rows - 1);  // rows past the end are computed but never stored
      a_voff[i] = (uint32_t)(((int64_t)(m - m0) * a.ldx + slot * 8) * 2);  // relative to the tile's first row: fits 32 bits
      a2_voff[i] = (uint32_t)(((int64_t)(m - m0) * a.ldx2 + slot * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < kWPW; ++i) {
      const int row = (wave * kWPW + i) * 8 + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      const int n = min(n0 + row, a.O - 1);
      w_voff[i] = (uint32_t)(((int64_t)(n - n0) * a.ldw + slot * 8) * 2);
    }
  };
  // The pieces of a K-tile are issued one by one between MFMAs, so the issue itself must be branch-free: everything
  // uniform (source bases, destination stage, "is there still a K-tile to fetch") is fixed by begin_issue() before the
  // MFMA block; past the last K-tile the pieces are re-fetched into the dummy KiB so that the vmcnt bookkeeping of all
  // K-steps stays identical.
  int ig = 0, ikt = 0, ij = 0;
  const char* abase = xb;
  const char* wbase = wb;
  uint32_t sdst = smem_l;
  bool s_valid = true, s_first = true;
  auto begin_issue = [&]() {
    s_valid = ig < total_g;
    if (s_valid) {
      if (ikt == 0) setup_issue_tile(ij);
      const int k0 = ikt * BK;
      s_first = k0 < a.K1;
      abase = s_first ? xb + x_toff + (int64_t)k0 * 2 : x2b + x2_toff + (int64_t)(k0 - a.K1) * 2;
      wbase = wb + w_toff + (int64_t)k0 * 2;
      sdst = smem_l + (ig % STAGES) * kStageBytes;
    }
  };
  auto issue_piece = [&](int i) {  // i: compile-time piece index 0 .. kPPW-1
    if (i < kAPW) {
      const int pc = wave * kAPW + i;
      const uint32_t dst = (s_valid && pc < kAPieces) ? sdst + pc * 1024 : dummy_l;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(abase + (s_first ? a_voff[i] : a2_voff[i])), (lds_void_t*)(size_t)dst, 16, 0, 0);
    } else {
      const int pc = wave * kWPW + (i - kAPW);
      const uint32_t dst = s_valid ? sdst + TBM * BK * 2 + pc * 1024 : dummy_l;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(wbase + w_voff[i - kAPW]), (lds_void_t*)(size_t)dst, 16, 0, 0);
    }
  };
  auto end_issue = [&]() {
    if (s_valid) {
      ++ig;
      if (++ikt == nk) {
        ikt = 0;
        ++ij;
      }
    }
  };
  begin_issue();
#pragma unroll
  for (int i = 0; i < kPPW; ++i) issue_piece(i);
  end_issue();
  // the tail rows ride in the shadow of the first (cold) K-tile's flight time
  if (a.tail_rows > 0) tail_rows_valu<T, (EPI & EPI_PRE) != 0>(a, a.n_rows, a.n_rows + a.tail_rows, (int)blockIdx.x * NW + wave, G * NW, lane);

  const int frow = lane & 15, fslot = lane >> 4;
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_rd[ks] = lds_off(wr * (16 * MI) + frow, fslot + 4 * ks);
    w_rd[ks] = TBM * BK * 2 + lds_off(wc * 64 + frow, fslot + 4 * ks);
  }

  int g = 0;
  bool counted_stores = false;
  for (int j = 0; j < my_tiles; ++j) {
    int m0, n0;
    tile_origin(j, m0, n0);
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* ln_rows = reinterpret_cast<float*>(smem + STAGES * kStageBytes + 1024) + (j & 1) * TBM * 2;
    GatherIdx<kPreIdx ? MI : 1> gidx;
    const GatherIdx<MI>* gidx_p = nullptr;
    if constexpr (kPreIdx) gidx_p = &gidx;
    if constexpr ((EPI & EPI_LNFOLD) != 0) {
      // mean / rstd of the tile's rows from the producer's strip sums (fixed order), one thread per row, while the first
      // K-tile is in flight; double-buffered over tiles, published by the K-loop's barriers
      for (int r = tid; r < TBM; r += 64 * NW) {
        const float* st = a.stats_in + (int64_t)min(m0 + r, a.n_rows - 1) * a.ln_strips * 2;
        float s1 = 0.f, s2 = 0.f;
        if (a.ln_strips == 8) {  // D = 512: four 16-byte loads, same summation order as the generic loop
          f32x4 v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const f32x4*>(st)[q];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            s1 += v[q][0];
            s2 += v[q][1];
            s1 += v[q][2];
            s2 += v[q][3];
          }
        } else {
          for (int q = 0; q < a.ln_strips; ++q) {
            s1 += st[2 * q];
            s2 += st[2 * q + 1];
          }
        }
        const float inv = 1.0f / (float)a.ln_D, mu = s1 * inv;
        ln_rows[2 * r] = mu;
        ln_rows[2 * r + 1] = rsqrtf(fmaxf(s2 * inv - mu * mu, 0.f) + a.ln_eps);
      }
    }

    for (int kt = 0; kt < nk; ++kt, ++g) {
      // K-tile g must have landed (it is the only DMA in flight); right after an interior epilogue its stores may stay
      if (counted_stores && kt == 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      counted_stores = false;
      __builtin_amdgcn_s_barrier();  // also: every wave has finished reading K-tile g-1, whose stage is refilled below
      asm volatile("" ::: "memory");
      if constexpr (kPreIdx) {
        // the epilogue's gather indices: requested here (after the wait above, so the counted waits stay exact; the next K-step's
        // vmcnt(0) covers them), used ~nk K-steps later
        if (kt == 0) {
          const int mr = m0 + wr * (16 * MI) + (lane >> 3);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
              const int m = min(mr + mi * 16 + it * 8, a.n_rows - 1);
              gidx.i1[mi][it] = a.idx1[m];
              gidx.i2[mi][it] = a.g2 != nullptr ? a.idx2[m] : 0;
            }
        }
      }
      const unsigned char* st = smem + (g % STAGES) * kStageBytes;
      begin_issue();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        frag8 fw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const frag8*>(st + w_rd[ks] + i * 16 * BK * 2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const frag8 fa = *reinterpret_cast<const frag8*>(st + a_rd[ks] + mi * 16 * BK * 2);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16<T>(fw[ni], fa, acc[mi][ni]);  // D^T tile: rows n, cols m
          if (ks * MI + mi < kPPW) issue_piece(ks * MI + mi);
        }
      }
      end_issue();
    }
    // epilogue in the stage that was just read: every wave must be done reading it
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool interior = (m0 + TBM <= a.n_rows) && (n0 + TBN <= a.O);
    mfma_epilogue_band<T, EPI, MI, true>(a, acc, m0, n0, wr, wc, lane, smem + ((g - 1) % STAGES) * kStageBytes + wave * 4096, interior, ln_rows,
                                         gidx_p);
    counted_stores = interior && (EPI & EPI_PRE) == 0;  // exactly kStores stores per wave, issued after the DMAs of the next K-tile
  }
}

// ---------------------------------------------------------------------------------------------- dispatch
static bool al(const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <typename T>
static bool mfma_eligible(const LinArgs& a) {
  if (sizeof(T) != 2) return false;
  if (a.K1 % 8 || a.K2 % 8 || a.O % 4) return false;
  if (a.ldx % 8 || a.ldw % 8 || (a.x2 && a.ldx2 % 8)) return false;                         // 16-byte operand rows
  if (a.ldy % 4 || (a.residual && a.ldr % 4) || (a.g1 && a.ldg1 % 4) || (a.g2 && a.ldg2 % 4)) return false;  // 8-byte epilogue
  return al(a.x, 16) && al(a.x2, 16) && al(a.w, 16) && al(a.y, 8) && al(a.residual, 8) && al(a.bias, 8) && al(a.g1, 8) && al(a.g2, 8);
}

template <typename T>
static int launch_generic(const LinArgs& a, hipStream_t st) {
  const dim3 grid((a.O + GT - 1) / GT, (a.n_rows + GT - 1) / GT);
  hipLaunchKernelGGL((linear_generic_kernel<T>), grid, dim3(256), 0, st, a);
  return check_launch("linear_generic_kernel");
}

template <typename T>
static bool ring_eligible(const LinArgs& a) {
  // K-tiles are whole, and the per-lane 32-bit byte offsets of the DMA addressing (relative to a tile's first row) cover
  // one tile of each operand (at most 320 rows)
  const int64_t lim = (int64_t)1 << 31;
  const bool k_ok = a.K1 % BK == 0 && a.K2 % BK == 0 && 320 * a.ldx * 2 < lim && 320 * a.ldw * 2 < lim &&
                    (a.x2 == nullptr || 320 * a.ldx2 * 2 < lim);
  // 16-byte epilogue accesses
  const bool e_ok = a.ldy % 8 == 0 && (!a.residual || a.ldr % 8 == 0) && (!a.g1 || a.ldg1 % 8 == 0) && (!a.g2 || a.ldg2 % 8 == 0) &&
                    al(a.y, 16) && al(a.residual, 16) && al(a.bias, 16) && al(a.g1, 16) && al(a.g2, 16);
  return k_ok && e_ok;
}

template <typename T, int EPI, int WM, bool PP, int WN = 2, int ST = 3>
static int launch_persistent_wm(const LinArgs& a, hipStream_t st) {
  // (64*WM) x (64*WN) tile, WM*WN waves, ST-stage ring (3 stages = 144 KiB at 4 x 2: one workgroup per CU)
  constexpr int smem_bytes = ST * (64 * WM + 64 * WN) * BK * 2 + 1024;
  static PerDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_mfma_persistent_kernel<T, WM, WN, ST, EPI, PP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
  });
  const int tm = (a.n_rows + 64 * WM - 1) / (64 * WM), tn = (a.O + 64 * WN - 1) / (64 * WN);
  const int nt = tm * tn * a.splits;
  const int per_cu = (160 * 1024) / smem_bytes < 1 ? 1 : (160 * 1024) / smem_bytes;  // small tiles: several workgroups per CU
  const int cap = 256 * (per_cu > 4 ? 4 : per_cu);
  const int grid = nt < cap ? nt : cap;
  hipLaunchKernelGGL((linear_mfma_persistent_kernel<T, WM, WN, ST, EPI, PP>), dim3(grid), dim3(64 * WM * WN), smem_bytes, st, a, tn, nt);
  return check_launch("linear_mfma_persistent_kernel");
}

template <typename T, int EPI, int MI = 4, int WR = 1, int KG = 4, int STAGES = kSStages, int SKW = SK>
static int launch_splitwave(const LinArgs& a, hipStream_t st) {
  constexpr int TM = 16 * MI * WR;
  constexpr int ring = STAGES * (TM + SN) * SKW * 2 + 1024;  // + the dummy KiB of the trailing DMA pieces
  constexpr int fin = (KG - 1) * WR * 2 * MI * 4 * 1024 + WR * 2 * 4096;  // K-group sums + epilogue bands (reuse the ring)
  constexpr int smem_bytes = ring > fin ? ring : fin;
  static_assert(smem_bytes <= 160 * 1024, "LDS");
  static PerDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_mfma_splitwave_kernel<T, EPI, MI, WR, KG, STAGES, SKW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
  });
  const int tm = (a.n_rows + TM - 1) / TM, tn = (a.O + SN - 1) / SN;
  hipLaunchKernelGGL((linear_mfma_splitwave_kernel<T, EPI, MI, WR, KG, STAGES, SKW>), dim3(tm * tn), dim3(512), smem_bytes, st, a, tn, tm * tn);
  return check_launch("linear_mfma_splitwave_kernel");
}

template <typename T, int EPI, int MI>
static int launch_bigtile(const LinArgs& a, hipStream_t st) {
  constexpr int TBM = 32 * MI, TBN = 256;
  constexpr int smem_bytes = 2 * (TBM + TBN) * BK * 2 + 1024 + ((EPI & EPI_LNFOLD) ? 2 * TBM * 2 * 4 : 0);
  static PerDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_mfma_bigtile_kernel<T, MI, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
  });
  const int tm = (a.n_rows + TBM - 1) / TBM, tn = (a.O + TBN - 1) / TBN;
  const int nt = tm * tn;
  const int grid = nt < 256 ? nt : 256;
  hipLaunchKernelGGL((linear_mfma_bigtile_kernel<T, MI, EPI>), dim3(grid), dim3(512), smem_bytes, st, a, tn, nt);
  return check_launch("linear_mfma_bigtile_kernel");
}

// Estimated duration [us] of one launch on 256 CUs.  Both terms are measured rates: the K-loop moves (TBM + TBN) * 128 B
// of operands per K-step at the ~44 GB/s per CU the LDS-DMA path sustains next to running MFMAs, the epilogue writes
// TBM * TBN outputs at ~0.09 ns each; every round of tiles pays both.
static double tile_cost_us(int tbm, int tbn, int rows, int O, int nk) {
  const int64_t tiles = (int64_t)((rows + tbm - 1) / tbm) * ((O + tbn - 1) / tbn);
  const double rounds = (double)((tiles + 255) / 256);
  return rounds * (nk * (tbm + tbn) * 2.9e-3 + (double)tbm * tbn * 0.09e-3);
}

template <typename T, int EPI>
static int launch_persistent(const LinArgs& a, hipStream_t st) {
  const int nk = (a.K1 + a.K2) / BK;
  const double c4 = tile_cost_us(256, 128, a.n_rows, a.O, nk), c3 = tile_cost_us(192, 128, a.n_rows, a.O, nk);
  // 320 x 256 tiles; a few rows beyond a multiple of 320 (the icosphere's 10 * 4^r + 2 nodes) are computed on the VALU
  // at the end of the same kernel (a column per wave) instead of costing a whole extra round of tiles
  constexpr int kBigM = 320, kTail = 32;
  const int rem = a.n_rows % kBigM;
  const bool split = rem > 0 && rem <= kTail && a.n_rows > kBigM;
  const int main_rows = split ? a.n_rows - rem : a.n_rows;
  const double cb = tile_cost_us(kBigM, 256, main_rows, a.O, nk) + (split ? 0.5 : 0.0);
  static const int force_big = [] { const char* e = getenv("ANEMOI_GEMM_BIG"); return e ? atoi(e) : -1; }();
  const bool big = force_big >= 0 ? (force_big != 0 && a.O >= 64) : cb < 0.95 * (c3 < c4 ? c3 : c4);
  if (big) {
    LinArgs m = a;
    m.n_rows = main_rows;
    m.tail_rows = split ? rem : 0;
    // Up to one round of 320 x 256 tiles: every CU ends its only tile at the same moment and the whole output (42 MB at
    // [10242 x 512] -> 2048) is written behind the last K-step: ~8 us at the ~5 TB/s HBM takes writes, with nothing to overlap
    // (tools/gemm_phase_timing.py).  160 x 256 tiles instead: two per CU, the first tile's output drains under the second
    // tile's K-loop, whose rate is set by the MFMAs (power-limited clock: 1.9 us per 320-row K-step on random data against
    // 0.7 us for its DMA, tools/dma_rate_probe.hip), not by the 44 % extra operand bytes.
    static const int half_mi = [] { const char* e = getenv("ANEMOI_GEMM_BIG_MI"); return env_int(e, 5, 5, 10); }();
    const int64_t t320 = (int64_t)((main_rows + 319) / 320) * ((a.O + 255) / 256);
    // ... and up to two rounds of them (GraphConv's [81840 x 512] -> 512 edge GEMMs: 8.66 -> 8.36 ms per GNN forward); beyond
    // that the drain is hidden anyway and the 320-row tile's lower operand traffic wins (N320: 15.5 against 15.85 ms)
    static const int max_t320 = [] { const char* e = getenv("ANEMOI_GEMM_BIG_MI5_T320"); return env_int(e, 512, 0, 1 << 30); }();
    if (half_mi == 5 && t320 <= max_t320) return launch_bigtile<T, EPI, 5>(m, st);
    return launch_bigtile<T, EPI, 10>(m, st);
  }
  // narrow outputs with a long K in ONE round of 160 x 128 tiles (MLP-2 of the hidden mesh: [10242 x 2048] -> 512 = 64 x 4 tiles + 2
  // tail rows): 128-wide K stages, K split over two wave groups (linear_mfma_splitwave_kernel<.., 5, 2, 2, 2>).  Measured on
  // MI355X: 33.3 us against 34.5 on the 192 x 128 ring kernel; at K = 512 (projection) it is 1.2 us SLOWER, hence K >= 1024.
  if constexpr ((EPI & (EPI_STATS | EPI_LNFOLD | EPI_PRE)) == 0) {
    static const bool narrow = [] { const char* e = getenv("ANEMOI_GEMM_NARROW"); return !(e && e[0] == '0'); }();
    constexpr int TM = 160;
    const int rem160 = a.n_rows % TM;
    const bool split160 = rem160 > 0 && rem160 <= 32 && a.n_rows > TM;
    const int rows160 = split160 ? a.n_rows - rem160 : a.n_rows;
    const int64_t t160 = (int64_t)((rows160 + TM - 1) / TM) * ((a.O + SN - 1) / SN);
    if (narrow && t160 > 128 && t160 <= 256 && a.K2 == 0 && a.K1 >= 1024 && a.K1 % SK == 0 && a.splits == 1 && !a.f32_atomic && !big) {
      LinArgs m = a;
      m.n_rows = rows160;
      m.tail_rows = split160 ? rem160 : 0;
      static const int n64 = [] { const char* e = getenv("ANEMOI_GEMM_NARROW64"); return e ? atoi(e) : 0; }();
      if (n64) return launch_splitwave<T, EPI, 5, 2, 2, 4, 64>(m, st);  // 64-wide stages, 4-deep ring
      return launch_splitwave<T, EPI, 5, 2, 2, 2>(m, st);
    }
  }
  // few tiles (small M, e.g. one rank's rows of a sharded mesh): 64 x 128 tiles on more CUs; the K-loop of a lone tile
  // is bound by the ~40 cycles a CU needs per 1-KiB LDS-DMA piece, i.e. by the tile's operand bytes, like the model says
  const double c1 = tile_cost_us(64, 128, a.n_rows, a.O, nk);
  if (c1 < 0.9 * (c3 < c4 ? c3 : c4)) {
    if constexpr ((EPI & EPI_LNFOLD) == 0) {
      // at most one round of 64 x 128 tiles: give every tile 8 waves (K split over wave groups) instead of 2 (also with the
      // row-statistics epilogue: the projection of a sharded mesh's block)
      static const bool sw = [] { const char* e = getenv("ANEMOI_GEMM_SPLITWAVE"); return !(e && e[0] == '0'); }();
      const int64_t t1 = (int64_t)((a.n_rows + SM - 1) / SM) * ((a.O + SN - 1) / SN);
      if (sw && t1 <= 256 && a.K2 == 0 && a.K1 % SK == 0 && a.splits == 1 && !a.f32_atomic) return launch_splitwave<T, EPI>(a, st);
    }
    return launch_persistent_wm<T, EPI, 1, false, 2>(a, st);
  }
  static const bool pp = [] { const char* e = getenv("ANEMOI_GEMM_PP"); return !(e && e[0] == '0'); }();
  if (pp) {
    if (c3 < c4) return launch_persistent_wm<T, EPI, 3, true>(a, st);
    return launch_persistent_wm<T, EPI, 4, true>(a, st);
  }
  if (c3 < c4) return launch_persistent_wm<T, EPI, 3, false>(a, st);
  return launch_persistent_wm<T, EPI, 4, false>(a, st);
}

template <typename T>
static int launch_mfma(const LinArgs& a, hipStream_t st) {
  if (ring_eligible<T>(a)) {
    const int epi = (a.residual ? EPI_RES : 0) | (a.g1 ? EPI_GATHER : 0) | (a.act == ANEMOI_ACT_GELU ? EPI_GELU : 0);
    if (a.y_pre != nullptr) {  // act == GELU checked by the caller
      switch (epi) {
        case 4: return launch_persistent<T, 4 | EPI_PRE>(a, st);
        case 5: return launch_persistent<T, 5 | EPI_PRE>(a, st);
        case 6: return launch_persistent<T, 6 | EPI_PRE>(a, st);
        default: return launch_persistent<T, 7 | EPI_PRE>(a, st);
      }
    }
    switch (epi) {
      case 0: return launch_persistent<T, 0>(a, st);
      case 1: return launch_persistent<T, 1>(a, st);
      case 2: return launch_persistent<T, 2>(a, st);
      case 3: return launch_persistent<T, 3>(a, st);
      case 4: return launch_persistent<T, 4>(a, st);
      case 5: return launch_persistent<T, 5>(a, st);
      case 6: return launch_persistent<T, 6>(a, st);
      default: return launch_persistent<T, 7>(a, st);
    }
  }
  const int tiles_m = (a.n_rows + BM - 1) / BM, tiles_n = (a.O + BN - 1) / BN;
  const int num_tiles = tiles_m * tiles_n;
  static PerDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_mfma_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kTileBytes);
  });
  hipLaunchKernelGGL((linear_mfma_kernel<T>), dim3(num_tiles), dim3(256), 4 * kTileBytes, st, a, tiles_n, num_tiles);
  return check_launch("linear_mfma_kernel");
}

template <typename T, bool RES = true>
static int launch_stats_producer(const LinArgs& a, hipStream_t st) {
  // y = x W^T + b + residual, plus the row statistics of y for the LayerNorm the next GEMM folds in (O = 512-class outputs):
  // the kernel choice of launch_persistent for this shape, with the statistics epilogue
  const int nk = (a.K1 + a.K2) / BK;
  const double c4 = tile_cost_us(256, 128, a.n_rows, a.O, nk), c3 = tile_cost_us(192, 128, a.n_rows, a.O, nk);
  constexpr int EPI = (RES ? EPI_RES : 0) | EPI_STATS;  // without a residual: the embedding in front of a mapper's LayerNorm
  {
    constexpr int TM = 160;
    // ONE tail rule on both sides of the fold: the consumer (launch_lnfold_consumer) recomputes the statistics of a row
    // from the row itself only for the rows beyond a multiple of 320 (at most 32 of them) - so only those may be peeled
    // here without strip sums.  A remainder of 161..192 mod 320 (<= 32 mod 160) goes through a ragged last tile, which
    // writes the strip sums of every valid row.
    const int rem320 = a.n_rows % 320;
    const bool split160 = rem320 > 0 && rem320 <= 32 && a.n_rows > 320;
    const int rem160 = split160 ? rem320 : 0;
    const int rows160 = a.n_rows - rem160;
    const int64_t t160 = (int64_t)((rows160 + TM - 1) / TM) * ((a.O + SN - 1) / SN);
    static const bool narrow = [] { const char* e = getenv("ANEMOI_GEMM_NARROW"); return !(e && e[0] == '0'); }();
    if (narrow && t160 > 128 && t160 <= 256 && a.K2 == 0 && a.K1 >= 1024 && a.K1 % SK == 0) {
      LinArgs m = a;  // tail rows: computed a column per wave, WITHOUT strip sums (the consumer takes their statistics from the rows)
      m.n_rows = rows160;
      m.tail_rows = split160 ? rem160 : 0;
      static const int n64 = [] { const char* e = getenv("ANEMOI_GEMM_NARROW64"); return e ? atoi(e) : 0; }();
      if (n64) return launch_splitwave<T, EPI, 5, 2, 2, 4, 64>(m, st);
      return launch_splitwave<T, EPI, 5, 2, 2, 2>(m, st);
    }
  }
  (void)c3;
  (void)c4;
  return launch_persistent<T, EPI>(a, st);  // the kernel choice of the same shape without statistics (big tiles at 40 320 rows)
}

template <typename T>
static int launch_lnfold_consumer(const LinArgs& a, hipStream_t st) {
  constexpr int kBigM = 320, kTail = 32;
  const int rem = a.n_rows % kBigM;
  LinArgs m = a;
  if (rem > 0 && rem <= kTail && a.n_rows > kBigM) {  // as in launch_persistent: tail rows on the VALU, in the same kernel
    m.n_rows = a.n_rows - rem;
    m.tail_rows = rem;
  }
  const int64_t t320 = (int64_t)((m.n_rows + 319) / 320) * ((a.O + 255) / 256);
  // few rows (one rank's share of a sharded mesh, small meshes): a round of big tiles leaves most of the chip idle (642 rows =
  // 3 x 8 tiles on 256 CUs) - the 64 x 128 kernels take the fold through their epilogue, statistics read from L1
  static const int small_rows = [] { return env_int(getenv("ANEMOI_LNFOLD_SMALL_ROWS"), 4096, 0, 1 << 30); }();
  if (a.n_rows < small_rows) {
    LinArgs a2 = a;
    if (rem > 0 && rem <= kTail && a.n_rows > kBigM) a2.ln_tail_begin = a.n_rows - rem;
    const LinArgs& a = a2;
    const double c34 = [&] { const int nk = a.K1 / BK; const double c4 = tile_cost_us(256, 128, a.n_rows, a.O, nk), c3 = tile_cost_us(192, 128, a.n_rows, a.O, nk); return c3 < c4 ? c3 : c4; }();
    if (tile_cost_us(64, 128, a.n_rows, a.O, a.K1 / BK) < 0.9 * c34) {
      const int64_t t1 = (int64_t)((a.n_rows + SM - 1) / SM) * ((a.O + SN - 1) / SN);
      const bool gelu = a.act == ANEMOI_ACT_GELU;
      if (t1 <= 256 && a.K1 % SK == 0)
        return gelu ? launch_splitwave<T, EPI_LNFOLD | EPI_GELU>(a, st) : launch_splitwave<T, EPI_LNFOLD>(a, st);
      return gelu ? launch_persistent_wm<T, EPI_LNFOLD | EPI_GELU, 1, false, 2>(a, st) : launch_persistent_wm<T, EPI_LNFOLD, 1, false, 2>(a, st);
    }
    // in between (a few thousand rows: the res-4 mesh, 2 562): the 192 x 128 kernel of the plain GEMM of the shape, lock-step schedule (the ping-pong one spills 80 registers with the fold)
    const bool gelu = a.act == ANEMOI_ACT_GELU;
    return gelu ? launch_persistent_wm<T, EPI_LNFOLD | EPI_GELU, 3, false, 2>(a, st) : launch_persistent_wm<T, EPI_LNFOLD, 3, false, 2>(a, st);
  }
  // 160-row tiles also beyond one round (40 320-row mapper GEMMs): the fold's epilogue has no registers to spare at 160
  // accumulators per lane (MI = 10: +7 us on [40320 x 512] -> 1024), at 80 it is free; ANEMOI_LNFOLD_MI5=0 restores the rule
  static const int always5 = [] { const char* e = getenv("ANEMOI_LNFOLD_MI5"); return e ? atoi(e) : 1; }();
  if ((t320 <= 256 || always5) && m.n_rows % 160 == 0)  // two 160 x 256 tiles per CU (see launch_persistent)
    return a.act == ANEMOI_ACT_GELU ? launch_bigtile<T, EPI_LNFOLD | EPI_GELU, 5>(m, st) : launch_bigtile<T, EPI_LNFOLD, 5>(m, st);
  return a.act == ANEMOI_ACT_GELU ? launch_bigtile<T, EPI_LNFOLD | EPI_GELU, 10>(m, st) : launch_bigtile<T, EPI_LNFOLD, 10>(m, st);
}

template <typename T>
static int launch_splitk(const LinArgs& a, hipStream_t st) {
  // 64 x 128 tiles (many tiles from a small output), lock-step schedule, plain epilogue
  return launch_persistent_wm<T, 0, 1, false, 2>(a, st);
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_linear_splitk_f32(const void* x, int64_t ldx, const void* w, int64_t ldw, float* y, int64_t ldy, int32_t n_rows,
                                        int32_t O, int32_t K, int32_t splits, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows > 0 && O > 0 && K > 0 && splits >= 1, "linear_splitk_f32: bad sizes n_rows=%d O=%d K=%d splits=%d", n_rows, O, K, splits);
  ANEMOI_REQUIRE(x && w && y, "linear_splitk_f32: null x/w/y");
  ANEMOI_REQUIRE(K % (BK * splits) == 0, "linear_splitk_f32: K=%d must be a multiple of %d * splits", K, BK);
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "linear_splitk_f32: 16-bit operands only");
  ANEMOI_REQUIRE(ldx >= K && ldw >= K && ldy >= O && ldx % 8 == 0 && ldw % 8 == 0 && al(x, 16) && al(w, 16) && al(y, 16) && ldy % 4 == 0,
                 "linear_splitk_f32: operands must be 16-byte aligned rows (ldx, ldw multiples of 8; ldy of 4)");
  LinArgs a{x, ldx, K, nullptr, 0, 0, w, ldw, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, y, ldy, n_rows, O, (int)ANEMOI_ACT_NONE};
  a.splits = splits;
  a.f32_atomic = 1;
  hipStream_t st = as_stream(stream);
  return dtype == ANEMOI_BF16 ? launch_splitk<bf16_t>(a, st) : launch_splitk<f16_t>(a, st);
}

extern "C" int anemoi_linear_stats_fwd(const void* x, int64_t ldx, int32_t K, const void* w, int64_t ldw, const void* bias,
                                       const void* residual, int64_t ldr, void* y, int64_t ldy, float* stats_out, int32_t n_rows,
                                       int32_t O, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows > 0 && O > 0 && K > 0 && x && w && y && stats_out, "linear_stats_fwd: bad arguments");
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "linear_stats_fwd: 16-bit operands only");
  ANEMOI_REQUIRE(O % 64 == 0 && K % BK == 0, "linear_stats_fwd: O=%d and K=%d must be multiples of 64", O, K);
  LinArgs a{x, ldx, K, nullptr, 0, 0, w, ldw, bias, nullptr, 0, nullptr, nullptr, 0, nullptr, residual, ldr, y, ldy, n_rows, O, (int)ANEMOI_ACT_NONE};
  a.stats_out = stats_out;
  const bool ok = dtype == ANEMOI_BF16 ? (mfma_eligible<bf16_t>(a) && ring_eligible<bf16_t>(a)) : (mfma_eligible<f16_t>(a) && ring_eligible<f16_t>(a));
  if (!ok) {
    set_error("linear_stats_fwd: operands not eligible for the ring kernel (alignment)");
    return ANEMOI_E_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  if (residual == nullptr) return dtype == ANEMOI_BF16 ? launch_stats_producer<bf16_t, false>(a, st) : launch_stats_producer<f16_t, false>(a, st);
  return dtype == ANEMOI_BF16 ? launch_stats_producer<bf16_t>(a, st) : launch_stats_producer<f16_t>(a, st);
}

extern "C" int anemoi_linear_lnfold_fwd(const void* x, int64_t ldx, int32_t K, const void* w_scaled, int64_t ldw, const float* ln_c,
                                        const float* ln_d, const float* stats_in, int32_t strips, float eps, anemoi_act_t act, void* y,
                                        int64_t ldy, int32_t n_rows, int32_t O, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows > 0 && O > 0 && K > 0 && x && w_scaled && y && ln_c && ln_d && stats_in && strips > 0, "linear_lnfold_fwd: bad arguments");
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "linear_lnfold_fwd: 16-bit operands only");
  ANEMOI_REQUIRE(O % 8 == 0 && K % BK == 0 && strips * 64 == K, "linear_lnfold_fwd: O %% 8, K %% 64 and strips * 64 == K required (K=%d strips=%d)", K, strips);
  ANEMOI_REQUIRE(act == ANEMOI_ACT_NONE || act == ANEMOI_ACT_GELU, "linear_lnfold_fwd: unknown activation");
  LinArgs a{x, ldx, K, nullptr, 0, 0, w_scaled, ldw, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, y, ldy, n_rows, O, (int)act};
  a.stats_in = stats_in;
  a.ln_c = ln_c;
  a.ln_d = ln_d;
  a.ln_strips = strips;
  a.ln_D = K;
  a.ln_eps = eps;
  const bool ok = dtype == ANEMOI_BF16 ? (mfma_eligible<bf16_t>(a) && ring_eligible<bf16_t>(a)) : (mfma_eligible<f16_t>(a) && ring_eligible<f16_t>(a));
  if (!ok || (reinterpret_cast<uintptr_t>(ln_c) & 15) || (reinterpret_cast<uintptr_t>(ln_d) & 15)) {
    set_error("linear_lnfold_fwd: operands not eligible for the big-tile kernel (alignment)");
    return ANEMOI_E_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  const int rc = dtype == ANEMOI_BF16 ? launch_lnfold_consumer<bf16_t>(a, st) : launch_lnfold_consumer<f16_t>(a, st);
  if (rc == ANEMOI_E_UNSUPPORTED) set_error("linear_lnfold_fwd: n_rows=%d leaves tail rows (not a multiple of 320 within 32)", n_rows);
  return rc;
}

static int linear_fwd_impl(const void* x, int64_t ldx, int32_t K1, const void* x2, int64_t ldx2, int32_t K2,
                           const void* w, int64_t ldw, const void* bias, const void* g1, int64_t ldg1,
                           const int32_t* idx1, const void* g2, int64_t ldg2, const int32_t* idx2,
                           const void* residual, int64_t ldr, void* y, int64_t ldy, void* y_pre, int64_t ldy_pre, int32_t n_rows,
                           int32_t O, anemoi_act_t act, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && O > 0 && K1 > 0 && K2 >= 0, "linear_fwd: bad sizes n_rows=%d O=%d K1=%d K2=%d", n_rows, O, K1, K2);
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && w && y, "linear_fwd: null x/w/y");
  ANEMOI_REQUIRE((K2 == 0) == (x2 == nullptr), "linear_fwd: x2 and K2 must be given together");
  ANEMOI_REQUIRE(ldx >= K1 && ldw >= K1 + K2 && ldy >= O && (!x2 || ldx2 >= K2), "linear_fwd: leading dimension too small");
  ANEMOI_REQUIRE((g1 == nullptr) == (idx1 == nullptr) && (g2 == nullptr) == (idx2 == nullptr), "linear_fwd: gather term needs both table and index");
  ANEMOI_REQUIRE(act == ANEMOI_ACT_NONE || act == ANEMOI_ACT_GELU, "linear_fwd: unknown activation %d", (int)act);
  LinArgs a{x, ldx, K1, x2, ldx2, K2, w, ldw, bias, g1, ldg1, idx1, g2, ldg2, idx2, residual, ldr, y, ldy, n_rows, O, (int)act};
  static const int fast_epi = [] { const char* e = getenv("ANEMOI_GEMM_FAST_EPI"); return (e && e[0] == '0') ? 0 : 1; }();
  a.fast_epi = fast_epi;
  hipStream_t st = as_stream(stream);
  if (y_pre != nullptr) {  // only the DMA-ring kernels (shared epilogue) store the pre-activation
    a.y_pre = y_pre;
    a.ldy_pre = ldy_pre;
    const bool ok = act == ANEMOI_ACT_GELU && ldy_pre >= O && ldy_pre % 8 == 0 && al(y_pre, 16) &&
                    ((dtype == ANEMOI_BF16 && mfma_eligible<bf16_t>(a) && ring_eligible<bf16_t>(a)) ||
                     (dtype == ANEMOI_F16 && mfma_eligible<f16_t>(a) && ring_eligible<f16_t>(a)));
    if (!ok) {
      set_error("linear_fwd_pre: the pre-activation output needs act = GELU on a DMA-ring GEMM shape (16-bit, K %% 64 == 0)");
      return ANEMOI_E_UNSUPPORTED;
    }
  }
  switch (dtype) {
    case ANEMOI_F32: return launch_generic<float>(a, st);
    case ANEMOI_BF16: return mfma_eligible<bf16_t>(a) ? launch_mfma<bf16_t>(a, st) : launch_generic<bf16_t>(a, st);
    case ANEMOI_F16: return mfma_eligible<f16_t>(a) ? launch_mfma<f16_t>(a, st) : launch_generic<f16_t>(a, st);
    default: set_error("unknown dtype %d", (int)dtype); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_linear_fwd(const void* x, int64_t ldx, int32_t K1, const void* x2, int64_t ldx2, int32_t K2,
                                 const void* w, int64_t ldw, const void* bias, const void* g1, int64_t ldg1,
                                 const int32_t* idx1, const void* g2, int64_t ldg2, const int32_t* idx2,
                                 const void* residual, int64_t ldr, void* y, int64_t ldy, int32_t n_rows, int32_t O,
                                 anemoi_act_t act, anemoi_dtype_t dtype, void* stream) {
  return linear_fwd_impl(x, ldx, K1, x2, ldx2, K2, w, ldw, bias, g1, ldg1, idx1, g2, ldg2, idx2, residual, ldr, y, ldy, nullptr, 0, n_rows, O,
                         act, dtype, stream);
}

extern "C" int anemoi_linear_fwd_pre(const void* x, int64_t ldx, int32_t K1, const void* x2, int64_t ldx2, int32_t K2,
                                     const void* w, int64_t ldw, const void* bias, const void* g1, int64_t ldg1,
                                     const int32_t* idx1, const void* g2, int64_t ldg2, const int32_t* idx2,
                                     const void* residual, int64_t ldr, void* y, int64_t ldy, void* y_pre, int64_t ldy_pre,
                                     int32_t n_rows, int32_t O, anemoi_act_t act, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(y_pre != nullptr, "linear_fwd_pre: null y_pre");
  return linear_fwd_impl(x, ldx, K1, x2, ldx2, K2, w, ldw, bias, g1, ldg1, idx1, g2, ldg2, idx2, residual, ldr, y, ldy, y_pre, ldy_pre,
                         n_rows, O, act, dtype, stream);
}
