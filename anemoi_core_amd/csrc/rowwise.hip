// Row-wise HBM-bound kernels: LayerNorm, GraphConv edge epilogue (LayerNorm + residual + dst segment sum),
// row gather.  One 64-lane wavefront per row; a lane owns VEC contiguous elements per 64*VEC chunk, so every
// global access is a full-width coalesced vector load/store (16 B per lane for bf16 at VEC=8).
//
// Reference semantics: torch.nn.LayerNorm (eps inside sqrt, biased variance, affine) as instantiated by
// layer_kernels.LayerNorm (models/src/anemoi/models/layers/utils.py:107-121); GraphConv's
// "edge_mlp(...) + edge_attr" followed by scatter-sum (models/src/anemoi/models/layers/conv.py:73-81).
#include <stdlib.h>

#include "common.h"

namespace anemoi {

constexpr int kRowWaves = 4;   // waves (rows) per block
constexpr int kMaxChunksLimit = 8;  // register-resident chunks per lane (template CH): D <= 64*VEC*CH

// Load a row into registers as float: x[lane*VEC + t*64*VEC + j], t < nchunks.
template <typename T, int VEC, int CH>
__device__ __forceinline__ void load_row(const T* __restrict__ p, int D, int lane, float (&r)[CH][VEC]) {
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = (t * 64 + lane) * VEC;
    if (c < D) {
      load_vec<T, VEC>(p + c, r[t]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) r[t][j] = 0.f;
    }
  }
}

// In-register LayerNorm of one row held across the wave: two-pass (mean, then centred variance).
template <typename T, int VEC, int CH>
__device__ __forceinline__ void normalise_row(float (&r)[CH][VEC], int D, int lane, const T* __restrict__ gamma,
                                              const T* __restrict__ beta, float eps, float gadd = 0.f) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < CH; ++t)
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += r[t][j];
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = (t * 64 + lane) * VEC;
    if (c < D) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float dlt = r[t][j] - mean;
        ss = fmaf(dlt, dlt, ss);
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = (t * 64 + lane) * VEC;
    if (c < D) {
      float g[VEC], b[VEC];
      load_vec<T, VEC>(gamma + c, g);
      if (beta != nullptr) {
        load_vec<T, VEC>(beta + c, b);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) b[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) r[t][j] = fmaf((r[t][j] - mean) * rstd, g[j] + gadd, b[j]);
    }
  }
}

template <typename T, int VEC, int CH>
__device__ __forceinline__ void store_row(T* __restrict__ p, int D, int lane, const float (&r)[CH][VEC]) {
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = (t * 64 + lane) * VEC;
    if (c < D) store_vec<T, VEC>(p + c, r[t]);
  }
}

template <typename T, int VEC, int CH>
__global__ __launch_bounds__(64 * kRowWaves) void layernorm_fwd_kernel(const T* __restrict__ x, int64_t ldx,
                                                                       const T* __restrict__ gamma,
                                                                       const T* __restrict__ beta,
                                                                       const T* __restrict__ residual, int64_t ldr,
                                                                       T* __restrict__ y, int64_t ldy, int n_rows,
                                                                       int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int rowi = blockIdx.x * kRowWaves + (threadIdx.x >> 6);
  if (rowi >= n_rows) return;
  float r[CH][VEC];
  load_row<T, VEC, CH>(x + (int64_t)rowi * ldx, D, lane, r);
  normalise_row<T, VEC, CH>(r, D, lane, gamma, beta, eps);
  if (residual != nullptr) {
    float rs[CH][VEC];
    load_row<T, VEC, CH>(residual + (int64_t)rowi * ldr, D, lane, rs);
#pragma unroll
    for (int t = 0; t < CH; ++t)
#pragma unroll
      for (int j = 0; j < VEC; ++j) r[t][j] += rs[t][j];
  }
  store_row<T, VEC, CH>(y + (int64_t)rowi * ldy, D, lane, r);
}

// Quarter-wave variant for D = 16 * 8 * CHQ 16-bit elements (D = 512: CHQ = 4): FOUR rows per wave, 16 lanes per row, every lane
// owns CHQ 16-byte chunks of its row (chunk c covers columns (c * 16 + l16) * 8 ..): 4x fewer waves than one row per wave, CHQ
// independent loads in flight per lane, and the two reductions stay inside a DPP row (4 butterfly steps instead of 6).
template <typename T, int CHQ, int LPR = 16>
__global__ __launch_bounds__(64 * kRowWaves) void layernorm_fwd_q_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ gamma,
                                                                         const T* __restrict__ beta, const T* __restrict__ residual,
                                                                         int64_t ldr, T* __restrict__ y, int64_t ldy, int n_rows, float eps) {
  constexpr int D = LPR * 8 * CHQ, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, l16 = lane % LPR;
  const int rowi = (blockIdx.x * kRowWaves + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = rowi < n_rows;
  const int64_t r = live ? rowi : n_rows - 1;  // lanes past the end recompute the last row and do not store
  float v[CHQ][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHQ; ++c) {
    load_vec<T, 8>(x + r * ldx + (c * LPR + l16) * 8, v[c]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[c][j];
  }
  const float mean = group_sum<LPR>(s) * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CHQ; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[c][j] -= mean;
      ss = fmaf(v[c][j], v[c][j], ss);
    }
  const float rstd = rsqrtf(group_sum<LPR>(ss) * (1.0f / D) + eps);
#pragma unroll
  for (int c = 0; c < CHQ; ++c) {
    const int col = (c * LPR + l16) * 8;
    float g[8], b[8], o[8];
    load_vec<T, 8>(gamma + col, g);
    if (beta != nullptr) {
      load_vec<T, 8>(beta + col, b);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaf(v[c][j] * rstd, g[j], b[j]);
    if (residual != nullptr) {
      float rs[8];
      load_vec<T, 8>(residual + r * ldr + col, rs);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += rs[j];
    }
    if (live) store_vec<T, 8>(y + r * ldy + col, o);
  }
}

// ConditionalLayerNorm (reference layers/normalization.py:34-94): y = LN(x) * (scale[row] + 1) + shift[row] with per-row
// modulation tensors (the two Linear maps of the conditioning, computed by one fused GEMM); lds = 0 broadcasts one row.
template <typename T, int VEC, int CH>
__global__ __launch_bounds__(64 * kRowWaves) void cond_layernorm_fwd_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ scale,
                                                                            int64_t lds, const T* __restrict__ shift, int64_t ldsh,
                                                                            T* __restrict__ y, int64_t ldy, int n_rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int rowi = blockIdx.x * kRowWaves + (threadIdx.x >> 6);
  if (rowi >= n_rows) return;
  float r[CH][VEC];
  load_row<T, VEC, CH>(x + (int64_t)rowi * ldx, D, lane, r);
  normalise_row<T, VEC, CH>(r, D, lane, scale + (int64_t)rowi * lds, shift + (int64_t)rowi * ldsh, eps, 1.0f);
  store_row<T, VEC, CH>(y + (int64_t)rowi * ldy, D, lane, r);
}

// e_new = LN(z) + e_old (LN optional) ; agg[d] = sum over the in-edges of d (dst-sorted => contiguous rows).
template <typename T, int VEC, int CH>
__global__ __launch_bounds__(64 * kRowWaves) void edge_ln_res_segsum_kernel(
    const T* __restrict__ z, int64_t ldz, const T* __restrict__ e_old, int64_t lde, const T* __restrict__ gamma,
    const T* __restrict__ beta, float eps, const int32_t* __restrict__ colptr, T* __restrict__ e_new, int64_t ldn,
    T* __restrict__ agg, int64_t ldagg, int n_dst, int D) {
  const int lane = threadIdx.x & 63;
  const int d = __builtin_amdgcn_readfirstlane(blockIdx.x * kRowWaves + (threadIdx.x >> 6));
  if (d >= n_dst) return;
  const int beg = __builtin_amdgcn_readfirstlane(colptr[d]);
  const int end = __builtin_amdgcn_readfirstlane(colptr[d + 1]);
  float sum[CH][VEC];
#pragma unroll
  for (int t = 0; t < CH; ++t)
#pragma unroll
    for (int j = 0; j < VEC; ++j) sum[t][j] = 0.f;
  for (int m = beg; m < end; ++m) {
    float r[CH][VEC], eo[CH][VEC];
    load_row<T, VEC, CH>(z + (int64_t)m * ldz, D, lane, r);
    load_row<T, VEC, CH>(e_old + (int64_t)m * lde, D, lane, eo);
    if (gamma != nullptr) normalise_row<T, VEC, CH>(r, D, lane, gamma, beta, eps);
#pragma unroll
    for (int t = 0; t < CH; ++t)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        r[t][j] += eo[t][j];
      }
    store_row<T, VEC, CH>(e_new + (int64_t)m * ldn, D, lane, r);
    // accumulate what was actually stored (storage precision), matching scatter(sum) over the stored e_new
#pragma unroll
    for (int t = 0; t < CH; ++t)
#pragma unroll
      for (int j = 0; j < VEC; ++j) sum[t][j] += to_float(from_float<T>(r[t][j]));
  }
  store_row<T, VEC, CH>(agg + (int64_t)d * ldagg, D, lane, sum);
}

// Quarter-wave variant for 512 16-bit channels (the GNN's edge rows): still one wave per destination, but FOUR of its in-edges at
// a time - 16 lanes per edge row, CHQ 16-byte chunks per lane, so 2 * CHQ loads per lane are in flight instead of 2, the LayerNorm
// reductions stay inside a DPP row, and the loads of the next four rows are issued before the current four are normalised.  A
// destination of the processor mesh has ~8 in-edges: two trips through the loop instead of eight dependent ones.  The per-row
// arithmetic is the one of the kernel above; the segment sum adds the lane groups' partial sums (fp32) at the end.
// <T, 2, 32> is the half-wave form (two rows at a time, 32 lanes per row, half the registers).
template <typename T, int CHQ, int LPR = 16>
__global__ __launch_bounds__(64 * kRowWaves) void edge_ln_res_segsum_kernel_mr(
    const T* __restrict__ z, int64_t ldz, const T* __restrict__ e_old, int64_t lde, const T* __restrict__ gamma,
    const T* __restrict__ beta, float eps, const int32_t* __restrict__ colptr, T* __restrict__ e_new, int64_t ldn,
    T* __restrict__ agg, int64_t ldagg, int n_dst) {
  constexpr int D = LPR * 8 * CHQ, RPW = 64 / LPR;
  using V8 = Vec<T, 8>;
  const int lane = threadIdx.x & 63, l16 = lane % LPR, g = lane / LPR;
  const int d = __builtin_amdgcn_readfirstlane(blockIdx.x * kRowWaves + (threadIdx.x >> 6));
  if (d >= n_dst) return;
  const int beg = __builtin_amdgcn_readfirstlane(colptr[d]);
  const int end = __builtin_amdgcn_readfirstlane(colptr[d + 1]);
  V8 gq[CHQ], bq[CHQ];
#pragma unroll
  for (int c = 0; c < CHQ; ++c) {
    gq[c] = *reinterpret_cast<const V8*>(gamma + (c * LPR + l16) * 8);
    if (beta != nullptr) {
      bq[c] = *reinterpret_cast<const V8*>(beta + (c * LPR + l16) * 8);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) bq[c].v[j] = from_float<T>(0.f);
    }
  }
  float sum[CHQ][8];
#pragma unroll
  for (int c = 0; c < CHQ; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[c][j] = 0.f;
  V8 zq[CHQ], eq[CHQ];
  const auto fetch = [&](int m0, V8(&zz)[CHQ], V8(&ee)[CHQ]) {  // groups past the segment's end re-read its last row and drop the result
    const int64_t m = min(m0 + g, end - 1);
#pragma unroll
    for (int c = 0; c < CHQ; ++c) {
      zz[c] = *reinterpret_cast<const V8*>(z + m * ldz + (c * LPR + l16) * 8);
      ee[c] = *reinterpret_cast<const V8*>(e_old + m * lde + (c * LPR + l16) * 8);
    }
  };
  if (beg < end) fetch(beg, zq, eq);
  for (int m0 = beg; m0 < end; m0 += RPW) {
    V8 zn[CHQ], en[CHQ];
    const bool more = m0 + RPW < end;  // wave-uniform
    if (more) fetch(m0 + RPW, zn, en);
    const bool live = m0 + g < end;
    float v[CHQ][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHQ; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[c][j] = to_float(zq[c].v[j]);
        s += v[c][j];
      }
    const float mean = group_sum<LPR>(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CHQ; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[c][j] -= mean;
        ss = fmaf(v[c][j], v[c][j], ss);
      }
    const float rstd = rsqrtf(group_sum<LPR>(ss) * (1.0f / D) + eps);
#pragma unroll
    for (int c = 0; c < CHQ; ++c) {
      V8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o.v[j] = from_float<T>(fmaf(v[c][j] * rstd, to_float(gq[c].v[j]), to_float(bq[c].v[j])) + to_float(eq[c].v[j]));
        if (live) sum[c][j] += to_float(o.v[j]);  // what is stored (storage precision), as scatter(sum) over the stored e_new sees it
      }
      if (live) *reinterpret_cast<V8*>(e_new + (int64_t)(m0 + g) * ldn + (c * LPR + l16) * 8) = o;
    }
    if (more) {
#pragma unroll
      for (int c = 0; c < CHQ; ++c) {
        zq[c] = zn[c];
        eq[c] = en[c];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CHQ; ++c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (LPR <= 16) sum[c][j] += __shfl_xor(sum[c][j], 16, 64);
      if constexpr (LPR <= 32) sum[c][j] += __shfl_xor(sum[c][j], 32, 64);
    }
    if (g == 0) store_vec<T, 8>(agg + (int64_t)d * ldagg + (c * LPR + l16) * 8, sum[c]);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(64 * kRowWaves) void gather_rows_kernel(const T* __restrict__ x, int64_t ldx,
                                                                     const int32_t* __restrict__ idx,
                                                                     T* __restrict__ out, int64_t ldo, int n_out, int D) {
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * kRowWaves + (threadIdx.x >> 6));
  if (i >= n_out) return;
  const int s = __builtin_amdgcn_readfirstlane(idx[i]);
  for (int c = lane * VEC; c < D; c += 64 * VEC) {
    *reinterpret_cast<Vec<T, VEC>*>(out + (int64_t)i * ldo + c) = *reinterpret_cast<const Vec<T, VEC>*>(x + (int64_t)s * ldx + c);
  }
}

// Output boundings at the model edge (reference layers/bounding.py:81-307), all configured boundings in ONE pass: a thread
// owns a row and applies the column program in order (ops: int32 [n_ops][4] = kind, column, total column, unused;
// params: fp32 [n_ops][2] = min / max or the normalised minimum).  kind: 1 relu, 2 leaky relu, 3 relu above a minimum,
// 4 leaky relu above a minimum, 5 hardtanh, 6 leaky hardtanh, 7 hardtanh * x[total], 8 leaky hardtanh * x[total]
// (slopes 0.01 as in torch / layers/activations.py:16-42), 9 de-normalisation (x - p0) / p1 (preprocessing/normalizer.py:217-252).
template <typename T>
__global__ void bound_columns_kernel(T* __restrict__ x, int64_t ldx, int n_rows, const int32_t* __restrict__ ops,
                                     const float* __restrict__ params, int n_ops) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  T* row = x + (int64_t)r * ldx;
  for (int i = 0; i < n_ops; ++i) {
    const int kind = ops[4 * i], col = ops[4 * i + 1], tot = ops[4 * i + 2];
    const float p0 = params[2 * i], p1 = params[2 * i + 1];
    float v = to_float(row[col]);
    auto leaky = [](float t) { return t > 0.f ? t : 0.01f * t; };
    auto lht = [&](float t) { return t < p0 ? p0 + 0.01f * (t - p0) : (t > p1 ? p1 + 0.01f * (t - p1) : t); };
    switch (kind) {
      case 1: v = fmaxf(v, 0.f); break;
      case 2: v = leaky(v); break;
      case 3: v = fmaxf(v - p0, 0.f) + p0; break;
      case 4: v = leaky(v - p0) + p0; break;
      case 5: v = fminf(fmaxf(v, p0), p1); break;
      case 6: v = lht(v); break;
      case 7: v = to_float(from_float<T>(fminf(fmaxf(v, p0), p1))) * to_float(row[tot]); break;
      case 8: v = to_float(from_float<T>(lht(v))) * to_float(row[tot]); break;
      case 9: v = to_float(from_float<T>(v - p0)) / p1; break;  // InputNormalizer.inverse_transform: x.subtract_(add).div_(mul) (normalizer.py:246-252)
      default: break;
    }
    row[col] = from_float<T>(v);
  }
}

// Output assembly at the model edge (reference models/encoder_processor_decoder.py:145-163 for batch = ensemble = time = 1):
// out[n, v] = x_out[n, v] + (col_map[v] >= 0 ? x_skip[n, col_map[v]] : 0) - the residual (SkipConnection) added onto the
// prognostic columns - in one pass instead of clone + index_select + index_add_.
template <typename T, int Q>
__global__ void assemble_output_kernel(const T* __restrict__ x_out, int64_t ldx, const T* __restrict__ skip, int64_t lds,
                                       const int32_t* __restrict__ col_map, T* __restrict__ out, int64_t ldo, int n_rows, int n_cols) {
  const int per_row = n_cols / Q;  // Q columns per thread (8-byte moves for 16-bit types when the widths / strides allow)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * per_row) return;
  const int n = (int)(i / per_row), v0 = (int)(i % per_row) * Q;
  Vec<T, Q> a = *reinterpret_cast<const Vec<T, Q>*>(x_out + (int64_t)n * ldx + v0), o;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    float val = to_float(a.v[j]);
    const int m = col_map[v0 + j];
    if (m >= 0) val = to_float(from_float<T>(val)) + to_float(skip[(int64_t)n * lds + m]);
    o.v[j] = from_float<T>(val);
  }
  *reinterpret_cast<Vec<T, Q>*>(out + (int64_t)n * ldo + v0) = o;
}

// Input assembly at the model edge for batch = ensemble = 1 (reference models/encoder_processor_decoder.py:98-143,
// "batch time ensemble grid vars -> (batch ensemble grid) (time vars)" + cat with the node attributes):
//   out[n, :] = [ x[0, n, :] | ... | x[T-1, n, :] | attrs[n, :] | 0 ... ]   (width W >= T*V + A: the alignment zeros of the embedding GEMMs)
// in one pass instead of a permute-copy and a cat.  Q elements per thread (Q = 4 when V, A, W and the row strides allow 8-byte moves).
template <typename T, int Q>
__global__ void assemble_input_kernel(const T* __restrict__ x, int64_t ld_t, int64_t ldx, int T_steps, int V, const T* __restrict__ attrs,
                                      int64_t lda, int A, T* __restrict__ out, int64_t ldo, int W, int n_rows) {
  const int per_row = W / Q;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * per_row) return;
  const int n = (int)(i / per_row), c = (int)(i % per_row) * Q;
  Vec<T, Q> v{};
  if (c < T_steps * V) {
    const int t = c / V, cv = c % V;  // V % Q == 0: a group never straddles two time steps
    v = *reinterpret_cast<const Vec<T, Q>*>(x + t * ld_t + (int64_t)n * ldx + cv);
  } else if (c < T_steps * V + A) {
    v = *reinterpret_cast<const Vec<T, Q>*>(attrs + (int64_t)n * lda + (c - T_steps * V));
  }
  *reinterpret_cast<Vec<T, Q>*>(out + (int64_t)n * ldo + c) = v;
}

// x * m + a with the two roundings of torch's `x.mul_(m).add_(a)` (no FMA contraction): bit-equal to the reference's
// InputNormalizer.transform in fp32 (preprocessing/normalizer.py:154-190).
__device__ __forceinline__ float mul_then_add(float x, float m, float a) {
#pragma clang fp contract(off)
  const float t = x * m;
  return t + a;
}

// Input assembly WITH the input normaliser as a column program (scope row f4): as assemble_input_kernel, and the time /
// variable columns become x * mul[v] + add[v] (fp32 arithmetic, one pass over the input).  The input may be wider than
// the model dtype (fp32 data into a 16-bit model): TI -> fp32 -> TO.
template <typename TI, typename TO>
__global__ void assemble_input_norm_kernel(const TI* __restrict__ x, int64_t ld_t, int64_t ldx, int T_steps, int V, const float* __restrict__ mul,
                                           const float* __restrict__ add, const TO* __restrict__ attrs, int64_t lda, int A, TO* __restrict__ out,
                                           int64_t ldo, int W, int n_rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * W) return;
  const int n = (int)(i / W), c = (int)(i % W);
  float v = 0.f;
  if (c < T_steps * V) {
    const int t = c / V, cv = c % V;
    v = to_float(x[t * ld_t + (int64_t)n * ldx + cv]);
    if (mul != nullptr) v = mul_then_add(v, mul[cv], add[cv]);
  } else if (c < T_steps * V + A) {
    v = to_float(attrs[(int64_t)n * lda + (c - T_steps * V)]);
  }
  out[(int64_t)n * ldo + c] = from_float<TO>(v);
}

// The same for 16-bit outputs whose width is a multiple of 8: a thread produces 8 consecutive columns and stores them as one 16-byte
// vector (the GNN embeddings' zero-padded [M, 128] inputs are mostly padding: 25 us -> a few at 81 840 rows).
template <typename TI, typename TO>
__global__ void assemble_input_norm_vec8_kernel(const TI* __restrict__ x, int64_t ld_t, int64_t ldx, int T_steps, int V, const float* __restrict__ mul,
                                                const float* __restrict__ add, const TO* __restrict__ attrs, int64_t lda, int A, TO* __restrict__ out,
                                                int64_t ldo, int W, int n_rows) {
  const int per_row = W >> 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * per_row) return;
  const int n = (int)(i / per_row), c0 = (int)(i % per_row) << 3;
  Vec<TO, 8> o;
  if (T_steps == 1 && A == 0) {  // a plain cast + zero-pad of [N, V] rows (the GNN embeddings' inputs): no divisions
    const TI* xr = x + (int64_t)n * ldx;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      float v = c < V ? to_float(xr[c]) : 0.f;
      if (mul != nullptr && c < V) v = mul_then_add(v, mul[c], add[c]);
      o.v[j] = from_float<TO>(v);
    }
    *reinterpret_cast<Vec<TO, 8>*>(out + (int64_t)n * ldo + c0) = o;
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    float v = 0.f;
    if (c < T_steps * V) {
      const int t = c / V, cv = c % V;
      v = to_float(x[t * ld_t + (int64_t)n * ldx + cv]);
      if (mul != nullptr) v = mul_then_add(v, mul[cv], add[cv]);
    } else if (c < T_steps * V + A) {
      v = to_float(attrs[(int64_t)n * lda + (c - T_steps * V)]);
    }
    o.v[j] = from_float<TO>(v);
  }
  *reinterpret_cast<Vec<TO, 8>*>(out + (int64_t)n * ldo + c0) = o;
}

// Output assembly with the skip connection taken from the RAW input (normalised on the fly) - the counterpart of
// assemble_input_norm_kernel: out[n, v] = x_out[n, v] + (col_map[v] >= 0 ? skip[n, m] * mul[m] + add[m] : 0), m = col_map[v].
// x_out is in the model dtype (TM), skip and out in the caller's data dtype (TS): the reference adds the residual in the
// input's dtype (x_out.to(dtype=x.dtype), models/encoder_processor_decoder.py:145-158).
template <typename TM, typename TS>
__global__ void assemble_output_norm_kernel(const TM* __restrict__ x_out, int64_t ldx, const TS* __restrict__ skip, int64_t lds,
                                            const int32_t* __restrict__ col_map, const float* __restrict__ mul, const float* __restrict__ add,
                                            TS* __restrict__ out, int64_t ldo, int n_rows, int n_cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * n_cols) return;
  const int n = (int)(i / n_cols), v = (int)(i % n_cols);
  float val = to_float(from_float<TS>(to_float(x_out[(int64_t)n * ldx + v])));
  const int m = col_map[v];
  if (m >= 0) {
    float sk = to_float(skip[(int64_t)n * lds + m]);
    if (mul != nullptr) sk = to_float(from_float<TS>(mul_then_add(sk, mul[m], add[m])));
    val += sk;
  }
  out[(int64_t)n * ldo + v] = from_float<TS>(val);
}

// y[r, c] = x[r, c] * mul[c] + add[c] (inverse = 0) or (x[r, c] - add[c]) / mul[c] (inverse = 1): InputNormalizer.transform /
// inverse_transform as a stand-alone kernel (preprocessing/normalizer.py:154-252); may run in place (y == x).
template <typename T>
__global__ void affine_columns_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, const float* __restrict__ mul,
                                      const float* __restrict__ add, int inverse, int n_rows, int V) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * V) return;
  const int r = (int)(i / V), c = (int)(i % V);
  const float v = to_float(x[(int64_t)r * ldx + c]);
  float o;
  if (inverse) {
    o = to_float(from_float<T>(v - add[c])) / mul[c];  // subtract_ then div_: two roundings in T like torch's in-place ops
  } else {
    if constexpr (sizeof(T) == 4) {
      o = mul_then_add(v, mul[c], add[c]);  // two fp32 roundings, never an FMA
    } else {
      o = to_float(from_float<T>(v * mul[c]));  // mul_ rounds to T, then add_ rounds again
      o = o + add[c];
    }
  }
  y[(int64_t)r * ldy + c] = from_float<T>(o);
}

// Pick the widest vector width (in elements) such that rows stay 16-byte-or-narrower aligned and D % VEC == 0.
template <typename T>
static int pick_vec(int D, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
  int vec = 16 / (int)sizeof(T);  // 16-byte accesses
  auto ok = [&](int v) {
    if (D % v) return false;
    for (int64_t ld : lds)
      if (ld % v) return false;
    for (const void* p : ptrs)
      if (p && (reinterpret_cast<uintptr_t>(p) % (v * sizeof(T)))) return false;
    return true;
  };
  while (vec > 1 && !ok(vec)) vec >>= 1;
  return vec;
}

// smallest power-of-two chunk count covering D, or 0 if the row does not fit in registers
static int pick_chunks(int D, int vec) {
  for (int ch = 1; ch <= kMaxChunksLimit; ch *= 2)
    if (D <= 64 * vec * ch) return ch;
  return 0;
}

#define ALL_VEC_CH(M) \
  M(1, 1) M(1, 2) M(1, 4) M(1, 8) M(2, 1) M(2, 2) M(2, 4) M(2, 8) M(4, 1) M(4, 2) M(4, 4) M(4, 8) M(8, 1) M(8, 2) M(8, 4) M(8, 8)

template <typename T>
static int layernorm_launch(const void* x, int64_t ldx, const void* gamma, const void* beta, const void* residual,
                            int64_t ldr, void* y, int64_t ldy, int n_rows, int D, float eps, hipStream_t st) {
  const int vec = pick_vec<T>(D, {ldx, ldy, residual ? ldr : (int64_t)0}, {x, y, gamma, beta, residual});
  const int ch = pick_chunks(D, vec);
  ANEMOI_REQUIRE(ch > 0, "layernorm_fwd: D=%d too large for the register-resident row (max %d at vector width %d)", D, 64 * vec * kMaxChunksLimit, vec);
  static const int quarter = [] { const char* e = getenv("ANEMOI_LN_QUARTER"); return e ? atoi(e) : 1; }();
  if (quarter && sizeof(T) == 2 && vec == 8 && D == 512 && n_rows > 0) {  // the 512-channel rows of the hot path (8 rows per wave,
    // 8 lanes per row, measured slower: +55 us per forward)
    const int rows_per_block = 4 * kRowWaves;
    hipLaunchKernelGGL((layernorm_fwd_q_kernel<T, 4>), dim3((n_rows + rows_per_block - 1) / rows_per_block), dim3(64 * kRowWaves), 0, st,
                       (const T*)x, ldx, (const T*)gamma, (const T*)beta, (const T*)residual, ldr, (T*)y, ldy, n_rows, eps);
    return check_launch("layernorm_fwd_q_kernel");
  }
  const dim3 grid((n_rows + kRowWaves - 1) / kRowWaves), block(64 * kRowWaves);
#define LN_CASE(V, C)                                                                                                   \
  case V * 16 + C:                                                                                                      \
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, V, C>), grid, block, 0, st, (const T*)x, ldx, (const T*)gamma,          \
                       (const T*)beta, (const T*)residual, ldr, (T*)y, ldy, n_rows, D, eps);                            \
    break;
  switch (vec * 16 + ch) {
    ALL_VEC_CH(LN_CASE)
    default: set_error("layernorm_fwd: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef LN_CASE
  return check_launch("layernorm_fwd_kernel");
}

template <typename T>
static int cond_layernorm_launch(const void* x, int64_t ldx, const void* scale, int64_t lds, const void* shift, int64_t ldsh, void* y,
                                 int64_t ldy, int n_rows, int D, float eps, hipStream_t st) {
  const int vec = pick_vec<T>(D, {ldx, ldy, lds, ldsh}, {x, y, scale, shift});
  const int ch = pick_chunks(D, vec);
  ANEMOI_REQUIRE(ch > 0, "cond_layernorm_fwd: D=%d too large for the register-resident row", D);
  const dim3 grid((n_rows + kRowWaves - 1) / kRowWaves), block(64 * kRowWaves);
#define CLN_CASE(V, C)                                                                                                      \
  case V * 16 + C:                                                                                                         \
    hipLaunchKernelGGL((cond_layernorm_fwd_kernel<T, V, C>), grid, block, 0, st, (const T*)x, ldx, (const T*)scale, lds,   \
                       (const T*)shift, ldsh, (T*)y, ldy, n_rows, D, eps);                                                 \
    break;
  switch (vec * 16 + ch) {
    ALL_VEC_CH(CLN_CASE)
    default: set_error("cond_layernorm_fwd: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef CLN_CASE
  return check_launch("cond_layernorm_fwd_kernel");
}

template <typename T>
static int edge_launch(const void* z, int64_t ldz, const void* e_old, int64_t lde, const void* gamma, const void* beta,
                       float eps, const int32_t* colptr, void* e_new, int64_t ldn, void* agg, int64_t ldagg, int n_dst,
                       int D, hipStream_t st) {
  const int vec = pick_vec<T>(D, {ldz, lde, ldn, ldagg}, {z, e_old, e_new, agg, gamma, beta});
  const int ch = pick_chunks(D, vec);
  ANEMOI_REQUIRE(ch > 0, "edge_ln_residual_segment_sum_fwd: D=%d too large (max %d)", D, 64 * vec * kMaxChunksLimit);
  const dim3 grid((n_dst + kRowWaves - 1) / kRowWaves), block(64 * kRowWaves);
  static const int quarter = env_int(getenv("ANEMOI_SEGSUM_QUARTER"), 2, 0, 3);  // rows per wave at a time: 1 -> four, 2 -> two, 3 -> one
  if constexpr (sizeof(T) == 2) {
    if (quarter && vec == 8 && D == 512 && gamma != nullptr && n_dst > 0) {  // the GNN's 512-channel edge rows
      if (quarter == 1)
        hipLaunchKernelGGL((edge_ln_res_segsum_kernel_mr<T, 4, 16>), grid, block, 0, st, (const T*)z, ldz, (const T*)e_old, lde, (const T*)gamma,
                           (const T*)beta, eps, colptr, (T*)e_new, ldn, (T*)agg, ldagg, n_dst);
      else if (quarter == 2)
        hipLaunchKernelGGL((edge_ln_res_segsum_kernel_mr<T, 2, 32>), grid, block, 0, st, (const T*)z, ldz, (const T*)e_old, lde, (const T*)gamma,
                           (const T*)beta, eps, colptr, (T*)e_new, ldn, (T*)agg, ldagg, n_dst);
      else
        hipLaunchKernelGGL((edge_ln_res_segsum_kernel_mr<T, 1, 64>), grid, block, 0, st, (const T*)z, ldz, (const T*)e_old, lde, (const T*)gamma,
                           (const T*)beta, eps, colptr, (T*)e_new, ldn, (T*)agg, ldagg, n_dst);
      return check_launch("edge_ln_res_segsum_kernel_mr");
    }
  }
#define E_CASE(V, C)                                                                                                    \
  case V * 16 + C:                                                                                                      \
    hipLaunchKernelGGL((edge_ln_res_segsum_kernel<T, V, C>), grid, block, 0, st, (const T*)z, ldz, (const T*)e_old,     \
                       lde, (const T*)gamma, (const T*)beta, eps, colptr, (T*)e_new, ldn, (T*)agg, ldagg, n_dst, D);    \
    break;
  switch (vec * 16 + ch) {
    ALL_VEC_CH(E_CASE)
    default: set_error("edge_ln_residual_segment_sum_fwd: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef E_CASE
  return check_launch("edge_ln_res_segsum_kernel");
}

template <typename T>
static int gather_launch(const void* x, int64_t ldx, const int32_t* idx, void* out, int64_t ldo, int n_out, int D,
                         hipStream_t st) {
  const int vec = pick_vec<T>(D, {ldx, ldo}, {x, out});
  const dim3 grid((n_out + kRowWaves - 1) / kRowWaves), block(64 * kRowWaves);
#define G_CASE(V)                                                                                                       \
  case V:                                                                                                               \
    hipLaunchKernelGGL((gather_rows_kernel<T, V>), grid, block, 0, st, (const T*)x, ldx, idx, (T*)out, ldo, n_out, D);  \
    break;
  switch (vec) {
    G_CASE(1) G_CASE(2) G_CASE(4) G_CASE(8)
    default: set_error("gather_rows: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef G_CASE
  return check_launch("gather_rows_kernel");
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_layernorm_fwd(const void* x, int64_t ldx, const void* gamma, const void* beta, const void* residual,
                                    int64_t ldr, void* y, int64_t ldy, int32_t n_rows, int32_t D, float eps,
                                    anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldx >= D && ldy >= D && (!residual || ldr >= D), "layernorm_fwd: bad sizes n_rows=%d D=%d", n_rows, D);
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && y && gamma, "layernorm_fwd: null pointer");
  switch (dtype) {
    case ANEMOI_F32: return layernorm_launch<float>(x, ldx, gamma, beta, residual, ldr, y, ldy, n_rows, D, eps, as_stream(stream));
    case ANEMOI_BF16: return layernorm_launch<bf16_t>(x, ldx, gamma, beta, residual, ldr, y, ldy, n_rows, D, eps, as_stream(stream));
    case ANEMOI_F16: return layernorm_launch<f16_t>(x, ldx, gamma, beta, residual, ldr, y, ldy, n_rows, D, eps, as_stream(stream));
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_edge_ln_residual_segment_sum_fwd(const void* z, int64_t ldz, const void* e_old, int64_t lde,
                                                       const void* gamma, const void* beta, float eps,
                                                       const int32_t* colptr, void* e_new, int64_t ldn, void* agg,
                                                       int64_t ldagg, int32_t n_dst, int32_t D, anemoi_dtype_t dtype,
                                                       void* stream) {
  ANEMOI_REQUIRE(n_dst >= 0 && D > 0 && ldz >= D && lde >= D && ldn >= D && ldagg >= D, "edge_ln_residual_segment_sum_fwd: bad sizes");
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(z && e_old && e_new && agg && colptr, "edge_ln_residual_segment_sum_fwd: null pointer");
  switch (dtype) {
    case ANEMOI_F32: return edge_launch<float>(z, ldz, e_old, lde, gamma, beta, eps, colptr, e_new, ldn, agg, ldagg, n_dst, D, as_stream(stream));
    case ANEMOI_BF16: return edge_launch<bf16_t>(z, ldz, e_old, lde, gamma, beta, eps, colptr, e_new, ldn, agg, ldagg, n_dst, D, as_stream(stream));
    case ANEMOI_F16: return edge_launch<f16_t>(z, ldz, e_old, lde, gamma, beta, eps, colptr, e_new, ldn, agg, ldagg, n_dst, D, as_stream(stream));
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_gather_rows(const void* x, int64_t ldx, const int32_t* idx, void* out, int64_t ldo, int32_t n_out,
                                  int32_t D, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_out >= 0 && D > 0 && ldx >= D && ldo >= D, "gather_rows: bad sizes");
  if (n_out == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && idx && out, "gather_rows: null pointer");
  switch (dtype) {
    case ANEMOI_F32: return gather_launch<float>(x, ldx, idx, out, ldo, n_out, D, as_stream(stream));
    case ANEMOI_BF16: return gather_launch<bf16_t>(x, ldx, idx, out, ldo, n_out, D, as_stream(stream));
    case ANEMOI_F16: return gather_launch<f16_t>(x, ldx, idx, out, ldo, n_out, D, as_stream(stream));
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_cond_layernorm_fwd(const void* x, int64_t ldx, const void* scale, int64_t lds, const void* shift, int64_t ldsh,
                                         void* y, int64_t ldy, int32_t n_rows, int32_t D, float eps, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldx >= D && ldy >= D && (lds == 0 || lds >= D) && (ldsh == 0 || ldsh >= D), "cond_layernorm_fwd: bad sizes");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && y && scale && shift, "cond_layernorm_fwd: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return cond_layernorm_launch<float>(x, ldx, scale, lds, shift, ldsh, y, ldy, n_rows, D, eps, st);
    case ANEMOI_BF16: return cond_layernorm_launch<bf16_t>(x, ldx, scale, lds, shift, ldsh, y, ldy, n_rows, D, eps, st);
    case ANEMOI_F16: return cond_layernorm_launch<f16_t>(x, ldx, scale, lds, shift, ldsh, y, ldy, n_rows, D, eps, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_bound_columns(void* x, int64_t ldx, int32_t n_rows, int32_t n_cols, const int32_t* ops, const float* params,
                                    int32_t n_ops, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && n_cols > 0 && ldx >= n_cols && n_ops >= 0, "bound_columns: bad sizes");
  if (n_rows == 0 || n_ops == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && ops && params, "bound_columns: null pointer");
  const dim3 grid((n_rows + 255) / 256), block(256);
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: hipLaunchKernelGGL((bound_columns_kernel<float>), grid, block, 0, st, (float*)x, ldx, n_rows, ops, params, n_ops); break;
    case ANEMOI_BF16: hipLaunchKernelGGL((bound_columns_kernel<bf16_t>), grid, block, 0, st, (bf16_t*)x, ldx, n_rows, ops, params, n_ops); break;
    case ANEMOI_F16: hipLaunchKernelGGL((bound_columns_kernel<f16_t>), grid, block, 0, st, (f16_t*)x, ldx, n_rows, ops, params, n_ops); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
  return check_launch("bound_columns_kernel");
}

extern "C" int anemoi_assemble_output(const void* x_out, int64_t ldx, const void* x_skip, int64_t lds, const int32_t* col_map, void* out,
                                      int64_t ldo, int32_t n_rows, int32_t n_cols, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && n_cols > 0 && ldx >= n_cols && ldo >= n_cols, "assemble_output: bad sizes");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x_out && x_skip && col_map && out, "assemble_output: null pointer");
  const size_t es = dtype == ANEMOI_F32 ? 4 : 2;
  const bool q4 = n_cols % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && reinterpret_cast<uintptr_t>(x_out) % (4 * es) == 0 &&
                  reinterpret_cast<uintptr_t>(out) % (4 * es) == 0;
  const int64_t n = (int64_t)n_rows * (n_cols / (q4 ? 4 : 1));
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = as_stream(stream);
#define AO_LAUNCH(TT, QQ)                                                                                                              \
  hipLaunchKernelGGL((assemble_output_kernel<TT, QQ>), grid, block, 0, st, (const TT*)x_out, ldx, (const TT*)x_skip, lds, col_map, (TT*)out, \
                     ldo, n_rows, n_cols)
  switch (dtype) {
    case ANEMOI_F32: if (q4) AO_LAUNCH(float, 4); else AO_LAUNCH(float, 1); break;
    case ANEMOI_BF16: if (q4) AO_LAUNCH(bf16_t, 4); else AO_LAUNCH(bf16_t, 1); break;
    case ANEMOI_F16: if (q4) AO_LAUNCH(f16_t, 4); else AO_LAUNCH(f16_t, 1); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
#undef AO_LAUNCH
  return check_launch("assemble_output_kernel");
}

extern "C" int anemoi_assemble_input(const void* x, int64_t ld_t, int64_t ldx, int32_t T_steps, int32_t V, const void* attrs, int64_t lda,
                                     int32_t A, void* out, int64_t ldo, int32_t W, int32_t n_rows, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && T_steps > 0 && V > 0 && A >= 0 && W >= T_steps * V + A && ldo >= W && ldx >= V && (A == 0 || lda >= A),
                 "assemble_input: bad sizes");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && out && (A == 0 || attrs), "assemble_input: null pointer");
  const size_t es = dtype == ANEMOI_F32 ? 4 : 2;
  auto al = [&](const void* p, int q) { return p == nullptr || reinterpret_cast<uintptr_t>(p) % (q * es) == 0; };
  const bool q4 = V % 4 == 0 && A % 4 == 0 && W % 4 == 0 && ld_t % 4 == 0 && ldx % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && al(x, 4) &&
                  al(attrs, 4) && al(out, 4);
  const int q = q4 ? 4 : 1;
  const int64_t n = (int64_t)n_rows * (W / q);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = as_stream(stream);
#define AI_LAUNCH(TT, QQ)                                                                                                        \
  hipLaunchKernelGGL((assemble_input_kernel<TT, QQ>), grid, block, 0, st, (const TT*)x, ld_t, ldx, T_steps, V, (const TT*)attrs, lda, A, \
                     (TT*)out, ldo, W, n_rows)
  switch (dtype) {
    case ANEMOI_F32: if (q4) AI_LAUNCH(float, 4); else AI_LAUNCH(float, 1); break;
    case ANEMOI_BF16: if (q4) AI_LAUNCH(bf16_t, 4); else AI_LAUNCH(bf16_t, 1); break;
    case ANEMOI_F16: if (q4) AI_LAUNCH(f16_t, 4); else AI_LAUNCH(f16_t, 1); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
#undef AI_LAUNCH
  return check_launch("assemble_input_kernel");
}

extern "C" int anemoi_assemble_input_norm(const void* x, anemoi_dtype_t x_dtype, int64_t ld_t, int64_t ldx, int32_t T_steps, int32_t V,
                                          const float* col_mul, const float* col_add, const void* attrs, int64_t lda, int32_t A, void* out,
                                          int64_t ldo, int32_t W, int32_t n_rows, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && T_steps > 0 && V > 0 && A >= 0 && W >= T_steps * V + A && ldo >= W && ldx >= V && (A == 0 || lda >= A),
                 "assemble_input_norm: bad sizes");
  ANEMOI_REQUIRE((col_mul == nullptr) == (col_add == nullptr), "assemble_input_norm: col_mul and col_add go together");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && out && (A == 0 || attrs), "assemble_input_norm: null pointer");
  ANEMOI_REQUIRE(x_dtype == dtype || x_dtype == ANEMOI_F32, "assemble_input_norm: the input is in the model dtype or fp32");
  hipStream_t st = as_stream(stream);
  if (dtype != ANEMOI_F32 && W % 8 == 0 && ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int64_t n8 = (int64_t)n_rows * (W / 8);
    const dim3 grid8((unsigned)((n8 + 255) / 256)), block8(256);
#define AIN8_LAUNCH(TI, TO)                                                                                                                    \
  hipLaunchKernelGGL((assemble_input_norm_vec8_kernel<TI, TO>), grid8, block8, 0, st, (const TI*)x, ld_t, ldx, T_steps, V, col_mul, col_add,  \
                     (const TO*)attrs, lda, A, (TO*)out, ldo, W, n_rows)
    if (dtype == ANEMOI_BF16) {
      if (x_dtype == ANEMOI_F32) AIN8_LAUNCH(float, bf16_t); else AIN8_LAUNCH(bf16_t, bf16_t);
    } else {
      if (x_dtype == ANEMOI_F32) AIN8_LAUNCH(float, f16_t); else AIN8_LAUNCH(f16_t, f16_t);
    }
#undef AIN8_LAUNCH
    return check_launch("assemble_input_norm_vec8_kernel");
  }
  const int64_t n = (int64_t)n_rows * W;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define AIN_LAUNCH(TI, TO)                                                                                                             \
  hipLaunchKernelGGL((assemble_input_norm_kernel<TI, TO>), grid, block, 0, st, (const TI*)x, ld_t, ldx, T_steps, V, col_mul, col_add,  \
                     (const TO*)attrs, lda, A, (TO*)out, ldo, W, n_rows)
  switch (dtype) {
    case ANEMOI_F32: AIN_LAUNCH(float, float); break;
    case ANEMOI_BF16: if (x_dtype == ANEMOI_F32) AIN_LAUNCH(float, bf16_t); else AIN_LAUNCH(bf16_t, bf16_t); break;
    case ANEMOI_F16: if (x_dtype == ANEMOI_F32) AIN_LAUNCH(float, f16_t); else AIN_LAUNCH(f16_t, f16_t); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
#undef AIN_LAUNCH
  return check_launch("assemble_input_norm_kernel");
}

extern "C" int anemoi_assemble_output_norm(const void* x_out, int64_t ldx, anemoi_dtype_t model_dtype, const void* x_skip, int64_t lds,
                                           const int32_t* col_map, const float* col_mul, const float* col_add, void* out, int64_t ldo,
                                           int32_t n_rows, int32_t n_cols, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && n_cols > 0 && ldx >= n_cols && ldo >= n_cols, "assemble_output_norm: bad sizes");
  ANEMOI_REQUIRE((col_mul == nullptr) == (col_add == nullptr), "assemble_output_norm: col_mul and col_add go together");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x_out && x_skip && col_map && out, "assemble_output_norm: null pointer");
  ANEMOI_REQUIRE(model_dtype == dtype || dtype == ANEMOI_F32, "assemble_output_norm: the output is in the model dtype or fp32");
  const int64_t n = (int64_t)n_rows * n_cols;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = as_stream(stream);
#define AON_LAUNCH(TM, TS)                                                                                                              \
  hipLaunchKernelGGL((assemble_output_norm_kernel<TM, TS>), grid, block, 0, st, (const TM*)x_out, ldx, (const TS*)x_skip, lds, col_map, \
                     col_mul, col_add, (TS*)out, ldo, n_rows, n_cols)
  switch (model_dtype) {
    case ANEMOI_F32: AON_LAUNCH(float, float); break;
    case ANEMOI_BF16: if (dtype == ANEMOI_F32) AON_LAUNCH(bf16_t, float); else AON_LAUNCH(bf16_t, bf16_t); break;
    case ANEMOI_F16: if (dtype == ANEMOI_F32) AON_LAUNCH(f16_t, float); else AON_LAUNCH(f16_t, f16_t); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
#undef AON_LAUNCH
  return check_launch("assemble_output_norm_kernel");
}

extern "C" int anemoi_affine_columns(const void* x, int64_t ldx, void* y, int64_t ldy, const float* col_mul, const float* col_add,
                                     int32_t inverse, int32_t n_rows, int32_t V, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && V > 0 && ldx >= V && ldy >= V, "affine_columns: bad sizes");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && y && col_mul && col_add, "affine_columns: null pointer");
  const int64_t n = (int64_t)n_rows * V;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: hipLaunchKernelGGL((affine_columns_kernel<float>), grid, block, 0, st, (const float*)x, ldx, (float*)y, ldy, col_mul, col_add, inverse, n_rows, V); break;
    case ANEMOI_BF16: hipLaunchKernelGGL((affine_columns_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, col_mul, col_add, inverse, n_rows, V); break;
    case ANEMOI_F16: hipLaunchKernelGGL((affine_columns_kernel<f16_t>), grid, block, 0, st, (const f16_t*)x, ldx, (f16_t*)y, ldy, col_mul, col_add, inverse, n_rows, V); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
  return check_launch("affine_columns_kernel");
}
