// Backward of the row-wise ops (scope row f1): LayerNorm, GELU, bias (column sums).  Same lane layout as rowwise.hip:
// one wave64 per row, a lane owns VEC contiguous elements per 64*VEC chunk.
//
// Reference semantics: autograd of torch.nn.LayerNorm / torch.nn.GELU (exact erf form) / the bias of torch.nn.Linear as
// the reference's layer_kernels instantiate them (models/src/anemoi/models/layers/utils.py:107-121).
//   x^ = (x - mean) rstd;  g = dy * gamma;  dx = rstd (g - mean(g) - x^ mean(g x^));  dgamma = sum_rows dy x^;  dbeta = sum_rows dy
// Column sums are deterministic: a persistent grid of kPartialBlocks x kBwdWaves waves walks the rows, each wave keeps its
// partial column sums in registers, the waves of a block add theirs in LDS in wave order and the block writes one row of an
// fp32 workspace [kPartialBlocks][2][D]; a second tiny kernel adds the partials in a fixed order (no atomics).
#include "common.h"

namespace anemoi {
namespace {

constexpr int kWaves = 4;             // waves per block
constexpr int kBwdWaves = 8;          // waves per block of the persistent row-walking kernels (16 waves per CU hide the
constexpr int kPartialBlocks = 512;   // four dependent wave reductions of a LayerNorm-backward row): two blocks per CU
constexpr int kMaxChunks = 8;

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

// MODE 0: LayerNorm backward (dx + partial dgamma/dbeta).  MODE 1: column sums of x only (bias gradient).
template <typename T, int VEC, int CH, int MODE>
__global__ __launch_bounds__(64 * kBwdWaves) void rowwise_bwd_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ gamma,
                                                                  const T* __restrict__ dy, int64_t lddy, T* __restrict__ dx,
                                                                  int64_t lddx, float* __restrict__ part, int n_rows, int D,
                                                                  float eps) {
  const int lane = threadIdx.x & 63;
  extern __shared__ float block_sums[];  // [2][D]
  const int wave = threadIdx.x >> 6;
  const int w = blockIdx.x * kBwdWaves + wave;
  const int nw = gridDim.x * kBwdWaves;
  float sg[CH][VEC], sb[CH][VEC], g[CH][VEC];
#pragma unroll
  for (int t = 0; t < CH; ++t)
#pragma unroll
    for (int j = 0; j < VEC; ++j) sg[t][j] = sb[t][j] = g[t][j] = 0.f;
  if constexpr (MODE == 0) load_row<T, VEC, CH>(gamma, D, lane, g);
  const float inv_d = 1.0f / (float)D;
  for (int r = w; r < n_rows; r += nw) {
    float xv[CH][VEC];
    load_row<T, VEC, CH>(x + (int64_t)r * ldx, D, lane, xv);
    if constexpr (MODE == 1) {
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) sb[t][j] += xv[t][j];
    } else {
      float gv[CH][VEC];
      load_row<T, VEC, CH>(dy + (int64_t)r * lddy, D, lane, gv);
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += xv[t][j];
      const float mean = wave_sum(s) * inv_d;
      float ss = 0.f;
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int c = (t * 64 + lane) * VEC;
        if (c < D) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            xv[t][j] -= mean;
            ss = fmaf(xv[t][j], xv[t][j], ss);
          }
        }
      }
      const float rstd = rsqrtf(wave_sum(ss) * inv_d + eps);
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          xv[t][j] *= rstd;  // x^ (0 outside the row)
          sb[t][j] += gv[t][j];
          sg[t][j] = fmaf(gv[t][j], xv[t][j], sg[t][j]);
          gv[t][j] *= g[t][j];  // g = dy * gamma
          c1 += gv[t][j];
          c2 = fmaf(gv[t][j], xv[t][j], c2);
        }
      c1 = wave_sum(c1) * inv_d;
      c2 = wave_sum(c2) * inv_d;
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int c = (t * 64 + lane) * VEC;
        if (c < D) {
          float o[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] = rstd * (gv[t][j] - c1 - xv[t][j] * c2);
          store_vec<T, VEC>(dx + (int64_t)r * lddx + c, o);
        }
      }
    }
  }
  if (part != nullptr) {  // block-uniform
    for (int k = 0; k < kBwdWaves; ++k) {  // fixed order: wave 0 stores, waves 1.. add
      if (wave == k) {
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          const int c = (t * 64 + lane) * VEC;
          if (c < D) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              block_sums[c + j] = k == 0 ? sg[t][j] : block_sums[c + j] + sg[t][j];
              block_sums[D + c + j] = k == 0 ? sb[t][j] : block_sums[D + c + j] + sb[t][j];
            }
          }
        }
      }
      __syncthreads();
    }
    float* p = part + (int64_t)blockIdx.x * 2 * D;
    for (int i = threadIdx.x; i < 2 * D; i += 64 * kBwdWaves) p[i] = block_sums[i];
  }
}

// ConditionalLayerNorm backward (reference layers/normalization.py:34-94): y = x^ (scale[row] + 1) + shift[row].
//   g = dy (scale + 1);  dx = rstd (g - mean(g) - x^ mean(g x^));  d_scale[row] = dy x^;  (d_shift[row] = dy: no kernel)
template <typename T, int VEC, int CH>
__global__ __launch_bounds__(64 * kWaves) void cond_layernorm_bwd_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ scale,
                                                                         int64_t lds, const T* __restrict__ dy, int64_t lddy,
                                                                         T* __restrict__ dx, int64_t lddx, T* __restrict__ dscale,
                                                                         int64_t ldds, int n_rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  float xv[CH][VEC], gv[CH][VEC], sc[CH][VEC];
  load_row<T, VEC, CH>(x + (int64_t)r * ldx, D, lane, xv);
  load_row<T, VEC, CH>(dy + (int64_t)r * lddy, D, lane, gv);
  load_row<T, VEC, CH>(scale + (int64_t)r * lds, D, lane, sc);
  const float inv_d = 1.0f / (float)D;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < CH; ++t)
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += xv[t][j];
  const float mean = wave_sum(s) * inv_d;
  float ss = 0.f;
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = (t * 64 + lane) * VEC;
    if (c < D) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        xv[t][j] -= mean;
        ss = fmaf(xv[t][j], xv[t][j], ss);
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) * inv_d + eps);
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = (t * 64 + lane) * VEC;
    float ds[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      xv[t][j] *= rstd;
      ds[j] = gv[t][j] * xv[t][j];
      gv[t][j] *= (c < D) ? sc[t][j] + 1.0f : 0.f;
      c1 += gv[t][j];
      c2 = fmaf(gv[t][j], xv[t][j], c2);
    }
    if (c < D) store_vec<T, VEC>(dscale + (int64_t)r * ldds + c, ds);
  }
  c1 = wave_sum(c1) * inv_d;
  c2 = wave_sum(c2) * inv_d;
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int c = (t * 64 + lane) * VEC;
    if (c < D) {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = rstd * (gv[t][j] - c1 - xv[t][j] * c2);
      store_vec<T, VEC>(dx + (int64_t)r * lddx + c, o);
    }
  }
}

// out0[c] = sum_w part[w][0][c], out1[c] = sum_w part[w][1][c].  One block per 16 columns of the [2D] row: 64 groups of 16
// threads each add every 64th partial row with four independent accumulators (loads in flight together), then a fixed-order
// tree over the 64 groups in LDS: deterministic.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ part, int nw, int D, float* __restrict__ out0,
                                                               float* __restrict__ out1) {
  __shared__ float red[64][17];
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * D) {
    const int64_t ld = 2 * (int64_t)D;
    int w = grp;
    for (; w + 192 < nw; w += 256) {
      s0 += part[(int64_t)w * ld + c];
      s1 += part[(int64_t)(w + 64) * ld + c];
      s2 += part[(int64_t)(w + 128) * ld + c];
      s3 += part[(int64_t)(w + 192) * ld + c];
    }
    for (; w < nw; w += 64) s0 += part[(int64_t)w * ld + c];
  }
  red[grp][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    if (grp < step) red[grp][cl] += red[grp + step][cl];
    __syncthreads();
  }
  if (grp == 0 && c < 2 * D) {
    if (c < D) {
      if (out0) out0[c] = red[0][cl];
    } else if (out1) {
      out1[c - D] = red[0][cl];
    }
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ pre, int64_t ldp, const T* __restrict__ dy, int64_t lddy,
                                                       T* __restrict__ dpre, int64_t lddp, int n_rows, int D) {
  const int per_row = D / VEC;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * per_row) return;
  const int r = (int)(i / per_row), c = (int)(i % per_row) * VEC;
  float x[VEC], g[VEC], o[VEC];
  load_vec<T, VEC>(pre + (int64_t)r * ldp + c, x);
  load_vec<T, VEC>(dy + (int64_t)r * lddy + c, g);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float cdf = 0.5f * (1.0f + fast_erf(x[j] * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x[j] * x[j]);
    o[j] = g[j] * fmaf(x[j], pdf, cdf);  // d/dx [x Phi(x)] = Phi(x) + x phi(x)
  }
  store_vec<T, VEC>(dpre + (int64_t)r * lddp + c, o);
}

// y = gelu(x), exact erf form (torch.nn.GELU default): the stand-alone activation of the layer_kernels plug-in point
// (kernels.GELU inside an unmodified reference MLP: layers/mlp.py:158-169).  The fused blocks never launch it.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int n_rows, int D) {
  const int per_row = D / VEC;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * per_row) return;
  const int r = (int)(i / per_row), c = (int)(i % per_row) * VEC;
  float v[VEC];
  load_vec<T, VEC>(x + (int64_t)r * ldx + c, v);
#pragma unroll
  for (int j = 0; j < VEC; ++j) v[j] = gelu_erf(v[j]);
  store_vec<T, VEC>(y + (int64_t)r * ldy + c, v);
}

// out[r] = sum_{i in [ptr[r], ptr[r+1])} x[ids ? ids[i] : i]   (fp32 accumulation, one wave per output row).
// ids == NULL: contiguous segments (edges sorted by destination); ids = the reverse-CSR edge list: rows grouped by source.
// Adjoint of the row gathers in the GraphConv GEMM epilogue and of gather_rows (deterministic, no atomics).
template <typename T, int VEC>
__global__ __launch_bounds__(64 * kWaves) void segment_sum_rows_kernel(const T* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                                                                       const int32_t* __restrict__ ids, T* __restrict__ out, int64_t ldo,
                                                                       int n_out, int D) {
  const int lane = threadIdx.x & 63;
  const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + (threadIdx.x >> 6));
  if (r >= n_out) return;
  const int beg = ptr[r], end = ptr[r + 1];
  for (int c = lane * VEC; c < D; c += 64 * VEC) {
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int i = beg; i < end; ++i) {
      const int row = ids ? ids[i] : i;
      float v[VEC];
      load_vec<T, VEC>(x + (int64_t)row * ldx + c, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += v[j];
    }
    store_vec<T, VEC>(out + (int64_t)r * ldo + c, acc);
  }
}

// out[i] = a[i] + b[idx[i]]
template <typename T, int VEC>
__global__ __launch_bounds__(64 * kWaves) void gather_add_rows_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb,
                                                                      const int32_t* __restrict__ idx, T* __restrict__ out, int64_t ldo,
                                                                      int n_out, int D) {
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + (threadIdx.x >> 6));
  if (i >= n_out) return;
  const int s = idx[i];
  for (int c = lane * VEC; c < D; c += 64 * VEC) {
    float u[VEC], v[VEC];
    load_vec<T, VEC>(a + (int64_t)i * lda + c, u);
    load_vec<T, VEC>(b + (int64_t)s * ldb + c, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) u[j] += v[j];
    store_vec<T, VEC>(out + (int64_t)i * ldo + c, u);
  }
}

// out[c][r] = x[r][c] for r < n_rows, 0 for n_rows <= r < n_pad: the K-contiguous, zero-padded operands of the dW GEMM
// (reduction over the rows).  64 x 64 tiles through LDS (+1 padding: conflict-free both ways), 16-bit or 32-bit elements.
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ out, int64_t ldo,
                                                            int n_rows, int n_cols, int n_pad) {
  __shared__ T tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    T v = from_float<T>(0.f);
    if (r < n_rows && c < n_cols) v = x[(int64_t)r * ldx + c];
    tile[ty + 4 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < n_cols && r < n_pad) out[(int64_t)c * ldo + r] = tile[tx][ty + 4 * i];
  }
}

// Gated feed-forward layers (reference layers/mlp.py:25-59): out = act(gate) * value with [gate | value] the two halves of
// one fused projection.  kind: 0 sigmoid (GLU), 1 SiLU (SwiGLU), 2 GELU erf (GeGLU), 3 ReLU (ReGLU).
__device__ __forceinline__ float glu_act(float g, int kind) {
  switch (kind) {
    case 0: return 1.0f / (1.0f + __expf(-g));
    case 1: return g / (1.0f + __expf(-g));
    case 2: return gelu_erf(g);
    default: return fmaxf(g, 0.f);
  }
}
__device__ __forceinline__ float glu_act_grad(float g, int kind) {
  switch (kind) {
    case 0: {
      const float sg = 1.0f / (1.0f + __expf(-g));
      return sg * (1.0f - sg);
    }
    case 1: {
      const float sg = 1.0f / (1.0f + __expf(-g));
      return sg * (1.0f + g * (1.0f - sg));
    }
    case 2: return 0.5f * (1.0f + fast_erf(g * 0.70710678118654752440f)) + g * 0.39894228040143267794f * __expf(-0.5f * g * g);
    default: return g > 0.f ? 1.0f : 0.f;
  }
}

// BWD = false: out[r, c] = act(gv[r, c]) * gv[r, D + c].   BWD = true: d_gv[r, c] = d_out * value * act'(gate),
// d_gv[r, D + c] = d_out * act(gate)
template <typename T, int VEC, bool BWD>
__global__ __launch_bounds__(256) void glu_kernel(const T* __restrict__ gv, int64_t ldgv, const T* __restrict__ d_out, int64_t lddo,
                                                  T* __restrict__ out, int64_t ldo, int n_rows, int D, int kind) {
  const int per_row = D / VEC;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * per_row) return;
  const int r = (int)(i / per_row), c = (int)(i % per_row) * VEC;
  float g[VEC], v[VEC];
  load_vec<T, VEC>(gv + (int64_t)r * ldgv + c, g);
  load_vec<T, VEC>(gv + (int64_t)r * ldgv + D + c, v);
  if constexpr (!BWD) {
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = glu_act(g[j], kind) * v[j];
    store_vec<T, VEC>(out + (int64_t)r * ldo + c, o);
  } else {
    float d[VEC], dg[VEC], dv[VEC];
    load_vec<T, VEC>(d_out + (int64_t)r * lddo + c, d);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      dg[j] = d[j] * v[j] * glu_act_grad(g[j], kind);
      dv[j] = d[j] * glu_act(g[j], kind);
    }
    store_vec<T, VEC>(out + (int64_t)r * ldo + c, dg);
    store_vec<T, VEC>(out + (int64_t)r * ldo + D + c, dv);
  }
}

template <typename T>
int pick_vec(int D, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
  int vec = 16 / (int)sizeof(T);
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

int pick_chunks(int D, int vec) {
  for (int ch = 1; ch <= kMaxChunks; ch *= 2)
    if (D <= 64 * vec * ch) return ch;
  return 0;
}

#define ALL_VEC_CH(M) \
  M(1, 1) M(1, 2) M(1, 4) M(1, 8) M(2, 1) M(2, 2) M(2, 4) M(2, 8) M(4, 1) M(4, 2) M(4, 4) M(4, 8) M(8, 1) M(8, 2) M(8, 4) M(8, 8)

template <typename T, int MODE>
int launch_rowwise_bwd(const void* x, int64_t ldx, const void* gamma, const void* dy, int64_t lddy, void* dx, int64_t lddx,
                       float* out0, float* out1, float* ws, int n_rows, int D, float eps, hipStream_t st) {
  const int vec = MODE == 0 ? pick_vec<T>(D, {ldx, lddy, lddx}, {x, dy, dx, gamma}) : pick_vec<T>(D, {ldx}, {x});
  const int ch = pick_chunks(D, vec);
  ANEMOI_REQUIRE(ch > 0, "rowwise backward: D=%d too large for the register-resident row", D);
  int blocks = (n_rows + kBwdWaves - 1) / kBwdWaves;
  blocks = blocks < kPartialBlocks ? blocks : kPartialBlocks;
  const bool want_sums = out0 != nullptr || out1 != nullptr;
  if (n_rows == 0) {  // no rows: the sums are zero
    if (!want_sums) return ANEMOI_OK;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * D + 15) / 16), dim3(1024), 0, st, ws, 0, D, out0, out1);
    return check_launch("reduce_partials_kernel");
  }
#define RB_CASE(V, C)                                                                                                          \
  case V * 16 + C:                                                                                                             \
    hipLaunchKernelGGL((rowwise_bwd_kernel<T, V, C, MODE>), dim3(blocks), dim3(64 * kBwdWaves), 2 * D * sizeof(float), st, (const T*)x, ldx, \
                       (const T*)gamma, (const T*)dy, lddy, (T*)dx, lddx, want_sums ? ws : nullptr, n_rows, D, eps);           \
    break;
  switch (vec * 16 + ch) {
    ALL_VEC_CH(RB_CASE)
    default: set_error("rowwise backward: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef RB_CASE
  int rc = check_launch("rowwise_bwd_kernel");
  if (rc != ANEMOI_OK || !want_sums) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * D + 15) / 16), dim3(1024), 0, st, ws, blocks, D, out0, out1);
  return check_launch("reduce_partials_kernel");
}

template <typename T>
int launch_gelu_bwd(const void* pre, int64_t ldp, const void* dy, int64_t lddy, void* dpre, int64_t lddp, int n_rows, int D, hipStream_t st) {
  const int vec = pick_vec<T>(D, {ldp, lddy, lddp}, {pre, dy, dpre});
  const int64_t n = (int64_t)n_rows * (D / vec);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define GB_CASE(V)                                                                                                          \
  case V:                                                                                                                   \
    hipLaunchKernelGGL((gelu_bwd_kernel<T, V>), grid, block, 0, st, (const T*)pre, ldp, (const T*)dy, lddy, (T*)dpre, lddp, n_rows, D); \
    break;
  switch (vec) {
    GB_CASE(1) GB_CASE(2) GB_CASE(4) GB_CASE(8)
    default: set_error("gelu_bwd: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef GB_CASE
  return check_launch("gelu_bwd_kernel");
}

template <typename T>
int launch_gelu_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int n_rows, int D, hipStream_t st) {
  const int vec = pick_vec<T>(D, {ldx, ldy}, {x, y});
  const int64_t n = (int64_t)n_rows * (D / vec);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define GF_CASE(V)                                                                                            \
  case V:                                                                                                     \
    hipLaunchKernelGGL((gelu_fwd_kernel<T, V>), grid, block, 0, st, (const T*)x, ldx, (T*)y, ldy, n_rows, D); \
    break;
  switch (vec) {
    GF_CASE(1) GF_CASE(2) GF_CASE(4) GF_CASE(8)
    default: set_error("gelu_fwd: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef GF_CASE
  return check_launch("gelu_fwd_kernel");
}

template <typename T>
int launch_segment_sum(const void* x, int64_t ldx, const int32_t* ptr, const int32_t* ids, void* out, int64_t ldo, int n_out, int D, hipStream_t st) {
  const int vec = pick_vec<T>(D, {ldx, ldo}, {x, out});
  const dim3 grid((n_out + kWaves - 1) / kWaves), block(64 * kWaves);
#define SS_CASE(V)                                                                                                          \
  case V:                                                                                                                   \
    hipLaunchKernelGGL((segment_sum_rows_kernel<T, V>), grid, block, 0, st, (const T*)x, ldx, ptr, ids, (T*)out, ldo, n_out, D); \
    break;
  switch (vec) {
    SS_CASE(1) SS_CASE(2) SS_CASE(4) SS_CASE(8)
    default: set_error("segment_sum_rows: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef SS_CASE
  return check_launch("segment_sum_rows_kernel");
}

template <typename T>
int launch_gather_add(const void* a, int64_t lda, const void* b, int64_t ldb, const int32_t* idx, void* out, int64_t ldo, int n_out, int D,
                      hipStream_t st) {
  const int vec = pick_vec<T>(D, {lda, ldb, ldo}, {a, b, out});
  const dim3 grid((n_out + kWaves - 1) / kWaves), block(64 * kWaves);
#define GA_CASE(V)                                                                                                          \
  case V:                                                                                                                   \
    hipLaunchKernelGGL((gather_add_rows_kernel<T, V>), grid, block, 0, st, (const T*)a, lda, (const T*)b, ldb, idx, (T*)out, ldo, n_out, D); \
    break;
  switch (vec) {
    GA_CASE(1) GA_CASE(2) GA_CASE(4) GA_CASE(8)
    default: set_error("gather_add_rows: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef GA_CASE
  return check_launch("gather_add_rows_kernel");
}

template <typename T, bool BWD>
int launch_glu(const void* gv, int64_t ldgv, const void* d_out, int64_t lddo, void* out, int64_t ldo, int n_rows, int D, int kind, hipStream_t st) {
  const int vec = BWD ? pick_vec<T>(D, {ldgv, lddo, ldo}, {gv, d_out, out}) : pick_vec<T>(D, {ldgv, ldo}, {gv, out});
  const int64_t n = (int64_t)n_rows * (D / vec);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define GLU_CASE(V)                                                                                                          \
  case V:                                                                                                                    \
    hipLaunchKernelGGL((glu_kernel<T, V, BWD>), grid, block, 0, st, (const T*)gv, ldgv, (const T*)d_out, lddo, (T*)out, ldo, n_rows, D, kind); \
    break;
  switch (vec) {
    GLU_CASE(1) GLU_CASE(2) GLU_CASE(4) GLU_CASE(8)
    default: set_error("glu: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef GLU_CASE
  return check_launch("glu_kernel");
}

template <typename T>
int launch_cond_ln_bwd(const void* x, int64_t ldx, const void* scale, int64_t lds, const void* dy, int64_t lddy, void* dx, int64_t lddx,
                       void* dscale, int64_t ldds, int n_rows, int D, float eps, hipStream_t st) {
  const int vec = pick_vec<T>(D, {ldx, lds, lddy, lddx, ldds}, {x, scale, dy, dx, dscale});
  const int ch = pick_chunks(D, vec);
  ANEMOI_REQUIRE(ch > 0, "cond_layernorm_bwd: D=%d too large for the register-resident row", D);
  const dim3 grid((n_rows + kWaves - 1) / kWaves), block(64 * kWaves);
#define CB_CASE(V, C)                                                                                                        \
  case V * 16 + C:                                                                                                           \
    hipLaunchKernelGGL((cond_layernorm_bwd_kernel<T, V, C>), grid, block, 0, st, (const T*)x, ldx, (const T*)scale, lds,     \
                       (const T*)dy, lddy, (T*)dx, lddx, (T*)dscale, ldds, n_rows, D, eps);                                  \
    break;
  switch (vec * 16 + ch) {
    ALL_VEC_CH(CB_CASE)
    default: set_error("cond_layernorm_bwd: bad vector width"); return ANEMOI_E_INVALID;
  }
#undef CB_CASE
  return check_launch("cond_layernorm_bwd_kernel");
}

}  // namespace
}  // namespace anemoi

using namespace anemoi;

extern "C" int64_t anemoi_reduce_workspace_bytes(int32_t D) { return (int64_t)kPartialBlocks * 2 * (D > 0 ? D : 0) * (int64_t)sizeof(float); }

extern "C" int anemoi_layernorm_bwd(const void* x, int64_t ldx, const void* gamma, const void* d_y, int64_t lddy, void* d_x,
                                    int64_t lddx, float* d_gamma, float* d_beta, float* workspace, int32_t n_rows, int32_t D,
                                    float eps, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldx >= D && lddy >= D && lddx >= D, "layernorm_bwd: bad sizes n_rows=%d D=%d", n_rows, D);
  ANEMOI_REQUIRE(x && gamma && d_y && d_x, "layernorm_bwd: null pointer");
  ANEMOI_REQUIRE(workspace || (!d_gamma && !d_beta), "layernorm_bwd: d_gamma/d_beta need the workspace (anemoi_reduce_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_rowwise_bwd<float, 0>(x, ldx, gamma, d_y, lddy, d_x, lddx, d_gamma, d_beta, workspace, n_rows, D, eps, st);
    case ANEMOI_BF16: return launch_rowwise_bwd<bf16_t, 0>(x, ldx, gamma, d_y, lddy, d_x, lddx, d_gamma, d_beta, workspace, n_rows, D, eps, st);
    case ANEMOI_F16: return launch_rowwise_bwd<f16_t, 0>(x, ldx, gamma, d_y, lddy, d_x, lddx, d_gamma, d_beta, workspace, n_rows, D, eps, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_colsum(const void* x, int64_t ldx, float* out, float* workspace, int32_t n_rows, int32_t D,
                             anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldx >= D, "colsum: bad sizes n_rows=%d D=%d", n_rows, D);
  ANEMOI_REQUIRE(x && out && workspace, "colsum: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_rowwise_bwd<float, 1>(x, ldx, nullptr, nullptr, 0, nullptr, 0, nullptr, out, workspace, n_rows, D, 0.f, st);
    case ANEMOI_BF16: return launch_rowwise_bwd<bf16_t, 1>(x, ldx, nullptr, nullptr, 0, nullptr, 0, nullptr, out, workspace, n_rows, D, 0.f, st);
    case ANEMOI_F16: return launch_rowwise_bwd<f16_t, 1>(x, ldx, nullptr, nullptr, 0, nullptr, 0, nullptr, out, workspace, n_rows, D, 0.f, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_gelu_bwd(const void* pre, int64_t ldp, const void* d_y, int64_t lddy, void* d_pre, int64_t lddp,
                               int32_t n_rows, int32_t D, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldp >= D && lddy >= D && lddp >= D, "gelu_bwd: bad sizes");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(pre && d_y && d_pre, "gelu_bwd: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_gelu_bwd<float>(pre, ldp, d_y, lddy, d_pre, lddp, n_rows, D, st);
    case ANEMOI_BF16: return launch_gelu_bwd<bf16_t>(pre, ldp, d_y, lddy, d_pre, lddp, n_rows, D, st);
    case ANEMOI_F16: return launch_gelu_bwd<f16_t>(pre, ldp, d_y, lddy, d_pre, lddp, n_rows, D, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_gelu_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t n_rows, int32_t D, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldx >= D && ldy >= D, "gelu_fwd: bad sizes");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && y, "gelu_fwd: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_gelu_fwd<float>(x, ldx, y, ldy, n_rows, D, st);
    case ANEMOI_BF16: return launch_gelu_fwd<bf16_t>(x, ldx, y, ldy, n_rows, D, st);
    case ANEMOI_F16: return launch_gelu_fwd<f16_t>(x, ldx, y, ldy, n_rows, D, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_segment_sum_rows(const void* x, int64_t ldx, const int32_t* ptr, const int32_t* ids, void* out, int64_t ldo,
                                       int32_t n_out, int32_t D, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_out >= 0 && D > 0 && ldx >= D && ldo >= D, "segment_sum_rows: bad sizes");
  if (n_out == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && ptr && out, "segment_sum_rows: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_segment_sum<float>(x, ldx, ptr, ids, out, ldo, n_out, D, st);
    case ANEMOI_BF16: return launch_segment_sum<bf16_t>(x, ldx, ptr, ids, out, ldo, n_out, D, st);
    case ANEMOI_F16: return launch_segment_sum<f16_t>(x, ldx, ptr, ids, out, ldo, n_out, D, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_gather_add_rows(const void* a, int64_t lda, const void* b, int64_t ldb, const int32_t* idx, void* out, int64_t ldo,
                                      int32_t n_out, int32_t D, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_out >= 0 && D > 0 && lda >= D && ldb >= D && ldo >= D, "gather_add_rows: bad sizes");
  if (n_out == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(a && b && idx && out, "gather_add_rows: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_gather_add<float>(a, lda, b, ldb, idx, out, ldo, n_out, D, st);
    case ANEMOI_BF16: return launch_gather_add<bf16_t>(a, lda, b, ldb, idx, out, ldo, n_out, D, st);
    case ANEMOI_F16: return launch_gather_add<f16_t>(a, lda, b, ldb, idx, out, ldo, n_out, D, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_transpose_pad(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t n_rows, int32_t n_cols, int32_t n_pad,
                                    anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && n_cols > 0 && n_pad >= n_rows && ldx >= n_cols && ldo >= n_pad, "transpose_pad: bad sizes");
  if (n_pad == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && out, "transpose_pad: null pointer");
  const dim3 grid((n_pad + 63) / 64, (n_cols + 63) / 64), block(256);
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: hipLaunchKernelGGL((transpose_pad_kernel<float>), grid, block, 0, st, (const float*)x, ldx, (float*)out, ldo, n_rows, n_cols, n_pad); break;
    case ANEMOI_BF16: hipLaunchKernelGGL((transpose_pad_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, n_rows, n_cols, n_pad); break;
    case ANEMOI_F16: hipLaunchKernelGGL((transpose_pad_kernel<f16_t>), grid, block, 0, st, (const f16_t*)x, ldx, (f16_t*)out, ldo, n_rows, n_cols, n_pad); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
  return check_launch("transpose_pad_kernel");
}

extern "C" int anemoi_glu_fwd(const void* gate_value, int64_t ldgv, void* out, int64_t ldo, int32_t n_rows, int32_t D, int32_t kind,
                              anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldgv >= 2 * D && ldo >= D && kind >= 0 && kind <= 3, "glu_fwd: bad sizes / kind");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(gate_value && out, "glu_fwd: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_glu<float, false>(gate_value, ldgv, nullptr, 0, out, ldo, n_rows, D, kind, st);
    case ANEMOI_BF16: return launch_glu<bf16_t, false>(gate_value, ldgv, nullptr, 0, out, ldo, n_rows, D, kind, st);
    case ANEMOI_F16: return launch_glu<f16_t, false>(gate_value, ldgv, nullptr, 0, out, ldo, n_rows, D, kind, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_glu_bwd(const void* gate_value, int64_t ldgv, const void* d_out, int64_t lddo, void* d_gate_value, int64_t lddgv,
                              int32_t n_rows, int32_t D, int32_t kind, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldgv >= 2 * D && lddo >= D && lddgv >= 2 * D && kind >= 0 && kind <= 3, "glu_bwd: bad sizes / kind");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(gate_value && d_out && d_gate_value, "glu_bwd: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_glu<float, true>(gate_value, ldgv, d_out, lddo, d_gate_value, lddgv, n_rows, D, kind, st);
    case ANEMOI_BF16: return launch_glu<bf16_t, true>(gate_value, ldgv, d_out, lddo, d_gate_value, lddgv, n_rows, D, kind, st);
    case ANEMOI_F16: return launch_glu<f16_t, true>(gate_value, ldgv, d_out, lddo, d_gate_value, lddgv, n_rows, D, kind, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}

extern "C" int anemoi_cond_layernorm_bwd(const void* x, int64_t ldx, const void* scale, int64_t lds, const void* d_y, int64_t lddy,
                                         void* d_x, int64_t lddx, void* d_scale, int64_t ldds, int32_t n_rows, int32_t D, float eps,
                                         anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && D > 0 && ldx >= D && lds >= D && lddy >= D && lddx >= D && ldds >= D, "cond_layernorm_bwd: bad sizes");
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && scale && d_y && d_x && d_scale, "cond_layernorm_bwd: null pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_cond_ln_bwd<float>(x, ldx, scale, lds, d_y, lddy, d_x, lddx, d_scale, ldds, n_rows, D, eps, st);
    case ANEMOI_BF16: return launch_cond_ln_bwd<bf16_t>(x, ldx, scale, lds, d_y, lddy, d_x, lddx, d_scale, ldds, n_rows, D, eps, st);
    case ANEMOI_F16: return launch_cond_ln_bwd<f16_t>(x, ldx, scale, lds, d_y, lddy, d_x, lddx, d_scale, ldds, n_rows, D, eps, st);
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
}
