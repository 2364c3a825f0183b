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

  // Epilogue.  acc[mi][ni][r] = out[m = m0 + wr*64 + mi*16 + (lane & 15)][n = n0 + wc*64 + ni*16 + (lane>>4)*4 + r]
  const T* __restrict__ bias = (const T*)a.bias;
  const T* __restrict__ g1 = (const T*)a.g1;
  const T* __restrict__ g2 = (const T*)a.g2;
  const T* __restrict__ res = (const T*)a.residual;
  T* __restrict__ y = (T*)a.y;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wr * 64 + mi * 16 + (lane & 15);
    if (m >= a.n_rows) continue;
    const int64_t r1 = g1 ? (int64_t)a.idx1[m] * a.ldg1 : 0;
    const int64_t r2 = g2 ? (int64_t)a.idx2[m] * a.ldg2 : 0;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + (lane >> 4) * 4;
      if (n >= a.O) continue;  // O % 4 == 0: a 4-column group is entirely inside or outside
      float vv[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      float t[4];
      if (bias) {
        load_vec<T, 4>(bias + n, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] += t[r];
      }
      if (g1) {
        load_vec<T, 4>(g1 + r1 + n, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] += t[r];
      }
      if (g2) {
        load_vec<T, 4>(g2 + r2 + n, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] += t[r];
      }
      if (a.act == ANEMOI_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] = gelu_erf(vv[r]);
      }
      if (res) {
        load_vec<T, 4>(res + (int64_t)m * a.ldr + n, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] += t[r];
      }
      store_vec<T, 4>(y + (int64_t)m * a.ldy + n, vv);
    }
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
static int launch_mfma(const LinArgs& a, hipStream_t st) {
  const int tiles_m = (a.n_rows + BM - 1) / BM, tiles_n = (a.O + BN - 1) / BN;
  const int num_tiles = tiles_m * tiles_n;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_mfma_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kTileBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL((linear_mfma_kernel<T>), dim3(num_tiles), dim3(256), 4 * kTileBytes, st, a, tiles_n, num_tiles);
  return check_launch("linear_mfma_kernel");
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_linear_fwd(const void* x, int64_t ldx, int32_t K1, const void* x2, int64_t ldx2, int32_t K2,
                                 const void* w, int64_t ldw, const void* bias, const void* g1, int64_t ldg1,
                                 const int32_t* idx1, const void* g2, int64_t ldg2, const int32_t* idx2,
                                 const void* residual, int64_t ldr, void* y, int64_t ldy, int32_t n_rows, int32_t O,
                                 anemoi_act_t act, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && O > 0 && K1 > 0 && K2 >= 0, "linear_fwd: bad sizes n_rows=%d O=%d K1=%d K2=%d", n_rows, O, K1, K2);
  if (n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(x && w && y, "linear_fwd: null x/w/y");
  ANEMOI_REQUIRE((K2 == 0) == (x2 == nullptr), "linear_fwd: x2 and K2 must be given together");
  ANEMOI_REQUIRE(ldx >= K1 && ldw >= K1 + K2 && ldy >= O && (!x2 || ldx2 >= K2), "linear_fwd: leading dimension too small");
  ANEMOI_REQUIRE((g1 == nullptr) == (idx1 == nullptr) && (g2 == nullptr) == (idx2 == nullptr), "linear_fwd: gather term needs both table and index");
  ANEMOI_REQUIRE(act == ANEMOI_ACT_NONE || act == ANEMOI_ACT_GELU, "linear_fwd: unknown activation %d", (int)act);
  LinArgs a{x, ldx, K1, x2, ldx2, K2, w, ldw, bias, g1, ldg1, idx1, g2, ldg2, idx2, residual, ldr, y, ldy, n_rows, O, (int)act};
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ANEMOI_F32: return launch_generic<float>(a, st);
    case ANEMOI_BF16: return mfma_eligible<bf16_t>(a) ? launch_mfma<bf16_t>(a, st) : launch_generic<bf16_t>(a, st);
    case ANEMOI_F16: return mfma_eligible<f16_t>(a) ? launch_mfma<f16_t>(a, st) : launch_generic<f16_t>(a, st);
    default: set_error("unknown dtype %d", (int)dtype); return ANEMOI_E_INVALID;
  }
}
