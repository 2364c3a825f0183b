// Weight gradient of a linear layer for gfx950:  dW[o, i] = sum_r dZ[r, o] * X[r, i]   (autograd of torch.nn.Linear as used by
// the reference's blocks, models/src/anemoi/models/layers/block.py:623-635, mlp.py:158-169 — "next" row f1 of the scope table).
//
// Both operands are stored with the REDUCTION index (the node / edge row r) outermost, i.e. the MFMA K dimension is strided in
// memory.  Instead of transposing dZ and X in HBM first (two extra passes over the activations) the tiles travel HBM -> LDS
// row-major by LDS-DMA and the fragments are read with the hardware transpose read ds_read_b64_tr_b16:
//
//  * tile 128 (o) x 128 (i) x 64 rows per step, 256 threads = 4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16x32 tiles;
//  * one DMA piece (64 lanes x 16 B = 1 KiB) is an [8 rows][64 columns] block: a lane fetches 8 consecutive columns of one
//    row, 8 lanes cover 128 contiguous bytes of that row.  Within a piece the 16-byte granule c of row k sits at column slot
//    c ^ (((k >> 1) & 3) << 1): the eight 32-byte row segments that two 16-lane groups of a transpose read touch then fall
//    into eight different 32-byte bank windows (conflict-free for the 64-bank ds_read_b64 classes);
//  * one ds_read_b64_tr_b16 hands every lane 4 reduction rows of its column; two of them make the 8-row operand of one MFMA.
//    The row -> operand-element assignment is the same permutation for both operands, which is all a dot product needs;
//  * rows past n_rows and columns past O / I are fetched from a zeroed global line (the source address of a DMA lane is free),
//    so ragged sizes need no masking in the loop;
//  * the bias gradient (column sums of dZ) is one extra MFMA per 16 columns against an all-ones operand - the loop is bound by
//    the LDS-DMA, so it is free and replaces a separate two-kernel column reduction per layer;
//  * the reduction is split over workgroups (a 10k-row, 2048x512 gradient has only 64 tiles for 256 CUs); every split writes
//    its fp32 partial tile and a second kernel sums the partials in fixed order and converts: deterministic, no atomics.
#include <stdlib.h>

#include "common.h"

namespace anemoi {
namespace {

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int kPiece = 1024;                 // bytes per DMA piece: [8 rows][64 cols] of 16-bit

using frag8 = __attribute__((ext_vector_type(8))) short;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __attribute__((aligned(256))) unsigned char g_zero_line[256];  // zero-initialised: source of out-of-range lanes

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

struct WgradArgs {
  const void* dz;  // [n_rows, O]
  int64_t lddz;
  const void* x;  // [n_rows, I]
  int64_t ldx;
  float* partial;  // [splits][O][I], then (bias gradient) [splits][O]
  int n_rows, O, I;
  int tiles_m, tiles_n, splits, steps_per_split;
};

template <typename T>
__device__ __forceinline__ frag8 ones_frag();
template <>
__device__ __forceinline__ frag8 ones_frag<bf16_t>() {
  return frag8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
}
template <>
__device__ __forceinline__ frag8 ones_frag<f16_t>() {
  return frag8{0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00};
}

// BIAS: the column sums of dZ (the bias gradient) ride along as one more MFMA per 16 columns against an all-ones operand in
// the waves that own the first 64 input columns of the first tile column - the kernel is bound by the LDS-DMA, not the MFMAs.
// TKS rows per LDS stage (64 or 32), STAGES-deep ring: the DMA is bound by bytes in flight x latency, so more, smaller stages in
// the same 64 KiB keep more of it in flight.
template <typename T, bool BIAS, int TKS, int STAGES>
__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradArgs a) {
  constexpr int kOpBytes = TKS * TM * 2;      // one operand of one stage: TKS / 8 row groups x 2 column halves of 1 KiB
  constexpr int kStage = 2 * kOpBytes;
  constexpr int PPW = TKS / 16;               // pieces per wave per operand per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  const int tiles = a.tiles_m * a.tiles_n;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int m0 = (tile / a.tiles_n) * TM, n0 = (tile % a.tiles_n) * TN;
  const int nk_total = (a.n_rows + TKS - 1) / TKS;
  const int kt0 = split * a.steps_per_split * (TK / TKS);  // the plan counts 64-row steps
  const int nk = min(a.steps_per_split * (TK / TKS), nk_total - kt0);

  // ---- DMA assignment: wave w fills pieces PPW w .. PPW w + PPW - 1 of each operand: piece p = row group p >> 1, column half p & 1
  const int k_in = lane >> 3;
  const int gran = (lane & 7) ^ (((k_in >> 1) & 3) << 1);  // granule (8 columns) this lane fetches for its LDS slot
  const unsigned char* zero = g_zero_line + (lane & 7) * 16;
  const unsigned char* dzp = (const unsigned char*)a.dz;
  const unsigned char* xp = (const unsigned char*)a.x;
  int64_t a_col[2], b_col[2];  // byte offset of the lane's granule inside a row, or -1
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int m = m0 + h * 64 + gran * 8, n = n0 + h * 64 + gran * 8;
    a_col[h] = m < a.O ? (int64_t)m * 2 : -1;
    b_col[h] = n < a.I ? (int64_t)n * 2 : -1;
  }
  const uint32_t smem_l = (uint32_t)(size_t)(lds_void_t*)smem;
  const int64_t a_row_bytes = a.lddz * 2, b_row_bytes = a.ldx * 2;

  auto issue = [&](int kt, int stage) {
    const uint32_t base = smem_l + stage * kStage + wave * PPW * kPiece;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int row = (kt0 + kt) * TKS + ((wave * PPW + j) >> 1) * 8 + k_in;
      const bool rv = row < a.n_rows;
      const unsigned char* sa = (rv && a_col[j & 1] >= 0) ? dzp + row * a_row_bytes + a_col[j & 1] : zero;
      const unsigned char* sb = (rv && b_col[j & 1] >= 0) ? xp + row * b_row_bytes + b_col[j & 1] : zero;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)sa, (lds_void_t*)(size_t)(base + j * kPiece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)sb, (lds_void_t*)(size_t)(base + kOpBytes + j * kPiece), 16, 0, 0);
    }
  };

  // ---- fragment read offsets: 16-lane group g reads rows (g & 1) * 4 + (i >> 2) of row group (g >> 1) [+ 2 r + 4 kb]
  const int g = lane >> 4, i16 = lane & 15, q = i16 & 3;
  const int fk = (g & 1) * 4 + (i16 >> 2);
  const int s2 = (fk >> 1) & 3;
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int slot = ((t ^ s2) << 1) | (q >> 1);
    const uint32_t in_piece = fk * 128 + slot * 16 + (q & 1) * 8;
    a_off[t] = ((g >> 1) * 2 + wr) * kPiece + in_piece;
    b_off[t] = kOpBytes + ((g >> 1) * 2 + wc) * kPiece + in_piece;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool bias_wave = BIAS && n0 == 0 && wc == 0;

  auto read_frag = [&](uint32_t addr) -> frag8 {
    // rows r*16 .. r*16+15 of a 32-row MFMA block: row groups +0 / +2 -> +0 / +4 KiB
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(size_t)addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(size_t)(addr + 4 * kPiece));
    return frag8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  auto compute = [&](int stage) {
    const uint32_t base = smem_l + stage * kStage;
#pragma unroll
    for (int kb = 0; kb < TKS / 32; ++kb) {  // 32-row MFMA blocks: row groups 4 kb .. 4 kb + 3 -> + 8 KiB
      frag8 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = read_frag(base + kb * 8 * kPiece + a_off[t]);
        fb[t] = read_frag(base + kb * 8 * kPiece + b_off[t]);
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = mfma16<T>(fb[tn], fa[tm], acc[tm][tn]);  // D[i-col][o-row]: 4 consecutive i per lane
      if constexpr (BIAS) {
        if (bias_wave) {
#pragma unroll
          for (int tm = 0; tm < 4; ++tm) accb[tm] = mfma16<T>(ones_frag<T>(), fa[tm], accb[tm]);  // every row = column sums
        }
      }
    }
  };

  // ring: stages kt+1 .. kt+STAGES-1 are in flight while stage kt is consumed
#pragma unroll
  for (int p = 0; p < STAGES - 1; ++p)
    if (p < nk) issue(p, p);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + STAGES - 2 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * 2 * PPW) : "memory");  // all but the newest STAGES-2 steps landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // everybody's pieces of step kt are in LDS, and everybody finished reading step kt-1 (raw: a
                                   // __syncthreads() would drain the DMA queue)
    if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    compute(kt % STAGES);
  }

  // ---- partial tile: lane holds dW[m][n .. n+3], m = .. + (lane & 15), n = .. + 4 (lane >> 4)
  float* __restrict__ out = a.partial + (int64_t)split * a.O * a.I;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + wr * 64 + tm * 16 + i16;
    if (m >= a.O) continue;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = n0 + wc * 64 + tn * 16 + 4 * g;
      if (n < a.I) *reinterpret_cast<f32x4*>(out + (int64_t)m * a.I + n) = acc[tm][tn];
    }
  }
  if constexpr (BIAS) {
    if (bias_wave && g == 0) {
      float* __restrict__ outb = a.partial + (int64_t)a.splits * a.O * a.I + (int64_t)split * a.O;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        const int m = m0 + wr * 64 + tm * 16 + i16;
        if (m < a.O) outb[m] = accb[tm][0];
      }
    }
  }
}

// dw[m][n] = sum over splits (fixed order) of partial[s][m][n], converted to T, 4 columns per thread; the threads past the
// matrix do the same for the bias gradient db[m].
template <typename T>
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, T* __restrict__ dw, int64_t lddw, T* __restrict__ db, int O, int I,
                                    int splits) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t quads = (int64_t)O * I / 4;
  if (idx >= quads) {
    const int64_t m = idx - quads;
    if (db == nullptr || m >= O) return;
    const float* pb = partial + (int64_t)splits * O * I + m;
    float s = pb[0];
    for (int p = 1; p < splits; ++p) s += pb[(int64_t)p * O];
    db[m] = from_float<T>(s);
    return;
  }
  const int64_t e = idx * 4;
  f32x4 s = *reinterpret_cast<const f32x4*>(partial + e);
  for (int p = 1; p < splits; ++p) s += *reinterpret_cast<const f32x4*>(partial + (int64_t)p * O * I + e);
  const int m = (int)(e / I), n = (int)(e % I);
  const float v[4] = {s[0], s[1], s[2], s[3]};
  store_vec<T, 4>(dw + (int64_t)m * lddw + n, v);
}

struct Plan {
  int tiles_m, tiles_n, splits, steps_per_split;
};

Plan make_plan(int n_rows, int O, int I) {
  Plan p;
  p.tiles_m = (O + TM - 1) / TM;
  p.tiles_n = (I + TN - 1) / TN;
  const int nk = (n_rows + TK - 1) / TK, tiles = p.tiles_m * p.tiles_n;
  static const int target = [] {  // workgroups to aim for: two per CU (developer override ANEMOI_WGRAD_WGS)
    const char* e = getenv("ANEMOI_WGRAD_WGS");
    const int v = e != nullptr ? atoi(e) : 0;
    return v > 0 ? v : 512;
  }();
  int want = (target + tiles - 1) / tiles;
  want = want < 1 ? 1 : want;
  const int max_splits = nk / 4 > 0 ? nk / 4 : 1;  // at least four steps per split
  if (want > max_splits) want = max_splits;
  p.steps_per_split = (nk + want - 1) / want;
  p.splits = (nk + p.steps_per_split - 1) / p.steps_per_split;
  return p;
}

hipStream_t as_hip_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T, bool BIAS, int TKS, int STAGES>
int launch_wgrad_cfg(const WgradArgs& a, hipStream_t st) {
  constexpr int lds = STAGES * 2 * TKS * TM * 2;
  static PerDeviceOnce attr_once;
  attr_once.run([&] { (void)hipFuncSetAttribute((const void*)wgrad_tn_kernel<T, BIAS, TKS, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
  hipLaunchKernelGGL((wgrad_tn_kernel<T, BIAS, TKS, STAGES>), dim3(a.tiles_m * a.tiles_n * a.splits), dim3(256), lds, st, a);
  return check_launch("wgrad_tn_kernel");
}

template <typename T, bool BIAS>
int launch_wgrad_kernel(const WgradArgs& a, hipStream_t st) {
  static const int cfg = [] {  // developer switch: 0 = 2 stages of 64 rows, 1 = 4 stages of 32 rows (same 64 KiB)
    const char* e = getenv("ANEMOI_WGRAD_RING");
    return e != nullptr ? atoi(e) : 0;  // measured: the deeper ring is 5-15 % slower (twice the barriers)
  }();
  return cfg == 0 ? launch_wgrad_cfg<T, BIAS, 64, 2>(a, st) : launch_wgrad_cfg<T, BIAS, 32, 4>(a, st);
}

template <typename T>
int launch_wgrad(const WgradArgs& a, T* dw, int64_t lddw, T* db, hipStream_t st) {
  const int rc = db != nullptr ? launch_wgrad_kernel<T, true>(a, st) : launch_wgrad_kernel<T, false>(a, st);
  if (rc != ANEMOI_OK) return rc;
  const int64_t threads = (int64_t)a.O * a.I / 4 + (db != nullptr ? a.O : 0);
  hipLaunchKernelGGL((wgrad_reduce_kernel<T>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a.partial, dw, lddw, db, a.O, a.I,
                     a.splits);
  return check_launch("wgrad_reduce_kernel");
}

}  // namespace
}  // namespace anemoi

using namespace anemoi;

extern "C" int64_t anemoi_linear_wgrad_workspace_bytes(int32_t n_rows, int32_t O, int32_t I) {
  if (n_rows <= 0 || O <= 0 || I <= 0) return 0;
  const Plan p = make_plan(n_rows, O, I);
  return (int64_t)p.splits * ((int64_t)O * I + O) * (int64_t)sizeof(float);
}

extern "C" int anemoi_linear_wgrad(const void* dz, int64_t lddz, const void* x, int64_t ldx, void* dw, int64_t lddw, void* db, void* workspace,
                                   int32_t n_rows, int32_t O, int32_t I, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_rows > 0 && O > 0 && I > 0, "linear_wgrad: bad sizes n_rows=%d O=%d I=%d", n_rows, O, I);
  ANEMOI_REQUIRE(dz && x && dw && workspace, "linear_wgrad: null pointer");
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "linear_wgrad: 16-bit operands only");
  ANEMOI_REQUIRE(O % 8 == 0 && I % 8 == 0 && lddz % 8 == 0 && ldx % 8 == 0 && lddz >= O && ldx >= I && lddw >= I && lddw % 4 == 0,
                 "linear_wgrad: O, I and the operand row strides must be multiples of 8 (16-byte granules)");
  ANEMOI_REQUIRE(reinterpret_cast<uintptr_t>(dz) % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(dw) % 8 == 0 && reinterpret_cast<uintptr_t>(workspace) % 16 == 0,
                 "linear_wgrad: operands must be 16-byte aligned");
  const Plan p = make_plan(n_rows, O, I);
  WgradArgs a{dz, lddz, x, ldx, (float*)workspace, n_rows, O, I, p.tiles_m, p.tiles_n, p.splits, p.steps_per_split};
  hipStream_t st = as_hip_stream(stream);
  return dtype == ANEMOI_BF16 ? launch_wgrad<bf16_t>(a, (bf16_t*)dw, lddw, (bf16_t*)db, st) : launch_wgrad<f16_t>(a, (f16_t*)dw, lddw, (f16_t*)db, st);
}
