// Row-resident embedding chain for gfx950 (round 6): one side of a GraphTransformer mapper in ONE launch -
//
//     y     = x W_e^T + b_e                         (emb_nodes_src / emb_nodes_dst: Linear(in, 512), layers/mapper.py:480-597, 600-704 of the reference)
//     q_out = LN(y) [W_a; W_b ...]^T + b            (the block's layer_norm_attention_src + [lin_key; lin_value], or layer_norm_attention_dest +
//                                                    [lin_query; lin_self]: layers/block.py:981-984)
//
// instead of the embedding GEMM (which wrote y and its row statistics) + the LayerNorm-fold GEMM (which read y back): for the 40 320-row sides
// of the O96 mappers y is a 41-MB round trip through HBM, and on the SOURCE side of the encoder it is needed by nobody else (the block
// returns the source rows untouched): with x_out = NULL y never exists in memory.
//
// The machinery is csrc/gt_chain2.hip's (chain2_core.h): a workgroup keeps a panel of <= 48 rows in LDS, weights are fragment-major images
// streamed L2 -> registers -> MFMA, the LayerNorm is the plain fp32 LayerNorm of the ROUNDED 16-bit rows applied without its affine part (the
// caller folds gamma / beta into the projection: wq = image of W diag(gamma), dq = W beta + b), accumulators start at their bias.  Per panel:
//
//     S0  all: x rows -> bufA (columns beyond in_features zero: the image of W_e is zero-padded to a multiple of 128 columns)
//     E   all eight waves: y = x W_e^T (48 x 64 tile per wave, K = 128 ng) + b_e, rounded -> bufC, per-wave row statistics
//     L   all: LayerNorm (no affine) of y from registers -> bufB;  group B: y rows -> global (if wanted);  the NEXT panel's x rows requested
//     Q_c group A: chunk 2c, group B: chunk 2c+1 of the projection: acc = dq[chunk]; GEMM on bufB; rounded -> the group's staging buffer (A: bufA,
//         B: bufC, each wave its own 128 columns) -> whole 256-byte row pieces to global
#include "chain2_core.h"

namespace anemoi {

struct RowChainArgs {
  const void* x;   int64_t ld_x;  int k_in;  // [n_rows, k_in] input rows (k_in % 8 == 0, <= 512)
  const char* we;  int ng;                   // embedding, fragment-major [512, 128 ng] (zero columns beyond k_in)
  const char* wq;  int qc;                   // projection with the LayerNorm's gamma folded in, fragment-major [512 qc, 512]
  const void* vec;                           // [b_e (512) | dq (512 qc)], model dtype
  float eps;
  void* xout;      int64_t ld_out;           // optional [n_rows, 512]: y
  void* qout;      int64_t ld_q;             // [n_rows, 512 qc]
  int n_rows, rows_per_tile, n_tiles;
};
constexpr int kRcRedOff = 3 * kBufBytes;                   // [48 rows][8 waves][2] fp32 LayerNorm partials
constexpr int kRcVecOff = kRcRedOff + kPanel * 8 * 2 * 4;  // the per-column vectors (16-bit): 512 + 512 qc <= 2560
constexpr int kRcVecMax = 2560;
constexpr int kRowChainSmem = kRcVecOff + kRcVecMax * 2;
static_assert(kRowChainSmem <= 160 * 1024, "LDS budget");

// The wave's 48 x 64 block (acc[mi][0..3]) + vec[column], rounded to the model dtype into the panel buffer `dst`; acc keeps the ROUNDED
// values; per-wave (mean, M2) of every row over the wave's 64 columns -> red[row][w8]   (round_rows64_add_stats without the skip rows)
template <typename T>
__device__ __forceinline__ void round_rows64_bias_stats(f32x4 (&acc)[3][8], unsigned char* dst, float* red, int lane, int w8, const unsigned char* vec) {
  const LaneCtx lc = lane_ctx(lane, w8);
  u32x2 rb[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) rb[ni] = *reinterpret_cast<const u32x2*>(vec + (w8 * 64 + ni * 16 + lc.g * 4) * 2);
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) {
    unsigned char* drow = dst + (mi * 16 + lc.x) * kRowBytes;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float o[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      float b[4];
      unpack4<T>(rb[ni], b);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] += b[k];
      const u32x2 pk = pack4<T>(o);
      *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pk;
      unpack4<T>(pk, o);
      acc[mi][ni] = f32x4{o[0], o[1], o[2], o[3]};
    }
    float s = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) s += (acc[mi][ni][0] + acc[mi][ni][1]) + (acc[mi][ni][2] + acc[mi][ni][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mw = s * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[mi][ni][r] - mw;
        q = fmaf(d, d, q);
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (lc.g == 0) *reinterpret_cast<float2*>(red + ((mi * 16 + lc.x) * 8 + w8) * 2) = make_float2(mw, q);
  }
}

// A panel of input rows: 48 rows x spr = 16 ng sixteen-byte slots (slots beyond the row's k_in / 8 are zero), shared out among the 512
// threads (<= 6 slots each), requested into registers and stored to the swizzled panel later - the request of the NEXT panel rides under
// the projection GEMMs of this one.
struct XRows {
  u32x4 v[6];
  __device__ __forceinline__ void request(const void* x, int64_t ld, int k_in, int ng, int r0, int nr, int tid, int es) {
    const int spr = 16 * ng, n = kPanel * spr, kin16 = k_in >> 3;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = tid + 512 * k;
      if (k * 512 < n) {  // (wave-uniform)
        const int row = min(i / spr, kPanel - 1), slot = i % spr;
        const bool live = row < nr && slot < kin16 && i < n;
        const unsigned char* p = reinterpret_cast<const unsigned char*>(x) + ((int64_t)(r0 + min(row, nr - 1)) * ld + min(slot, kin16 - 1) * 8) * es;
        const u32x4 t = stream_load(reinterpret_cast<const u32x4*>(p));
        v[k] = live ? t : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* buf, int ng, int tid) {
    const int spr = 16 * ng, n = kPanel * spr;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = tid + 512 * k;
      if (i < n) {
        const int row = i / spr, slot = i % spr;
        *reinterpret_cast<u32x4*>(buf + row * kRowBytes + ((slot ^ (row & 15)) << 4)) = v[k];
      }
    }
  }
};

template <typename T>
__global__ __launch_bounds__(512, 1) void gt_rowchain_kernel(RowChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const bufA = smem;
  unsigned char* const bufB = smem + kBufBytes;
  unsigned char* const bufC = smem + 2 * kBufBytes;
  float* const red = reinterpret_cast<float*>(smem + kRcRedOff);
  const unsigned char* const vec = smem + kRcVecOff;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6), wq = w8 & 3, grp = w8 >> 2;  // waves wq and wq + 4 share a SIMD
  const uint32_t loff = lane * 16;
  const int qc = a.qc, ng = a.ng;
  const int64_t se = (int64_t)ng * 16384;  // one 64-column slab of the embedding image: 4 ng K-steps x 4 KiB
  const char* const wes = a.we + (int64_t)w8 * se;
  auto wqc = [&](int k) { return a.wq + (int64_t)(8 * k + 2 * wq) * kSlab; };
  int tile = blockIdx.x;
  if (tile >= a.n_tiles) return;
  frag8 ring[2][8];
  f32x4 acc[3][8];
  XRows xr;
  // the first panel's rows, then the per-column vectors and the weight ring's first fragments behind them (loads return in order)
  {
    const int r0 = tile * a.rows_per_tile;
    xr.request(a.x, a.ld_x, a.k_in, ng, r0, min(a.rows_per_tile, a.n_rows - r0), tid, (int)sizeof(T));
    const int n16 = (512 + 512 * qc) / 8;  // <= 320
    u32x4 vv = reinterpret_cast<const u32x4*>(a.vec)[min(tid, n16 - 1)];
    ring_prologue64(ring, wes, loff);
    xr.store(bufA, ng, tid);
    if (tid < n16) reinterpret_cast<u32x4*>(smem + kRcVecOff)[tid] = vv;
    lds_barrier();
  }
  const bool mine_any = grp < qc;  // this group has at least one chunk of the projection
  for (;;) {
    const int r0 = tile * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    // E: y = x W_e^T + b_e -> bufC (rounded), row statistics
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm64<T>(bufA, lane, ring, wes, mine_any ? wqc(grp) : wes, mine_any ? (int64_t)kSlab : (int64_t)8192, loff, acc, ng);
    round_rows64_bias_stats<T>(acc, bufC, red, lane, w8, vec);
    lds_barrier();  // y and the partials are complete; every wave is behind its last read of the x rows
    // L: LayerNorm (no affine) -> bufB; y -> global by group B (each wave its own 128 columns: the columns it will stage its chunk in)
    normalise_rows64<T>(acc, red, a.eps, bufB, lane, w8);
    if (a.xout != nullptr && grp == 1) store_staged<T>(bufC, (T*)a.xout + (int64_t)r0 * a.ld_out, a.ld_out, nr, lane, wq);
    const int tile_next = tile + (int)gridDim.x;
    const bool more = tile_next < a.n_tiles;
    if (more) {
      const int rn = tile_next * a.rows_per_tile;
      xr.request(a.x, a.ld_x, a.k_in, ng, rn, min(a.rows_per_tile, a.n_rows - rn), tid, (int)sizeof(T));
    }
    lds_barrier();
    // Q: this group's chunks of the projection
    unsigned char* const stage = grp == 0 ? bufA : bufC;
    for (int k = grp; k < qc; k += 2) {
      init_acc<T, false>(acc, vec, 512 + 512 * k, nullptr, lane, wq);
      const bool last = k + 2 >= qc;
      gemm128<T>(bufB, lane, ring, wqc(k), kSlab, last ? wes : wqc(k + 2), last ? (int64_t)8192 : (int64_t)kSlab, loff, acc);
      round_rows<T, false>(acc, stage, nullptr, lane, wq);
      store_staged<T>(stage, (T*)a.qout + (int64_t)r0 * a.ld_q + k * kCh, a.ld_q, nr, lane, wq);
    }
    lds_barrier();  // every wave is behind its last read of bufB and of its staging columns
    if (!more) break;
    tile = tile_next;
    xr.store(bufA, ng, tid);
    lds_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------- several panel rounds: pipelined
// With more panels than CUs a workgroup walks a LIST of panels, and the single-panel schedule above spends ~21 us per round of which only ~12 are the
// weight stream (1.2 MB per panel through the CU's L1 path).  Here the two wave groups work on DIFFERENT panels: group A (waves 0-3, 128 columns each)
// computes y = x W_e^T + b_e, its statistics and LayerNorm for panel s while group B (waves 4-7) runs the projection chunks of panel s - 1:
//
//     phase 1   A: LN(y)(s-1), parked in its accumulators since the last step -> bufN; the rows of x(s), requested a step ago -> bufX
//     phase 2   A: request x(s+1); acc = b_e; GEMM on bufX; rounded y in registers + row statistics     B: chunk 0 of panel s-1 on bufN -> staged -> global
//     phase 3   A: y -> global (staged through bufX, if wanted); LN(y) in registers                    B: chunks 1 .. of panel s-1
//
// one s_barrier behind each phase (both groups: the hardware barrier counts all eight waves), n + 1 steps for n panels.  LDS: bufX, bufN, group B's
// staging buffer (48 KB each) + the partials + the vectors.  in_features <= 256 (the parked rows of x(s+1) are 6 registers per lane).
constexpr int kRc2Red = 3 * kBufBytes;                  // [48 rows][4 waves][2] fp32
constexpr int kRc2Vec = kRc2Red + kPanel * 4 * 2 * 4;
constexpr int kRowChain2Smem = kRc2Vec + kRcVecMax * 2;
static_assert(kRowChain2Smem <= 160 * 1024, "LDS budget");

struct XRowsA {  // a panel of input rows shared out among group A's 256 threads: <= 6 sixteen-byte slots each (in_features <= 256)
  u32x4 v[6];
  __device__ __forceinline__ void request(const void* x, int64_t ld, int k_in, int ng, int r0, int nr, int t, int es) {
    const int spr = 16 * ng, n = kPanel * spr, kin16 = k_in >> 3;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = t + 256 * k;
      if (k * 256 < n) {  // (wave-uniform)
        const int row = min(i / spr, kPanel - 1), slot = i % spr;
        const bool live = row < nr && slot < kin16 && i < n;
        const unsigned char* p = reinterpret_cast<const unsigned char*>(x) + ((int64_t)(r0 + min(row, nr - 1)) * ld + min(slot, kin16 - 1) * 8) * es;
        const u32x4 tv = stream_load(reinterpret_cast<const u32x4*>(p));
        v[k] = live ? tv : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* buf, int ng, int t) {
    const int spr = 16 * ng, n = kPanel * spr;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = t + 256 * k;
      if (i < n) {
        const int row = i / spr, slot = i % spr;
        *reinterpret_cast<u32x4*>(buf + row * kRowBytes + ((slot ^ (row & 15)) << 4)) = v[k];
      }
    }
  }
};

struct PipeCtx {
  int lane, wq, tid;
  uint32_t loff;
  int b0, grid, n;  // this workgroup's panels: b0, b0 + grid, ... (n of them)
};
__device__ __forceinline__ void pipe_rows(const RowChainArgs& a, const PipeCtx& c, int s, int& r0, int& nr) {
  r0 = (c.b0 + s * c.grid) * a.rows_per_tile;
  nr = min(a.rows_per_tile, a.n_rows - r0);
}

// Both roles execute the SAME barriers: one behind the prologue, three per step (behind phases 1, 2, 3), n + 1 steps.
template <typename T>
__device__ __forceinline__ void pipe_role_a(const RowChainArgs& a, const PipeCtx& c, unsigned char* smem) {
  unsigned char* const bufX = smem;
  unsigned char* const bufN = smem + kBufBytes;
  float* const red = reinterpret_cast<float*>(smem + kRc2Red);
  const unsigned char* const vec = smem + kRc2Vec;
  const int lane = c.lane, wq = __builtin_amdgcn_readfirstlane(c.wq), ng = a.ng, n = c.n;
  const int64_t se = (int64_t)ng * 16384;  // one 64-column slab of the embedding image
  const char* const wes = a.we + (int64_t)(2 * wq) * se;
  frag8 ring[2][8];
  f32x4 acc[3][8];
  XRowsA xr;
  {
    int r0, nr;
    pipe_rows(a, c, 0, r0, nr);
    xr.request(a.x, a.ld_x, a.k_in, ng, r0, nr, c.tid, (int)sizeof(T));
    ring_prologue(ring, wes, se, c.loff);
    xr.store(bufX, ng, c.tid);
    lds_barrier();
  }
  for (int s = 0; s <= n; ++s) {
    // phase 1: LN(y)(s-1), parked in the accumulators -> bufN (rounded here); the rows of x(s) -> bufX
    if (s >= 1) {
      round_rows<T, false>(acc, bufN, nullptr, lane, wq);
      if (s < n) xr.store(bufX, ng, c.tid);
    }
    lds_barrier();
    // phase 2: y = x W_e^T + b_e, rounded in registers, per-wave row statistics; x(s+1) requested
    if (s < n) {
      if (s + 1 < n) {
        int rn, nrn;
        pipe_rows(a, c, s + 1, rn, nrn);
        xr.request(a.x, a.ld_x, a.k_in, ng, rn, nrn, c.tid, (int)sizeof(T));
      }
      init_acc<T, false>(acc, vec, 0, nullptr, lane, wq);
      gemm128<T>(bufX, lane, ring, wes, se, wes, se, c.loff, acc, 2 * ng);
      round_rows<T, true, false, false>(acc, nullptr, red, lane, wq);
    }
    lds_barrier();
    // phase 3: y -> global (staged through bufX: every wave of the group is behind its last read of the x rows); LN(y) in registers
    if (s < n) {
      if (a.xout != nullptr) {
        int r0, nr;
        pipe_rows(a, c, s, r0, nr);
        round_rows<T, false>(acc, bufX, nullptr, lane, wq);
        store_staged<T>(bufX, (T*)a.xout + (int64_t)r0 * a.ld_out, a.ld_out, nr, lane, wq);
      }
      normalise_regs<T>(acc, red, a.eps, lane, wq);
    }
    lds_barrier();
  }
}

template <typename T>
__device__ __forceinline__ void pipe_role_b(const RowChainArgs& a, const PipeCtx& c, unsigned char* smem) {
  unsigned char* const bufN = smem + kBufBytes;
  unsigned char* const bufS = smem + 2 * kBufBytes;
  const unsigned char* const vec = smem + kRc2Vec;
  const int lane = c.lane, wq = __builtin_amdgcn_readfirstlane(c.wq), qc = a.qc, n = c.n;
  auto wqc = [&](int k) { return a.wq + (int64_t)(8 * k + 2 * wq) * kSlab; };
  frag8 ring[2][8];
  f32x4 acc[3][8];
  ring_prologue(ring, wqc(0), kSlab, c.loff);
  lds_barrier();
  for (int s = 0; s <= n; ++s) {
    int rp, nrp;
    pipe_rows(a, c, s - 1, rp, nrp);
    lds_barrier();  // phase 1 is group A's
    // phase 2: chunk 0 of the projection of panel s - 1
    if (s >= 1) {
      init_acc<T, false>(acc, vec, 512, nullptr, lane, wq);
      gemm128<T>(bufN, lane, ring, wqc(0), kSlab, qc > 1 ? wqc(1) : wqc(0), kSlab, c.loff, acc);
      round_rows<T, false>(acc, bufS, nullptr, lane, wq);
      store_staged<T>(bufS, (T*)a.qout + (int64_t)rp * a.ld_q, a.ld_q, nrp, lane, wq);
    }
    lds_barrier();
    // phase 3: its other chunks
    if (s >= 1) {
      for (int k = 1; k < qc; ++k) {
        init_acc<T, false>(acc, vec, 512 + 512 * k, nullptr, lane, wq);
        gemm128<T>(bufN, lane, ring, wqc(k), kSlab, k + 1 < qc ? wqc(k + 1) : wqc(0), kSlab, c.loff, acc);
        round_rows<T, false>(acc, bufS, nullptr, lane, wq);
        store_staged<T>(bufS, (T*)a.qout + (int64_t)rp * a.ld_q + k * kCh, a.ld_q, nrp, lane, wq);
      }
    }
    lds_barrier();
  }
}

template <typename T>
__global__ __launch_bounds__(512, 1) void gt_rowchain_pipe_kernel(RowChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  PipeCtx c;
  c.tid = tid & 255;
  c.lane = tid & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.wq = w8 & 3;
  c.loff = c.lane * 16;
  c.b0 = (int)blockIdx.x;
  c.grid = (int)gridDim.x;
  if (c.b0 >= a.n_tiles) return;
  c.n = (a.n_tiles - c.b0 + c.grid - 1) / c.grid;
  {  // the per-column vectors -> LDS (visible behind the prologue's barrier)
    const int n16 = (512 + 512 * a.qc) / 8;  // <= 320
    if (tid < n16) reinterpret_cast<u32x4*>(smem + kRc2Vec)[tid] = reinterpret_cast<const u32x4*>(a.vec)[tid];
  }
  if (w8 < 4) pipe_role_a<T>(a, c, smem);
  else pipe_role_b<T>(a, c, smem);
}

template <typename T>
static int launch_rowchain(const RowChainArgs& a, hipStream_t st) {
  static PerDeviceOnce once;
  once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_rowchain_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kRowChainSmem); });
  // several rounds: as many workgroups as make the rounds even (the CUs of an XCD share that L2's bandwidth; gt_chain2.hip's rule)
  int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  if (a.n_tiles > 256) {
    const int rounds = (a.n_tiles + 255) / 256;
    grid = (a.n_tiles + rounds - 1) / rounds;
  }
  if (a.n_tiles > grid && a.k_in <= 256) {  // several rounds of panels: the two wave groups on different panels
    static PerDeviceOnce once2;
    once2.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_rowchain_pipe_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kRowChain2Smem); });
    hipLaunchKernelGGL((gt_rowchain_pipe_kernel<T>), dim3(grid), dim3(512), kRowChain2Smem, st, a);
    return check_launch("gt_rowchain_pipe_kernel");
  }
  hipLaunchKernelGGL((gt_rowchain_kernel<T>), dim3(grid), dim3(512), kRowChainSmem, st, a);
  return check_launch("gt_rowchain_kernel");
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_rowchain_fwd(const anemoi_gt_rowchain_args_t* p, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(p != nullptr, "gt_rowchain_fwd: null argument block");
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "gt_rowchain_fwd: 16-bit model dtypes only");
  ANEMOI_REQUIRE(p->n_rows >= 0 && p->channels == kCh, "gt_rowchain_fwd: channels=%d (this kernel is built for %d)", p->channels, kCh);
  if (p->n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(p->in_features > 0 && p->in_features <= kCh && p->in_features % 8 == 0,
                 "gt_rowchain_fwd: in_features=%d must be a multiple of 8 up to %d (rows move as 16-byte pieces)", p->in_features, kCh);
  ANEMOI_REQUIRE(p->q_out_features > 0 && p->q_out_features % kCh == 0 && p->q_out_features <= 4 * kCh,
                 "gt_rowchain_fwd: q_out_features=%d must be a multiple of %d up to %d", p->q_out_features, kCh, 4 * kCh);
  ANEMOI_REQUIRE(p->x && p->we && p->wq && p->vec && p->q_out, "gt_rowchain_fwd: null operand");
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  ANEMOI_REQUIRE(al16(p->x) && al16(p->we) && al16(p->wq) && al16(p->vec) && al16(p->x_out) && al16(p->q_out), "gt_rowchain_fwd: operands must be 16-byte aligned");
  ANEMOI_REQUIRE(p->ld_x >= p->in_features && p->ld_x % 8 == 0 && p->ld_q >= p->q_out_features && p->ld_q % 8 == 0 &&
                     (p->x_out == nullptr || (p->ld_out >= kCh && p->ld_out % 8 == 0)),
                 "gt_rowchain_fwd: leading dimensions too small or not multiples of 8 elements");
  RowChainArgs a{};
  a.x = p->x; a.ld_x = p->ld_x; a.k_in = p->in_features;
  a.we = (const char*)p->we; a.ng = (p->in_features + 127) / 128;
  a.wq = (const char*)p->wq; a.qc = p->q_out_features / kCh;
  a.vec = p->vec;
  a.eps = p->ln_eps;
  a.xout = p->x_out; a.ld_out = p->ld_out;
  a.qout = p->q_out; a.ld_q = p->ld_q;
  a.n_rows = p->n_rows;
  a.rows_per_tile = p->rows_per_tile > 0 ? p->rows_per_tile : kPanel;
  ANEMOI_REQUIRE(a.rows_per_tile <= kPanel, "gt_rowchain_fwd: rows_per_tile=%d exceeds the %d-row panel", a.rows_per_tile, kPanel);
  a.n_tiles = (a.n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  hipStream_t st = as_stream(stream);
  return dtype == ANEMOI_BF16 ? launch_rowchain<bf16_t>(a, st) : launch_rowchain<f16_t>(a, st);
}
