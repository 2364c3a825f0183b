// Cluster chain for gfx950 (round 6): the row-local tail of a GraphTransformer block - what csrc/gt_chain2.hip computes,
//
//     x1 = attn W_p^T + b_p + x                      (projection + skip,            layers/block.py:1263-1266 of the reference)
//     h  = GELU(LN_mlp(x1) W_1^T + b_1)               (node_dst_mlp, first Linear,   layers/block.py:1268-1271, layers/mlp.py:158-169)
//     x2 = h W_2^T + b_2 + x1 [+ latent skip]        (second Linear + skip;         encoder_processor_decoder.py:295-296 for the skip)
//     qkvs' = LN_attn'(x2) [W_q; W_k; W_v; W_s]'^T + b' (the NEXT block's fused projections, layers/block.py:1237-1245)
//
// - for block tails of FEW rows: a rank's share of a sharded mesh (1 281 + halo rows at 8 ranks), the hidden meshes of res 3 / 4.  There the
// row-resident chain has one 48-row panel for every ninth CU and each of those streams the layer's whole 6.5 MiB through its own L1 path
// (~60 us) while the rest of the chip idles, and the launch-per-GEMM path runs five launches of ~15 us at 5 % of the MFMA roof.
//
// Here FOUR CUs OF ONE XCD form a cluster that owns a panel, as a tensor-parallel group over the MLP's hidden width: member m keeps the
// whole panel (the projection is computed by every member: 512 KiB of weights, no exchange, the LayerNorm stays local) and takes hidden
// chunk m - the first Linear's 512 columns [512 m, 512 m + 512) and the matching K-slice of the second Linear - so that per layer a CU
// streams 4 x 512 KiB instead of 6.5 MiB and FOUR times as many CUs work.  The second Linear's partial sums [48 x 512] fp32 are the ONE
// exchange per panel: every member writes its partial to a scratch slot, raises the cluster's counter, reads the other three and adds all
// four in member order (so the four copies of x2 are bit-identical); then member m computes chunk m of the trailing projection and stores
// a quarter of the x2 rows.
//
// The exchange goes through memory with sc1 (agent-scope) stores and loads plus an agent-scope atomic counter - the placement-independent
// recipe of /opt/skills/guides/MI355X_MICROARCH.md (Correctness boundaries); that workgroups b, b + 8, b + 16, b + 24 share an XCD (observed:
// workgroup b runs on XCD b % 8) only makes it an L2 round trip (tools/pair_exchange_probe.hip: 2.3 us per 48 KiB).  The members of a
// cluster must be resident together: the grid is at most one workgroup per CU (160 KB of LDS each) and a cluster's four workgroups are
// dispatched within 32 consecutive ids; a member that waits longer than ~1 s traps (the launch FAILS - never a silent wrong result).
#include "chain2_core.h"

namespace anemoi {

constexpr int kClusterSize = 4;

struct ClusterArgs {
  const void* attn;  int64_t ld_attn;
  const void* xres;  int64_t ld_x;
  const char* wp;                    // projection, fragment-major [512, 512]
  const char* w1;                    // MLP-1 with LN_mlp's gamma folded in, fragment-major [2048, 512]
  const char* w2;                    // MLP-2 fragment-major [512, 2048]
  const char* wq;    int qc;         // trailing projection with LN_attn' gamma folded in, [512 qc, 512]; qc <= 4
  const void* vec;                   // [b_p 512 | d1 2048 | b_2 512 | dq 512 qc]
  float eps1, epsq;
  const void* extra; int64_t ld_extra;
  void* xout;        int64_t ld_out;
  void* qout;        int64_t ld_q;
  void* lnout;       int64_t ld_ln;  // optional: LN_attn'(x2) WITHOUT its affine part, [n_rows, 512]
  void* qout2;       int64_t ld_q2;  int q_split;  // optional: chunks >= q_split of the trailing projection go to qout2 (column 512 (chunk - q_split))
  float* scratch;                    // [clusters][2][4 members][8 waves][3][4][64 lanes][4] fp32
  unsigned* counters;                // [clusters], zero at allocation, then monotonic (4 per panel)
  int n_rows, rows_per_tile, n_tiles;
};
constexpr int kClRedOff = 3 * kBufBytes;
constexpr int kClVecOff = kClRedOff + kPanel * 8 * 2 * 4;
constexpr int kClVecMax = 5120;  // 512 + 2048 + 512 + 2048
constexpr int kClusterSmem = kClVecOff + kClVecMax * 2;
static_assert(kClusterSmem <= 160 * 1024, "LDS budget");
constexpr size_t kClSlotFloats = (size_t)8 * 3 * 4 * 64 * 4;  // one member's partial [48 x 512] in accumulator layout

// acc[mi][0..3] = vec[col0 + 64 w8 + column]
template <typename T>
__device__ __forceinline__ void init_acc64(f32x4 (&acc)[3][8], const unsigned char* vec, int col0, int lane, int w8) {
  const LaneCtx lc = lane_ctx(lane, w8);
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    float b[4];
    unpack4<T>(*reinterpret_cast<const u32x2*>(vec + (col0 + w8 * 64 + ni * 16 + lc.g * 4) * 2), b);
#pragma unroll
    for (int mi = 0; mi < 3; ++mi) acc[mi][ni] = f32x4{b[0], b[1], b[2], b[3]};
  }
}
// GELU (GELU = false: nothing) of the wave's 48 x 64 block, rounded to the model dtype into the panel buffer `dst` (its own columns)
template <typename T, bool GELU>
__device__ __forceinline__ void round_rows64(const f32x4 (&acc)[3][8], unsigned char* dst, int lane, int w8) {
  const LaneCtx lc = lane_ctx(lane, w8);
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) {
    unsigned char* drow = dst + (mi * 16 + lc.x) * kRowBytes;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float o[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      if (GELU) {
        gelu_fast2(o[0], o[1]);
        gelu_fast2(o[2], o[3]);
      }
      *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pack4<T>(o);
      if (ni & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// the wave's staged 48 x 64 block (panel layout, its own columns) to global memory as 128-byte row pieces: 8 lanes per row
template <typename T>
__device__ __forceinline__ void store_staged64(const unsigned char* strip, T* out, int64_t ld, int nr, int lane, int w8) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave reads back only what it wrote itself: no barrier
  asm volatile("" : "+v"(lane), "+s"(w8));
  const int rl = lane >> 3, sl = lane & 7;
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int row = it * 8 + rl;
    const u32x4 v = *reinterpret_cast<const u32x4*>(strip + row * kRowBytes + (((w8 * 8 + sl) ^ (row & 15)) << 4));
    if (row < nr) stream_store(v, reinterpret_cast<u32x4*>(out + (int64_t)row * ld + w8 * 64 + sl * 8));
  }
}
__device__ __forceinline__ void store_sc1(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 load_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <typename T>
__global__ __launch_bounds__(512, 1) void gt_cluster_chain_kernel(ClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const bufH = smem;                  // the member's hidden chunk; then the staging of its projection chunk
  unsigned char* const bufB = smem + kBufBytes;      // attention rows, then LN(x1), then LN'(x2)
  unsigned char* const bufC = smem + 2 * kBufBytes;  // skip rows, then x1, then x2
  float* const red = reinterpret_cast<float*>(smem + kClRedOff);
  const unsigned char* const vec = smem + kClVecOff;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t loff = lane * 16;
  // cluster (xcd, j) = workgroups xcd + 8 (4 j + m), m = 0..3
  const int b = (int)blockIdx.x, xcd = b & 7, iq = b >> 3;
  const int m = iq & 3, cluster = (iq >> 2) * 8 + xcd, n_clusters = (int)gridDim.x >> 2;
  int tile = cluster;
  if (tile >= a.n_tiles) return;
  const int qc = a.qc;
  const bool myq = m < qc;
  const char* const wps = a.wp + (int64_t)w8 * kSlab;
  const char* const w1s = a.w1 + (int64_t)(8 * m + w8) * kSlab;
  const char* const w2s = a.w2 + (int64_t)w8 * (4 * (int64_t)kSlab) + (int64_t)m * kSlab;
  const char* const wqs = a.wq + (int64_t)(8 * m + w8) * kSlab;
  float* const my_slot = a.scratch + ((size_t)cluster * 2 * kClusterSize + m) * kClSlotFloats;  // + parity * 4 slots
  unsigned* const ctr = a.counters + cluster * 32;  // (a 128-byte line per cluster)
  frag8 ring[2][8];
  f32x4 acc[3][8];

  // 6 attention rows and 6 skip rows per wave -> the swizzled panels
  auto load_panel = [&](int r0, int nr, auto between) {
    int l0 = lane, w0 = w8;
    asm volatile("" : "+v"(l0), "+s"(w0));
    u32x4 va[6], vx[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = min(w0 * 6 + i, nr - 1);
      va[i] = stream_load(reinterpret_cast<const u32x4*>((const T*)a.attn + (int64_t)(r0 + row) * a.ld_attn + l0 * 8));
      vx[i] = stream_load(reinterpret_cast<const u32x4*>((const T*)a.xres + (int64_t)(r0 + row) * a.ld_x + l0 * 8));
    }
    __builtin_amdgcn_sched_barrier(0);
    between();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = w0 * 6 + i;
      const int off = row * kRowBytes + ((l0 ^ (row & 15)) << 4);
      *reinterpret_cast<u32x4*>(bufB + off) = row < nr ? va[i] : u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(bufC + off) = row < nr ? vx[i] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  {
    const int r0 = tile * a.rows_per_tile;
    const int n16 = (3072 + 512 * qc) / 8;  // <= 640
    u32x4 v0, v1;
    load_panel(r0, min(a.rows_per_tile, a.n_rows - r0), [&] {
      v0 = reinterpret_cast<const u32x4*>(a.vec)[min(tid, n16 - 1)];
      v1 = reinterpret_cast<const u32x4*>(a.vec)[min(tid + 512, n16 - 1)];
      ring_prologue64(ring, wps, loff);
    });
    reinterpret_cast<u32x4*>(smem + kClVecOff)[tid] = v0;
    if (tid + 512 < n16) reinterpret_cast<u32x4*>(smem + kClVecOff)[tid + 512] = v1;
    lds_barrier();
  }
  unsigned round = 0;
  for (;;) {
    const int r0 = tile * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    // P (every member): x1 = attn Wp^T + b_p + x -> bufC, LayerNorm_mlp(x1) without its affine part -> bufB
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm64<T>(bufB, lane, ring, wps, w1s, 8192, loff, acc);
    round_rows64_add_stats<T>(acc, bufC, red, lane, w8, vec);
    lds_barrier();
    normalise_rows64<T>(acc, red, a.eps1, bufB, lane, w8);
    lds_barrier();
    // M1: this member's hidden chunk h_m = GELU(LN(x1) W1[chunk m]^T + d1[chunk m]) -> bufH
    init_acc64<T>(acc, vec, 512 + 512 * m, lane, w8);
    gemm64<T>(bufB, lane, ring, w1s, w2s, 8192, loff, acc);
    round_rows64<T, true>(acc, bufH, lane, w8);
    lds_barrier();
    // M2: the member's partial of x2: h_m W2[:, chunk m]^T
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm64<T>(bufH, lane, ring, w2s, myq ? wqs : wps, 8192, loff, acc);
    // the ONE exchange: partials through memory (sc1 stores / loads), an agent-scope counter per cluster
    {
      float* mine = my_slot + (size_t)(round & 1) * kClusterSize * kClSlotFloats + (size_t)w8 * (3 * 4 * 64 * 4) + lane * 4;
#pragma unroll
      for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) store_sc1(mine + (mi * 4 + ni) * 256, acc[mi][ni]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (old / kClusterSize + 1u) * kClusterSize;
        bool ok = false;
        for (int it = 0; it < (1 << 23); ++it) {
          if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { ok = true; break; }
          __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) __builtin_trap();  // a member that never came (its workgroup not resident): fail the launch, never continue on partial sums
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // x2 = (p0 + p1) + (p2 + p3) + b_2 + x1 (rounded) -> bufC, row statistics; the members' copies are bit-identical
    {
      const LaneCtx lc = lane_ctx(lane, w8);
      const float* base = a.scratch + ((size_t)cluster * 2 + (round & 1)) * kClusterSize * kClSlotFloats + (size_t)w8 * (3 * 4 * 64 * 4) + lane * 4;
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        f32x4 p[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) p[c][ni] = (c == m) ? acc[mi][ni] : load_sc1(base + (size_t)c * kClSlotFloats + (mi * 4 + ni) * 256);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) asm volatile("" : "+v"(p[c][ni]));  // (the sums below must stay behind the wait: the compiler does not know these loads)
        unsigned char* drow = bufC + (mi * 16 + lc.x) * kRowBytes;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float bv[4], xv[4], o[4];
          unpack4<T>(*reinterpret_cast<const u32x2*>(vec + (2560 + w8 * 64 + ni * 16 + lc.g * 4) * 2), bv);
          unpack4<T>(*reinterpret_cast<const u32x2*>(drow + lc.coff[ni]), xv);
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = ((p[0][ni][k] + p[1][ni][k]) + (p[2][ni][k] + p[3][ni][k])) + (bv[k] + xv[k]);
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
    lds_barrier();  // x2 and the partials of its statistics are complete
    if (qc > 0 || a.lnout != nullptr) normalise_rows64<T>(acc, red, a.epsq, bufB, lane, w8);
    // x2 [+ latent skip] -> global: member m stores rows 12 m .. 12 m + 11 as whole 1-KiB rows (768 pieces of 16 bytes)
    {
      int t0 = tid;
      asm volatile("" : "+v"(t0));
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = t0 + 512 * k;
        const int row = m * 12 + (i >> 6), sl = i & 63;
        if (i < 768 && row < nr) {
          u32x4 v = *reinterpret_cast<const u32x4*>(bufC + row * kRowBytes + ((sl ^ (row & 15)) << 4));
          if (a.extra != nullptr) {
            // the latent skip rides on the last block's output, added to the block's ROUNDED output as `x + skip` does
            const u32x4 e = *reinterpret_cast<const u32x4*>((const T*)a.extra + (int64_t)(r0 + row) * a.ld_extra + sl * 8);
            float pp[4], ss[4];
            unpack4<T>(u32x2{v[0], v[1]}, pp);
            unpack4<T>(u32x2{e[0], e[1]}, ss);
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[r] += ss[r];
            const u32x2 lo = pack4<T>(pp);
            unpack4<T>(u32x2{v[2], v[3]}, pp);
            unpack4<T>(u32x2{e[2], e[3]}, ss);
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[r] += ss[r];
            const u32x2 hi = pack4<T>(pp);
            v = u32x4{lo[0], lo[1], hi[0], hi[1]};
          }
          stream_store(v, reinterpret_cast<u32x4*>((T*)a.xout + (int64_t)(r0 + row) * a.ld_out + sl * 8));
        }
      }
    }
    lds_barrier();  // LN'(x2) is complete in bufB; every wave is behind its reads of bufC and bufH
    if (a.lnout != nullptr) {
      int t0 = tid;
      asm volatile("" : "+v"(t0));
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = t0 + 512 * k;
        const int row = m * 12 + (i >> 6), sl = i & 63;
        if (i < 768 && row < nr) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(bufB + row * kRowBytes + ((sl ^ (row & 15)) << 4));
          *reinterpret_cast<u32x4*>((T*)a.lnout + (int64_t)(r0 + row) * a.ld_ln + sl * 8) = v;
        }
      }
    }
    const int tile_next = tile + n_clusters;
    const bool more = tile_next < a.n_tiles;
    // Q: chunk m of the trailing projection
    if (myq) {
      init_acc64<T>(acc, vec, 3072 + 512 * m, lane, w8);
      gemm64<T>(bufB, lane, ring, wqs, wps, 8192, loff, acc);
      round_rows64<T, false>(acc, bufH, lane, w8);
      if (a.qout2 != nullptr && m >= a.q_split) store_staged64<T>(bufH, (T*)a.qout2 + (int64_t)r0 * a.ld_q2 + (m - a.q_split) * kCh, a.ld_q2, nr, lane, w8);
      else store_staged64<T>(bufH, (T*)a.qout + (int64_t)r0 * a.ld_q + m * kCh, a.ld_q, nr, lane, w8);
    }
    if (!more) break;
    lds_barrier();  // every wave is behind its last read of bufB / bufH
    tile = tile_next;
    ++round;
    {
      const int rn = tile * a.rows_per_tile;
      load_panel(rn, min(a.rows_per_tile, a.n_rows - rn), [] {});
    }
    lds_barrier();
  }
}

}  // namespace anemoi

using namespace anemoi;

static int cluster_grid(int n_tiles) {
  int per_xcd = (n_tiles + 7) / 8;  // clusters per XCD
  if (per_xcd > 8) per_xcd = 8;     // 8 clusters x 4 members = the XCD's 32 CUs
  return per_xcd * 32;
}

extern "C" int64_t anemoi_gt_cluster_chain_workspace_bytes(void) {
  // 64 clusters x (2 parities x 4 members x [48 x 512] fp32 + a counter line)
  return (int64_t)64 * (2 * kClusterSize * (int64_t)kClSlotFloats * 4) + 64 * 128;
}

extern "C" int anemoi_gt_cluster_chain_fwd(const anemoi_gt_cluster_chain_args_t* p, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(p != nullptr, "gt_cluster_chain_fwd: null argument block");
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "gt_cluster_chain_fwd: 16-bit model dtypes only");
  ANEMOI_REQUIRE(p->n_rows >= 0 && p->channels == kCh, "gt_cluster_chain_fwd: channels=%d (this kernel is built for %d)", p->channels, kCh);
  if (p->n_rows == 0) return ANEMOI_OK;
  if (p->hidden != 4 * kCh) {
    set_error("gt_cluster_chain_fwd: hidden=%d (the cluster of four splits a hidden width of %d)", p->hidden, 4 * kCh);
    return ANEMOI_E_UNSUPPORTED;
  }
  ANEMOI_REQUIRE(p->q_out_features >= 0 && p->q_out_features % kCh == 0 && p->q_out_features <= 4 * kCh,
                 "gt_cluster_chain_fwd: q_out_features=%d must be a multiple of %d up to %d", p->q_out_features, kCh, 4 * kCh);
  ANEMOI_REQUIRE(p->attn && p->x_res && p->wp && p->w1 && p->w2 && p->vec && p->x_out && p->workspace, "gt_cluster_chain_fwd: null operand");
  // columns of the trailing projection that go to q_out (all of them without a second destination)
  const int q1_cols = p->q_out2 != nullptr ? (p->q_split < 0 ? 0 : (p->q_split * kCh < p->q_out_features ? p->q_split * kCh : p->q_out_features)) : p->q_out_features;
  ANEMOI_REQUIRE(p->q_out_features == 0 || (p->wq && (q1_cols == 0 || p->q_out)), "gt_cluster_chain_fwd: the trailing projection needs wq and q_out");
  ANEMOI_REQUIRE((p->q_out_features == 0 && p->ln_out == nullptr) || p->extra == nullptr,
                 "gt_cluster_chain_fwd: the trailing projection / LayerNorm output read x2 before a second residual is added: not both");
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  ANEMOI_REQUIRE(al16(p->attn) && al16(p->x_res) && al16(p->wp) && al16(p->w1) && al16(p->w2) && al16(p->x_out) && al16(p->wq) && al16(p->q_out) &&
                     al16(p->extra) && al16(p->vec) && al16(p->ln_out) && (reinterpret_cast<uintptr_t>(p->workspace) & 127) == 0,
                 "gt_cluster_chain_fwd: operands must be 16-byte aligned (workspace: 128)");
  ANEMOI_REQUIRE(p->ld_attn >= kCh && p->ld_x >= kCh && p->ld_out >= kCh && p->ld_attn % 8 == 0 && p->ld_x % 8 == 0 && p->ld_out % 8 == 0 &&
                     (p->extra == nullptr || (p->ld_extra >= kCh && p->ld_extra % 8 == 0)) && (p->ln_out == nullptr || (p->ld_ln >= kCh && p->ld_ln % 8 == 0)) &&
                     (q1_cols == 0 || (p->ld_q >= q1_cols && p->ld_q % 8 == 0)),
                 "gt_cluster_chain_fwd: leading dimensions too small or not multiples of 8 elements (rows move as 16-byte pieces)");
  ANEMOI_REQUIRE(p->workspace_bytes >= anemoi_gt_cluster_chain_workspace_bytes(), "gt_cluster_chain_fwd: workspace of %lld bytes, need %lld",
                 (long long)p->workspace_bytes, (long long)anemoi_gt_cluster_chain_workspace_bytes());
  ClusterArgs a{};
  a.attn = p->attn; a.ld_attn = p->ld_attn;
  a.xres = p->x_res; a.ld_x = p->ld_x;
  a.wp = (const char*)p->wp; a.w1 = (const char*)p->w1; a.w2 = (const char*)p->w2;
  a.wq = (const char*)p->wq; a.qc = p->q_out_features / kCh;
  a.vec = p->vec;
  a.eps1 = p->ln1_eps; a.epsq = p->lnq_eps;
  a.extra = p->extra; a.ld_extra = p->ld_extra;
  a.xout = p->x_out; a.ld_out = p->ld_out;
  a.qout = p->q_out; a.ld_q = p->ld_q;
  a.lnout = p->ln_out; a.ld_ln = p->ld_ln;
  a.qout2 = p->q_out2; a.ld_q2 = p->ld_q2; a.q_split = p->q_split;
  ANEMOI_REQUIRE(p->q_out2 == nullptr || (p->q_split >= 0 && p->q_split * kCh <= p->q_out_features && al16(p->q_out2) && p->ld_q2 % 8 == 0 &&
                                          p->ld_q2 >= p->q_out_features - p->q_split * kCh),
                 "gt_cluster_chain_fwd: q_out2 needs 0 <= 512 q_split <= q_out_features, 16-byte alignment and ld_q2 >= the columns it receives");
  // the counters first (their own 128-byte lines), the partial slots behind them
  a.counters = reinterpret_cast<unsigned*>(p->workspace);
  a.scratch = reinterpret_cast<float*>(reinterpret_cast<char*>(p->workspace) + 64 * 128);
  a.n_rows = p->n_rows;
  a.rows_per_tile = kPanel;
  a.n_tiles = (a.n_rows + kPanel - 1) / kPanel;
  const int grid = cluster_grid(a.n_tiles);
  hipStream_t st = as_stream(stream);
  if (dtype == ANEMOI_BF16) {
    static PerDeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_cluster_chain_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kClusterSmem); });
    hipLaunchKernelGGL((gt_cluster_chain_kernel<bf16_t>), dim3(grid), dim3(512), kClusterSmem, st, a);
  } else {
    static PerDeviceOnce once;
    once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_cluster_chain_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kClusterSmem); });
    hipLaunchKernelGGL((gt_cluster_chain_kernel<f16_t>), dim3(grid), dim3(512), kClusterSmem, st, a);
  }
  return check_launch("gt_cluster_chain_kernel");
}
