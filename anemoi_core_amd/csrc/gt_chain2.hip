// Role-split row-resident layer chain for gfx950 (round 5): the row-local part of a GraphTransformer block in ONE launch, as
// csrc/gt_chain.hip computes it -
//
//     x1 = attn W_p^T + b_p + x                      (projection + skip,            layers/block.py:1263-1266 of the reference)
//     h  = GELU(LN_mlp(x1) W_1^T + b_1)               (node_dst_mlp, first Linear,   layers/block.py:1268-1271, layers/mlp.py:158-169)
//     x2 = h W_2^T + b_2 + x1 [+ latent skip]        (second Linear + skip;         encoder_processor_decoder.py:295-296 for the skip)
//     qkvs' = LN_attn'(x2) [W_q; W_k; W_v; W_s]'^T + b' (the NEXT block's fused projections, layers/block.py:1237-1245)
//
// - but with the eight waves of the workgroup split into TWO GROUPS OF FOUR with different jobs, so that one group's epilogue (VALU,
// LDS, global stores) runs beside the other group's MFMA stream instead of with the matrix cores idle.
//
// What the round-4 kernel's in-kernel timeline said (profiles/r04_chain_timeline.txt): its eight waves are always in the same phase -
// of 114 us the thirteen GEMM segments are ~65, the rest are epilogues and barrier waits with the MFMA pipe AND the weight stream idle.
// What tools/role_split_probe.hip measured before this kernel was written (profiles/probes/r05_role_split_probe.txt): ONE group of four
// waves (one per SIMD), each owning a 48 x 128 output slab and streaming its 8 KiB per K-step through a register ring of only two
// K-steps, pulls the layer's 6.5 MiB through the CU's L1 path as fast as eight waves do (66 us per layer, 103 GB/s per CU), in 200
// registers; two such groups on the layer's schedule with GELU / convert / LDS-write epilogues and a barrier per step: 74 us.
//
// Roles.  Group A (waves 0-3): projection, the MLP's first Linear chunk by chunk (+ GELU), the even chunks of the trailing
// projection.  Group B (waves 4-7; wave w + 4 shares a SIMD with wave w): the MLP's second Linear, accumulating over the hidden
// chunks A produces (the two accumulator sets of the round-4 kernel's software pipeline now live in two different waves: 96
// registers each), the x2 epilogue, the odd chunks of the trailing projection.  Per panel, one s_barrier per step:
//
//     S0  A: attention rows -> bufB            B: skip rows -> bufC
//     S1  A: P = attn Wp^T, acc initialised with b_p + x; x1 rounded -> bufC, per-wave row statistics
//     S2  A: LayerNorm (no affine) of x1 from registers -> bufB      B: acc2 = b_2 + x1 (from bufC)
//     M_t (t = 0..hc)   A: M1(t): acc = d1[t]; GEMM on bufB; GELU -> h_t (bufA / bufC alternating)      B: M2(t-1): acc2 += h_(t-1) W2_t^T
//                       (t = hc, B: x2 rounded -> the free h buffer, per-wave row statistics)
//     S8  A: x2 rows [+ latent skip] -> global, whole 1-KiB rows      B: LayerNorm (no affine) of x2 from registers -> bufB
//     Q_c A: chunk 2c, B: chunk 2c+1 of the trailing projection: acc = dq[chunk]; GEMM on bufB; rounded -> the group's own h buffer ->
//         whole 256-byte row pieces to global
//
// LayerNorm.  The statistics are the plain fp32 LayerNorm statistics of the ROUNDED 16-bit rows (per-wave (mean, M2) over 128 columns,
// merged with Chan's formula: no E[x^2] - mean^2 cancellation); the normalised row (x - mean) * rstd is rounded to the model dtype
// and the AFFINE part is folded into the Linear that follows, on the host and once per parameter version (W diag(gamma) rounded to
// the model dtype, d = W beta + b): the GEMM then needs no epilogue arithmetic at all - its accumulators START at d.  Likewise the
// projection's start at b_p + x and the second Linear's at b_2 + x1, so x1 never goes to global memory.  The per-column vectors
// (b_p | d1 | b_2 | dq, model dtype) sit in LDS for the whole launch.
#include "chain_core.h"

namespace anemoi {

struct Chain2Args {
  const void* attn;  int64_t ld_attn;       // [n_rows, 512]  attention output + self term
  const void* xres;  int64_t ld_x;          // [n_rows, 512]  the block's input (skip)
  const char* wp;                           // projection, fragment-major
  const char* w1;    int hc;                // MLP-1 with LN_mlp's gamma folded in, [hidden, 512] fragment-major, hidden = 512 hc
  const char* w2;                           // MLP-2 [512, hidden] fragment-major
  const char* wq;    int qc;                // trailing projection with LN_attn' gamma folded in, [512 qc, 512] fragment-major; qc = 0: none
  const void* vec;                          // [512 b_p | 512 hc d1 | 512 b_2 | 512 qc dq]
  float eps1, epsq;
  const void* extra; int64_t ld_extra;      // optional second residual of x2
  void* xout;        int64_t ld_out;        // [n_rows, 512]  x2
  void* qout;        int64_t ld_q;          // [n_rows, 512 qc]
  int n_rows, rows_per_tile, n_tiles;
  int prio_a;                               // experiment: s_setprio of group A (0: none)
  int prio_q;                               // experiment: chunk-ordered priorities in the trailing projection
  int dbg;                                  // experiment (timing only, results are garbage): bit 0 no GELU arithmetic, 1 no LayerNorm statistics / normalisation, 2 no stores of the trailing projection
  int warm;                                 // the L2 warm-up of the next step's weights (ANEMOI_CHAIN2_WARM=0: off, for the A/B)
  unsigned long long* timeline;             // developer aid (TL instantiation only): [workgroups][8 waves][kTl2Slots] s_memtime stamps
};
constexpr int kTl2Slots = 48;
constexpr int kRed2Off = 3 * kBufBytes;                   // [48 rows][4 waves][2] fp32 LayerNorm partials
constexpr int kVecOff = kRed2Off + kPanel * 4 * 2 * 4;    // the per-column vectors (16-bit)
constexpr int kVecMaxElems = 6144;                        // 12 KiB: 512 + hidden + 512 + q_out <= 6144 (hidden = q_out = 2048: 5120)
constexpr int kChain2Smem = kVecOff + kVecMaxElems * 2;
constexpr int kVecMaxElemsTl = 5120;                      // the instrumented instantiation gives 2 KiB of the vector region to its stamps
constexpr int kTl2Off = kVecOff + kVecMaxElemsTl * 2;
static_assert(kChain2Smem <= 160 * 1024 && kTl2Off + 8 * kTl2Slots * 8 <= 160 * 1024, "LDS budget");

// 16 K-steps (K = 512) of this wave's 48 x 128 tile.  A fragments from the swizzled LDS panel (the next K-step's requested before this
// one's MFMAs), B fragments from a register ring of two K-steps x 8 fragments, each slot refilled right behind its three MFMAs with
// the fragment of two K-steps ahead - of this segment or, in its last pair, of the wave's NEXT segment (`nxt`).  The wave's 128
// columns are two adjacent 64-column slabs of the fragment-major image: streams `cur` and `cur + cs`.
template <typename T>
__device__ __forceinline__ void gemm128(const unsigned char* abuf, int lane, frag8 (&ring)[2][8], const char* cur, int64_t cs, const char* nxt,
                                        int64_t ns, uint32_t loff, f32x4 (&acc)[3][8]) {
  asm volatile("" : "+v"(lane));
  const int x = lane & 15, ks = lane >> 4;
  const unsigned char* arow = abuf + x * kRowBytes;
  frag8 fa[3];
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) fa[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16 * kRowBytes + ((ks ^ x) << 4));
#pragma unroll 1
  for (int q = 0; q < 8; ++q) {
    const char* p0 = q < 7 ? cur + (q + 1) * 8192 : nxt;
    const char* p1 = q < 7 ? cur + cs + (q + 1) * 8192 : nxt + ns;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int st = q * 2 + j;
      const int sn = st < 15 ? st + 1 : 15;  // (the last step re-reads its own fragments: no branch in the stream)
      frag8 fn[3];
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) fn[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16 * kRowBytes + (((sn * 4 + ks) ^ x) << 4));
      __builtin_amdgcn_sched_barrier(0);
      const gptr_t g0 = uniform_ptr(p0 + j * 4096), g1 = uniform_ptr(p1 + j * 4096);
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
#pragma unroll
        for (int mi = 0; mi < 3; ++mi) acc[mi][ni] = cmfma<T>(ring[j][ni], fa[mi], acc[mi][ni]);  // D^T: lane = row x, 4 consecutive columns
        ring[j][ni] = *reinterpret_cast<gfrag_t>((ni < 4 ? g0 : g1) + loff + (ni & 3) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) fa[mi] = fn[mi];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Per-lane coordinates of a 128-column wave (re-derived from an opaque lane id in every phase: see LaneCtx in chain_core.h)
struct Lane2 {
  int x, g;
  int coff[8];  // LDS byte offset (inside a panel row) of this lane's 4 columns of column block ni: slot = wq*16 + ni*2 + (g>>1), swizzled by the row
};
__device__ __forceinline__ Lane2 lane2(int lane, int wq) {
  asm volatile("" : "+v"(lane), "+s"(wq));
  Lane2 c;
  c.x = lane & 15;
  c.g = lane >> 4;
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) c.coff[ni] = (((wq * 16 + ni * 2 + (c.g >> 1)) ^ c.x) << 4) + (c.g & 1) * 8;
  return c;
}

// acc[mi][ni] = v[col] (+ the panel values at the lane's positions of `rows`): the accumulators of a GEMM start at its bias (+ residual)
template <typename T, bool ROWS>
__device__ __forceinline__ void init_acc(f32x4 (&acc)[3][8], const unsigned char* vec, int col0, const unsigned char* rows, int lane, int wq) {
  const Lane2 lc = lane2(lane, wq);
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) {
    float b[4];
    unpack4<T>(*reinterpret_cast<const u32x2*>(vec + (col0 + wq * 128 + ni * 16 + lc.g * 4) * 2), b);
#pragma unroll
    for (int mi = 0; mi < 3; ++mi) {
      if (ROWS) {
        float r[4];
        unpack4<T>(*reinterpret_cast<const u32x2*>(rows + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]), r);
        acc[mi][ni] = f32x4{b[0] + r[0], b[1] + r[1], b[2] + r[2], b[3] + r[3]};
      } else {
        acc[mi][ni] = f32x4{b[0], b[1], b[2], b[3]};
      }
    }
  }
}

// The wave's 48 x 128 block rounded to the model dtype into the panel buffer `dst` (its own columns); acc keeps the ROUNDED values.
// STATS: per-wave (mean, M2) of every row over the wave's 128 columns -> red[row][wq].
// ADD: first + vec[col0 + column] + the values the panel buffer `dst` holds at the same positions (the projection's bias and skip rows).
template <typename T, bool STATS, bool ADD = false>
__device__ __forceinline__ void round_rows(f32x4 (&acc)[3][8], unsigned char* dst, float* red, int lane, int wq, const unsigned char* vec = nullptr, int col0 = 0) {
  const Lane2 lc = lane2(lane, wq);
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) {
    unsigned char* drow = dst + (mi * 16 + lc.x) * kRowBytes;
    // (ADD: the reads of half a row band are requested together, pinned - the ring's next fragments are live here: all 48 reads at once spill,
    // one pair at a time exposes an LDS round trip twelve times)
    u32x2 rb[8], rr[8];
    if (ADD) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ni = 4 * h; ni < 4 * h + 4; ++ni) {
          rb[ni] = *reinterpret_cast<const u32x2*>(vec + (col0 + wq * 128 + ni * 16 + lc.g * 4) * 2);
          rr[ni] = *reinterpret_cast<const u32x2*>(drow + lc.coff[ni]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      float o[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      if (ADD) {
        float b[4], r[4];
        unpack4<T>(rb[ni], b);
        unpack4<T>(rr[ni], r);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += b[k] + r[k];
      }
      const u32x2 pk = pack4<T>(o);
      *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pk;
      if (STATS) {
        unpack4<T>(pk, o);
        acc[mi][ni] = f32x4{o[0], o[1], o[2], o[3]};
      }
    }
    if (STATS) {
      float s = 0.f;
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) s += (acc[mi][ni][0] + acc[mi][ni][1]) + (acc[mi][ni][2] + acc[mi][ni][3]);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mw = s * (1.0f / 128.0f);
      float q = 0.f;
#pragma unroll
      for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[mi][ni][r] - mw;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (lc.g == 0) *reinterpret_cast<float2*>(red + ((mi * 16 + lc.x) * 4 + wq) * 2) = make_float2(mw, q);
    }
  }
}

// GELU of the wave's 48 x 128 block, rounded to the model dtype into the panel buffer `dst` (its own columns).  Column block by column
// block, pinned: left alone the scheduler interleaves all 48 polynomial chains and spills 32 registers around them.
template <typename T>
__device__ __forceinline__ void gelu_rows(const f32x4 (&acc)[3][8], unsigned char* dst, int lane, int wq) {
  const Lane2 lc = lane2(lane, wq);
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) {
    unsigned char* drow = dst + (mi * 16 + lc.x) * kRowBytes;
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      float o[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      gelu_fast2(o[0], o[1]);
      gelu_fast2(o[2], o[3]);
      *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pack4<T>(o);
      if (ni & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// LayerNorm without the affine part: the four waves' (mean, M2) of each row merged in wave order (Chan et al.), the rounded values in
// acc normalised and stored (model dtype) into the panel buffer `dst`.
template <typename T>
__device__ __forceinline__ void normalise_rows(const f32x4 (&acc)[3][8], const float* red, float eps, unsigned char* dst, int lane, int wq) {
  const Lane2 lc = lane2(lane, wq);
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) {
    const f32x4* pr = reinterpret_cast<const f32x4*>(red + (mi * 16 + lc.x) * 8);
    const f32x4 p0 = pr[0], p1 = pr[1];
    const float mu = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * 0.25f;
    const float d0 = p0[0] - mu, d1 = p0[2] - mu, d2 = p1[0] - mu, d3 = p1[2] - mu;
    const float m2 = fmaf(128.0f, (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3), (p0[1] + p0[3]) + (p1[1] + p1[3]));
    const float rstd = rsqrtf(m2 * (1.0f / (float)kCh) + eps);
    const float nm = -mu * rstd;
    unsigned char* drow = dst + (mi * 16 + lc.x) * kRowBytes;
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaf(acc[mi][ni][r], rstd, nm);
      *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pack4<T>(o);
    }
  }
}

// What both roles share: the launch's constants and the wave's place in it
struct Ctx2 {
  int lane, wq, wave;
  uint32_t loff;
  int tl_n;
};
template <bool TL>
__device__ __forceinline__ void stamp2(Ctx2& c, unsigned char* smem) {
  if constexpr (TL) {
    if (c.tl_n < kTl2Slots) {
      const unsigned long long now = __builtin_readcyclecounter();
      if (c.lane == 0) reinterpret_cast<unsigned long long*>(smem + kTl2Off)[c.wave * kTl2Slots + c.tl_n] = now;
    }
    ++c.tl_n;
  }
}
// 12 whole rows of the panel (rows wq*12 ..) from global memory into a panel buffer; rows beyond the panel are zero
template <typename T, typename Between>
__device__ __forceinline__ void load_rows12(const T* src, int64_t ld, int r0, int nr, unsigned char* dst, int lane, int wq, Between between) {
  asm volatile("" : "+v"(lane), "+s"(wq));  // (else the twelve LDS addresses are hoisted out of the panel loop and spilled around the GEMM segments)
  u32x4 v[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int row = wq * 12 + i;
    v[i] = *reinterpret_cast<const u32x4*>(src + (int64_t)(r0 + min(row, nr - 1)) * ld + lane * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
  between();  // (loads the caller wants queued BEHIND the rows)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int row = wq * 12 + i;
    *reinterpret_cast<u32x4*>(dst + row * kRowBytes + ((lane ^ (row & 15)) << 4)) = row < nr ? v[i] : u32x4{0u, 0u, 0u, 0u};
  }
}
// the per-column vectors -> LDS, once per launch, by all 512 threads (requested behind the first panel's rows; visible behind the first barrier)
struct VecCopy {
  u32x4 v[2];
  __device__ __forceinline__ void request(const Chain2Args& a, int tid) {
    const int n16 = (1024 + 512 * (a.hc + a.qc)) / 8;  // <= 768
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] = reinterpret_cast<const u32x4*>(a.vec)[min(tid + 512 * k, n16 - 1)];
  }
  __device__ __forceinline__ void store(const Chain2Args& a, int tid, unsigned char* smem) {
    const int n16 = (1024 + 512 * (a.hc + a.qc)) / 8;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 512 * k < n16) reinterpret_cast<u32x4*>(smem + kVecOff)[tid + 512 * k] = v[k];
  }
};
// the wave's first weight fragments (two K-steps of both 64-column streams)
__device__ __forceinline__ void ring_prologue(frag8 (&ring)[2][8], const char* f0, int64_t fs, uint32_t loff) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const gptr_t g0 = uniform_ptr(f0 + j * 4096), g1 = uniform_ptr(f0 + fs + j * 4096);
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      ring[j][ni] = *reinterpret_cast<gfrag_t>((ni < 4 ? g0 : g1) + loff + (ni & 3) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// the wave's staged 48 x 128 block (panel layout, its own columns) to global memory as 256-byte row pieces: 16 lanes per row
template <typename T>
__device__ __forceinline__ void store_staged(const unsigned char* strip, T* out, int64_t ld, int nr, int lane, int wq) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave reads back only what it wrote itself: no barrier
  asm volatile("" : "+v"(lane), "+s"(wq));
  const int rl = lane >> 4, sl = lane & 15;
#pragma unroll
  for (int it = 0; it < 12; ++it) {
    const int row = it * 4 + rl;
    const u32x4 v = *reinterpret_cast<const u32x4*>(strip + row * kRowBytes + (((wq * 16 + sl) ^ (row & 15)) << 4));
    if (row < nr) *reinterpret_cast<u32x4*>(out + (int64_t)row * ld + wq * 128 + sl * 8) = v;
  }
}

// L2 warm-up of a weight segment one step ahead of its use.  All CUs of an XCD walk the same weight stream at the same time behind a ring
// of only two K-steps, so a line that misses the XCD's L2 (in a 16-layer model every layer's weights have left the caches since the
// last forward) exposes the HBM latency to all 32 of them at once.  Each CU therefore TOUCHES its 1/32 share of the segments of the
// NEXT step - one dword per 128-byte line, 64 lines per instruction - from group A's waves right behind their GEMM, where the
// epilogue that follows hides the latency; by the time a ring asks for the lines they are in the L2 (observed: workgroup b runs on XCD
// b % 8, so b / 8 numbers the CUs of an XCD - for speed only: a different placement warms less, nothing depends on it).
// A segment = the 512 x 512 weight of one group step: eight 64-KiB slabs `piece` bytes apart; `half` selects 64 of the CU's 128 lines.
__device__ __forceinline__ unsigned touch_share(const char* seg, int64_t piece, int half, int lane) {
  const int line = (((int)blockIdx.x >> 3) & 31) * 128 + half * 64 + lane;  // 0 .. 4095
  return *reinterpret_cast<const unsigned*>(seg + (int64_t)(line >> 9) * piece + (line & 511) * 128);
}
// (the touched value must stay "in use" until the next touch: the compiler then keeps its register and counts the load)
__device__ __forceinline__ void touch_done(unsigned& v) { asm volatile("" : "+v"(v)); }

// The trailing projection's chunks in strict priority order (chunk 0 > 1 > 2 > 3): the groups' chunks then take the weight stream one
// behind the other, each epilogue beside the other group's GEMM, instead of the older wave of a SIMD starving the younger one.
__device__ __forceinline__ void set_chunk_prio(int on, int k) {
  if (!on) return;
  if (k == 0) __builtin_amdgcn_s_setprio(3);
  else if (k == 1) __builtin_amdgcn_s_setprio(2);
  else if (k == 2) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// Both roles execute the SAME number of s_barrier per panel (the hardware barrier counts arrivals, not program locations):
// S0 | P GEMM | x1 | S2 | hc + 1 MLP steps | S8 | one behind the trailing projection (if any).
template <typename T, bool TL>
__device__ __forceinline__ void role_a(const Chain2Args& a, Ctx2& c, unsigned char* smem) {
  const int hc = a.hc, qc = a.qc, lane = c.lane, wq = __builtin_amdgcn_readfirstlane(c.wq);
  // bufA: the even hidden chunks (x2 / group A's staged projection outputs when hc is even); bufB: the A operand of P, M1 and the
  // trailing projection (attention rows, LN(x1), LN'(x2)); bufC: skip rows, then x1, then the odd hidden chunks
  unsigned char* const bufB = smem + kBufBytes;
  unsigned char* const bufC = smem + 2 * kBufBytes;
  float* const red = reinterpret_cast<float*>(smem + kRed2Off);
  const unsigned char* const vec = smem + kVecOff;
  auto hbuf = [&](int t) { return smem + (t & 1) * (2 * kBufBytes); };
  const int64_t s1 = kSlab;
  const char* const wpw = a.wp + (int64_t)(2 * wq) * kSlab;
  auto w1c = [&](int k) { return a.w1 + (int64_t)(8 * k + 2 * wq) * kSlab; };
  auto wqc = [&](int k) { return a.wq + (int64_t)(8 * k + 2 * wq) * kSlab; };
  frag8 ring[2][8];
  f32x4 acc[3][8];
  // the L2 warm-up: waves 0, 1 touch the CU's share of group A's next segment, waves 2, 3 of group B's (64 lines each)
  const int64_t sw2 = (int64_t)hc * kSlab;
  auto seg_a = [&](int t) { return t < hc ? a.w1 + (int64_t)(8 * t) * kSlab : (qc > 0 ? a.wq : a.wp); };  // M1(t); behind the last chunk: Q0 / the next panel's P
  unsigned warm = 0;
  auto touch = [&](const char* sa, const char* sb, int64_t pb) {
    touch_done(warm);
    if (!a.warm) return;
    if (wq < 2) { if (sa != nullptr) warm = touch_share(sa, kSlab, wq & 1, lane); }
    else if (sb != nullptr) warm = touch_share(sb, pb, wq & 1, lane);
  };
  // S0 of the first panel (peeled: straight-line code, so that the stores of the rows wait for the rows only): the attention rows ->
  // bufB, requested AHEAD of the vectors, the warm-up and the weight ring's first fragments - loads return in order
  {
    const int r0 = (int)blockIdx.x * a.rows_per_tile;
    load_rows12<T>((const T*)a.attn, a.ld_attn, r0, min(a.rows_per_tile, a.n_rows - r0), bufB, lane, wq, [&] {
      stamp2<TL>(c, smem);  // rows requested
      ring_prologue(ring, wpw, s1, c.loff);
      // this CU's share of the projection's weights (waves 0, 1) and of M1(0)'s (waves 2, 3)
      if (a.warm) warm = touch_share(wq < 2 ? a.wp : a.w1, kSlab, wq & 1, lane);
      stamp2<TL>(c, smem);  // everything requested
    });
    stamp2<TL>(c, smem);  // rows stored
    lds_barrier();
  }
  for (int tile = blockIdx.x;;) {
    const int r0 = tile * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    stamp2<TL>(c, smem);
    // S1: the projection on the attention rows alone - the skip rows and the per-column vectors come in through group B meanwhile (the
    // CU's memory pipe holds ~64 wave-instructions: every load in front of the first MFMA costs its share of a 2-us round, in-kernel
    // timeline) - then + b_p + x; x1 (rounded) over the skip rows it was computed from (each lane rewrites the positions it read)
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm128<T>(bufB, lane, ring, wpw, s1, w1c(0), s1, c.loff, acc);
    touch(seg_a(0), hc > 1 ? seg_a(1) : nullptr, kSlab);  // M1(0) (4 us away) and M1(1)
    stamp2<TL>(c, smem);
    lds_barrier();  // group B's skip rows and vectors are in LDS
    round_rows<T, true, true>(acc, bufC, red, lane, wq, vec, 0);
    stamp2<TL>(c, smem);
    lds_barrier();
    // S2: LayerNorm_mlp(x1) without its affine part -> bufB (every wave has read the attention rows)
    if (a.dbg & 2) round_rows<T, false>(acc, bufB, nullptr, lane, wq);
    else normalise_rows<T>(acc, red, a.eps1, bufB, lane, wq);
    stamp2<TL>(c, smem);
    lds_barrier();
    // the MLP's first Linear, chunk by chunk (+ GELU); step hc is group B's alone
    for (int t = 0; t < hc; ++t) {
      stamp2<TL>(c, smem);
      init_acc<T, false>(acc, vec, 512 + 512 * t, nullptr, lane, wq);
      const char* nxt = t + 1 < hc ? w1c(t + 1) : (qc > 0 ? wqc(0) : wpw);
      if (a.prio_a == 1) __builtin_amdgcn_s_setprio(2);
      gemm128<T>(bufB, lane, ring, w1c(t), s1, nxt, s1, c.loff, acc);
      if (a.prio_a == 1) __builtin_amdgcn_s_setprio(0);
      touch(t == 0 ? nullptr : seg_a(t + 1), a.w2 + (int64_t)t * kSlab, sw2);  // the next step's M1(t + 1) (t = 0: touched behind P) and M2(t)
      stamp2<TL>(c, smem);
      if (a.dbg & 1) round_rows<T, false>(acc, hbuf(t), nullptr, lane, wq);
      else gelu_rows<T>(acc, hbuf(t), lane, wq);
      stamp2<TL>(c, smem);
      lds_barrier();
    }
    stamp2<TL>(c, smem);
    if (qc > 1) touch(hc == 1 ? seg_a(hc) : nullptr, a.wq + (int64_t)8 * kSlab, kSlab);  // group B's first chunk of the trailing projection (Q0: behind M1(hc - 1))
    const int tile_next = tile + (int)gridDim.x;
    if (qc == 0 && tile_next < a.n_tiles) {
      // nothing to do in this step and bufB free (every wave of the group is behind its last M1 segment): the NEXT panel's attention rows
      const int rn = tile_next * a.rows_per_tile;
      load_rows12<T>((const T*)a.attn, a.ld_attn, rn, min(a.rows_per_tile, a.n_rows - rn), bufB, lane, wq, [] {});
    }
    lds_barrier();  // step hc
    stamp2<TL>(c, smem);
    // S8: x2 [+ latent skip] -> global as whole rows
    {
      const unsigned char* xb = hbuf(hc);
      int l0 = lane, w0 = wq;
      asm volatile("" : "+v"(l0), "+s"(w0));
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int row = w0 * 12 + i;
        u32x4 v = *reinterpret_cast<const u32x4*>(xb + row * kRowBytes + ((l0 ^ (row & 15)) << 4));
        if (row < nr) {
          if (a.extra != nullptr) {
            // the latent skip rides on the last block's output, added to the block's ROUNDED output as `x + skip` does
            const u32x4 e = *reinterpret_cast<const u32x4*>((const T*)a.extra + (int64_t)(r0 + row) * a.ld_extra + l0 * 8);
            float p[4], s[4];
            unpack4<T>(u32x2{v[0], v[1]}, p);
            unpack4<T>(u32x2{e[0], e[1]}, s);
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] += s[r];
            const u32x2 lo = pack4<T>(p);
            unpack4<T>(u32x2{v[2], v[3]}, p);
            unpack4<T>(u32x2{e[2], e[3]}, s);
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] += s[r];
            const u32x2 hi = pack4<T>(p);
            v = u32x4{lo[0], lo[1], hi[0], hi[1]};
          }
          *reinterpret_cast<u32x4*>((T*)a.xout + (int64_t)(r0 + row) * a.ld_out + l0 * 8) = v;
        }
      }
    }
    stamp2<TL>(c, smem);
    lds_barrier();
    // the even chunks of the trailing projection, staged (rounded) in x2's buffer
    for (int k = 0; k < qc; k += 2) {
      stamp2<TL>(c, smem);
      init_acc<T, false>(acc, vec, 1024 + 512 * hc + 512 * k, nullptr, lane, wq);
      set_chunk_prio(a.prio_q, k);
      gemm128<T>(bufB, lane, ring, wqc(k), s1, k + 2 < qc ? wqc(k + 2) : wpw, s1, c.loff, acc);
      if (a.prio_q) __builtin_amdgcn_s_setprio(0);
      touch(k + 2 < qc ? a.wq + (int64_t)(8 * (k + 2)) * kSlab : nullptr, k + 3 < qc ? a.wq + (int64_t)(8 * (k + 3)) * kSlab : nullptr, kSlab);
      stamp2<TL>(c, smem);
      round_rows<T, false>(acc, hbuf(hc), nullptr, lane, wq);
      if (!(a.dbg & 4)) store_staged<T>(hbuf(hc), (T*)a.qout + (int64_t)r0 * a.ld_q + k * kCh, a.ld_q, nr, lane, wq);
      stamp2<TL>(c, smem);
    }
    if (qc > 0) lds_barrier();  // (the groups' chunks are independent of each other: one barrier behind them all, for the next panel's S0)
    tile = tile_next;
    if (tile >= a.n_tiles) break;
    // S0 of the next panel (without a trailing projection its rows came in during step hc, two barriers ago)
    if (qc > 0) {
      const int rn = tile * a.rows_per_tile;
      load_rows12<T>((const T*)a.attn, a.ld_attn, rn, min(a.rows_per_tile, a.n_rows - rn), bufB, lane, wq, [] {});
      lds_barrier();
    }
  }
  touch_done(warm);
}

template <typename T, bool TL>
__device__ __forceinline__ void role_b(const Chain2Args& a, Ctx2& c, unsigned char* smem) {
  const int hc = a.hc, qc = a.qc, lane = c.lane, wq = __builtin_amdgcn_readfirstlane(c.wq);
  // bufA: the even hidden chunks (x2 / group A's staged projection outputs when hc is even); bufB: the A operand of P, M1 and the
  // trailing projection (attention rows, LN(x1), LN'(x2)); bufC: skip rows, then x1, then the odd hidden chunks
  unsigned char* const bufB = smem + kBufBytes;
  unsigned char* const bufC = smem + 2 * kBufBytes;
  float* const red = reinterpret_cast<float*>(smem + kRed2Off);
  const unsigned char* const vec = smem + kVecOff;
  auto hbuf = [&](int t) { return smem + (t & 1) * (2 * kBufBytes); };
  const int64_t s1 = kSlab, s2 = (int64_t)hc * kSlab;
  auto w2c = [&](int k) { return a.w2 + (int64_t)(2 * wq) * s2 + (int64_t)k * kSlab; };
  auto wqc = [&](int k) { return a.wq + (int64_t)(8 * k + 2 * wq) * kSlab; };
  frag8 ring[2][8];
  f32x4 acc[3][8];
  lds_barrier();  // S0 is group A's (the attention rows)
  for (int tile = blockIdx.x;;) {
    const int r0 = tile * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    stamp2<TL>(c, smem);
    // beside the projection: the skip rows -> bufC; in the first panel also the per-column vectors -> LDS and the weight ring's first fragments
    if (tile == (int)blockIdx.x) {
      VecCopy vc0, vc1;
      load_rows12<T>((const T*)a.xres, a.ld_x, r0, nr, bufC, lane, wq, [&] {
        vc0.request(a, wq * 64 + lane);
        vc1.request(a, 256 + wq * 64 + lane);
      });
      vc0.store(a, wq * 64 + lane, smem);
      vc1.store(a, 256 + wq * 64 + lane, smem);
      ring_prologue(ring, w2c(0), s2, c.loff);
    } else {
      load_rows12<T>((const T*)a.xres, a.ld_x, r0, nr, bufC, lane, wq, [] {});
    }
    stamp2<TL>(c, smem);
    lds_barrier();  // the skip rows are in LDS
    lds_barrier();  // x1 is group A's
    stamp2<TL>(c, smem);
    // S2: x2's accumulators start at b_2 + x1
    init_acc<T, true>(acc, vec, 512 + 512 * hc, bufC, lane, wq);
    stamp2<TL>(c, smem);
    lds_barrier();
    stamp2<TL>(c, smem);
    lds_barrier();  // step 0 is group A's
    // the MLP's second Linear, accumulating over the hidden chunks as group A delivers them
    for (int t = 1; t <= hc; ++t) {
      stamp2<TL>(c, smem);
      const char* nxt = t < hc ? w2c(t) : (qc > 1 ? wqc(1) : w2c(0));
      const int64_t ns = (t < hc || qc <= 1) ? s2 : s1;
      gemm128<T>(hbuf(t - 1), lane, ring, w2c(t - 1), s2, nxt, ns, c.loff, acc);
      stamp2<TL>(c, smem);
      if (t == hc) {  // x2 (rounded) -> the h buffer nobody reads any more, for group A to store
        if (qc > 0) round_rows<T, true>(acc, hbuf(hc), red, lane, wq);
        else round_rows<T, false>(acc, hbuf(hc), nullptr, lane, wq);
        stamp2<TL>(c, smem);
      }
      lds_barrier();
    }
    // S8: LayerNorm_attn'(x2) without its affine part -> bufB
    if (qc > 0) normalise_rows<T>(acc, red, a.epsq, bufB, lane, wq);
    stamp2<TL>(c, smem);
    lds_barrier();
    // the odd chunks of the trailing projection, staged in the other h buffer
    for (int k = 1; k < qc; k += 2) {
      stamp2<TL>(c, smem);
      init_acc<T, false>(acc, vec, 1024 + 512 * hc + 512 * k, nullptr, lane, wq);
      const bool last = k + 2 >= qc;
      set_chunk_prio(a.prio_q, k);
      gemm128<T>(bufB, lane, ring, wqc(k), s1, last ? w2c(0) : wqc(k + 2), last ? s2 : s1, c.loff, acc);
      if (a.prio_q) __builtin_amdgcn_s_setprio(0);
      stamp2<TL>(c, smem);
      round_rows<T, false>(acc, hbuf(hc + 1), nullptr, lane, wq);
      if (!(a.dbg & 4)) store_staged<T>(hbuf(hc + 1), (T*)a.qout + (int64_t)r0 * a.ld_q + k * kCh, a.ld_q, nr, lane, wq);
      stamp2<TL>(c, smem);
    }
    if (qc > 0) lds_barrier();
    tile += (int)gridDim.x;
    if (tile >= a.n_tiles) break;
    if (qc > 0) lds_barrier();  // S0 of the next panel is group A's
  }
}

template <typename T, bool TL = false>
__global__ __launch_bounds__(512, 1) void gt_chain2_kernel(Chain2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  Ctx2 c;
  c.lane = tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.wq = c.wave & 3;  // waves wq and wq + 4 share a SIMD
  c.loff = c.lane * 16;
  c.tl_n = 0;
  stamp2<TL>(c, smem);  // 0: entry
  if ((int)blockIdx.x >= a.n_tiles) return;
  if (c.wave < 4) {
    if (a.prio_a == 2) __builtin_amdgcn_s_setprio(2);
    role_a<T, TL>(a, c, smem);
  } else {
    role_b<T, TL>(a, c, smem);
  }
  if constexpr (TL) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c.lane < kTl2Slots)
      a.timeline[((size_t)blockIdx.x * 8 + c.wave) * kTl2Slots + c.lane] =
          c.lane < c.tl_n ? reinterpret_cast<const unsigned long long*>(smem + kTl2Off)[c.wave * kTl2Slots + c.lane] : 0ull;
  }
}

template <typename T>
static int launch_chain2(const Chain2Args& a, hipStream_t st) {
  static PerDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_chain2_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kChain2Smem);
  });
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  if (a.timeline != nullptr) {
    static PerDeviceOnce tl_once;
    tl_once.run([&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_chain2_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kTl2Off + 8 * kTl2Slots * 8);
    });
    hipLaunchKernelGGL((gt_chain2_kernel<T, true>), dim3(grid), dim3(512), kTl2Off + 8 * kTl2Slots * 8, st, a);
    return check_launch("gt_chain2_kernel<timeline>");
  }
  hipLaunchKernelGGL((gt_chain2_kernel<T>), dim3(grid), dim3(512), kChain2Smem, st, a);
  return check_launch("gt_chain2_kernel");
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_chain2_fwd(const anemoi_gt_chain2_args_t* p, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(p != nullptr, "gt_chain2_fwd: null argument block");
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "gt_chain2_fwd: 16-bit model dtypes only");
  ANEMOI_REQUIRE(p->n_rows >= 0 && p->channels == kCh, "gt_chain2_fwd: channels=%d (this kernel is built for %d)", p->channels, kCh);
  if (p->n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(p->hidden > 0 && p->hidden % kCh == 0 && p->q_out_features >= 0 && p->q_out_features % kCh == 0,
                 "gt_chain2_fwd: hidden=%d and q_out_features=%d must be multiples of %d", p->hidden, p->q_out_features, kCh);
  const int n_vec = 2 * kCh + p->hidden + p->q_out_features;
  if (n_vec > (p->timeline != nullptr ? kVecMaxElemsTl : kVecMaxElems)) return ANEMOI_E_UNSUPPORTED;  // the per-column vectors must fit their LDS region
  ANEMOI_REQUIRE(p->attn && p->x_res && p->wp && p->w1 && p->w2 && p->vec && p->x_out, "gt_chain2_fwd: null operand");
  ANEMOI_REQUIRE(p->q_out_features == 0 || (p->wq && p->q_out), "gt_chain2_fwd: the trailing projection needs wq and q_out");
  ANEMOI_REQUIRE(p->q_out_features == 0 || p->extra == nullptr, "gt_chain2_fwd: the trailing projection reads x2 before a second residual is added: not both");
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  ANEMOI_REQUIRE(al16(p->attn) && al16(p->x_res) && al16(p->wp) && al16(p->w1) && al16(p->w2) && al16(p->x_out) && al16(p->wq) && al16(p->q_out) &&
                     al16(p->extra) && al16(p->vec),
                 "gt_chain2_fwd: operands must be 16-byte aligned");
  ANEMOI_REQUIRE(p->ld_attn >= kCh && p->ld_x >= kCh && p->ld_out >= kCh && p->ld_attn % 8 == 0 && p->ld_x % 8 == 0 && p->ld_out % 8 == 0 &&
                     (p->extra == nullptr || (p->ld_extra >= kCh && p->ld_extra % 8 == 0)) &&
                     (p->q_out_features == 0 || (p->ld_q >= p->q_out_features && p->ld_q % 8 == 0)),
                 "gt_chain2_fwd: leading dimensions too small or not multiples of 8 elements (rows move as 16-byte pieces)");
  Chain2Args a{};
  a.attn = p->attn; a.ld_attn = p->ld_attn;
  a.xres = p->x_res; a.ld_x = p->ld_x;
  a.wp = (const char*)p->wp;
  a.w1 = (const char*)p->w1; a.hc = p->hidden / kCh;
  a.w2 = (const char*)p->w2;
  a.wq = (const char*)p->wq; a.qc = p->q_out_features / kCh;
  a.vec = p->vec;
  a.eps1 = p->ln1_eps; a.epsq = p->lnq_eps;
  a.extra = p->extra; a.ld_extra = p->ld_extra;
  a.xout = p->x_out; a.ld_out = p->ld_out;
  a.qout = p->q_out; a.ld_q = p->ld_q;
  a.timeline = reinterpret_cast<unsigned long long*>(p->timeline);
  static const int prio_a = env_int(getenv("ANEMOI_CHAIN2_PRIO_A"), 0, 0, 2);
  a.prio_a = prio_a;
  static const int dbg = env_int(getenv("ANEMOI_CHAIN2_DBG"), 0, 0, 7);
  a.dbg = dbg;
  static const int prio_q = env_int(getenv("ANEMOI_CHAIN2_PRIO_Q"), 0, 0, 1);
  a.prio_q = prio_q;
  static const int warm = env_int(getenv("ANEMOI_CHAIN2_WARM"), 1, 0, 1);
  a.warm = warm;
  a.n_rows = p->n_rows;
  a.rows_per_tile = p->rows_per_tile > 0 ? p->rows_per_tile : anemoi_gt_chain_rows_per_tile(p->n_rows);
  ANEMOI_REQUIRE(a.rows_per_tile <= kPanel, "gt_chain2_fwd: rows_per_tile=%d exceeds the %d-row panel", a.rows_per_tile, kPanel);
  a.n_tiles = (a.n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  hipStream_t st = as_stream(stream);
  return dtype == ANEMOI_BF16 ? launch_chain2<bf16_t>(a, st) : launch_chain2<f16_t>(a, st);
}
