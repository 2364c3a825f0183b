// Shared device code of the ROLE-SPLIT row-resident chain kernels (round 5; gt_chain2.hip: the GraphTransformer block tail; round 6:
// gt_rowchain.hip: the mappers' embedding -> LayerNorm -> projection chains; experiments/gnn_chain2.hip: GraphConv's edge MLP).  The workgroup's eight waves form two groups of four (waves w and w + 4 share a SIMD); a wave owns a 48 x 128
// output slab of its group's GEMM segment: 24 accumulator quads, the three A fragments of a K-step shared by eight column blocks, its
// 8 KiB of weights per K-step through a register ring of only two K-steps (tools/role_split_probe.hip: one such group pulls the weight
// stream through the CU's L1 path as fast as eight 64-column waves do, in 200 registers).  What the groups do differs per kernel; the
// pieces here are the GEMM segment, the accumulator start values, the epilogue forms and the L2 warm-up of weights one step ahead.
#pragma once
#include "chain_core.h"

namespace anemoi {

// Row streams (panel rows in, output rows out) are touched once per launch.  ANEMOI_CHAIN2_NT (build-time mask): 1 = the loads carry the
// non-temporal hint, so that in the XCD's L2 they do not push out the WEIGHTS, which all 32 CUs of the XCD read - and, in multi-round
// launches, read again (res 6 forward -2.4 %); 2 = the stores too (O96 +1.7 %: the attention launch behind reads them - off); 4 = the
// stores write through (sc1) instead of sitting dirty in this XCD's L2 until the end of the kernel (O96 -0.4 % on one box, 0 on the next).
// Default 1.
#ifndef ANEMOI_CHAIN2_NT
#define ANEMOI_CHAIN2_NT 1
#endif
__device__ __forceinline__ u32x4 stream_load(const u32x4* p) {
  if constexpr ((ANEMOI_CHAIN2_NT & 1) != 0) return __builtin_nontemporal_load(p);
  else return *p;
}
__device__ __forceinline__ void stream_store(u32x4 v, u32x4* p) {
  if constexpr ((ANEMOI_CHAIN2_NT & 2) != 0) __builtin_nontemporal_store(v, p);
  else if constexpr ((ANEMOI_CHAIN2_NT & 4) != 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");  // write-through
  else *p = v;
}

// 16 K-steps (K = 512) of this wave's 48 x 128 tile.  A fragments from the swizzled LDS panel (the next K-step's requested before this
// one's MFMAs), B fragments from a register ring of two K-steps x 8 fragments, each slot refilled right behind its three MFMAs with
// the fragment of two K-steps ahead - of this segment or, in its last pair, of the wave's NEXT segment (`nxt`).  The wave's 128
// columns are two adjacent 64-column slabs of the fragment-major image: streams `cur` and `cur + cs`.
// nq: pairs of K-steps (K = 64 nq; 8 = the 512-wide segment every caller but a narrow first GEMM uses).
template <typename T>
__device__ __forceinline__ void gemm128(const unsigned char* abuf, int lane, frag8 (&ring)[2][8], const char* cur, int64_t cs, const char* nxt,
                                        int64_t ns, uint32_t loff, f32x4 (&acc)[3][8], int nq = 8) {
  asm volatile("" : "+v"(lane));
  const int x = lane & 15, ks = lane >> 4;
  const unsigned char* arow = abuf + x * kRowBytes;
  frag8 fa[3];
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) fa[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16 * kRowBytes + ((ks ^ x) << 4));
  const int last = 2 * nq - 1;
#pragma unroll 1
  for (int q = 0; q < nq; ++q) {
    const char* p0 = q < nq - 1 ? cur + (q + 1) * 8192 : nxt;
    const char* p1 = q < nq - 1 ? cur + cs + (q + 1) * 8192 : nxt + ns;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int st = q * 2 + j;
      const int sn = st < last ? st + 1 : last;  // (the last step re-reads its own fragments: no branch in the stream)
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

// The same for a 48 x 64 tile (ONE 64-column slab of the image, acc[mi][0..3]) on the SAME ring registers, as four K-steps x 4 fragments:
// K-step j of the stream sits in ring[j & 1][(j >> 1) * 4 + ni].  Used where all eight waves of the workgroup share ONE segment (the
// projection: 512 columns = 8 waves x 64), so that its epilogue is spread over eight waves instead of four.  In its last group of four
// K-steps the slots are refilled with the first fragments of the wave's NEXT segment in the layout gemm128 expects (two K-steps of the
// two slabs `nxt`, `nxt + ns`); a gemm128 hands over to a gemm64 with ns = 8192 (K-steps 2, 3 of the one slab behind K-steps 0, 1).
// ng: groups of four K-steps (K = 128 ng; 4 = the 512-wide segment; fewer: a narrow first GEMM on a zero-padded operand).
template <typename T>
__device__ __forceinline__ void gemm64(const unsigned char* abuf, int lane, frag8 (&ring)[2][8], const char* cur, const char* nxt, int64_t ns,
                                       uint32_t loff, f32x4 (&acc)[3][8], int ng = 4) {
  asm volatile("" : "+v"(lane));
  const int x = lane & 15, ks = lane >> 4;
  const unsigned char* arow = abuf + x * kRowBytes;
  frag8 fa[3];
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) fa[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16 * kRowBytes + ((ks ^ x) << 4));
  const int last = ng * 4 - 1;
#pragma unroll 1
  for (int q = 0; q < ng; ++q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int st = q * 4 + j;
      const int sn = st < last ? st + 1 : last;  // (the last step re-reads its own fragments: no branch in the stream)
      frag8 fn[3];
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) fn[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16 * kRowBytes + (((sn * 4 + ks) ^ x) << 4));
      __builtin_amdgcn_sched_barrier(0);
      const char* pf = q < ng - 1 ? cur + (q + 1) * 16384 + j * 4096 : (j < 2 ? nxt + j * 4096 : nxt + ns + (j - 2) * 4096);
      const gptr_t g0 = uniform_ptr(pf);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
        for (int mi = 0; mi < 3; ++mi) acc[mi][ni] = cmfma<T>(ring[j & 1][(j >> 1) * 4 + ni], fa[mi], acc[mi][ni]);
        ring[j & 1][(j >> 1) * 4 + ni] = *reinterpret_cast<gfrag_t>(g0 + loff + ni * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) fa[mi] = fn[mi];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// the wave's first weight fragments of a gemm64 segment (four K-steps of the one slab)
__device__ __forceinline__ void ring_prologue64(frag8 (&ring)[2][8], const char* f0, uint32_t loff) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const gptr_t g0 = uniform_ptr(f0 + j * 4096);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      ring[j & 1][(j >> 1) * 4 + ni] = *reinterpret_cast<gfrag_t>(g0 + loff + ni * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// The wave's 48 x 64 block (acc[mi][0..3], wave w8 of eight: columns 64 w8 ..) + vec[column] + the values the panel buffer `dst` holds at
// the same positions (the projection's bias and the skip rows), rounded to the model dtype back into `dst`; acc keeps the ROUNDED values;
// per-wave (mean, M2) of every row over the wave's 64 columns -> red[row][w8].
template <typename T>
__device__ __forceinline__ void round_rows64_add_stats(f32x4 (&acc)[3][8], unsigned char* dst, float* red, int lane, int w8, const unsigned char* vec) {
  const LaneCtx lc = lane_ctx(lane, w8);
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) {
    unsigned char* drow = dst + (mi * 16 + lc.x) * kRowBytes;
    u32x2 rb[4], rr[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      rb[ni] = *reinterpret_cast<const u32x2*>(vec + (w8 * 64 + ni * 16 + lc.g * 4) * 2);
      rr[ni] = *reinterpret_cast<const u32x2*>(drow + lc.coff[ni]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float o[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      float b[4], r[4];
      unpack4<T>(rb[ni], b);
      unpack4<T>(rr[ni], r);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] += b[k] + r[k];
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
// LayerNorm without the affine part from eight 64-column partials per row (merged in wave order, Chan et al.): the rounded values in
// acc[mi][0..3] normalised and stored (model dtype) into the panel buffer `dst`.
template <typename T>
__device__ __forceinline__ void normalise_rows64(const f32x4 (&acc)[3][8], const float* red, float eps, unsigned char* dst, int lane, int w8) {
  const LaneCtx lc = lane_ctx(lane, w8);
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) {
    const f32x4* pr = reinterpret_cast<const f32x4*>(red + (mi * 16 + lc.x) * 16);
    const f32x4 p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
    const float mu = (((p0[0] + p0[2]) + (p1[0] + p1[2])) + ((p2[0] + p2[2]) + (p3[0] + p3[2]))) * 0.125f;
    float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + ((p2[1] + p2[3]) + (p3[1] + p3[3]));
    const float d0 = p0[0] - mu, d1 = p0[2] - mu, d2 = p1[0] - mu, d3 = p1[2] - mu;
    const float d4 = p2[0] - mu, d5 = p2[2] - mu, d6 = p3[0] - mu, d7 = p3[2] - mu;
    m2 = fmaf(64.0f, ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7)), m2);
    const float rstd = rsqrtf(m2 * (1.0f / (float)kCh) + eps);
    const float nm = -mu * rstd;
    unsigned char* drow = dst + (mi * 16 + lc.x) * kRowBytes;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaf(acc[mi][ni][r], rstd, nm);
      *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pack4<T>(o);
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
// STORE = false: the rounded values stay in acc only (dst unused) - the pipelined row chain keeps a panel in registers across a barrier.
template <typename T, bool STATS, bool ADD = false, bool STORE = true>
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
      if constexpr (STORE) *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pk;
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

// The same in REGISTERS: acc <- (acc - mean) rstd, unrounded (a round_rows<T, false> writes - and rounds - it later)
template <typename T>
__device__ __forceinline__ void normalise_regs(f32x4 (&acc)[3][8], const float* red, float eps, int lane, int wq) {
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
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = fmaf(acc[mi][ni][r], rstd, nm);
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
    v[i] = stream_load(reinterpret_cast<const u32x4*>(src + (int64_t)(r0 + min(row, nr - 1)) * ld + lane * 8));
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
    if (row < nr) stream_store(v, reinterpret_cast<u32x4*>(out + (int64_t)row * ld + wq * 128 + sl * 8));
  }
}

// L2 warm-up of a weight segment one step ahead of its use.  All CUs of an XCD walk the same weight stream at the same time behind a ring
// of only two K-steps, so a line that misses the XCD's L2 (in a 16-layer model every layer's weights have left the caches since the
// last forward) exposes the HBM latency to all 32 of them at once.  Each CU therefore TOUCHES its 1/32 share of the segments of the
// NEXT step - one dword per 128-byte line, 64 lines per instruction - from group A's waves right behind their GEMM, where the
// epilogue that follows hides the latency; by the time a ring asks for the lines they are in the L2 (observed: workgroup b runs on XCD
// b % 8, so b / 8 numbers the CUs of an XCD - for speed only: a different placement warms less, nothing depends on it).
// A segment = the 512 x 512 weight of one group step: eight 64-KiB slabs `piece` bytes apart; `half` selects 64 of the CU's 128 lines.
struct Warm {
  unsigned v[3];
  int first, end;  // PART: this lane's first line of the wave's share, one past the share's last line
  int n_it;        // PART: loads per touch (wave-uniform)
};
// PART = false: all 256 CUs work, workgroup b touches the fixed 1/32 share (b >> 3) & 31 (observed: b runs on XCD b % 8), its line recomputed
// from the workgroup id at every touch (nothing kept live).  PART = true (a launch of fewer than 256 workgroups): the launch's workgroups on
// my XCD share the segment's 4096 lines out among themselves - ALL of them (10 242 rows = 214 panels of 48: 26 or 27 CUs per XCD; with a
// fixed 1/32 share each, 16 % of every weight stayed cold: the O96 forward 2.81 instead of 2.55 ms, profiles/r05_chain2_touch_coverage.txt).
// Two waves per share (`half`), up to 192 lines per wave = full coverage from 11 workgroups per XCD on; the bounds are computed once per kernel.
template <bool PART>
__device__ __forceinline__ void warm_init(Warm& w, int half, int lane) {
  w.v[0] = w.v[1] = w.v[2] = 0u;
  if constexpr (PART) {
    const int xcd = (int)blockIdx.x & 7;
    const int n_cu = ((int)gridDim.x - xcd + 7) >> 3;
    const int s = (int)blockIdx.x >> 3;
    const int lo = (s * 4096) / n_cu, hi = ((s + 1) * 4096) / n_cu;
    const int mid = lo + ((hi - lo + 1) >> 1);
    w.first = (half ? mid : lo) + lane;
    w.end = half ? hi : mid;
    w.n_it = __builtin_amdgcn_readfirstlane((w.end - (half ? mid : lo) + 63) >> 6);
  }
}
template <bool PART>
__device__ __forceinline__ void touch_share(Warm& w, const char* seg, int64_t piece, int half, int lane) {
  if constexpr (!PART) {
    const int line = (((int)blockIdx.x >> 3) & 31) * 128 + half * 64 + lane;  // 0 .. 4095
    w.v[0] = *reinterpret_cast<const unsigned*>(seg + (int64_t)(line >> 9) * piece + (line & 511) * 128);
  } else {
    int line = w.first;
    asm volatile("" : "+v"(line));  // (else the line addresses of every call site are computed up front and live - spilled - through the kernel)
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      if (it < w.n_it) {  // (wave-uniform)
        const int l = min(line, w.end - 1);  // (lanes beyond the share re-touch its last line: no divergence)
        w.v[it] = *reinterpret_cast<const unsigned*>(seg + (int64_t)(l >> 9) * piece + (l & 511) * 128);
        line += 64;
      }
    }
  }
}
// (the touched values must stay "in use" until the next touch: the compiler then keeps their registers and counts the loads)
template <bool PART>
__device__ __forceinline__ void touch_done(Warm& w) {
  if constexpr (PART) asm volatile("" : "+v"(w.v[0]), "+v"(w.v[1]), "+v"(w.v[2]));
  else asm volatile("" : "+v"(w.v[0]));
}

}  // namespace anemoi
