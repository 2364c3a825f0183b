// Fused graph-transformer edge attention (forward) for gfx950.
//
// Spec: the reference's _gt_fwd Triton kernel (models/src/anemoi/models/triton/gt.py:81-179) and
// GraphTransformerConv (models/src/anemoi/models/layers/conv.py:84-147).  Not a port: the Triton kernel
// runs one program per destination with [H,C] register tiles; here one 64-lane wavefront owns one
// destination row of D = H*C channels, each lane holds VEC = D/64 contiguous channels (a 16-byte load
// for bf16 at D=512), a head spans LPH = C/VEC adjacent lanes and the per-head <q,k> dot is finished
// with a DPP butterfly inside the row.  Online softmax state (m, l) and the accumulator stay in fp32
// registers; the edge loop is software-pipelined one edge ahead.
//
// EDGE_FUSED variant: E = edge_attr @ W_e^T + b_e is never formed.  Using linearity,
//   <q_h, k_h + W_h a + b_h>       = <q_h,k_h> + sum_f a_f <q_h, W_h[:,f]> + <q_h,b_h>
//   sum_e p_e (v_h + W_h a_e + b_h) = sum_e p_e v_h + W_h (sum_e p_e a_e) + b_h sum_e p_e
// so per destination we build qw[h][f] once (W' = [W_e | b_e] staged in LDS), per edge we only touch the
// fe_pad fp32 edge features (wave-uniform -> scalar loads) and at the end apply W' to the weighted feature
// sum.  HBM traffic per layer drops from 2(4ND + MD) to 2*4ND + 4*M*fe_pad bytes.
#include <type_traits>

#include "common.h"

namespace anemoi {

#ifndef ANEMOI_ATTN_WPB
#define ANEMOI_ATTN_WPB 4
#endif
constexpr int kWavesPerBlock = ANEMOI_ATTN_WPB;
#ifndef ANEMOI_ATTN_MIN_WAVES
#define ANEMOI_ATTN_MIN_WAVES (16 / ANEMOI_ATTN_WPB)
#endif
using f32x2 = __attribute__((ext_vector_type(2))) float;
// Where a destination's loads are requested (profiles/r05_attention_header_ab.txt, same-box A/Bs): the source ids travel WITH the q slice,
// ahead of the qw set-up (-6 % of the launch), and the scalar part of the next header (id, edge range) one destination ahead.  The
// round-5 variants that were measured slower or equal - first header beside the W' staging, first ring fill ahead of the qw set-up,
// next header behind the edge loop, the lane split of the feature terms, the timing-only ablation builds - live in
// csrc/experiments/gt_attention_r05_variants.hip (tools/build_alt.sh builds them into an alternative library for same-box A/Bs).
constexpr int kAttnPF = 3;  // edges of K|V rows in flight per wave (modulo-unrolled ring, counted waits)

template <int VEC>
struct EdgeRow {
  float k[VEC];
  float v[VEC];
  float e[VEC];
};

// ---------------------------------------------------------------------------------------------- fast path
template <typename T, int VEC, int LPH, bool HAS_E>
__global__ __launch_bounds__(64 * kWavesPerBlock) void gt_attn_fwd_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    const T* __restrict__ e, int64_t lde, const int32_t* __restrict__ row, const int32_t* __restrict__ colptr,
    const T* __restrict__ addend, int64_t ldadd, T* __restrict__ out, int64_t ldo, float* __restrict__ lse, int n_dst,
    int H, float scale, float drop_p, uint64_t drop_seed) {
  const int lane = threadIdx.x & 63;
  const int d = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (d >= n_dst) return;
  const int beg = __builtin_amdgcn_readfirstlane(colptr[d]);
  const int end = __builtin_amdgcn_readfirstlane(colptr[d + 1]);
  const int c0 = lane * VEC;
  const float inv_keep = 1.0f / (1.0f - drop_p);

  float qv[VEC], acc[VEC];
  load_vec<T, VEC>(q + (int64_t)d * ldq + c0, qv);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    qv[j] *= scale;
    acc[j] = 0.f;
  }
  float m = -INFINITY, l = 0.f;

  EdgeRow<VEC> cur, nxt;
  auto fetch = [&](int ei, EdgeRow<VEC>& r) {
    const int s = __builtin_amdgcn_readfirstlane(row[ei]);
    load_vec<T, VEC>(k + (int64_t)s * ldk + c0, r.k);
    load_vec<T, VEC>(v + (int64_t)s * ldv + c0, r.v);
    if constexpr (HAS_E) load_vec<T, VEC>(e + (int64_t)ei * lde + c0, r.e);
  };
  if (beg < end) fetch(beg, nxt);
  for (int ei = beg; ei < end; ++ei) {
    cur = nxt;
    if (ei + 1 < end) fetch(ei + 1, nxt);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if constexpr (HAS_E) {
        cur.k[j] += cur.e[j];
        cur.v[j] += cur.e[j];
      }
      dot = fmaf(qv[j], cur.k[j], dot);
    }
    dot = group_sum<LPH>(dot);
    const float m_new = fmaxf(m, dot);
    const float corr = __expf(m - m_new);  // first edge: exp(-inf) = 0
    const float p = __expf(dot - m_new);
    l = fmaf(l, corr, p);
    // dropout acts on the NORMALISED weight: the denominator sums every edge, the numerator the kept ones (scaled by 1 / (1 - p))
    const float pv = drop_p > 0.f ? p * attn_dropout_scale(drop_seed, ei, lane / LPH, drop_p, inv_keep) : p;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(acc[j], corr, pv * cur.v[j]);
    m = m_new;
  }

  const float inv = (end > beg) ? 1.0f / l : 0.f;
  float o[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) o[j] = acc[j] * inv;
  if (addend != nullptr) {
    float a[VEC];
    load_vec<T, VEC>(addend + (int64_t)d * ldadd + c0, a);
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] += a[j];
  }
  store_vec<T, VEC>(out + (int64_t)d * ldo + c0, o);
  if (lse != nullptr && (lane % LPH) == 0) lse[(int64_t)d * H + lane / LPH] = (end > beg) ? m + __logf(l) : 0.f;
}

// <a, b> over a lane's VEC channels.  16-bit types use v_dot2c_f32_{bf16,f16} on the packed pairs as loaded (exact
// products, fp32 accumulation): no per-element conversion of the K row.
template <typename T, int VEC>
__device__ __forceinline__ float dot_rows(const Vec<T, VEC>& a, const Vec<T, VEC>& b) {
  float acc = 0.f;
  if constexpr (sizeof(T) == 2 && (VEC % 2 == 0)) {
    const uint32_t* pa = reinterpret_cast<const uint32_t*>(&a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) {
      if constexpr (std::is_same<T, bf16_t>::value) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pa[i]), __builtin_bit_cast(bf2, pb[i]), acc, false);
      } else {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pa[i]), __builtin_bit_cast(h2, pb[i]), acc, false);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc = fmaf(to_float(a.v[i]), to_float(b.v[i]), acc);
  }
  return acc;
}

// Deferred running max: the rescale of the accumulators only runs (wave-uniform branch) when some head's score
// exceeds its running max by more than this; otherwise p = exp(s - m_old) <= e^8 is accumulated directly (fp32
// accumulators; the value after the final division is the same).
constexpr float kDeferThr = 8.0f;

// ---------------------------------------------------------------------------------------------- fused lin_edge
// LDS image of W' = [W_e | b_e | 0]: per lane a chunk of VEC*FE_PAD floats (+4 floats of padding so that
// 16 consecutive lanes' ds_read_b128 hit 16 distinct 16-byte bank slots).
template <int VEC, int FE_PAD>
struct WLayout {
  static constexpr int kChunk = VEC * FE_PAD + 4;
  static constexpr int kFloats = 64 * kChunk;
};

template <typename T, int VEC, int LPH, int FE_PAD, bool KVADJ>
__global__ __launch_bounds__(64 * kWavesPerBlock, ANEMOI_ATTN_MIN_WAVES) void gt_attn_fused_edge_fwd_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
    const float* __restrict__ feat, const float* __restrict__ w_packed, const int32_t* __restrict__ row,
    const int32_t* __restrict__ colptr, const int32_t* __restrict__ order, const T* __restrict__ addend, int64_t ldadd,
    T* __restrict__ out, int64_t ldo, float* __restrict__ lse, int n_dst, int H, float scale, int out_wt) {
  using L = WLayout<VEC, FE_PAD>;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];  // [64][kChunk]
  const int lane = threadIdx.x & 63;
  const int c0 = lane * VEC;

  // XCD-aware persistent schedule.  Workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on
  // it).  Each XCD gets one CONTIGUOUS slice of the destination range, so the K/V rows its waves gather (the mesh
  // neighbourhood of that slice: nodes are latitude/longitude sorted) stay resident in that XCD's 4 MiB L2 instead of
  // every L2 pulling every row from the fabric (8x the traffic).  Inside a slice, waves take destinations round-robin.
  const int xcd = blockIdx.x & 7;
  const int blocks_in_xcd = (gridDim.x - xcd + 7) >> 3;
  const int waves_in_xcd = blocks_in_xcd * kWavesPerBlock;
  const int wave_in_xcd = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * kWavesPerBlock + (threadIdx.x >> 6));
  const int per_xcd = (n_dst + 7) >> 3;
  const int d_lo = xcd * per_xcd;
  const int d_hi = min(n_dst, d_lo + per_xcd);

  using Raw = Vec<T, VEC>;  // a row slice as loaded (converted to fp32 only when consumed)
  // The header of a destination: its id and edge range (scalar registers, requested ONE DESTINATION AHEAD - the first one's in front of
  // the W' staging - so that a destination starts with its q slice and source ids, not with colptr), its q row slice and the source
  // ids of its first 64 in-edges.
  const int i0 = d_lo + wave_in_xcd;
  int h_d = 0, h_beg = 0, h_end = 0, h_src = 0;
  Raw h_q;
  auto load_scalars = [&](int i, int& d, int& beg, int& end) {
    d = order ? __builtin_amdgcn_readfirstlane(order[i]) : i;
    beg = __builtin_amdgcn_readfirstlane(colptr[d]);
    end = __builtin_amdgcn_readfirstlane(colptr[d + 1]);
  };
  int n_d = 0, n_beg = 0, n_end = 0;
  if (i0 < d_hi) load_scalars(i0, n_d, n_beg, n_end);
  // Stage W' = [W_e | b_e | 0] (fp32 [D][FE_PAD], packed once on the host side of the ABI) into LDS: all 16-byte
  // loads are issued before the first write (one memory round trip per workgroup).
  {
    constexpr int kQ = FE_PAD / 4;                 // float4 per channel row
    constexpr int kTotal = 64 * VEC * kQ;          // float4 in the image
    constexpr int kIter = (kTotal + 64 * kWavesPerBlock - 1) / (64 * kWavesPerBlock);
    float4 tmp[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const int idx = threadIdx.x + it * 64 * kWavesPerBlock;
      tmp[it] = idx < kTotal ? reinterpret_cast<const float4*>(w_packed)[idx] : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const int idx = threadIdx.x + it * 64 * kWavesPerBlock;
      if (idx < kTotal) {
        const int c = idx / kQ, qq = idx % kQ;  // channel c, features 4 qq .. 4 qq + 3 -> chunk layout [feature][channel]
        float* dst = w_lds + (c / VEC) * L::kChunk + (qq * 4) * VEC + (c % VEC);
        dst[0] = tmp[it].x;
        dst[VEC] = tmp[it].y;
        dst[2 * VEC] = tmp[it].z;
        dst[3 * VEC] = tmp[it].w;
      }
    }
  }
  __syncthreads();
  const float* wl = w_lds + lane * L::kChunk;

  const float sl2e = scale * 1.4426950408889634f;  // p = 2^((s' - m') * scale * log2 e)
  const float thr = kDeferThr / scale;
  constexpr int PF = kAttnPF;

  // `order` (optional): the destination processed at position i.  The ~768 destinations an XCD works on at one moment are
  // then a compact patch of the mesh instead of a whole latitude ring, so the K|V rows they gather fit that XCD's L2
  // (layers/graphcache.py builds it from the graph; outputs are written at their own rows, the result does not depend on it).
  for (int i = i0; i < d_hi; i += waves_in_xcd) {
    // W' lives in LDS and is re-read per destination: without this barrier the compiler hoists all VEC*FE_PAD values
    // into registers across the loop (256 VGPRs, 1 wave/SIMD) and the kernel becomes latency-bound.
    asm volatile("" ::: "memory");
    h_d = n_d, h_beg = n_beg, h_end = n_end;
    h_q = *reinterpret_cast<const Raw*>(q + (int64_t)h_d * ldq + c0);
    h_src = (lane < h_end - h_beg) ? row[h_beg + lane] : 0;
    if (i + waves_in_xcd < d_hi) load_scalars(i + waves_in_xcd, n_d, n_beg, n_end);
    const int d = h_d, beg = h_beg, end = h_end;
    const Raw q_raw = h_q;
    float qv[VEC], acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      qv[j] = to_float(q_raw.v[j]);
      acc[j] = 0.f;
    }
    // Scores are kept in units of 1/scale (s' = <q, k + e>, the softmax argument is scale * s'): the scale rides in the
    // exponent's multiplier (one multiply per edge less).
    // Edge loop in chunks of 64: the source ids of a chunk are fetched with ONE coalesced load (lane j holds
    // row[chunk + j]) and broadcast with v_readlane, so row loads never wait on a dependent scalar load.
    int chunk = beg, n = min(64, end - beg), my_src = h_src;  // the first chunk's ids came with the header
    Raw kb[PF], vb[PF];
    float fb[PF][FE_PAD];
    auto fetch = [&](int j, Raw& kr, Raw& vr, float (&fr)[FE_PAD]) {
      j = min(j, n - 1);  // refills past the end re-read the last edge: an UNCONDITIONAL load keeps the ring registers
                          // free of select/copy code (a conditional one made the compiler wait for the load at once)
      const int s = __builtin_amdgcn_readlane(my_src, j);
      const float* a = feat + (int64_t)(chunk + j) * FE_PAD;  // wave-uniform address -> scalar loads
      if constexpr (KVADJ) {
        // v = the D columns after k in the same buffer (the fused projection's layout): ONE address and an immediate
        // offset; the row offset in 32 bits (checked at launch) - 13 scalar instructions fewer per edge, and this
        // kernel is bound by instruction issue
        const T* kp = k + (uint32_t)((uint32_t)s * (uint32_t)ldk) + c0;
        kr = *reinterpret_cast<const Raw*>(kp);
        vr = *reinterpret_cast<const Raw*>(kp + 64 * VEC);
      } else {
        kr = *reinterpret_cast<const Raw*>(k + (int64_t)s * ldk + c0);
        vr = *reinterpret_cast<const Raw*>(v + (int64_t)s * ldv + c0);
      }
#pragma unroll
      for (int f = 0; f < FE_PAD; ++f) fr[f] = a[f];
    };
    // qw[f] = (1/LPH) * sum over the head's channels of q[c] * W'[c][f]  (pre-divided: every lane of the
    // head adds the same edge-feature term before the head butterfly).
    float qw[FE_PAD], sf[FE_PAD];
#pragma unroll
    for (int f = 0; f < FE_PAD; ++f) {
      // W' is stored [feature][channel] per lane: the VEC channels of a feature are contiguous (16-byte LDS reads) and
      // the channel pairs map onto packed FMAs without shuffles, here and in the final W' * sum(p a)
      float t = 0.f;
      if constexpr (VEC % 2 == 0) {
        f32x2 t2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < VEC; j += 2)
          t2 = __builtin_elementwise_fma(f32x2{qv[j], qv[j + 1]}, *reinterpret_cast<const f32x2*>(wl + f * VEC + j), t2);
        t = t2[0] + t2[1];
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) t = fmaf(qv[j], wl[f * VEC + j], t);
      }
      qw[f] = group_sum<LPH>(t) * (1.0f / LPH);
      sf[f] = 0.f;
    }
    float m = -INFINITY, l = 0.f;

    for (; chunk < end; chunk += 64) {
      if (chunk != beg) {
        n = min(64, end - chunk);
        my_src = (lane < n) ? row[chunk + lane] : 0;
      }
#pragma unroll
      for (int st = 0; st < PF; ++st) fetch(st, kb[st], vb[st], fb[st]);
      // Edges in GROUPS of PF inside ONE basic block: the ring slots always hold loaded rows (refills past the end re-read the last edge), an edge
      // beyond the chunk gets the score -inf (weight 0).  The group's scores are formed first, the running max is checked ONCE per group, and
      // the scheduler can interleave the PF independent dependency chains (the per-edge form below is one basic block per edge: dependent
      // packed FMAs back to back, a wait state each, and a scalar bound check + two branches per edge).  The hidden mesh's in-degrees are
      // multiples of 6 and the decoder's exactly 3: no padded edge there.  Same-box A/B against the per-edge loop (round 5's, in
      // csrc/experiments/gt_attention_r05_variants.hip): O96 -0.2 %, res 6 -0.2 %, N320 -0.6 % per forward; 88 instead of 94 VGPRs; per edge 60 instead of
      // 68 issue slots (2 s_nop instead of 8, 0.7 branches instead of 3) - profiles/r06_attention_grouped_ab.txt.
      for (int j0 = 0; j0 < n; j0 += PF) {
        float dots[PF];
#pragma unroll
        for (int st = 0; st < PF; ++st) {
          float dot = dot_rows<T, VEC>(q_raw, kb[st]);
          if constexpr (FE_PAD % 2 == 0) {
            f32x2 d2[2] = {{dot, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int f = 0; f < FE_PAD; f += 2)
              d2[(f / 2) & 1] = __builtin_elementwise_fma(f32x2{fb[st][f], fb[st][f + 1]}, f32x2{qw[f], qw[f + 1]}, d2[(f / 2) & 1]);
            d2[0] += d2[1];
            dot = d2[0][0] + d2[0][1];
          } else {
#pragma unroll
            for (int f = 0; f < FE_PAD; ++f) dot = fmaf(fb[st][f], qw[f], dot);
          }
          dot = group_sum<LPH>(dot);
          dots[st] = (j0 + st < n) ? dot : -INFINITY;  // (wave-uniform condition)
        }
        float mx = dots[0];
#pragma unroll
        for (int st = 1; st < PF; ++st) mx = fmaxf(mx, dots[st]);
        if (__builtin_amdgcn_ballot_w64(mx > m + thr) != 0) {  // always in the first group, rare afterwards
          const float m_new = fmaxf(m, mx);
          const float corr = __builtin_amdgcn_exp2f((m - m_new) * sl2e);  // 2^(-inf) = 0 in the first group
          l *= corr;
#pragma unroll
          for (int jj = 0; jj < VEC; ++jj) acc[jj] *= corr;
#pragma unroll
          for (int f = 0; f < FE_PAD; ++f) sf[f] *= corr;
          m = m_new;
        }
#pragma unroll
        for (int st = 0; st < PF; ++st) {
          const float p = __builtin_amdgcn_exp2f((dots[st] - m) * sl2e);  // 2^(-inf) = 0 for an edge beyond the chunk
          l += p;
#pragma unroll
          for (int jj = 0; jj < VEC; ++jj) acc[jj] = fmaf(p, to_float(vb[st].v[jj]), acc[jj]);
#pragma unroll
          for (int f = 0; f < FE_PAD; ++f) sf[f] = fmaf(p, fb[st][f], sf[f]);
          fetch(j0 + st + PF, kb[st], vb[st], fb[st]);
        }
      }
    }

    asm volatile("" ::: "memory");
    const float inv = (end > beg) ? 1.0f / l : 0.f;
    float o[VEC];
    if constexpr (VEC % 2 == 0) {
      f32x2 o2[VEC / 2];
#pragma unroll
      for (int j = 0; j < VEC; j += 2) o2[j / 2] = f32x2{acc[j], acc[j + 1]};
#pragma unroll
      for (int f = 0; f < FE_PAD; ++f) {
        const f32x2 s2 = {sf[f], sf[f]};
#pragma unroll
        for (int j = 0; j < VEC; j += 2) o2[j / 2] = __builtin_elementwise_fma(s2, *reinterpret_cast<const f32x2*>(wl + f * VEC + j), o2[j / 2]);
      }
#pragma unroll
      for (int j = 0; j < VEC; j += 2) {
        o[j] = o2[j / 2][0] * inv;
        o[j + 1] = o2[j / 2][1] * inv;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = acc[j];
#pragma unroll
        for (int f = 0; f < FE_PAD; ++f) t = fmaf(sf[f], wl[f * VEC + j], t);
        o[j] = t * inv;
      }
    }
    if (addend != nullptr) {
      float ad[VEC];
      load_vec<T, VEC>(addend + (int64_t)d * ldadd + c0, ad);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += ad[j];
    }
    if constexpr (sizeof(T) * VEC == 16) {
      if (out_wt) {
        // write-through store (sc1): the output row is read by ANOTHER kernel, so keeping its line in this XCD's L2 only
        // evicts K|V rows that later destinations of the window still gather (ANEMOI_ATTN_OUT_WT, measured A/B in DESIGN.md)
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        Vec<T, VEC> tmp;
#pragma unroll
        for (int j = 0; j < VEC; ++j) tmp.v[j] = from_float<T>(o[j]);
        T* dstp = out + (int64_t)d * ldo + c0;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dstp), "v"(__builtin_bit_cast(u32x4, tmp)) : "memory");
      } else {
        store_vec<T, VEC>(out + (int64_t)d * ldo + c0, o);
      }
    } else {
      store_vec<T, VEC>(out + (int64_t)d * ldo + c0, o);
    }
    if (lse != nullptr && (lane % LPH) == 0) lse[(int64_t)d * H + lane / LPH] = (end > beg) ? m * scale + __logf(l) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------- generic path
// Any (H, C): one thread per (destination, head), serial over edges and channels.  Used for shapes the
// wave-per-row layout cannot express (D not a multiple of 64, non power-of-two lanes per head — e.g. the
// reference's own test shapes H in {2,6}, C in {4,6}).  Correctness path, not a performance path.
constexpr int kGenericMaxC = 256;

template <typename T>
__global__ void gt_attn_fwd_generic_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                           const T* __restrict__ v, int64_t ldv, const T* __restrict__ e, int64_t lde,
                                           const float* __restrict__ feat, int fe, int fe_pad,
                                           const float* __restrict__ w_packed,
                                           const int32_t* __restrict__ row, const int32_t* __restrict__ colptr,
                                           const T* __restrict__ addend, int64_t ldadd, T* __restrict__ out,
                                           int64_t ldo, float* __restrict__ lse, int n_dst, int H, int C, float scale,
                                           float drop_p, uint64_t drop_seed) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_dst * H) return;
  const int d = (int)(t / H), h = (int)(t % H);
  const int beg = colptr[d], end = colptr[d + 1];
  const float inv_keep = 1.0f / (1.0f - drop_p);
  float acc[kGenericMaxC];  // private (scratch-memory) fp32 accumulator
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  float m = -INFINITY, l = 0.f;
  const T* qp = q + (int64_t)d * ldq + h * C;
  for (int ei = beg; ei < end; ++ei) {
    const int s = row[ei];
    const T* kp = k + (int64_t)s * ldk + h * C;
    const T* vp = v + (int64_t)s * ldv + h * C;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
      float ee = 0.f;
      if (e != nullptr) ee = to_float(e[(int64_t)ei * lde + h * C + c]);
      if (feat != nullptr) {
        ee = 0.f;  // the bias rides on the constant-1 feature column fe
        for (int f = 0; f < fe_pad; ++f) ee = fmaf(feat[(int64_t)ei * fe_pad + f], w_packed[(int64_t)(h * C + c) * fe_pad + f], ee);
      }
      dot = fmaf(to_float(qp[c]) * scale, to_float(kp[c]) + ee, dot);
    }
    const float m_new = fmaxf(m, dot);
    const float corr = expf(m - m_new), p = expf(dot - m_new);
    l = l * corr + p;
    const float pv = drop_p > 0.f ? p * attn_dropout_scale(drop_seed, ei, h, drop_p, inv_keep) : p;
    for (int c = 0; c < C; ++c) {
      float ee = 0.f;
      if (e != nullptr) ee = to_float(e[(int64_t)ei * lde + h * C + c]);
      if (feat != nullptr) {
        ee = 0.f;  // the bias rides on the constant-1 feature column fe
        for (int f = 0; f < fe_pad; ++f) ee = fmaf(feat[(int64_t)ei * fe_pad + f], w_packed[(int64_t)(h * C + c) * fe_pad + f], ee);
      }
      acc[c] = acc[c] * corr + pv * (to_float(vp[c]) + ee);
    }
    m = m_new;
  }
  const float inv = (end > beg) ? 1.0f / l : 0.f;
  for (int c = 0; c < C; ++c) {
    float o = acc[c] * inv;
    if (addend != nullptr) o += to_float(addend[(int64_t)d * ldadd + h * C + c]);
    out[(int64_t)d * ldo + h * C + c] = from_float<T>(o);
  }
  if (lse != nullptr) lse[(int64_t)d * H + h] = (end > beg) ? m + logf(l) : 0.f;
}

// W' = [W_e | b_e | 0] as fp32 [D][fe_pad]
template <typename T>
__global__ void pack_edge_weights_kernel(const T* __restrict__ w, const T* __restrict__ b, float* __restrict__ out, int D,
                                         int fe, int fe_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D * fe_pad) return;
  const int c = i / fe_pad, f = i % fe_pad;
  out[i] = f < fe ? to_float(w[(int64_t)c * fe + f]) : ((f == fe && b != nullptr) ? to_float(b[c]) : 0.f);
}

template <typename T>
__global__ void pack_edge_features_kernel(const T* __restrict__ ea, int64_t ld, float* __restrict__ out, int M, int fe,
                                          int fe_pad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * fe_pad) return;
  const int m = (int)(i / fe_pad), f = (int)(i % fe_pad);
  out[i] = f < fe ? to_float(ea[(int64_t)m * ld + f]) : (f == fe ? 1.0f : 0.0f);
}

// ---------------------------------------------------------------------------------------------- dispatch
struct AttnArgs {
  const void *q, *k, *v, *e;
  int64_t ldq, ldk, ldv, lde;
  const float* feat;
  int fe, fe_pad;
  const float* w_packed;
  const int32_t *row, *colptr;
  const int32_t* order = nullptr;
  const void* addend;
  int64_t ldadd;
  void* out;
  int64_t ldo;
  float* lse;
  int n_dst, n_src, H, C;
  hipStream_t stream;
  float drop_p = 0.f;      // attention dropout (materialised-E op only): probability and the seed of the replayable mask
  uint64_t drop_seed = 0;
};

static bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

template <typename T, int VEC, int LPH>
static int launch_fast(const AttnArgs& a) {
  const float scale = 1.0f / sqrtf((float)a.C);
  const dim3 block(64 * kWavesPerBlock);
  if (a.feat == nullptr) {
    const dim3 grid((a.n_dst + kWavesPerBlock - 1) / kWavesPerBlock);
    if (a.e != nullptr)
      hipLaunchKernelGGL((gt_attn_fwd_kernel<T, VEC, LPH, true>), grid, block, 0, a.stream, (const T*)a.q, a.ldq,
                         (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, (const T*)a.e, a.lde, a.row, a.colptr,
                         (const T*)a.addend, a.ldadd, (T*)a.out, a.ldo, a.lse, a.n_dst, a.H, scale, a.drop_p, a.drop_seed);
    else
      hipLaunchKernelGGL((gt_attn_fwd_kernel<T, VEC, LPH, false>), grid, block, 0, a.stream, (const T*)a.q, a.ldq,
                         (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, (const T*)nullptr, (int64_t)0, a.row, a.colptr,
                         (const T*)a.addend, a.ldadd, (T*)a.out, a.ldo, a.lse, a.n_dst, a.H, scale, a.drop_p, a.drop_seed);
    return check_launch("gt_attn_fwd_kernel");
  }
  // Fused lin_edge: each wave walks `dst_per_wave` consecutive destinations so the W' staging is amortised,
  // while keeping >= ~8 blocks per CU worth of waves in flight.
  auto go = [&](auto fe_pad_c) {
    constexpr int FE_PAD = decltype(fe_pad_c)::value;
    using L = WLayout<VEC, FE_PAD>;
    if (L::kFloats * sizeof(float) > 64 * 1024) return 1;  // beyond the default dynamic-LDS limit: generic path
    // persistent grid: a few workgroups per CU (LDS 25 KiB each, 5-7 waves/SIMD by registers); every wave walks destinations
    // d, d + total_waves, ... so the W' staging is paid once per workgroup.  How many: with the locality-preserving work order
    // the number of destinations in flight per XCD is also the size of the window whose K|V rows should stay in the L2 -
    // measured on MI355X (profiles/r03_attention_blocks_per_cu.txt, three repetitions): 10 242 destinations 27.4 / 26.9 / 26.1 us
    // at 5 / 6 / 7 workgroups per CU, 40 962 destinations 80.6 / 89.6 / 84.2 us (5 workgroups = 640 waves per XCD = exactly 8
    // destinations per wave, and the smallest window); forwards: O96 2.991 / 2.991 / 2.973 ms, O96 -> res 6 8.69 / 8.81 / 8.76 ms,
    // N320 15.21 / 15.68 / 15.39 ms.  ANEMOI_ATTN_BLOCKS_PER_CU overrides the rule.
    static const int per_cu_env = [] { const char* e = getenv("ANEMOI_ATTN_BLOCKS_PER_CU"); return env_int(e, 0, 0, 32); }();
    const int per_cu = per_cu_env > 0 ? per_cu_env : (a.n_dst >= 20000 ? 5 : 7);
    static const int out_wt = [] { return env_int(getenv("ANEMOI_ATTN_OUT_WT"), 0, 0, 1); }();
    const int max_blocks = 256 * per_cu;
    int blocks = (a.n_dst + kWavesPerBlock - 1) / kWavesPerBlock;
    blocks = blocks < max_blocks ? blocks : max_blocks;
    blocks = (blocks + 7) & ~7;  // a whole number of workgroups per XCD
    const dim3 grid(blocks);
    // the fused [q|k|v|self] projection buffer: v sits right behind k in every row
    const bool kv_adjacent = (const T*)a.v == (const T*)a.k + 64 * VEC && a.ldv == a.ldk &&
                             (int64_t)a.n_src * a.ldk < (int64_t(1) << 32);
    auto kern = kv_adjacent ? gt_attn_fused_edge_fwd_kernel<T, VEC, LPH, FE_PAD, true> : gt_attn_fused_edge_fwd_kernel<T, VEC, LPH, FE_PAD, false>;
    hipLaunchKernelGGL(kern, grid, block, L::kFloats * sizeof(float),
                       a.stream, (const T*)a.q, a.ldq, (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, a.feat, a.w_packed,
                       a.row, a.colptr, a.order, (const T*)a.addend, a.ldadd, (T*)a.out, a.ldo, a.lse, a.n_dst, a.H, scale, out_wt);
    return check_launch("gt_attn_fused_edge_fwd_kernel");
  };
  switch (a.fe_pad) {
    case 4: return go(std::integral_constant<int, 4>{});
    case 8: return go(std::integral_constant<int, 8>{});
    case 12: return go(std::integral_constant<int, 12>{});
    case 16: return go(std::integral_constant<int, 16>{});
    default: return 1;  // not covered by the fast path
  }
}

template <typename T, int VEC>
static int launch_vec(const AttnArgs& a) {
  const int lph = a.C / VEC;
  switch (lph) {
    case 1: return launch_fast<T, VEC, 1>(a);
    case 2: return launch_fast<T, VEC, 2>(a);
    case 4: return launch_fast<T, VEC, 4>(a);
    case 8: return launch_fast<T, VEC, 8>(a);
    case 16: return launch_fast<T, VEC, 16>(a);
    default: return 1;
  }
}

template <typename T>
static int launch(const AttnArgs& a) {
  const int D = a.H * a.C;
  int rc = 1;
  // fast path: one wave covers the row exactly, rows 16-byte aligned for vector access
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool lds_ok = (a.ldq % 8 == 0) && (a.ldk % 8 == 0) && (a.ldv % 8 == 0) && (a.ldo % 8 == 0) &&
                      (a.e == nullptr || a.lde % 8 == 0) && (a.addend == nullptr || a.ldadd % 8 == 0) && al16(a.q) &&
                      al16(a.k) && al16(a.v) && al16(a.out) && al16(a.e) && al16(a.addend) && al16(a.feat) && al16(a.w_packed);
  if (D % 64 == 0 && lds_ok) {
    const int vec = D / 64;
    if (a.C % vec == 0 && pow2(a.C / vec) && a.C / vec <= 16) {
      if (vec == 1) rc = launch_vec<T, 1>(a);
      else if (vec == 2) rc = launch_vec<T, 2>(a);
      else if (vec == 4) rc = launch_vec<T, 4>(a);
      else if (vec == 8) rc = launch_vec<T, 8>(a);
      else if (vec == 16) rc = launch_vec<T, 16>(a);
    }
  }
  if (rc <= 0) return rc;
  // generic path
  if (a.C > kGenericMaxC) {
    set_error("gt_attention: channels per head C=%d not supported (fast path needs H*C %% 64 == 0 and a power-of-two "
              "C*64/(H*C) <= 16; generic path needs C <= %d)", a.C, kGenericMaxC);
    return ANEMOI_E_UNSUPPORTED;
  }
  const int64_t threads = (int64_t)a.n_dst * a.H;
  if (threads == 0) return ANEMOI_OK;
  const float scale = 1.0f / sqrtf((float)a.C);
  hipLaunchKernelGGL((gt_attn_fwd_generic_kernel<T>), dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, a.stream,
                     (const T*)a.q, a.ldq, (const T*)a.k, a.ldk, (const T*)a.v, a.ldv, (const T*)a.e, a.lde, a.feat, a.fe,
                     a.fe_pad, a.w_packed, a.row, a.colptr, (const T*)a.addend, a.ldadd,
                     (T*)a.out, a.ldo, a.lse, a.n_dst, a.H, a.C, scale, a.drop_p, a.drop_seed);
  return check_launch("gt_attn_fwd_generic_kernel");
}

static int dispatch(const AttnArgs& a, anemoi_dtype_t dtype) {
  switch (dtype) {
    case ANEMOI_F32: return launch<float>(a);
    case ANEMOI_BF16: return launch<bf16_t>(a);
    case ANEMOI_F16: return launch<f16_t>(a);
    default: set_error("unknown dtype %d", (int)dtype); return ANEMOI_E_INVALID;
  }
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_attention_dropout_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                       const void* e, int64_t lde, const int32_t* row, const int32_t* colptr,
                                       const void* addend, int64_t ldadd, void* out, int64_t ldo, float* lse,
                                       int32_t n_dst, int32_t n_src, int32_t H, int32_t C, float drop_p, uint64_t drop_seed,
                                       anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gt_attention_dropout_fwd: dropout probability %g outside [0, 1)", (double)drop_p);
  ANEMOI_REQUIRE(H < 65536, "gt_attention_dropout_fwd: the mask's counter holds 16 bits of head index, got H=%d", H);
  ANEMOI_REQUIRE(n_dst >= 0 && n_src >= 0 && H > 0 && C > 0, "gt_attention_dropout_fwd: bad sizes n_dst=%d n_src=%d H=%d C=%d", n_dst, n_src, H, C);
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(q && k && v && out && colptr, "gt_attention_dropout_fwd: null q/k/v/out/colptr");
  const int D = H * C;
  ANEMOI_REQUIRE(ldq >= D && ldk >= D && ldv >= D && ldo >= D && (!e || lde >= D) && (!addend || ldadd >= D),
                 "gt_attention_dropout_fwd: leading dimension smaller than H*C=%d", D);
  AttnArgs a{q, k, v, e, ldq, ldk, ldv, lde, nullptr, 0, 0, nullptr, row, colptr, nullptr, addend, ldadd, out, ldo, lse, n_dst, n_src, H, C, as_stream(stream)};
  a.drop_p = drop_p;
  a.drop_seed = drop_seed;
  return dispatch(a, dtype);
}

extern "C" int anemoi_gt_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                       const void* e, int64_t lde, const int32_t* row, const int32_t* colptr,
                                       const void* addend, int64_t ldadd, void* out, int64_t ldo, float* lse,
                                       int32_t n_dst, int32_t n_src, int32_t H, int32_t C, anemoi_dtype_t dtype,
                                       void* stream) {
  return anemoi_gt_attention_dropout_fwd(q, ldq, k, ldk, v, ldv, e, lde, row, colptr, addend, ldadd, out, ldo, lse, n_dst, n_src, H, C, 0.f, 0,
                                         dtype, stream);
}

// keep-scale of every (edge, head) as the kernels derive it: what a caller (or a test) needs to restate the op with an explicit mask
__global__ void attn_dropout_mask_kernel(float* __restrict__ out, int64_t n, int H, float p, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = attn_dropout_scale(seed, (int)(i / H), (int)(i % H), p, 1.0f / (1.0f - p));
}

extern "C" int anemoi_attention_dropout_mask(float* out, int32_t n_edges, int32_t H, float drop_p, uint64_t drop_seed, void* stream) {
  ANEMOI_REQUIRE(n_edges >= 0 && H > 0 && H < 65536 && drop_p >= 0.f && drop_p < 1.f, "attention_dropout_mask: bad arguments M=%d H=%d p=%g", n_edges, H, (double)drop_p);
  const int64_t n = (int64_t)n_edges * H;
  if (n == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(out, "attention_dropout_mask: null output");
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), out, n, H, drop_p, drop_seed);
  return check_launch("attn_dropout_mask_kernel");
}

extern "C" int anemoi_gt_attention_fused_edge_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                                  int64_t ldv, const float* edge_feat, int32_t fe_pad,
                                                  const float* w_packed, const int32_t* row, const int32_t* colptr,
                                                  const int32_t* dst_order, const void* addend, int64_t ldadd, void* out, int64_t ldo, float* lse,
                                                  int32_t n_dst, int32_t n_src, int32_t H, int32_t C,
                                                  anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(n_dst >= 0 && n_src >= 0 && H > 0 && C > 0, "gt_attention_fused_edge_fwd: bad sizes");
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(q && k && v && out && colptr && edge_feat && w_packed, "gt_attention_fused_edge_fwd: null pointer");
  ANEMOI_REQUIRE(fe_pad >= 4 && fe_pad % 4 == 0, "gt_attention_fused_edge_fwd: fe_pad must be a positive multiple of 4, got %d", fe_pad);
  const int D = H * C;
  ANEMOI_REQUIRE(ldq >= D && ldk >= D && ldv >= D && ldo >= D && (!addend || ldadd >= D), "gt_attention_fused_edge_fwd: leading dimension smaller than H*C=%d", D);
  AttnArgs a{q, k, v, nullptr, ldq, ldk, ldv, 0, edge_feat, fe_pad - 1, fe_pad, w_packed, row, colptr, dst_order, addend, ldadd, out, ldo, lse, n_dst, n_src, H, C, as_stream(stream)};
  return dispatch(a, dtype);
}

extern "C" int anemoi_pack_edge_weights(const void* w_edge, const void* b_edge, float* out, int32_t D, int32_t fe,
                                        int32_t fe_pad, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(D > 0 && fe > 0 && fe_pad == 4 * ((fe + 1 + 3) / 4), "pack_edge_weights: bad sizes D=%d fe=%d fe_pad=%d", D, fe, fe_pad);
  ANEMOI_REQUIRE(w_edge && out, "pack_edge_weights: null pointer");
  const int n = D * fe_pad;
  const dim3 grid((n + 255) / 256), block(256);
  switch (dtype) {
    case ANEMOI_F32: hipLaunchKernelGGL((pack_edge_weights_kernel<float>), grid, block, 0, as_stream(stream), (const float*)w_edge, (const float*)b_edge, out, D, fe, fe_pad); break;
    case ANEMOI_BF16: hipLaunchKernelGGL((pack_edge_weights_kernel<bf16_t>), grid, block, 0, as_stream(stream), (const bf16_t*)w_edge, (const bf16_t*)b_edge, out, D, fe, fe_pad); break;
    case ANEMOI_F16: hipLaunchKernelGGL((pack_edge_weights_kernel<f16_t>), grid, block, 0, as_stream(stream), (const f16_t*)w_edge, (const f16_t*)b_edge, out, D, fe, fe_pad); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
  return check_launch("pack_edge_weights_kernel");
}

extern "C" int anemoi_pack_edge_features(const void* edge_attr, int64_t ld, float* out, int32_t M, int32_t fe,
                                         int32_t fe_pad, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(M >= 0 && fe > 0 && fe_pad == 4 * ((fe + 1 + 3) / 4) && ld >= fe, "pack_edge_features: bad sizes M=%d fe=%d fe_pad=%d", M, fe, fe_pad);
  if (M == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(edge_attr && out, "pack_edge_features: null pointer");
  const int64_t n = (int64_t)M * fe_pad;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (dtype) {
    case ANEMOI_F32: hipLaunchKernelGGL((pack_edge_features_kernel<float>), grid, block, 0, as_stream(stream), (const float*)edge_attr, ld, out, M, fe, fe_pad); break;
    case ANEMOI_BF16: hipLaunchKernelGGL((pack_edge_features_kernel<bf16_t>), grid, block, 0, as_stream(stream), (const bf16_t*)edge_attr, ld, out, M, fe, fe_pad); break;
    case ANEMOI_F16: hipLaunchKernelGGL((pack_edge_features_kernel<f16_t>), grid, block, 0, as_stream(stream), (const f16_t*)edge_attr, ld, out, M, fe, fe_pad); break;
    default: set_error("unknown dtype"); return ANEMOI_E_INVALID;
  }
  return check_launch("pack_edge_features_kernel");
}
