// Shared device code of the row-resident chain kernels (gt_chain.hip: GraphTransformer block; gnn_chain.hip: GraphConv edge / node
// MLPs).  A workgroup of 8 waves owns a panel of <= 48 rows x 512 channels in LDS (16-byte slots XOR-swizzled by the row); every wave
// owns a 64-column slab of each GEMM's output and streams its B fragments from a fragment-major weight image straight into a
// register ring (see gt_chain.hip for the measurements behind each choice).
#pragma once
#include "common.h"

namespace anemoi {

using frag8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T>
__device__ __forceinline__ f32x4 cmfma(frag8 a, frag8 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 cmfma<bf16_t>(frag8 a, frag8 b, f32x4 c) {
  using bf8 = __attribute__((ext_vector_type(8))) __bf16;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 cmfma<f16_t>(frag8 a, frag8 b, f32x4 c) {
  using h8 = __attribute__((ext_vector_type(8))) _Float16;
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// four 16-bit values travel as two dwords (plain registers: a struct of four halves tempts the compiler into scratch)
template <typename T>
__device__ __forceinline__ void unpack4(u32x2 p, float (&o)[4]);
template <>
__device__ __forceinline__ void unpack4<bf16_t>(u32x2 p, float (&o)[4]) {
  o[0] = __uint_as_float(p[0] << 16);
  o[1] = __uint_as_float(p[0] & 0xffff0000u);
  o[2] = __uint_as_float(p[1] << 16);
  o[3] = __uint_as_float(p[1] & 0xffff0000u);
}
template <>
__device__ __forceinline__ void unpack4<f16_t>(u32x2 p, float (&o)[4]) {
  // (element-wise through 16-bit integers: with a bit_cast of each dword to a 2-vector of halves hipcc 7.2 dropped the second
  // dword and converted the first one twice - found by the f16 parity tests)
  const unsigned lo = p[0], hi = p[1];
  o[0] = (float)__builtin_bit_cast(_Float16, (unsigned short)(lo & 0xffffu));
  o[1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(lo >> 16));
  o[2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi & 0xffffu));
  o[3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi >> 16));
}
template <typename T>
__device__ __forceinline__ u32x2 pack4(const float (&v)[4]) {
  const T a = from_float<T>(v[0]), b = from_float<T>(v[1]), c = from_float<T>(v[2]), d = from_float<T>(v[3]);
  return u32x2{(unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16),
               (unsigned)__builtin_bit_cast(unsigned short, c) | ((unsigned)__builtin_bit_cast(unsigned short, d) << 16)};
}
// (halves: two conversions + one v_pack_b32_f16 per dword instead of a shift and an or - with 96 values per lane in an epilogue the
// longer form made the role-split chain's f16 instantiation spill)
template <>
__device__ __forceinline__ u32x2 pack4<f16_t>(const float (&v)[4]) {
  using h2 = __attribute__((ext_vector_type(2))) _Float16;
  const h2 lo = {(_Float16)v[0], (_Float16)v[1]}, hi = {(_Float16)v[2], (_Float16)v[3]};
  return u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}
// this lane's 4 values of each of the wave's 4 column blocks of a parameter vector (bias, gamma, beta), loaded EARLY - before
// the GEMM whose epilogue uses them: vmcnt retires in order, so a load issued behind the weight ring's prefetches could only be
// waited for together with them (a full L2 latency exposed at every epilogue)
template <typename T>
__device__ __forceinline__ void load_cols(const T* __restrict__ p, int wave, int g, u32x2 (&o)[4]) {
  asm volatile("" : "+v"(g));  // (else the per-lane address is hoisted to the kernel's entry and spilled around the GEMM segments)
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) o[ni] = *reinterpret_cast<const u32x2*>(p + wave * 64 + ni * 16 + g * 4);
}

constexpr int kCh = 512;                    // channels of the residual stream (8 waves x 64 columns)
constexpr int kPanel = 48;                  // panel rows (3 MFMA row bands)
constexpr int kRowBytes = kCh * 2;          // one 16-bit row in LDS
constexpr int kBufBytes = kPanel * kRowBytes;
constexpr int kSlab = 16 * 4096;            // one segment of a wave's weight stream: 16 K-steps x (4 fragments x 1 KiB)
constexpr int kRedOff = 3 * kBufBytes;      // [48][8][2] fp32 LayerNorm partials
constexpr int kChainSmem = kRedOff + kPanel * 8 * 2 * 4;
constexpr int kTlOff = kChainSmem;          // instrumented instantiation only: [8 waves][kTlSlots] stamps, copied out at the end

// a pointer the compiler must keep in scalar registers (it is wave-uniform by construction): the loads then take the
// "SGPR base + 32-bit VGPR offset + immediate" form instead of a 64-bit VGPR address per fragment group
typedef const __attribute__((address_space(1))) char* gptr_t;  // a GLOBAL pointer: an integer round trip must not degrade the loads to flat_load
__device__ __forceinline__ gptr_t uniform_ptr(const char* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<gptr_t>(((uint64_t)hi << 32) | lo);
}
typedef const __attribute__((address_space(1))) frag8* gfrag_t;

// 16 K-steps (K = 512) of this wave's 48 x 64 tile: A fragments from the swizzled LDS panel, B fragments from the register
// ring (filled 4 K-steps ago), the ring slot refilled right behind its MFMAs with the fragments of 4 K-steps ahead - of this
// segment or, in its last group, of the NEXT segment (`nxt`), so the stream never drains across the epilogues.
// sched_barrier pins the issue order: left alone the scheduler sinks all 16 loads to the end of the loop body and the
// waitcnt pass then drains the queue at the top (measured with tools/weight_stream_probe.hip).
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
template <typename T, int NB = 3, typename Hook = NoHook>
__device__ __forceinline__ void gemm_seg(const unsigned char* abuf, int lane, frag8 (&bq)[4][4], const char* cur, const char* nxt,
                                         uint32_t loff, f32x4 (&acc)[NB][4], Hook hook = Hook(), int nq = 4, int flip = -1) {  // cur / nxt: wave-uniform (SGPR) bases, loff = lane * 16
  // flip >= 0 (experiment): the wave raises its issue priority in every other group of 4 K-steps - flip = 0 / 1 for the two waves of a SIMD
  // nq: groups of 4 K-steps (K = 128 nq; 4 = the 512-wide segment every caller but the MLP chain's first GEMM uses)
  asm volatile("" : "+v"(lane));
  const int x = lane & 15, ks = lane >> 4;
  const unsigned char* arow = abuf + x * kRowBytes;
  // the A fragments of K-step st+1 are requested BEFORE the MFMAs of step st (12 more registers): with one set of fragment
  // registers every step exposed three LDS round trips in front of its MFMAs (a third of a segment's time, in-kernel timeline)
  frag8 fa[NB];
#pragma unroll
  for (int mi = 0; mi < NB; ++mi) fa[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16 * kRowBytes + ((ks ^ x) << 4));
  const int last = nq * 4 - 1;
#pragma unroll 1
  for (int q = 0; q < nq; ++q) {
    const char* pfg = q < nq - 1 ? cur + (q + 1) * 16384 : nxt;
    hook(q);  // side traffic of the caller, a quarter of it per group of 4 K-steps (the edge chain's next panel)
    if (flip >= 0) {
      if ((q + flip) & 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int st = q * 4 + j;
      const int sn = st < last ? st + 1 : last;  // (the last step re-reads its own fragments: no branch in the stream)
      frag8 fn[NB];
#pragma unroll
      for (int mi = 0; mi < NB; ++mi) fn[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16 * kRowBytes + (((sn * 4 + ks) ^ x) << 4));
      __builtin_amdgcn_sched_barrier(0);  // (else the scheduler sinks these reads behind the MFMAs, into the registers they free)
#pragma unroll
      for (int mi = 0; mi < NB; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = cmfma<T>(bq[j][ni], fa[mi], acc[mi][ni]);  // D^T: lane = row x, 4 consecutive columns
      {
        const gptr_t pj = uniform_ptr(pfg + j * 4096);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bq[j][ni] = *reinterpret_cast<gfrag_t>(pj + loff + ni * 1024);
      }
#pragma unroll
      for (int mi = 0; mi < NB; ++mi) fa[mi] = fn[mi];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (flip >= 0) __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ void lds_barrier() {  // LDS writes of all waves visible; global loads in flight (the weight ring) stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Per-lane coordinates, re-derived from an OPAQUE copy of the lane id at the start of every phase: the epilogues' address
// arithmetic (a few dozen registers of LDS / global offsets per phase) is invariant over the panel loop, LICM hoists all of it
// to the kernel's entry, and the allocator then spills it around the GEMM segments (scratch reloads retire in order behind the
// weight ring: every reload would drain it).  Behind the barrier the values are computed where they are used.
struct LaneCtx {
  int x, g;
  int coff[4];  // LDS byte offset (inside a panel row) of this lane's 4 columns of column block ni: slot = wave*8 + ni*2 + (g>>1), swizzled by the row
};
__device__ __forceinline__ LaneCtx lane_ctx(int lane, int wave) {
  asm volatile("" : "+v"(lane));
  LaneCtx c;
  c.x = lane & 15;
  c.g = lane >> 4;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) c.coff[ni] = (((wave * 8 + ni * 2 + (c.g >> 1)) ^ c.x) << 4) + (c.g & 1) * 8;
  return c;
}

template <typename T, int NB = 3>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NB][4]) {
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Row statistics of the panel rows held as v[mi][ni][r] (fp32 images of the ROUNDED row values; lane = row mi*16 + x, columns
// wave*64 + ni*16 + g*4 + r): per-wave (mean, M2) partials through LDS, merged in wave order with Chan's formula (no
// E[x^2] - mean^2 cancellation).  ONE barrier inside; `red` may be reused after the NEXT barrier of the caller.
template <typename T, int NB = 3>
__device__ __forceinline__ void panel_row_stats(const f32x4 (&v)[NB][4], float eps, float* red, int wave, int x, int g, float (&mean)[NB], float (&rstd)[NB]) {
#pragma unroll
  for (int mi = 0; mi < NB; ++mi) {
    float s = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) s += (v[mi][ni][0] + v[mi][ni][1]) + (v[mi][ni][2] + v[mi][ni][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mw = s * (1.0f / 64.0f);  // mean of this wave's 64 columns of the row
    float q = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = v[mi][ni][r] - mw;
        q = fmaf(d, d, q);
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (g == 0) *reinterpret_cast<float2*>(red + ((mi * 16 + x) * 8 + wave) * 2) = make_float2(mw, q);
  }
  lds_barrier();
#pragma unroll
  for (int mi = 0; mi < NB; ++mi) {
    // the 8 waves' (mean, M2) of the row, merged in wave order (Chan et al.): M2 = sum M2_w + 64 sum (mean_w - mean)^2
    const f32x4* pr = reinterpret_cast<const f32x4*>(red + (mi * 16 + x) * 16);
    const f32x4 p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
    const float mu = (((p0[0] + p0[2]) + (p1[0] + p1[2])) + ((p2[0] + p2[2]) + (p3[0] + p3[2]))) * 0.125f;
    float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + ((p2[1] + p2[3]) + (p3[1] + p3[3]));
    const float d0 = p0[0] - mu, d1 = p0[2] - mu, d2 = p1[0] - mu, d3 = p1[2] - mu;
    const float d4 = p2[0] - mu, d5 = p2[2] - mu, d6 = p3[0] - mu, d7 = p3[2] - mu;
    const float dm = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
    m2 = fmaf(64.0f, dm, m2);
    mean[mi] = mu;
    rstd[mi] = rsqrtf(m2 * (1.0f / (float)kCh) + eps);
  }
}

// LayerNorm of the panel rows held as v[mi][ni][r] and store of the normalised rows (model dtype) into the LDS panel `dst`.  gm / bt:
// this lane's gamma / beta (packed, loaded before the GEMM).  One barrier inside (the partials), one after (the panel is complete).
template <typename T>
__device__ __forceinline__ void panel_layernorm(f32x4 (&v)[3][4], const u32x2 (&gm)[4], const u32x2 (&bt)[4], float eps, unsigned char* dst,
                                                float* red, int wave, int lane) {
  const LaneCtx lc = lane_ctx(lane, wave);
  float mean[3], rstd[3];
  panel_row_stats<T>(v, eps, red, wave, lc.x, lc.g, mean, rstd);
#pragma unroll
  for (int mi = 0; mi < 3; ++mi) {
    unsigned char* drow = dst + (mi * 16 + lc.x) * kRowBytes;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float gv[4], bv[4], o[4];
      unpack4<T>(gm[ni], gv);
      unpack4<T>(bt[ni], bv);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaf((v[mi][ni][r] - mean[mi]) * rstd[mi], gv[r], bv[r]);
      *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pack4<T>(o);
    }
  }
  lds_barrier();
}

// The wave's 48 x 64 output block (packed rows pk[mi][ni]: lane = row mi*16 + x, 4 columns) to global memory as whole 128-byte
// lines: through a wave-private LDS strip (row m, 16-byte slot s at m*128 + ((s ^ ((m >> 1) & 7)) << 4): conflict-free for the
// 8-byte writes in the MFMA layout and for the 16-byte row-major read-back), 6 stores of 16 bytes per lane.  `out` = address of
// (panel row 0, the wave's first column); the wave's own LDS operations are ordered: no barrier.
template <typename T, int NB = 3>
__device__ __forceinline__ void store_block_via_strip(const u32x2 (&pk)[NB][4], unsigned char* strip, T* out, int64_t ld, int nr, int lane, int wave) {
  const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
  for (int mi = 0; mi < NB; ++mi) {
    const int m = mi * 16 + lc.x;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<u32x2*>(strip + m * 128 + (((ni * 2 + (lc.g >> 1)) ^ ((m >> 1) & 7)) << 4) + (lc.g & 1) * 8) = pk[mi][ni];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int rl = (lc.g << 1) | (lc.x >> 3), sl = lc.x & 7;  // lane >> 3, lane & 7
#pragma unroll
  for (int it = 0; it < 2 * NB; ++it) {
    const int m = it * 8 + rl;
    const u32x4 v = *reinterpret_cast<const u32x4*>(strip + m * 128 + ((sl ^ ((m >> 1) & 7)) << 4));
    if (m < nr) *reinterpret_cast<u32x4*>(out + (int64_t)m * ld + sl * 8) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip may be rewritten
}

}  // namespace anemoi
