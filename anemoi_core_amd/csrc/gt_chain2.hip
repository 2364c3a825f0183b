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
// (Round 6 built the projection as a step of ALL EIGHT waves - 48 x 64 tiles, gemm64 of chain2_core.h, x1 and the LayerNorm an eighth of the
// panel per wave; twice: skip rows through group B before the first barrier, and all rows six per wave with the skip rows parked in registers
// under the GEMM.  The instrumented launch got faster (91.9 against 95.3 us: projection 3.9-4.8 us, x1 2.0, LayerNorm 1.5-2.1) and the forwards
// SLOWER on the same box: O96 +1.1 % / +1.6 %, res 6 +-0 / +1.4 % - profiles/r06_chain2_shared_projection_ab.txt.  This is the round-5 schedule.)
//
// LayerNorm.  The statistics are the plain fp32 LayerNorm statistics of the ROUNDED 16-bit rows (per-wave (mean, M2) over 128 columns,
// merged with Chan's formula: no E[x^2] - mean^2 cancellation); the normalised row (x - mean) * rstd is rounded to the model dtype
// and the AFFINE part is folded into the Linear that follows, on the host and once per parameter version (W diag(gamma) rounded to
// the model dtype, d = W beta + b): the GEMM then needs no epilogue arithmetic at all - its accumulators START at d.  Likewise the
// projection's start at b_p + x and the second Linear's at b_2 + x1, so x1 never goes to global memory.  The per-column vectors
// (b_p | d1 | b_2 | dq, model dtype) sit in LDS for the whole launch.
#include "chain2_core.h"

namespace anemoi {

struct Chain2Args {
  const void* attn;  int64_t ld_attn;       // [n_rows, 512]  attention output + self term
  const void* xres;  int64_t ld_x;          // [n_rows, 512]  the block's input (skip)
  const char* wp;                           // projection, fragment-major
  const char* w1;    int hc;                // MLP-1 with LN_mlp's gamma folded in, [hidden, 512] fragment-major, hidden = 512 hc
  const char* w2;                           // MLP-2 [512, hidden] fragment-major
  const char* wq;    int qc;                // trailing projection with LN_attn' gamma folded in, [512 qc, 512] fragment-major; qc = 0: none
  int qn;                                   // NARROW trailing projection (qc == 1): only its first 128 qn columns exist (waves 0 .. qn-1 of group A); 0: all 512
  int q_cols;                               // columns of the trailing projection = entries of dq in vec
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
  int b_delay;                              // group B's head start handicap in the dual MLP steps, units of ~0.5 us (ANEMOI_CHAIN2_B_DELAY)
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
// the per-column vectors -> LDS, once per launch, by all 512 threads (requested behind the first panel's rows; visible behind the first barrier)
struct VecCopy {
  u32x4 v[2];
  __device__ __forceinline__ void request(const Chain2Args& a, int tid) {
    const int n16 = (1024 + 512 * a.hc + a.q_cols) / 8;  // <= 768
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] = reinterpret_cast<const u32x4*>(a.vec)[min(tid + 512 * k, n16 - 1)];
  }
  __device__ __forceinline__ void store(const Chain2Args& a, int tid, unsigned char* smem) {
    const int n16 = (1024 + 512 * a.hc + a.q_cols) / 8;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 512 * k < n16) reinterpret_cast<u32x4*>(smem + kVecOff)[tid + 512 * k] = v[k];
  }
};
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
template <typename T, bool TL, bool PART>
__device__ __forceinline__ void role_a(const Chain2Args& a, Ctx2& c, unsigned char* smem) {
  const int hc = a.hc, qc = a.qc, lane = c.lane, wq = __builtin_amdgcn_readfirstlane(c.wq);
  const int dbg = kExperiments ? a.dbg : 0;  // (timing experiments: compiled out of the product library)
  // bufA: the even hidden chunks (x2 / group A's staged projection outputs when hc is even); bufB: the A operand of P, M1 and the
  // trailing projection (attention rows, LN(x1), LN'(x2)); bufC: skip rows, then x1, then the odd hidden chunks
  unsigned char* const bufB = smem + kBufBytes;
  unsigned char* const bufC = smem + 2 * kBufBytes;
  float* const red = reinterpret_cast<float*>(smem + kRed2Off);
  const unsigned char* const vec = smem + kVecOff;
  auto hbuf = [&](int t) { return smem + (t & 1) * (2 * kBufBytes); };
  const int64_t s1 = kSlab;
  const char* const wpw = a.wp + (int64_t)(2 * wq) * kSlab;
  const int one = (dbg & 8) ? 0 : 1;  // (dbg & 8: every chunk reads chunk 0's weights - a 2-MB weight set that stays in the L2, timing experiments only)
  auto w1c = [&](int k) { return a.w1 + (int64_t)(8 * k * one + 2 * wq) * kSlab; };
  auto wqc = [&](int k) { return a.wq + (int64_t)(8 * k * one + 2 * wq) * kSlab; };
  frag8 ring[2][8];
  f32x4 acc[3][8];
  // the L2 warm-up: waves 0, 1 touch the CU's share of group A's next segment, waves 2, 3 of group B's (64 lines each)
  const int64_t sw2 = (int64_t)hc * kSlab;
  const bool narrow_idle = a.qn > 0 && wq >= a.qn;  // a narrow trailing projection leaves this wave without a chunk
  auto seg_a = [&](int t) { return t < hc ? a.w1 + (int64_t)(8 * t) * kSlab : ((qc > 0 && a.qn == 0) ? a.wq : a.wp); };  // M1(t); behind the last chunk: Q0 / the next panel's P (a narrow Q is not a whole segment: not touched)
  Warm warm;
  warm_init<PART>(warm, wq & 1, lane);
  auto touch = [&](const char* sa, const char* sb, int64_t pb) {
    touch_done<PART>(warm);
    if (!a.warm) return;
    if (wq < 2) { if (sa != nullptr) touch_share<PART>(warm, sa, kSlab, wq & 1, lane); }
    else if (sb != nullptr) touch_share<PART>(warm, sb, pb, wq & 1, lane);
  };
  // S0 of the first panel (peeled: straight-line code, so that the stores of the rows wait for the rows only): the attention rows ->
  // bufB, requested AHEAD of the vectors, the warm-up and the weight ring's first fragments - loads return in order
  {
    const int r0 = (int)blockIdx.x * a.rows_per_tile;
    load_rows12<T>((const T*)a.attn, a.ld_attn, r0, min(a.rows_per_tile, a.n_rows - r0), bufB, lane, wq, [&] {
      stamp2<TL>(c, smem);  // rows requested
      ring_prologue(ring, wpw, s1, c.loff);
      // this CU's share of the projection's weights (waves 0, 1) and of M1(0)'s (waves 2, 3)
      if (a.warm) touch_share<PART>(warm, wq < 2 ? a.wp : a.w1, kSlab, wq & 1, lane);
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
    if (dbg & 2) round_rows<T, false>(acc, bufB, nullptr, lane, wq);
    else normalise_rows<T>(acc, red, a.eps1, bufB, lane, wq);
    stamp2<TL>(c, smem);
    lds_barrier();
    // the MLP's first Linear, chunk by chunk (+ GELU); step hc is group B's alone
    for (int t = 0; t < hc; ++t) {
      stamp2<TL>(c, smem);
      init_acc<T, false>(acc, vec, 512 + 512 * t, nullptr, lane, wq);
      const char* nxt = t + 1 < hc ? w1c(t + 1) : ((qc > 0 && !narrow_idle) ? wqc(0) : wpw);
      if (kExperiments && a.prio_a == 1) __builtin_amdgcn_s_setprio(2);
      gemm128<T>(bufB, lane, ring, w1c(t), s1, nxt, s1, c.loff, acc);
      if (kExperiments && a.prio_a == 1) __builtin_amdgcn_s_setprio(0);
      touch(t == 0 ? nullptr : seg_a(t + 1), a.w2 + (int64_t)t * kSlab, sw2);  // the next step's M1(t + 1) (t = 0: touched behind P) and M2(t)
      stamp2<TL>(c, smem);
      if (dbg & 1) round_rows<T, false>(acc, hbuf(t), nullptr, lane, wq);
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
        if (row < nr && a.xout != nullptr) {
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
            // (a trailing projection behind the latent skip - the decoder's k | v - reads LayerNorm(x2 + skip): the sum goes back into the panel)
            if (qc > 0) *reinterpret_cast<u32x4*>(hbuf(hc) + row * kRowBytes + ((l0 ^ (row & 15)) << 4)) = v;
          }
          stream_store(v, reinterpret_cast<u32x4*>((T*)a.xout + (int64_t)(r0 + row) * a.ld_out + l0 * 8));
        }
      }
    }
    stamp2<TL>(c, smem);
    if (qc > 0 && a.extra != nullptr) {  // group B re-reads x2 + skip and takes its statistics: two barriers more, both roles
      lds_barrier();
      lds_barrier();
    }
    lds_barrier();
    // the even chunks of the trailing projection, staged (rounded) in x2's buffer
    for (int k = 0; k < qc; k += 2) {
      if (narrow_idle) break;  // (a narrow projection: this wave's columns do not exist)
      stamp2<TL>(c, smem);
      init_acc<T, false>(acc, vec, 1024 + 512 * hc + 512 * k, nullptr, lane, wq);
      set_chunk_prio(kExperiments && a.prio_q, k);
      gemm128<T>(bufB, lane, ring, wqc(k), s1, k + 2 < qc ? wqc(k + 2) : wpw, s1, c.loff, acc);
      if (kExperiments && a.prio_q) __builtin_amdgcn_s_setprio(0);
      touch(k + 2 < qc ? a.wq + (int64_t)(8 * (k + 2)) * kSlab : nullptr, k + 3 < qc ? a.wq + (int64_t)(8 * (k + 3)) * kSlab : nullptr, kSlab);
      stamp2<TL>(c, smem);
      round_rows<T, false>(acc, hbuf(hc), nullptr, lane, wq);
      if (!(dbg & 4)) store_staged<T>(hbuf(hc), (T*)a.qout + (int64_t)r0 * a.ld_q + k * kCh, a.ld_q, nr, lane, wq);
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
  touch_done<PART>(warm);
}

template <typename T, bool TL>
__device__ __forceinline__ void role_b(const Chain2Args& a, Ctx2& c, unsigned char* smem) {
  const int hc = a.hc, qc = a.qc, lane = c.lane, wq = __builtin_amdgcn_readfirstlane(c.wq);
  const int dbg = kExperiments ? a.dbg : 0;  // (timing experiments: compiled out of the product library)
  // bufA: the even hidden chunks (x2 / group A's staged projection outputs when hc is even); bufB: the A operand of P, M1 and the
  // trailing projection (attention rows, LN(x1), LN'(x2)); bufC: skip rows, then x1, then the odd hidden chunks
  unsigned char* const bufB = smem + kBufBytes;
  unsigned char* const bufC = smem + 2 * kBufBytes;
  float* const red = reinterpret_cast<float*>(smem + kRed2Off);
  const unsigned char* const vec = smem + kVecOff;
  auto hbuf = [&](int t) { return smem + (t & 1) * (2 * kBufBytes); };
  const int64_t s1 = kSlab, s2 = (int64_t)hc * kSlab;
  const int one = (dbg & 8) ? 0 : 1;
  auto w2c = [&](int k) { return a.w2 + (int64_t)(2 * wq) * s2 + (int64_t)(k * one) * kSlab; };
  auto wqc = [&](int k) { return a.wq + (int64_t)(8 * k * one + 2 * wq) * kSlab; };
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
      // In the dual steps group A is the critical path (its GEMM, then its GELU) while both groups pull weights through the same memory pipe:
      // this group starts its stream a little later, so that A's GEMM gets the pipe first and A's GELU runs beside the bulk of THIS GEMM
      if (kExperiments && t < hc)
        for (int i = 0; i < a.b_delay; ++i) __builtin_amdgcn_s_sleep(16);  // ~0.5 us each
      gemm128<T>(hbuf(t - 1), lane, ring, w2c(t - 1), s2, nxt, ns, c.loff, acc);
      stamp2<TL>(c, smem);
      if (t == hc) {  // x2 (rounded) -> the h buffer nobody reads any more, for group A to store
        if (qc > 0 && a.extra == nullptr) round_rows<T, true>(acc, hbuf(hc), red, lane, wq);
        else round_rows<T, false>(acc, hbuf(hc), nullptr, lane, wq);
        stamp2<TL>(c, smem);
      }
      lds_barrier();
    }
    if (qc > 0 && a.extra != nullptr) {
      // the latent skip AND a trailing projection: group A adds the skip to the rounded rows (whole rows, as it stores them) and writes the sums back;
      // this group reads them in its accumulator layout and takes the statistics of x2 + skip
      lds_barrier();
      {
        const Lane2 lc = lane2(lane, wq);
#pragma unroll
        for (int mi = 0; mi < 3; ++mi)
#pragma unroll
          for (int ni = 0; ni < 8; ++ni) {
            float o[4];
            unpack4<T>(*reinterpret_cast<const u32x2*>(hbuf(hc) + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]), o);
            acc[mi][ni] = f32x4{o[0], o[1], o[2], o[3]};
          }
      }
      round_rows<T, true>(acc, hbuf(hc), red, lane, wq);  // (the values are rounded already: this rewrites them unchanged and leaves the per-wave statistics)
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
      set_chunk_prio(kExperiments && a.prio_q, k);
      gemm128<T>(bufB, lane, ring, wqc(k), s1, last ? w2c(0) : wqc(k + 2), last ? s2 : s1, c.loff, acc);
      if (kExperiments && a.prio_q) __builtin_amdgcn_s_setprio(0);
      stamp2<TL>(c, smem);
      round_rows<T, false>(acc, hbuf(hc + 1), nullptr, lane, wq);
      if (!(dbg & 4)) store_staged<T>(hbuf(hc + 1), (T*)a.qout + (int64_t)r0 * a.ld_q + k * kCh, a.ld_q, nr, lane, wq);
      stamp2<TL>(c, smem);
    }
    if (qc > 0) lds_barrier();
    tile += (int)gridDim.x;
    if (tile >= a.n_tiles) break;
    if (qc > 0) lds_barrier();  // S0 of the next panel is group A's
  }
}

// PART: the launch has fewer than 256 workgroups (one round, not every CU busy): the L2 warm-up's shares follow the workgroup count.  Its own
// instantiation, so that full-grid (multi-round) launches run the code they were tuned with (the general form costs them 1.5 %, same box).
template <typename T, bool TL = false, bool PART = false>
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
    if (kExperiments && a.prio_a == 2) __builtin_amdgcn_s_setprio(2);
    role_a<T, TL, PART>(a, c, smem);
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_chain2_kernel<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kChain2Smem);
  });
  // Several rounds: as many workgroups as make the rounds even (854 panels: 4 rounds of 214 instead of 3 of 256 + 86) - the CUs of an XCD share
  // that L2's bandwidth, so a round of 27 CUs per XCD is faster than one of 32 (ANEMOI_CHAIN2_EVEN_GRID=0: always 256)
  static const int even_grid = ANEMOI_EXPERIMENT_ENV("ANEMOI_CHAIN2_EVEN_GRID", 1, 0, 1);
  int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  static const int max_grid = ANEMOI_EXPERIMENT_ENV("ANEMOI_CHAIN2_MAX_GRID", 256, 8, 256);  // (experiments: fewer CUs per round)
  if (even_grid && a.n_tiles > 256) {
    const int rounds = (a.n_tiles + max_grid - 1) / max_grid;
    grid = (a.n_tiles + rounds - 1) / rounds;
  }
#ifdef ANEMOI_EXPERIMENTS
  if (a.timeline != nullptr) {
    static PerDeviceOnce tl_once;
    tl_once.run([&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_chain2_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kTl2Off + 8 * kTl2Slots * 8);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_chain2_kernel<T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kTl2Off + 8 * kTl2Slots * 8);
    });
    if (grid < 256) hipLaunchKernelGGL((gt_chain2_kernel<T, true, true>), dim3(grid), dim3(512), kTl2Off + 8 * kTl2Slots * 8, st, a);
    else hipLaunchKernelGGL((gt_chain2_kernel<T, true>), dim3(grid), dim3(512), kTl2Off + 8 * kTl2Slots * 8, st, a);
    return check_launch("gt_chain2_kernel<timeline>");
  }
#endif
  if (grid < 256) hipLaunchKernelGGL((gt_chain2_kernel<T, false, true>), dim3(grid), dim3(512), kChain2Smem, st, a);
  else hipLaunchKernelGGL((gt_chain2_kernel<T>), dim3(grid), dim3(512), kChain2Smem, st, a);
  return check_launch("gt_chain2_kernel");
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_chain_rows_per_tile(int32_t n_rows) {
  // Rows per LDS panel of the chain launch: always the TALLEST panels (48 rows), i.e. the fewest CUs - every busy CU streams the whole
  // layer's weights whatever its panel's height, and the CUs of an XCD share that L2's bandwidth: 10 242 rows as 214 panels of 48 instead of
  // 250 of 41 is 4 % of the O96 forward; multi-round launches even out the ROUNDS instead (launch_chain2), not the panel heights
  // (profiles/r05_chain2_touch_coverage.txt).  ANEMOI_CHAIN_ROWS=n forces n rows (<= 48).
  static const int forced = env_int(getenv("ANEMOI_CHAIN_ROWS"), 0, 0, kPanel);
  (void)n_rows;
  return forced > 0 ? forced : kPanel;
}

extern "C" int anemoi_gt_chain2_fwd(const anemoi_gt_chain2_args_t* p, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(p != nullptr, "gt_chain2_fwd: null argument block");
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "gt_chain2_fwd: 16-bit model dtypes only");
  ANEMOI_REQUIRE(p->n_rows >= 0 && p->channels == kCh, "gt_chain2_fwd: channels=%d (this kernel is built for %d)", p->channels, kCh);
  if (p->n_rows == 0) return ANEMOI_OK;
  const bool narrow = p->q_out_features > 0 && p->q_out_features < kCh;  // a narrow trailing projection: 128, 256 or 384 columns
  ANEMOI_REQUIRE(p->hidden > 0 && p->hidden % kCh == 0 && p->q_out_features >= 0 && (narrow ? p->q_out_features % 128 == 0 : p->q_out_features % kCh == 0),
                 "gt_chain2_fwd: hidden=%d must be a multiple of %d, q_out_features=%d a multiple of %d (or 128, 256, 384)", p->hidden, kCh, p->q_out_features, kCh);
  const int n_vec = 2 * kCh + p->hidden + p->q_out_features;
  if (n_vec > (p->timeline != nullptr ? kVecMaxElemsTl : kVecMaxElems)) return ANEMOI_E_UNSUPPORTED;  // the per-column vectors must fit their LDS region
  ANEMOI_REQUIRE(p->attn && p->x_res && p->wp && p->w1 && p->w2 && p->vec && (p->x_out || p->q_out_features > 0), "gt_chain2_fwd: null operand");
  ANEMOI_REQUIRE(p->q_out_features == 0 || (p->wq && p->q_out), "gt_chain2_fwd: the trailing projection needs wq and q_out");
  ANEMOI_REQUIRE(p->extra == nullptr || p->x_out != nullptr, "gt_chain2_fwd: a second residual needs x_out");
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  ANEMOI_REQUIRE(al16(p->attn) && al16(p->x_res) && al16(p->wp) && al16(p->w1) && al16(p->w2) && al16(p->x_out) && al16(p->wq) && al16(p->q_out) &&
                     al16(p->extra) && al16(p->vec),
                 "gt_chain2_fwd: operands must be 16-byte aligned");
  ANEMOI_REQUIRE(p->ld_attn >= kCh && p->ld_x >= kCh && (p->x_out == nullptr || (p->ld_out >= kCh && p->ld_out % 8 == 0)) && p->ld_attn % 8 == 0 && p->ld_x % 8 == 0 &&
                     (p->extra == nullptr || (p->ld_extra >= kCh && p->ld_extra % 8 == 0)) &&
                     (p->q_out_features == 0 || (p->ld_q >= p->q_out_features && p->ld_q % 8 == 0)),
                 "gt_chain2_fwd: leading dimensions too small or not multiples of 8 elements (rows move as 16-byte pieces)");
  Chain2Args a{};
  a.attn = p->attn; a.ld_attn = p->ld_attn;
  a.xres = p->x_res; a.ld_x = p->ld_x;
  a.wp = (const char*)p->wp;
  a.w1 = (const char*)p->w1; a.hc = p->hidden / kCh;
  a.w2 = (const char*)p->w2;
  a.wq = (const char*)p->wq; a.qc = narrow ? 1 : p->q_out_features / kCh;
  a.qn = narrow ? p->q_out_features / 128 : 0; a.q_cols = p->q_out_features;
  a.vec = p->vec;
  a.eps1 = p->ln1_eps; a.epsq = p->lnq_eps;
  a.extra = p->extra; a.ld_extra = p->ld_extra;
  a.xout = p->x_out; a.ld_out = p->ld_out;
  a.qout = p->q_out; a.ld_q = p->ld_q;
  if (!kExperiments && p->timeline != nullptr) {
    set_error("gt_chain2_fwd: the in-kernel timeline is part of the experiments build only (python -m anemoi_core_amd.build --experiments)");
    return ANEMOI_E_UNSUPPORTED;
  }
  a.timeline = reinterpret_cast<unsigned long long*>(p->timeline);
  static const int prio_a = ANEMOI_EXPERIMENT_ENV("ANEMOI_CHAIN2_PRIO_A", 0, 0, 2);
  a.prio_a = prio_a;
  static const int dbg = ANEMOI_EXPERIMENT_ENV("ANEMOI_CHAIN2_DBG", 0, 0, 15);
  a.dbg = dbg;
  static const int prio_q = ANEMOI_EXPERIMENT_ENV("ANEMOI_CHAIN2_PRIO_Q", 0, 0, 1);
  a.prio_q = prio_q;
  static const int warm = ANEMOI_EXPERIMENT_ENV("ANEMOI_CHAIN2_WARM", 1, 0, 1);
  a.warm = warm;
  static const int b_delay = ANEMOI_EXPERIMENT_ENV("ANEMOI_CHAIN2_B_DELAY", 0, 0, 40);
  a.b_delay = b_delay;
  a.n_rows = p->n_rows;
  a.rows_per_tile = p->rows_per_tile > 0 ? p->rows_per_tile : anemoi_gt_chain_rows_per_tile(p->n_rows);
  ANEMOI_REQUIRE(a.rows_per_tile <= kPanel, "gt_chain2_fwd: rows_per_tile=%d exceeds the %d-row panel", a.rows_per_tile, kPanel);
  a.n_tiles = (a.n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  hipStream_t st = as_stream(stream);
  return dtype == ANEMOI_BF16 ? launch_chain2<bf16_t>(a, st) : launch_chain2<f16_t>(a, st);
}
