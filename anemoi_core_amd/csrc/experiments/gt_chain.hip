// Row-resident layer chain for gfx950: everything of a GraphTransformer block that is LOCAL TO A NODE ROW, in ONE launch -
//
//     x1 = attn W_p^T + b_p + x                      (projection + skip,            layers/block.py:1263-1266 of the reference)
//     h  = GELU(LN_mlp(x1) W_1^T + b_1)               (node_dst_mlp, first Linear,   layers/block.py:1268-1271, layers/mlp.py:158-169)
//     x2 = h W_2^T + b_2 + x1 [+ latent skip]        (second Linear + skip;         encoder_processor_decoder.py:295-296 for the skip)
//     qkvs' = LN_attn'(x2) [W_q; W_k; W_v; W_s]'^T + b' (the NEXT block's fused projections, layers/block.py:1237-1245)
//
// instead of four GEMM launches (each with its launch, cold first K-tile, epilogue and write tail: half of their time at
// 10 242 rows) plus the LayerNorm statistics hand-off between them.
//
// Structure.  A workgroup (8 waves, one per CU) owns a panel of <= 48 node rows for the whole chain.  The panel's activations
// never leave the CU: three [48][512] 16-bit buffers in LDS (A operand of the running GEMM, the next A operand, the residual),
// 16-byte slots XOR-swizzled by the row so that both the 8-byte epilogue writes and the fragment ds_read_b128 are conflict-free.
// What streams is the WEIGHTS: every wave owns a 64-column slab of each GEMM's output and reads its B fragments straight from
// L2 into registers - the weights are pre-packed fragment-major on the host (ops.pack_weight_frag: one contiguous KiB per
// 16 columns x 32 k), so a fragment is ONE fully coalesced global_load_dwordx4 and never touches LDS.  A register ring of
// 4 K-steps x 4 fragments (16 loads = 16 KiB per wave, 128 KiB per CU in flight) covers the L2 latency; no barrier inside a
// 16-step segment, the two waves of a SIMD interleave MFMA and load issue by themselves.  tools/weight_stream_probe.hip
// measured this loop on MI355X: 6.5 MiB of weights per CU and layer at 97 GB/s per CU next to 1.2 PFLOP/s of MFMAs = 70 us
// per layer for all four GEMMs, against 112 us for the four launches.  The hidden activations [rows x 2048] never exist:
// MLP-1 is produced in 512-column chunks that are consumed at once as K-chunks of MLP-2 (two accumulator sets in registers).
// LayerNorm is the real thing (fp32 statistics of the rounded 16-bit rows, per-wave (mean, M2) partials merged with Chan's
// formula: no E[x^2] - mean^2 cancellation), its output rounded to the model dtype as the reference's autocast does.
#include "chain_core.h"
#include "anemoi_hip_experiments.h"

namespace anemoi {

struct ChainArgs {
  const void* attn;  int64_t ld_attn;       // [n_rows, 512]  attention output + self term
  const void* xres;  int64_t ld_x;          // [n_rows, 512]  the block's input (skip)
  const char* wp;    const void* bp;        // projection, fragment-major / [512]
  const void* ln1_g; const void* ln1_b; float ln1_eps;
  const char* w1;    const void* b1;  int hc;   // MLP-1 [hidden, 512] fragment-major, hidden = 512 hc
  const char* w2;    const void* b2;        // MLP-2 [512, hidden] fragment-major
  const void* extra; int64_t ld_extra;      // optional second residual of x2
  void* xout;        int64_t ld_out;        // [n_rows, 512]  x2
  const void* lnq_g; const void* lnq_b; float lnq_eps;
  const char* wq;    const void* bq;  int qc;   // trailing projection [512 qc, 512] fragment-major; qc = 0: none
  void* qout;        int64_t ld_q;          // [n_rows, 512 qc]
  int n_rows, rows_per_tile, n_tiles;
  int prio_young;                           // experiment: s_setprio of waves 4-7 (0: none)
  int dbg;                                  // experiment (timing only, results are garbage): bit 0 skip the MLP epilogues, 1 skip LayerNorms, 2 skip q stores, 3 skip x1/x2 epilogues
  unsigned long long* timeline;             // developer aid (TL instantiation only): [workgroups][8 waves][kTlSlots] s_memtime stamps
};
constexpr int kTlSlots = 48;

// TL: a second instantiation that stamps s_memtime at the phase boundaries of each workgroup's FIRST panel (every wave, lane 0)
// into a.timeline - tools/chain_timeline.py; the production instantiation carries none of it.
//
// Schedule of one panel (what the in-kernel timeline of the first version asked for: the two waves of a SIMD do not progress
// evenly - the older one takes a 16-step segment in ~5.7 us, the younger in 7.5-9 us - so every barrier right behind a GEMM
// segment exposes that skew, and an epilogue behind a barrier runs with the matrix cores idle):
//   panel loads are issued BEFORE the weight ring's (the ring's first use comes later);
//   P GEMM -> x1 epilogue (skip rows preloaded into registers; x1 parked in the OUTPUT rows in global memory, read back by the
//     same lane before the last MLP-2 segment) -> LayerNorm (2 barriers: the partials, the panel);
//   MLP software-pipelined over the hidden chunks, h double-buffered in LDS (bufA / bufC):
//       M1(0) w(h0) | M1(1) B w(h1) M2(0) | M1(2) B w(h2) M2(1) | M1(3) B w(h3) M2(2) | B M2(3)
//     ONE barrier per chunk, between a wave's M1(c) segment (+ its GELU, done in registers BEFORE the barrier) and its write of
//     h_c: behind it every wave has finished M2(c-2) (so h_c may overwrite h_(c-2)) and w(h_(c-1)) (so M2(c-1) may read it) -
//     and every wave arrives with a whole segment of slack;
//   x2 epilogue -> LayerNorm' (2 barriers) -> the trailing projection's chunks without barriers, outputs transposed through a
//     wave-private LDS strip into whole 128-byte lines (16-byte stores).
template <typename T, bool TL = false>
__global__ __launch_bounds__(512, 1) void gt_chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const bufA = smem;                  // A operand: attention rows, then the even hidden chunks
  unsigned char* const bufB = smem + kBufBytes;      // A operand: LayerNorm'd rows (MLP-1, trailing projection)
  unsigned char* const bufC = smem + 2 * kBufBytes;  // the odd hidden chunks; staging strips of the trailing projection's stores
  float* const red = reinterpret_cast<float*>(smem + kRedOff);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;

  // this wave's weight stream: wave-uniform bases (scalar registers) + the lane's 16 bytes of every fragment
  const uint32_t loff = lane * 16;
  const char* const wp0 = a.wp + (int64_t)wave * kSlab;
  const char* const w10 = a.w1 + (int64_t)wave * kSlab;                 // + c * 8 slabs
  const char* const w20 = a.w2 + (int64_t)wave * a.hc * kSlab;          // + c slabs
  const char* const wq0 = a.qc > 0 ? a.wq + (int64_t)wave * kSlab : wp0;  // + c * 8 slabs

  [[maybe_unused]] int tl_n = 0;
  // stamps go to LDS (a global store per stamp sits in the vmcnt queue of the weight ring and distorts what it measures)
  auto stamp = [&]() {
    if constexpr (TL) {
      if (tl_n < kTlSlots) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0) reinterpret_cast<unsigned long long*>(smem + kTlOff)[wave * kTlSlots + tl_n] = now;
      }
      ++tl_n;
    }
  };
  stamp();  // 0: kernel entry
  if (wave >= 4) {
    if (a.prio_young == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio_young == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio_young == 3) __builtin_amdgcn_s_setprio(3);
  }
  const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};

  int tile = blockIdx.x;
  if (tile >= a.n_tiles) return;
  // the first panel's attention rows (coalesced 16-byte loads), requested ahead of the weight ring
  u32x4 va[6];
  auto request_panel = [&](int t) {
    const int r0 = t * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + 512 * i;
      const int rr = min(idx >> 6, nr - 1), slot = idx & 63;
      va[i] = *reinterpret_cast<const u32x4*>((const T*)a.attn + (int64_t)(r0 + rr) * a.ld_attn + slot * 8);
    }
  };
  request_panel(tile);
  __builtin_amdgcn_sched_barrier(0);
  frag8 bq[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      bq[j][ni] = *reinterpret_cast<gfrag_t>(uniform_ptr(wp0 + j * 4096) + loff + ni * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }

  for (;;) {
    const int r0 = tile * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    // ---- the panel: attention rows -> bufA (rows beyond the panel are zero).  The skip rows and the epilogue's parameter columns are
    // requested BEHIND the panel's barrier: 46 loads per lane in front of it took 5-8 us just to issue (in-kernel timeline)
    stamp();  // loads issued
    if constexpr (TL) {
      asm volatile("" : "+v"(va[5]));
      __builtin_amdgcn_sched_barrier(0);
      stamp();  // panel rows arrived
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + 512 * i;
      const int row = idx >> 6, slot = idx & 63;
      *reinterpret_cast<u32x4*>(bufA + row * kRowBytes + ((slot ^ (row & 15)) << 4)) = row < nr ? va[i] : zero4;
    }
    lds_barrier();
    stamp();  // 1: panel in LDS
    u32x2 xr[3][4];  // this lane's skip values (P epilogue), later x1 (MLP-2 epilogue)
    u32x2 pb[4], pg[4], pt[4];  // packed parameter columns of the coming epilogue (see load_cols)
    load_cols<T>((const T*)a.bp, wave, g, pb);
    load_cols<T>((const T*)a.ln1_g, wave, g, pg);
    if (a.ln1_b != nullptr) {
      load_cols<T>((const T*)a.ln1_b, wave, g, pt);
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) pt[ni] = u32x2{0u, 0u};
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[3][4], acc2[3][4];
    // ---- projection + skip -> x1 (parked in the output rows), LayerNorm_mlp(x1) -> bufB
    zero_acc<T>(acc);
    // the skip rows are requested behind the GEMM's 12th K-step: loads retire in order, and in front of the GEMM these twelve 8-byte row
    // loads per lane (HBM latency) stood before every weight-ring wait from its 5th step on (found in the GraphConv edge chain's timeline;
    // the ring loads issued behind them here are the next segment's, first waited for behind the epilogue that consumes the rows anyway)
    gemm_seg<T>(bufA, lane, bq, wp0, wp0 + 3 * 16384, loff, acc, NoHook(), 3);
    {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const T* xrow = (const T*)a.xres + (int64_t)(r0 + min(mi * 16 + lc.x, nr - 1)) * a.ld_x + wave * 64 + lc.g * 4;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) xr[mi][ni] = *reinterpret_cast<const u32x2*>(xrow + ni * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    gemm_seg<T>(bufA + 48 * 16, lane, bq, wp0 + 3 * 16384, w10, loff, acc, NoHook(), 1);  // K = 384 .. 511 (slots 48 + s: the swizzle stays in the low 4 bits)
    stamp();  // 2: projection GEMM done
    if (!(a.dbg & 8)) {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const int m = mi * 16 + lc.x;
        T* orow = (T*)a.xout + (int64_t)(r0 + min(m, nr - 1)) * a.ld_out + wave * 64 + lc.g * 4;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float bias[4], res[4], o[4];
          unpack4<T>(pb[ni], bias);
          unpack4<T>(xr[mi][ni], res);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (acc[mi][ni][r] + bias[r]) + res[r];
          const u32x2 pk = pack4<T>(o);
          if (m < nr) *reinterpret_cast<u32x2*>(orow + ni * 16) = pk;  // x1, read back by this same lane as MLP-2's skip
          unpack4<T>(pk, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mi][ni][r] = o[r];
        }
      }
    }
    if (!(a.dbg & 2)) panel_layernorm<T>(acc, pg, pt, a.ln1_eps, bufB, red, wave, lane);
    else { lds_barrier(); lds_barrier(); }
    stamp();  // 3: x1 + LayerNorm done

    // ---- MLP, software-pipelined over the hidden chunks (see the schedule above)
    zero_acc<T>(acc2);
    for (int c = 0; c < a.hc; ++c) {
      load_cols<T>((const T*)a.b1 + c * kCh, wave, g, pb);
      __builtin_amdgcn_sched_barrier(0);
      zero_acc<T>(acc);
      // the segment behind M1(c) is M2(c-1) (c > 0), else M1(1) - or M2(0) when there is a single chunk
      const char* nx1 = c > 0 ? w20 + (int64_t)(c - 1) * kSlab : (a.hc > 1 ? w10 + (int64_t)8 * kSlab : w20);
      gemm_seg<T>(bufB, lane, bq, w10 + (int64_t)c * 8 * kSlab, nx1, loff, acc);
      stamp();  // MLP-1 chunk GEMM done
      u32x2 hp[3][4];
      if (a.dbg & 1) {
#pragma unroll
        for (int mi = 0; mi < 3; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) hp[mi][ni] = u32x2{__float_as_uint(acc[mi][ni][0]), __float_as_uint(acc[mi][ni][1])};
      } else
#pragma unroll
      for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float bias[4], t[4];
          unpack4<T>(pb[ni], bias);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = acc[mi][ni][r] + bias[r];
          gelu_fast2(t[0], t[1]);
          gelu_fast2(t[2], t[3]);
          hp[mi][ni] = pack4<T>(t);
        }
      stamp();  // GELU in registers
      if (c > 0) lds_barrier();  // all waves: M2(c-2) read and h_(c-1) written (c = 0: bufA was last read by the projection, two barriers ago)
      stamp();  // barrier passed
      {
        const LaneCtx lc = lane_ctx(lane, wave);
        unsigned char* const hb = (c & 1) ? bufC : bufA;
#pragma unroll
        for (int mi = 0; mi < 3; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<u32x2*>(hb + (mi * 16 + lc.x) * kRowBytes + lc.coff[ni]) = hp[mi][ni];
      }
      if (c > 0) {
        // M2(c-1); behind it M1(c+1), or the last MLP-2 segment
        const char* nx2 = c + 1 < a.hc ? w10 + (int64_t)(c + 1) * 8 * kSlab : w20 + (int64_t)c * kSlab;
        gemm_seg<T>(((c - 1) & 1) ? bufC : bufA, lane, bq, w20 + (int64_t)(c - 1) * kSlab, nx2, loff, acc2);
        stamp();  // MLP-2 K-chunk done
      }
    }
    // the epilogue behind the last MLP-2 segment: its parameters and this lane's x1 values, requested ahead of the segment
    load_cols<T>((const T*)a.b2, wave, g, pb);
    if (a.qc > 0) {
      load_cols<T>((const T*)a.lnq_g, wave, g, pg);
      if (a.lnq_b != nullptr) {
        load_cols<T>((const T*)a.lnq_b, wave, g, pt);
      } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) pt[ni] = u32x2{0u, 0u};
      }
    }
    {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const T* orow = (const T*)a.xout + (int64_t)(r0 + min(mi * 16 + lc.x, nr - 1)) * a.ld_out + wave * 64 + lc.g * 4;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) xr[mi][ni] = *reinterpret_cast<const u32x2*>(orow + ni * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();  // the last hidden chunk is complete
    stamp();
    gemm_seg<T>(((a.hc - 1) & 1) ? bufC : bufA, lane, bq, w20 + (int64_t)(a.hc - 1) * kSlab, a.qc > 0 ? wq0 : wp0, loff, acc2);
    stamp();  // last MLP-2 K-chunk done
    // ---- x2 = h W2^T + b2 + x1 [+ extra] -> global; LayerNorm_attn'(x2) -> bufB
    if (!(a.dbg & 8)) {
      const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const int m = mi * 16 + lc.x;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = wave * 64 + ni * 16 + lc.g * 4;
          float bias[4], res[4], t[4];
          unpack4<T>(pb[ni], bias);
          unpack4<T>(xr[mi][ni], res);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = (acc2[mi][ni][r] + bias[r]) + res[r];
          u32x2 pk = pack4<T>(t);
          if (a.extra != nullptr && m < nr) {
            // the latent skip rides on the last block's output, added to the block's ROUNDED output as `x + skip` does
            float e[4];
            unpack4<T>(pk, t);
            unpack4<T>(*reinterpret_cast<const u32x2*>((const T*)a.extra + (int64_t)(r0 + m) * a.ld_extra + n), e);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] += e[r];
            pk = pack4<T>(t);
          }
          if (m < nr) *reinterpret_cast<u32x2*>((T*)a.xout + (int64_t)(r0 + m) * a.ld_out + n) = pk;
          unpack4<T>(pk, t);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc2[mi][ni][r] = t[r];
        }
      }
    }
    stamp();  // x2 epilogue done
    const int next_tile = tile + (int)gridDim.x;
    const bool more = next_tile < a.n_tiles;
    if (a.qc > 0) {
      if (!(a.dbg & 2)) panel_layernorm<T>(acc2, pg, pt, a.lnq_eps, bufB, red, wave, lane);
      else { lds_barrier(); lds_barrier(); }
      stamp();  // LayerNorm' done
      // behind LayerNorm's barriers no wave reads bufA / bufC any more: they stage the stores (a 48 x 128-byte strip per wave)
      unsigned char* const strip = ((wave & 4) ? bufC : bufA) + (wave & 3) * (kPanel * 128);
      for (int c = 0; c < a.qc; ++c) {
        load_cols<T>((const T*)a.bq + c * kCh, wave, g, pb);
        __builtin_amdgcn_sched_barrier(0);
        zero_acc<T>(acc);
        const char* nxt = c + 1 < a.qc ? wq0 + (int64_t)(c + 1) * 8 * kSlab : wp0;  // (the next panel's projection; after the last panel: a harmless re-read)
        gemm_seg<T>(bufB, lane, bq, wq0 + (int64_t)c * 8 * kSlab, nxt, loff, acc);
        if (!(a.dbg & 4)) {
          // strip row m, 16-byte slot s (8 per row) at m*128 + ((s ^ ((m >> 1) & 7)) << 4): conflict-free for the 8-byte writes in
          // the MFMA layout and for the 16-byte row-major read-back (8 lanes per row, 8 rows per instruction)
          const LaneCtx lc = lane_ctx(lane, wave);
#pragma unroll
          for (int mi = 0; mi < 3; ++mi) {
            const int m = mi * 16 + lc.x;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              float bias[4], o[4];
              unpack4<T>(pb[ni], bias);
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = acc[mi][ni][r] + bias[r];
              *reinterpret_cast<u32x2*>(strip + m * 128 + (((ni * 2 + (lc.g >> 1)) ^ ((m >> 1) & 7)) << 4) + (lc.g & 1) * 8) = pack4<T>(o);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private strip: the wave's own LDS operations are ordered, no barrier
          const int rl = (lc.g << 1) | (lc.x >> 3);  // lane >> 3: row within a group of 8
          const int sl = lane & 7;
#pragma unroll
          for (int it = 0; it < 6; ++it) {
            const int m = it * 8 + rl;
            const u32x4 v = *reinterpret_cast<const u32x4*>(strip + m * 128 + ((sl ^ ((m >> 1) & 7)) << 4));
            if (m < nr) *reinterpret_cast<u32x4*>((T*)a.qout + (int64_t)(r0 + m) * a.ld_q + c * kCh + wave * 64 + sl * 8) = v;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip is rewritten by the next chunk
        }
        stamp();  // trailing projection chunk c done (GEMM + stores issued)
      }
    }
    if constexpr (TL) {
      if (tile == (int)blockIdx.x) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < kTlSlots && lane < tl_n)
          a.timeline[((size_t)blockIdx.x * 8 + wave) * kTlSlots + lane] = reinterpret_cast<const unsigned long long*>(smem + kTlOff)[wave * kTlSlots + lane];
      }
    }
    if (!more) break;
    tile = next_tile;
    request_panel(tile);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();  // every wave is done with this panel's LDS (bufA: staging strips / the last hidden chunk's buffer)
  }
}

template <typename T>
static int launch_chain(const ChainArgs& a, hipStream_t st) {
  static PerDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_chain_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainSmem);
  });
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  if (a.timeline != nullptr) {
    static PerDeviceOnce tl_once;
    tl_once.run([&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gt_chain_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainSmem + 8 * kTlSlots * 8);
    });
    hipLaunchKernelGGL((gt_chain_kernel<T, true>), dim3(grid), dim3(512), kChainSmem + 8 * kTlSlots * 8, st, a);
    return check_launch("gt_chain_kernel<timeline>");
  }
  hipLaunchKernelGGL((gt_chain_kernel<T>), dim3(grid), dim3(512), kChainSmem, st, a);
  return check_launch("gt_chain_kernel");
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_chain_fwd(const anemoi_gt_chain_args_t* p, anemoi_dtype_t dtype, void* stream) {
  ANEMOI_REQUIRE(p != nullptr, "gt_chain_fwd: null argument block");
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 || dtype == ANEMOI_F16, "gt_chain_fwd: 16-bit model dtypes only");
  ANEMOI_REQUIRE(p->n_rows >= 0 && p->channels == kCh, "gt_chain_fwd: channels=%d (this kernel is built for %d)", p->channels, kCh);
  if (p->n_rows == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(p->hidden > 0 && p->hidden % kCh == 0 && p->q_out_features >= 0 && p->q_out_features % kCh == 0,
                 "gt_chain_fwd: hidden=%d and q_out_features=%d must be multiples of %d", p->hidden, p->q_out_features, kCh);
  ANEMOI_REQUIRE(p->attn && p->x_res && p->wp && p->bp && p->ln1_w && p->w1 && p->b1 && p->w2 && p->b2 && p->x_out, "gt_chain_fwd: null operand");
  ANEMOI_REQUIRE(p->q_out_features == 0 || (p->lnq_w && p->wq && p->bq && p->q_out), "gt_chain_fwd: the trailing projection needs lnq_w, wq, bq, q_out");
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  ANEMOI_REQUIRE(al16(p->attn) && al16(p->x_res) && al16(p->wp) && al16(p->w1) && al16(p->w2) && al16(p->x_out) && al16(p->wq) && al16(p->q_out) &&
                     al16(p->extra) && (reinterpret_cast<uintptr_t>(p->bp) & 7) == 0 && (reinterpret_cast<uintptr_t>(p->b1) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(p->b2) & 7) == 0 && (reinterpret_cast<uintptr_t>(p->bq) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(p->ln1_w) & 7) == 0 && (reinterpret_cast<uintptr_t>(p->ln1_b) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(p->lnq_w) & 7) == 0 && (reinterpret_cast<uintptr_t>(p->lnq_b) & 7) == 0,
                 "gt_chain_fwd: operands must be 16-byte aligned (vectors: 8-byte)");
  ANEMOI_REQUIRE(p->ld_attn >= kCh && p->ld_x >= kCh && p->ld_out >= kCh && p->ld_attn % 8 == 0 && p->ld_x % 8 == 0 && p->ld_out % 4 == 0 &&
                     (p->extra == nullptr || (p->ld_extra >= kCh && p->ld_extra % 4 == 0)) &&
                     (p->q_out_features == 0 || (p->ld_q >= p->q_out_features && p->ld_q % 8 == 0)),  // (q_out leaves as 16-byte pieces; x_out / extra as 8-byte ones)
                 "gt_chain_fwd: leading dimensions too small or not vector-aligned");
  ChainArgs a{};
  a.attn = p->attn; a.ld_attn = p->ld_attn;
  a.xres = p->x_res; a.ld_x = p->ld_x;
  a.wp = (const char*)p->wp; a.bp = p->bp;
  a.ln1_g = p->ln1_w; a.ln1_b = p->ln1_b; a.ln1_eps = p->ln1_eps;
  a.w1 = (const char*)p->w1; a.b1 = p->b1; a.hc = p->hidden / kCh;
  a.w2 = (const char*)p->w2; a.b2 = p->b2;
  a.extra = p->extra; a.ld_extra = p->ld_extra;
  a.xout = p->x_out; a.ld_out = p->ld_out;
  a.lnq_g = p->lnq_w; a.lnq_b = p->lnq_b; a.lnq_eps = p->lnq_eps;
  a.wq = (const char*)p->wq; a.bq = p->bq; a.qc = p->q_out_features / kCh;
  a.qout = p->q_out; a.ld_q = p->ld_q;
  a.timeline = reinterpret_cast<unsigned long long*>(p->timeline);
  static const int prio_young = env_int(getenv("ANEMOI_CHAIN_PRIO_YOUNG"), 0, 0, 3);
  a.prio_young = prio_young;
  static const int dbg = env_int(getenv("ANEMOI_CHAIN_DBG"), 0, 0, 255);
  a.dbg = dbg;
  a.n_rows = p->n_rows;
  a.rows_per_tile = p->rows_per_tile > 0 ? p->rows_per_tile : anemoi_gt_chain_rows_per_tile(p->n_rows);
  ANEMOI_REQUIRE(a.rows_per_tile <= kPanel, "gt_chain_fwd: rows_per_tile=%d exceeds the %d-row panel", a.rows_per_tile, kPanel);
  a.n_tiles = (a.n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  hipStream_t st = as_stream(stream);
  return dtype == ANEMOI_BF16 ? launch_chain<bf16_t>(a, st) : launch_chain<f16_t>(a, st);
}
