// Two-group form of GraphConv's edge-MLP chain for gfx950 (round 5) - what gnn_chain.hip's gnn_edge_chain_kernel computes,
//
//     e' = LayerNorm(W_2 gelu(W_1 gelu(W_e e + (x W_i^T)[dst] + (x W_j^T)[src] + b_0) + b_1) + b_2) + e        (reference layers/conv.py:29-81)
//
// (and its MLP instantiation: an embedding MLP Linear-GELU-Linear-GELU-Linear-LayerNorm, layers/mlp.py:29-100), on the role-split
// machinery of chain2_core.h: the workgroup's eight waves are TWO INDEPENDENT GROUPS of four (one wave of each per SIMD), each running
// the whole chain in place on its own panel of <= 48 rows (e -> h1 -> h2 -> the staged output in ONE LDS buffer), a wave owning a
// 48 x 128 slab.  Nothing ties the groups together - the only synchronisation is a barrier among a group's four waves (a monotonic LDS
// counter: the hardware barrier counts all eight) - so they drift apart and one's epilogues, gathers and row loads run beside the
// other's GEMM segments.  b_0 and the gathered node-level rows enter as the START values of the first GEMM's accumulators (an fp32
// re-association of the symmetric kernel's arithmetic); the per-column vectors sit in LDS.
//
// MEASURED (profiles/r05_gnn_edge_chain_role_split.txt): parity-green and SLOWER than the symmetric kernel - 212 against 195 us at
// 81 840 rows (a lock-step two-role form with three 48-row buffers, built first: 198-205 us).  The chain is bound by the CU's L1 path
// times the number of passes over the 1.5 MB of weights, and 40 / 48-row panels need 7-8 passes where the symmetric kernel's 64-row
// panels need 5.  EXPERIMENTS BUILD ONLY (python -m anemoi_core_amd.build --experiments -> lib/libanemoi_hip_exp.so, then
// ANEMOI_HIP_LIB=.../libanemoi_hip_exp.so ANEMOI_GNN_CHAIN_V2=1 selects this kernel); the product library does not contain it.
#include "chain2_core.h"
#include "gnn_chain_args.h"

namespace anemoi {

// A barrier among the FOUR waves of one group (the hardware barrier counts all eight): a monotonic LDS counter.  LDS operations of a
// wave complete in order, so whoever sees a wave's increment sees what it wrote before.  Bounded: a miscount must not hang the GPU - and must not go unnoticed either (it traps).
__device__ __forceinline__ void group4_barrier(unsigned* ctr, unsigned& epoch, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  epoch += 4;
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  bool ok = false;
  for (int it = 0; it < (1 << 16); ++it) {
    const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int)(v - epoch) >= 0) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (!ok) __builtin_trap();  // a miscount must neither hang the GPU nor carry on over an incomplete panel: the launch FAILS (check_launch / the next sync)
  asm volatile("" ::: "memory");
}


constexpr int kG2RedOff = 2 * kBufBytes;                     // per group [48 rows][4 waves][2] fp32 LayerNorm partials
constexpr int kG2VecOff = kG2RedOff + 2 * kPanel * 4 * 2 * 4;  // b_0 | b_1 | b_2 | gamma | beta (16-bit)
constexpr int kG2CtrOff = kG2VecOff + 5 * kCh * 2;           // the counters of the two groups' barriers
constexpr int kG2Smem = kG2CtrOff + 32;
static_assert(kG2Smem <= 160 * 1024, "LDS budget");

// 12 rows of `nslots` 16-byte slots each (a panel of a narrow first GEMM: K = 8 nslots) - load_rows12 with a width
template <typename T>
__device__ __forceinline__ void load_rows12_w(const T* src, int64_t ld, int r0, int nr, unsigned char* dst, int lane, int wq, int nslots) {
  asm volatile("" : "+v"(lane), "+s"(wq));
  u32x4 v[12];
  if (lane < nslots) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int row = wq * 12 + i;
      v[i] = *reinterpret_cast<const u32x4*>(src + (int64_t)(r0 + min(row, nr - 1)) * ld + lane * 8);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int row = wq * 12 + i;  // (the swizzle permutes the 16 slots of a 128-column group among themselves)
      *reinterpret_cast<u32x4*>(dst + row * kRowBytes + ((lane ^ (row & 15)) << 4)) = row < nr ? v[i] : u32x4{0u, 0u, 0u, 0u};
    }
  }
}

// One group's whole chain on its own panels (group gid of 2 x gridDim.x, panels gid, gid + 2 gridDim.x, ...), in place in ONE panel
// buffer: e -> h1 -> h2 -> the staged output.  Only the group's own barrier: nothing ties the two groups of a CU together, so they
// drift apart and one's epilogues, gathers and row loads run beside the other's GEMM segments.
template <typename T, bool MLP>
__device__ __forceinline__ void edge_group_chain(const EdgeChainArgs& a, unsigned char* buf, float* red, const unsigned char* vec, unsigned* ctr, int gid,
                                                 int lane, int wq) {
  const uint32_t loff = lane * 16;
  const int nq0 = MLP ? 2 * a.k0_groups : 8;       // pairs of K-steps of the first GEMM
  const int64_t cs0 = (int64_t)nq0 * 8192;         // one 64-column slab of w0
  const char* const w0w = a.w0 + (int64_t)(2 * wq) * cs0;
  const char* const w1w = a.w1 + (int64_t)(2 * wq) * kSlab;
  const char* const w2w = a.w2 + (int64_t)(2 * wq) * kSlab;
  const T* const resp = MLP ? (const T*)a.res : (const T*)a.e;
  const int64_t ld_res = MLP ? a.ld_res : a.ld_e;
  frag8 ring[2][8];
  f32x4 acc[3][8];
  unsigned epoch = 0;
  ring_prologue(ring, w0w, cs0, loff);
  for (int tile = gid; tile < a.n_tiles; tile += 2 * (int)gridDim.x) {
    const int r0 = tile * a.rows_per_tile;
    const int nr = min(a.rows_per_tile, a.n_rows - r0);
    // ---- the e rows -> the panel; the start values of G1's accumulators: b_0 (+ the two gathered node-level rows of every panel row:
    // all six row indices first, then the rows of two bands in flight at a time)
    load_rows12_w<T>((const T*)a.e, a.ld_e, r0, nr, buf, lane, wq, nq0 * 8);
    {
      const Lane2 lc = lane2(lane, wq);
      if constexpr (MLP) {
        init_acc<T, false>(acc, vec, 0, nullptr, lane, wq);
      } else {
        int i1[3], i2[3];
#pragma unroll
        for (int mi = 0; mi < 3; ++mi) {
          const int m = r0 + min(mi * 16 + lc.x, nr - 1);
          i1[mi] = a.idx1[m];
          i2[mi] = a.idx2[m];
        }
        u32x2 ga[2][8], gb[2][8];
        auto request = [&](int mi, u32x2 (&pa)[8], u32x2 (&pb)[8]) {
          const T* r1 = (const T*)a.g1 + (int64_t)i1[mi] * a.ld_g1 + wq * 128 + lc.g * 4;
          const T* r2 = (const T*)a.g2 + (int64_t)i2[mi] * a.ld_g2 + wq * 128 + lc.g * 4;
#pragma unroll
          for (int ni = 0; ni < 8; ++ni) {
            if (a.dbg & 2) {  // (timing experiment: no gathered rows)
              pa[ni] = pb[ni] = u32x2{0u, 0u};
            } else {
              pa[ni] = *reinterpret_cast<const u32x2*>(r1 + ni * 16);
              pb[ni] = *reinterpret_cast<const u32x2*>(r2 + ni * 16);
            }
          }
        };
        auto consume = [&](int mi, const u32x2 (&pa)[8], const u32x2 (&pb)[8]) {
#pragma unroll
          for (int ni = 0; ni < 8; ++ni) {
            float b[4], t1[4], t2[4];
            unpack4<T>(*reinterpret_cast<const u32x2*>(vec + (wq * 128 + ni * 16 + lc.g * 4) * 2), b);
            unpack4<T>(pa[ni], t1);
            unpack4<T>(pb[ni], t2);
            acc[mi][ni] = f32x4{b[0] + (t1[0] + t2[0]), b[1] + (t1[1] + t2[1]), b[2] + (t1[2] + t2[2]), b[3] + (t1[3] + t2[3])};
          }
        };
        request(0, ga[0], gb[0]);
        request(1, ga[1], gb[1]);
        __builtin_amdgcn_sched_barrier(0);
        consume(0, ga[0], gb[0]);
        __builtin_amdgcn_sched_barrier(0);
        request(2, ga[0], gb[0]);
        __builtin_amdgcn_sched_barrier(0);
        consume(1, ga[1], gb[1]);
        consume(2, ga[0], gb[0]);
      }
    }
    group4_barrier(ctr, epoch, lane);  // the panel is complete
    // ---- h1 = gelu(e W_e^T + ...) in place
    gemm128<T>(buf, lane, ring, w0w, cs0, w1w, kSlab, loff, acc, nq0);
    group4_barrier(ctr, epoch, lane);  // every wave of the group is behind its last fragment read
    if (a.dbg & 1) round_rows<T, false>(acc, buf, nullptr, lane, wq);
    else gelu_rows<T>(acc, buf, lane, wq);
    init_acc<T, false>(acc, vec, kCh, nullptr, lane, wq);
    group4_barrier(ctr, epoch, lane);
    // ---- h2 = gelu(h1 W_1^T + b_1) in place
    gemm128<T>(buf, lane, ring, w1w, kSlab, w2w, kSlab, loff, acc);
    group4_barrier(ctr, epoch, lane);
    if (a.dbg & 1) round_rows<T, false>(acc, buf, nullptr, lane, wq);
    else gelu_rows<T>(acc, buf, lane, wq);
    init_acc<T, false>(acc, vec, 2 * kCh, nullptr, lane, wq);
    group4_barrier(ctr, epoch, lane);
    // ---- z = h2 W_2^T + b_2 (rounded, as the Linear's output is); e' = LayerNorm(z) + e -> global
    gemm128<T>(buf, lane, ring, w2w, kSlab, w0w, cs0, loff, acc);
    {
      // this lane's values of the residual rows (L2-hot: the group read the same rows for the panel): in flight under the rounding and the statistics
      const Lane2 lc = lane2(lane, wq);
      u32x2 er[3][8];
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const T* erow = resp != nullptr ? resp + (int64_t)(r0 + min(mi * 16 + lc.x, nr - 1)) * ld_res + wq * 128 + lc.g * 4 : nullptr;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) er[mi][ni] = (resp != nullptr && !(a.dbg & 4)) ? *reinterpret_cast<const u32x2*>(erow + ni * 16) : u32x2{0u, 0u};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          float o[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
          unpack4<T>(pack4<T>(o), o);
          acc[mi][ni] = f32x4{o[0], o[1], o[2], o[3]};
        }
        float sm = 0.f;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) sm += (acc[mi][ni][0] + acc[mi][ni][1]) + (acc[mi][ni][2] + acc[mi][ni][3]);
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        const float mw = sm * (1.0f / 128.0f);
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
      group4_barrier(ctr, epoch, lane);  // the partials of all four waves; and every wave is behind its last fragment read of the panel
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const f32x4* pr = reinterpret_cast<const f32x4*>(red + (mi * 16 + lc.x) * 8);
        const f32x4 p0 = pr[0], p1 = pr[1];
        const float mu = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * 0.25f;
        const float d0 = p0[0] - mu, d1 = p0[2] - mu, d2 = p1[0] - mu, d3 = p1[2] - mu;
        const float m2 = fmaf(128.0f, (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3), (p0[1] + p0[3]) + (p1[1] + p1[3]));
        const float rstd = rsqrtf(m2 * (1.0f / (float)kCh) + a.ln_eps);
        unsigned char* drow = buf + (mi * 16 + lc.x) * kRowBytes;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          float gv[4], bv[4], ev[4], o[4];
          unpack4<T>(*reinterpret_cast<const u32x2*>(vec + (3 * kCh + wq * 128 + ni * 16 + lc.g * 4) * 2), gv);
          unpack4<T>(*reinterpret_cast<const u32x2*>(vec + (4 * kCh + wq * 128 + ni * 16 + lc.g * 4) * 2), bv);
          unpack4<T>(er[mi][ni], ev);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaf((acc[mi][ni][r] - mu) * rstd, gv[r], bv[r]) + ev[r];  // edge_ln_res_segsum's arithmetic
          *reinterpret_cast<u32x2*>(drow + lc.coff[ni]) = pack4<T>(o);
        }
      }
      store_staged<T>(buf, (T*)a.e_new + (int64_t)r0 * a.ld_o, a.ld_o, (a.dbg & 8) ? 0 : nr, lane, wq);
    }
    group4_barrier(ctr, epoch, lane);  // every wave has read its staged columns back (and the partials): the next panel's rows may come in
  }
}

template <typename T, bool MLP>
__global__ __launch_bounds__(512, 1) void gnn_edge_chain2_kernel(EdgeChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = __builtin_amdgcn_readfirstlane(wave & 3), grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  // the per-column vectors -> LDS (b_0 | b_1 | b_2 | gamma | beta: 5 x 64 pieces of 16 bytes), the two groups' barrier counters
  if (tid < 320) {
    const int v = tid >> 6, i = tid & 63;
    const void* src = v == 0 ? a.b0 : v == 1 ? a.b1 : v == 2 ? a.b2 : v == 3 ? a.ln_g : a.ln_b;
    reinterpret_cast<u32x4*>(smem + kG2VecOff)[tid] = src != nullptr ? reinterpret_cast<const u32x4*>(src)[i] : u32x4{0u, 0u, 0u, 0u};
  }
  if (tid < 8) reinterpret_cast<unsigned*>(smem + kG2CtrOff)[tid] = 0u;
  // L2 warm-up: this CU's 1/32 share of the three weights (chain2_core.h)
  Warm warm;
  warm_init<false>(warm, wave & 1, lane);
  if (wave >= 2) {
    const int which = (wave - 2) >> 1;
    const char* seg = which == 0 ? a.w0 : which == 1 ? a.w1 : a.w2;
    if (which > 0 || !MLP || a.k0_groups == 4) touch_share<false>(warm, seg, kSlab, wave & 1, lane);
  }
  lds_barrier();  // the ONE workgroup barrier: the vectors and the counters
  edge_group_chain<T, MLP>(a, smem + grp * kBufBytes, reinterpret_cast<float*>(smem + kG2RedOff) + grp * (kPanel * 8), smem + kG2VecOff,
                           reinterpret_cast<unsigned*>(smem + kG2CtrOff) + grp * 4, (int)blockIdx.x * 2 + grp, lane, wq);
  touch_done<false>(warm);
}

template <typename T, bool MLP>
static int launch2(const EdgeChainArgs& a, hipStream_t st) {
  static PerDeviceOnce once;
  once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gnn_edge_chain2_kernel<T, MLP>), hipFuncAttributeMaxDynamicSharedMemorySize, kG2Smem); });
  const int grid = a.n_tiles < 512 ? (a.n_tiles + 1) / 2 : 256;
  hipLaunchKernelGGL((gnn_edge_chain2_kernel<T, MLP>), dim3(grid), dim3(512), kG2Smem, st, a);
  return check_launch(MLP ? "gnn_mlp_chain2_kernel" : "gnn_edge_chain2_kernel");
}

int launch_edge_chain2(const EdgeChainArgs& a0, int dtype, void* stream, bool mlp) {
  EdgeChainArgs a = a0;
  // this kernel copies the per-column vectors as 16-byte pieces (the symmetric kernel, whose entry points validate the operands, reads 8-byte ones)
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  ANEMOI_REQUIRE(al16(a.b0) && al16(a.b1) && al16(a.b2) && al16(a.ln_g) && al16(a.ln_b), "gnn_edge_chain2: the bias / LayerNorm vectors must be 16-byte aligned");
  // experiments (timing only, results are garbage): bit 0 no GELU arithmetic, 1 no gathered rows, 2 no residual rows, 3 no global stores
  static const int dbg = env_int(getenv("ANEMOI_EDGE_CHAIN2_DBG"), 0, 0, 15);
  a.dbg = dbg;
  static const int rows = env_int(getenv("ANEMOI_EDGE_CHAIN2_ROWS"), 0, 0, kPanel);
  // whole rounds of panels over the 512 groups (two per CU), panels as even as the 48-row limit allows
  const int64_t rounds = ((int64_t)a.n_rows + 512 * kPanel - 1) / (512 * kPanel);
  const int64_t r = ((int64_t)a.n_rows + 512 * rounds - 1) / (512 * rounds);
  a.rows_per_tile = rows > 0 ? rows : (int)(r < 1 ? 1 : (r > kPanel ? kPanel : r));
  a.n_tiles = (a.n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  hipStream_t st = as_stream(stream);
  if (dtype == ANEMOI_BF16) return mlp ? launch2<bf16_t, true>(a, st) : launch2<bf16_t, false>(a, st);
  return mlp ? launch2<f16_t, true>(a, st) : launch2<f16_t, false>(a, st);
}

}  // namespace anemoi
