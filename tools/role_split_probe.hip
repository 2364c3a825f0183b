// Developer probe for the role-split layer chain (csrc/gt_chain2.hip, round 5): the same 6.5 MiB weight stream per CU and layer as
// tools/weight_stream_probe.hip, but worked by TWO GROUPS of four waves (one wave of each group per SIMD).  A wave owns a
// 48 x 128 output slab (24 accumulator quads, 3 A fragments per K-step shared by 8 column blocks) and streams its own 8 KiB per
// K-step through a register ring of RD K-steps; the groups run DIFFERENT segments (as the M1 / M2 roles of the MLP pipeline do), so one
// group's epilogue (VALU, LDS writes) runs beside the other group's MFMA stream.
//   hipcc --offload-arch=gfx950 -O3 tools/role_split_probe.hip -o /tmp/rsp && /tmp/rsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using frag8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

typedef const __attribute__((address_space(1))) char* gptr_t;
__device__ __forceinline__ gptr_t uniform_ptr(const char* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<gptr_t>(((uint64_t)hi << 32) | lo);
}
typedef const __attribute__((address_space(1))) frag8* gfrag_t;
__device__ __forceinline__ frag8 ldw(const char* base, int off, unsigned loff) {  // base: wave-uniform; off: compile-time
  const gptr_t b = uniform_ptr(base + (off & ~4095));
  return *reinterpret_cast<gfrag_t>(b + loff + (off & 4095));
}

constexpr int kWaveSeg = 16 * 8192;       // one wave-segment: 16 K-steps x 8 fragments x 1 KiB
constexpr int kGroupSeg = 4 * kWaveSeg;   // 512 KiB = one [512 x 512] weight

// MODE 0: both groups free-running (A 7 segments, B 6), no synchronisation
// MODE 1: the layer's lock-step schedule with s_barrier between steps: A | - | A | AB AB AB | B | - | AB AB
// MODE 2: MODE 1 + a GELU-like VALU epilogue + LDS writes behind A's four "MLP-1" segments and a cvt + LDS write behind every other one
// ORDER 0: MFMAs row-band-major, the A fragment of a band re-read right behind its block (no second fragment set)
// ORDER 1: MFMAs column-block-major, a ring slot refilled right behind its three MFMAs, A fragments double-buffered
template <int RD, int MODE, int ORDER, int SOLO /* 1: only group A works (all 13 segments) */>
__global__ __launch_bounds__(512, 1) void probe(const char* __restrict__ W, float* sink, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  for (int i = threadIdx.x; i < 2 * 48 * 1024 / 4; i += 512) {
    const unsigned h = (i + 1) * 2654435761u, h2 = (i + 77) * 40503u * 2654435761u;
    reinterpret_cast<unsigned*>(smem)[i] = (0x3f80u | ((h >> 9) & 0x7fu) | ((h >> 31) << 15)) | ((0x3f80u | ((h2 >> 9) & 0x7fu) | ((h2 >> 31) << 15)) << 16);
  }
  __syncthreads();
  unsigned char* const outbuf = smem + 48 * 1024;
  f32x4 acc[3][8];
  frag8 ring[RD][8];
  const int x = lane & 15, ks = lane >> 4;
  const unsigned char* arow = smem + x * 1024;
  const unsigned loff = lane * 16;

  auto seg_ptr = [&](int s) { return W + (size_t)(s % 13) * kGroupSeg + (size_t)wq * kWaveSeg; };
  auto prefetch = [&](const char* p) {
#pragma unroll
    for (int j = 0; j < RD; ++j)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        ring[j][ni] = ldw(p, j * 8192 + ni * 1024, loff);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  auto gemm = [&](const char* cur, const char* nxt) {
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag8 fa[3];
#pragma unroll
    for (int mi = 0; mi < 3; ++mi) fa[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16384 + ((ks ^ x) << 4));
    constexpr int NQ = 16 / RD;
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
      const char* pfg = q < NQ - 1 ? cur + (q + 1) * RD * 8192 : nxt;
#pragma unroll
      for (int j = 0; j < RD; ++j) {
        const int st = q * RD + j;
        const int sn = st < 15 ? st + 1 : 15;
        if (ORDER == 0) {
#pragma unroll
          for (int mi = 0; mi < 3; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[j][ni], fa[mi], acc[mi][ni], 0, 0, 0);
            fa[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16384 + (((sn * 4 + ks) ^ x) << 4));
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int ni = 0; ni < 8; ++ni) ring[j][ni] = ldw(pfg, j * 8192 + ni * 1024, loff);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          frag8 fn[3];
#pragma unroll
          for (int mi = 0; mi < 3; ++mi) fn[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16384 + (((sn * 4 + ks) ^ x) << 4));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < 8; ++ni) {
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[j][ni], fa[mi], acc[mi][ni], 0, 0, 0);
            ring[j][ni] = ldw(pfg, j * 8192 + ni * 1024, loff);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int mi = 0; mi < 3; ++mi) fa[mi] = fn[mi];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  auto epilogue = [&](bool gelu) {
    // lane = panel row x of band mi, 4 consecutive columns of column block ni
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[mi][ni][r];
          if (gelu) {
            const float a = fabsf(v);
            float p = fmaf(a, -0.0113f, 0.0721f);
            p = fmaf(p, a, -0.3312f);
            p = fmaf(p, a, -1.1283f);
            p = fmaf(p, a, -1.0f);
            v = fmaxf(v, 0.f) - a * __builtin_amdgcn_exp2f(p * a);
          }
          t[r] = v;
        }
        const unsigned lo = (__float_as_uint(t[0]) >> 16) | (__float_as_uint(t[1]) & 0xffff0000u);
        const unsigned hi = (__float_as_uint(t[2]) >> 16) | (__float_as_uint(t[3]) & 0xffff0000u);
        *reinterpret_cast<u32x2*>(outbuf + (mi * 16 + x) * 1024 + (((wq * 16 + ni * 2 + (ks >> 1)) ^ x) << 4) + (ks & 1) * 8) = u32x2{lo, hi};
      }
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  float tsum = 0.f;
  auto fold = [&]() {
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) tsum += (acc[mi][ni][0] + acc[mi][ni][1]) + (acc[mi][ni][2] + acc[mi][ni][3]);
  };
  for (int rep = 0; rep < reps; ++rep) {
    if (SOLO) {
      if (grp == 0) {
        prefetch(seg_ptr(0));
        for (int s = 0; s < 13; ++s) {
          gemm(seg_ptr(s), seg_ptr(s + 1));
          if (MODE == 2) epilogue(s >= 1 && s <= 4);
          fold();
        }
      }
      continue;
    }
    if (MODE == 0) {
      const int n = grp == 0 ? 7 : 6, s0 = grp == 0 ? 0 : 7;
      prefetch(seg_ptr(s0));
      for (int s = 0; s < n; ++s) {
        gemm(seg_ptr(s0 + s), seg_ptr(s0 + (s + 1 < n ? s + 1 : s)));
        fold();
      }
    } else {
      // steps: 0 A(P) | 1 - (LayerNorm) | 2 A(M1_0) | 3,4,5 A(M1_c+1) B(M2_c) | 6 B(M2_3) | 7 - (LayerNorm') | 8,9 A(Q) B(Q)
      int sa = 0, sb = 7;
      prefetch(seg_ptr(grp == 0 ? 0 : 7));
      for (int step = 0; step < 10; ++step) {
        const bool a_on = step == 0 || (step >= 2 && step <= 5) || step >= 8;
        const bool b_on = (step >= 3 && step <= 6) || step >= 8;
        if (grp == 0 && a_on) {
          gemm(seg_ptr(sa), seg_ptr(sa + 1 < 7 ? sa + 1 : 0));
          if (MODE == 2) epilogue(step >= 2 && step <= 5);
          fold();
          ++sa;
        }
        if (grp == 1 && b_on) {
          gemm(seg_ptr(sb), seg_ptr(sb + 1 < 13 ? sb + 1 : 7));
          if (MODE == 2 && (step == 6 || step >= 8)) epilogue(false);
          fold();
          ++sb;
        }
        bar();
      }
    }
  }
  if (tsum == 123.456f) sink[0] = tsum;
}

template <int RD, int MODE, int ORDER, int SOLO = 0>
static void run(const char* name, const char* W, float* sink) {
  const int lds = 96 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<RD, MODE, ORDER, SOLO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 16;
  probe<RD, MODE, ORDER, SOLO><<<256, 512, lds>>>(W, sink, 1);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    probe<RD, MODE, ORDER, SOLO><<<256, 512, lds>>>(W, sink, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double us_layer = best * 1e3 / reps;
  const double bytes_cu = 13.0 * kGroupSeg;
  const double flop = 256.0 * 13 * 4 * 16 * 24 * 2.0 * 16 * 16 * 32;
  printf("%-52s %7.2f us per layer | %6.1f GB/s per CU | %6.1f TFLOP/s (48-row panel)\n", name, us_layer, bytes_cu / us_layer * 1e-3, flop / us_layer * 1e-6);
}

int main() {
  const size_t bytes = (size_t)13 * kGroupSeg;
  std::vector<unsigned short> h(bytes / 2);
  unsigned s = 12345u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (unsigned short)(0x3c00u + ((s >> 9) & 0x3ffu) + ((s >> 31) << 15));
  }
  char* W;
  float* sink;
  hipMalloc(&W, bytes);
  hipMalloc(&sink, 64);
  hipMemcpy(W, h.data(), bytes, hipMemcpyHostToDevice);
  run<4, 0, 0>("free-running, ring 4, band-major", W, sink);
  run<4, 0, 1>("free-running, ring 4, block-major", W, sink);
  run<2, 0, 0>("free-running, ring 2, band-major", W, sink);
  run<2, 0, 1>("free-running, ring 2, block-major", W, sink);
  run<4, 0, 0, 1>("one group alone (13 segments), ring 4, band-major", W, sink);
  run<4, 0, 1, 1>("one group alone (13 segments), ring 4, block-major", W, sink);
  run<2, 0, 1, 1>("one group alone (13 segments), ring 2, block-major", W, sink);
  run<4, 1, 0>("layer schedule + barriers, ring 4, band-major", W, sink);
  run<4, 1, 1>("layer schedule + barriers, ring 4, block-major", W, sink);
  run<4, 2, 0>("  + epilogues (GELU, cvt, LDS), band-major", W, sink);
  run<4, 2, 1>("  + epilogues (GELU, cvt, LDS), block-major", W, sink);
  run<2, 2, 1>("  + epilogues, ring 2, block-major", W, sink);
  run<4, 2, 1, 1>("one group alone + epilogues, ring 4, block-major", W, sink);
  return 0;
}
