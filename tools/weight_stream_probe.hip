// Developer probe for the row-resident layer kernel (csrc/gt_chain.hip): how fast can EVERY CU stream the same fragment-major
// weight image (6.5 MB per processor layer: Wp, W1, W2, Wqkvs) from L2 straight into registers and multiply it against a
// 48-row activation panel that stays in LDS?  8 waves per CU, every wave owns a 64-column slab: per 32-wide K-step 4 coalesced
// 1-KiB global_load_dwordx4 (the B fragments of 4 x 16 columns, prefetched 4 K-steps = 16 loads ahead in a register ring),
// 3 ds_read_b128 (the A fragments of 3 x 16 rows) and 12 MFMAs.  No barrier inside a segment of 16 K-steps.
//   hipcc --offload-arch=gfx950 -O3 tools/weight_stream_probe.hip -o /tmp/wsp && /tmp/wsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using frag8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kSegBytes = 16 * 4096;  // one segment: 16 K-steps x 4 KiB per wave

template <bool LOADS, bool MFMA, bool LDSR, int BAR = 0 /* s_barrier behind every BAR-th segment */, int PRIO = 0 /* s_setprio of waves 4-7 */, int VAR = 0>
__global__ __launch_bounds__(512, 1) void probe(const char* __restrict__ W, int segs_per_wave, int n_slabs, float* sink, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // fill the A panel (48 rows x 1 KiB) with something non-constant
  for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 512) {
    unsigned v = 0x3f803f80u ^ (i * 2654435761u & 0x007f007fu);
    if (VAR & 2) {  // random-ish bf16 in +-[1, 2): every mantissa and sign bit toggles
      const unsigned h = (i + 1) * 2654435761u, h2 = (i + 77) * 40503u * 2654435761u;
      v = (0x3f80u | ((h >> 9) & 0x7fu) | ((h >> 31) << 15)) | ((0x3f80u | ((h2 >> 9) & 0x7fu) | ((h2 >> 31) << 15)) << 16);
    }
    reinterpret_cast<unsigned*>(smem)[i] = v;
  }
  __syncthreads();
  f32x4 acc[3][4], accb[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = accb[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  frag8 bq[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bq[j][ni] = frag8{};
  const int x = lane & 15, ks = lane >> 4;
  const unsigned char* arow = smem + x * 1024;
  if (PRIO > 0 && wave >= 4) __builtin_amdgcn_s_setprio(PRIO);
  for (int rep = 0; rep < reps; ++rep) {
    // the wave's segment sequence: slab (seg * 8 + wave) % n_slabs of the image
    auto seg_base = [&](int s) { return W + (size_t)((s * 8 + wave) % n_slabs) * kSegBytes + lane * 16; };
    const char* pf = seg_base(0);
    if (LOADS) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) { bq[j][ni] = *reinterpret_cast<const frag8*>(pf + j * 4096 + ni * 1024); __builtin_amdgcn_sched_barrier(0); }
    }
    auto seg = [&](int s, f32x4 (&ac)[3][4]) {
      const char* base = seg_base(s);
      const char* nxt = seg_base(s + 1 < segs_per_wave ? s + 1 : s);
      frag8 fa[3];
      if ((VAR & 1) && LDSR) {
#pragma unroll
        for (int mi = 0; mi < 3; ++mi) fa[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16384 + ((ks ^ x) << 4));
      }
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        const char* pfg = q < 3 ? base + (q + 1) * 16384 : nxt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int st = q * 4 + j;
          frag8 fn[3];
          if ((VAR & 1) && LDSR) {
            const int sn = st < 15 ? st + 1 : 15;
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) fn[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16384 + (((sn * 4 + ks) ^ x) << 4));
            __builtin_amdgcn_sched_barrier(0);
          } else if (LDSR) {
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) fa[mi] = *reinterpret_cast<const frag8*>(arow + mi * 16384 + (((st * 4 + ks) ^ x) << 4));
          } else {
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) fa[mi] = bq[j][mi];
          }
          if (MFMA) {
#pragma unroll
            for (int mi = 0; mi < 3; ++mi)
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) ac[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j][ni], fa[mi], ac[mi][ni], 0, 0, 0);
          } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) ac[0][ni] += __builtin_bit_cast(f32x4, bq[j][ni]);
          }
          if (LOADS) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bq[j][ni] = *reinterpret_cast<const frag8*>(pfg + j * 4096 + ni * 1024);
          }
          if ((VAR & 1) && LDSR) {
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) fa[mi] = fn[mi];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (BAR > 0 && (s + 1) % BAR == 0) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    };
    for (int s = 0; s < segs_per_wave; ++s) {
      if ((VAR & 4) && (s & 1)) seg(s, accb);
      else seg(s, acc);
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3] + accb[i][j][0] + accb[i][j][3];
  if (t == 123.456f) sink[0] = t;
}

template <bool L, bool M, bool R, int BAR = 0, int PRIO = 0, int VAR = 0>
static void run(const char* name, const char* W, int segs, int n_slabs, float* sink, int grid) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<L, M, R, BAR, PRIO, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, ((VAR & 8) ? 150 : 64) * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 16;  // 16 "layers" per launch
  probe<L, M, R, BAR, PRIO, VAR><<<grid, 512, ((VAR & 8) ? 150 : 64) * 1024>>>(W, segs, n_slabs, sink, 1);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    probe<L, M, R, BAR, PRIO, VAR><<<grid, 512, ((VAR & 8) ? 150 : 64) * 1024>>>(W, segs, n_slabs, sink, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double us_layer = best * 1e3 / reps;
  const double bytes_cu = (double)segs * 8 * kSegBytes;  // per CU and layer
  const double flop = (double)grid * 8 * segs * 16 * 12 * 2.0 * 16 * 16 * 32;
  printf("%-34s grid %3d: %7.2f us per layer  | %6.1f GB/s per CU, %5.2f TB/s chip | %6.1f TFLOP/s (48-row panel)\n", name, grid, us_layer,
         L ? bytes_cu / us_layer * 1e-3 : 0.0, L ? bytes_cu * grid / us_layer * 1e-6 : 0.0, M ? flop / us_layer * 1e-6 : 0.0);
}

int main() {
  // one layer's weights: 13 segments of 16 K-steps per wave (Wp 1, W1 4, W2 4, Wqkvs 4) = 104 slabs of 64 KiB = 6.5 MiB
  const int segs = 13, n_slabs = 104;
  const size_t bytes = (size_t)n_slabs * kSegBytes;
  std::vector<unsigned short> h(bytes / 2);
  unsigned s = 12345u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (unsigned short)(0x3c00u + ((s >> 9) & 0x3ffu) + ((s >> 31) << 15));  // random-ish bf16 around +-1
  }
  char* W;
  float* sink;
  hipMalloc(&W, bytes);
  hipMalloc(&sink, 64);
  hipMemcpy(W, h.data(), bytes, hipMemcpyHostToDevice);
  for (int grid : {256}) {
    run<true, false, false>("loads only", W, segs, n_slabs, sink, grid);
    run<false, true, true>("MFMA + LDS reads only", W, segs, n_slabs, sink, grid);
    run<true, true, false>("loads + MFMA", W, segs, n_slabs, sink, grid);
    run<true, true, true>("loads + MFMA + LDS reads", W, segs, n_slabs, sink, grid);
    run<true, true, true, 1>("  + barrier every segment", W, segs, n_slabs, sink, grid);
    run<true, true, true, 1, 0, 1>("  + barrier/seg, A prefetch", W, segs, n_slabs, sink, grid);
    run<true, true, true, 1, 0, 2>("  + barrier/seg, random A panel", W, segs, n_slabs, sink, grid);
    run<false, true, true, 0, 0, 2>("  MFMA + LDS only, random A panel", W, segs, n_slabs, sink, grid);
    run<true, true, true, 1, 0, 4>("  + barrier/seg, two acc sets", W, segs, n_slabs, sink, grid);
    run<true, true, true, 1, 0, 8>("  + barrier/seg, 150 KiB LDS", W, segs, n_slabs, sink, grid);
    run<true, true, true, 1, 0, 15>("  + barrier/seg, all four", W, segs, n_slabs, sink, grid);
  }
  return 0;
}
