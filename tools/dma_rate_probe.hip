// Developer probe: what bounds the K-loop of the 320 x 256 big-tile GEMM (csrc/linear.hip)?  Every CU streams the operand
// tiles of a [10240 x 512] x [512 -> 2048] GEMM through the 2-stage LDS ring by global_load_lds_dwordx4, exactly as the kernel
// does (72 pieces of 1 KiB per K-step, 9 per wave), with NO fragment reads and optionally the K-step's 80 MFMAs per wave on
// register operands.  Variants: operand rows at their natural 1-KiB stride (a piece = 8 rows x 128 B) vs. K-tile-major packed
// operands (a piece = 1 contiguous KiB); default vs. non-temporal cache policy; 2 x 72 KiB stages vs. 4 x 36 KiB (BK = 32).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate_probe.hip -o /tmp/dmap && /tmp/dmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
using frag8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int N = 10240, K = 512, O = 2048;  // bf16 operands: rows of 1 KiB
constexpr int TBM = 320, TBN = 256;

// MODE 0: row-major operands (row stride K*2 bytes).  MODE 1: K-tile-major: [K/64][rows][128 B].
template <int MODE, int AUX, int MFMA /* MFMAs per wave and 64-wide K-step (the GEMM has 80) */, int BKB /* bytes of K per stage row: 128 (BK=64) or 64 (BK=32) */, bool DMA = true>
__global__ __launch_bounds__(512, 1) void probe(const char* __restrict__ A, const char* __restrict__ W, int steps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGES = 144 * 1024 / ((TBM + TBN) * BKB);
  constexpr int kStage = (TBM + TBN) * BKB;
  constexpr int kRowsPerPiece = 1024 / BKB;                  // 8 (BK=64) or 16 (BK=32)
  constexpr int kAP = TBM / kRowsPerPiece, kWP = TBN / kRowsPerPiece;  // pieces per stage
  constexpr int kAPW = (kAP + 7) / 8, kWPW = (kWP + 7) / 8;
  constexpr int kKT = K * 2 / BKB;                           // K-tiles per full K
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int id = blockIdx.x;
  {
    const int xcd = id & 7, pos = id >> 3;
    id = xcd * 32 + pos;
  }
  const int m0 = (id / 8) * TBM, n0 = (id % 8) * TBN;
  const uint32_t smem_l = (uint32_t)(size_t)(lds_void_t*)smem;
  const int lanes_per_row = BKB / 16, prow = lane / lanes_per_row, pcol = (lane % lanes_per_row) * 16;

  auto issue = [&](int g) {
    const int kt = g % kKT;
    const uint32_t st = smem_l + (g % STAGES) * kStage;
#pragma unroll
    for (int i = 0; i < kAPW; ++i) {
      const int p = wave * kAPW + i;
      if (p < kAP) {
        const char* src = MODE == 0 ? A + (size_t)(m0 + p * kRowsPerPiece + prow) * (K * 2) + kt * BKB + pcol
                                    : A + (size_t)kt * N * BKB + (size_t)(m0 + p * kRowsPerPiece) * BKB + lane * 16;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(size_t)(st + p * 1024), 16, 0, AUX);
      }
    }
#pragma unroll
    for (int i = 0; i < kWPW; ++i) {
      const int p = wave * kWPW + i;
      if (p < kWP) {
        const char* src = MODE == 0 ? W + (size_t)(n0 + p * kRowsPerPiece + prow) * (K * 2) + kt * BKB + pcol
                                    : W + (size_t)kt * O * BKB + (size_t)(n0 + p * kRowsPerPiece) * BKB + lane * 16;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(size_t)(st + TBM * BKB + p * 1024), 16, 0, AUX);
      }
    }
  };
  f32x4 acc[20];
  frag8 fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[i] = (__bf16)(float)(lane + i);
    fb[i] = (__bf16)(float)(lane * 3 + i);
  }
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (DMA) for (int s = 0; s < STAGES - 1; ++s) issue(s);
  for (int g = 0; g < steps; ++g) {
    // stage g landed; STAGES-2 newer stages may stay in flight (every wave issues the same number of pieces per stage or fewer)
    if constexpr (STAGES == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (kAPW + kWPW)) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (DMA && g + STAGES - 1 < steps) issue(g + STAGES - 1);
    if constexpr (MFMA > 0) {
      constexpr int kM = MFMA * BKB / 128;  // MFMAs per wave per stage
#pragma unroll
      for (int i = 0; i < kM; ++i) acc[i % 20] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i % 20], 0, 0, 0);
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 12345.678f) sink[threadIdx.x] = t;  // keep the MFMAs alive
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- full K-step emulation of the big-tile kernel: DMA + the fragment ds_read_b128 pattern + 80 MFMAs per wave ---------------
// PF: A fragments fetched PF bands ahead of their MFMAs (0 = right before use, as the product kernel); PLACE: where the 9 DMA
// pieces of the next stage are issued (0 = one per band in the first K-half, as the product; 1 = all before the first MFMA;
// 2 = one every other band over both K-halves); PRIO: s_setprio 1 around the MFMAs.
template <int PF, int PLACE, int PRIO, int EP = 0>
__global__ __launch_bounds__(512, 1) void kstep(const char* __restrict__ A, const char* __restrict__ W, int steps, float* sink, unsigned short* Y = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MI = 10, kStage = (TBM + TBN) * 128, kAPW = 5, kWPW = 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int id = blockIdx.x;
  id = (id & 7) * 32 + (id >> 3);
  const int m0 = (id / 8) * TBM, n0 = (id % 8) * TBN;
  const uint32_t smem_l = (uint32_t)(size_t)(lds_void_t*)smem;
  const int prow = lane >> 3, pslot = lane & 7;
  const char* asrc[kAPW];
  const char* wsrc[kWPW];
#pragma unroll
  for (int i = 0; i < kAPW; ++i) {
    const int row = (wave * kAPW + i) * 8 + prow;
    asrc[i] = A + (size_t)(m0 + row) * (K * 2) + ((pslot ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < kWPW; ++i) {
    const int row = (wave * kWPW + i) * 8 + prow;
    wsrc[i] = W + (size_t)(n0 + row) * (K * 2) + ((pslot ^ ((row >> 1) & 7)) << 4);
  }
  int ig = 0;
  uint32_t sdst = smem_l;
  int koff = 0;
  auto begin_issue = [&]() {
    sdst = smem_l + (ig & 1) * kStage;
    koff = (ig & 7) * 128;
  };
  auto piece = [&](int i) {
    if (i < kAPW)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[i] + koff), (lds_void_t*)(size_t)(sdst + (wave * kAPW + i) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(wsrc[i - kAPW] + koff), (lds_void_t*)(size_t)(sdst + TBM * 128 + (wave * kWPW + i - kAPW) * 1024), 16, 0, 0);
  };
  begin_issue();
#pragma unroll
  for (int i = 0; i < 9; ++i) piece(i);
  ++ig;
  const int frow = lane & 15, fslot = lane >> 4;
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ra = wr * 160 + frow, rw = wc * 64 + frow;
    a_rd[ks] = ra * 128 + (((fslot + 4 * ks) ^ ((ra >> 1) & 7)) << 4);
    w_rd[ks] = TBM * 128 + rw * 128 + (((fslot + 4 * ks) ^ ((rw >> 1) & 7)) << 4);
  }
  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < steps; ++g) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned char* st = smem + (g & 1) * kStage;
    begin_issue();
    if (PLACE == 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) piece(i);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag8 fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const frag8*>(st + w_rd[ks] + i * 16 * 128);
      frag8 fa[MI];
      if (PF > 0) {
#pragma unroll
        for (int i = 0; i < PF; ++i) fa[i] = *reinterpret_cast<const frag8*>(st + a_rd[ks] + i * 16 * 128);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if (PF == 0) fa[mi] = *reinterpret_cast<const frag8*>(st + a_rd[ks] + mi * 16 * 128);
        else if (mi + PF < MI) fa[mi + PF] = *reinterpret_cast<const frag8*>(st + a_rd[ks] + (mi + PF) * 16 * 128);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[mi][ni], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (PLACE == 0 && ks == 0 && mi < 9) piece(mi);
        if (PLACE == 2 && (mi & 1) == 0 && ks * 5 + mi / 2 < 9) piece(ks * 5 + mi / 2);
      }
    }
    ++ig;
  }
  if constexpr (EP == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) sink[threadIdx.x] = t;
  } else {
    __builtin_amdgcn_s_barrier();
    unsigned char* epi = smem + wave * 4096;
    const int cp = lane & 7, wrow = lane & 15;
    unsigned short* ylane = Y + (size_t)(m0 + wr * 160 + (lane >> 3)) * O + n0 + wc * 64 + cp * 8;
    float keep = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      f32x4 c[2][2];
      if constexpr (EP == 3) {  // no LDS: the accumulators as they are
        c[0][0] = acc[mi][0]; c[0][1] = acc[mi][1]; c[1][0] = acc[mi][2]; c[1][1] = acc[mi][3];
      } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int slot = ni * 4 + (lane >> 4);
          *reinterpret_cast<f32x4*>(epi + wrow * 256 + ((((slot >> 1) ^ (wrow & 7)) << 1) | (slot & 1)) * 16) = acc[mi][ni];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3);
          c[it][0] = *reinterpret_cast<const f32x4*>(epi + row * 256 + ((cp ^ (row & 7)) << 5));
          c[it][1] = *reinterpret_cast<const f32x4*>(epi + row * 256 + ((cp ^ (row & 7)) << 5) + 16);
        }
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        u4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const __bf16 lo = (__bf16)(c[it][(2 * r) >> 2][(2 * r) & 3] + 1.0f), hi = (__bf16)(c[it][(2 * r + 1) >> 2][(2 * r + 1) & 3] + 1.0f);
          o[r] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        if constexpr (EP == 1) keep += __builtin_bit_cast(float, o[0] ^ o[1] ^ o[2] ^ o[3]);
        else *reinterpret_cast<u4*>(ylane + (size_t)(mi * 16 + it * 8) * O) = o;
      }
    }
    if (EP == 1 && keep == 12345.678f) sink[threadIdx.x] = keep;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PF, int PLACE, int PRIO, int EP = 0>
static void run2(const char* name, const char* A, const char* W, float* sink, unsigned short* Y = nullptr) {
  auto k = kstep<PF, PLACE, PRIO, EP>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int ksweeps : {1, 16}) {
    const int steps = ksweeps * 8;
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 144 * 1024, 0, A, W, steps, sink, Y);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 1 && ms < best) best = ms;
    }
    printf("%-58s K-sweeps %2d: %8.1f us  %6.2f us per 64-wide K-step\n", name, ksweeps, best * 1e3, best * 1e3 / steps);
  }
}

template <int MODE, int AUX, int MFMA, int BKB, bool DMA = true>
static void run(const char* name, const char* A, const char* W, float* sink) {
  auto k = probe<MODE, AUX, MFMA, BKB, DMA>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int kt_per_k = K * 2 / BKB;
  for (int ksweeps : {1, 16}) {
    const int steps = ksweeps * kt_per_k;
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 144 * 1024, 0, A, W, steps, sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 1 && ms < best) best = ms;
    }
    const double bytes_per_cu = (double)steps * (TBM + TBN) * BKB;
    printf("%-58s K-sweeps %2d: %8.1f us  %6.2f us per 64-wide K-step  %6.1f GB/s per CU\n", name, ksweeps, best * 1e3,
           best * 1e3 / (steps * BKB / 128.0), bytes_per_cu / (best * 1e-3) * 1e-9);
  }
}

int main() {
  char *A, *W;
  float* sink;
  hipMalloc(&A, (size_t)N * K * 2);
  hipMalloc(&W, (size_t)O * K * 2);
  hipMalloc(&sink, 4096);
  hipMemset(A, 0x11, (size_t)N * K * 2);
  hipMemset(W, 0x22, (size_t)O * K * 2);
  if (getenv("PROBE_RANDOM")) {  // bf16 N(0,1)-like random operands: DVFS lowers the clock against constant data (guide rule 25)
    auto fill = [](char* d, size_t n) {
      unsigned short* h = (unsigned short*)malloc(n * 2);
      unsigned s = 12345u;
      for (size_t i = 0; i < n; ++i) {
        float acc = 0.f;
        for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; acc += (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
        const float v = acc * 1.732f;  // ~N(0,1)
        unsigned u; memcpy(&u, &v, 4);
        h[i] = (unsigned short)(u >> 16);
      }
      hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
      free(h);
    };
    fill(A, (size_t)N * K);
    fill(W, (size_t)O * K);
    printf("operands: random ~N(0,1) bf16\n");
  }
  run<0, 0, 0, 128>("row-major, DMA only", A, W, sink);
  run<1, 0, 0, 128>("K-tile-major (contiguous KiB pieces), DMA only", A, W, sink);
  run<0, 0, 80, 128, false>("80 MFMA per wave and K-step, NO DMA", A, W, sink);
  run<0, 0, 20, 128>("row-major, DMA + 20 MFMA", A, W, sink);
  run<0, 0, 40, 128>("row-major, DMA + 40 MFMA", A, W, sink);
  run<0, 0, 60, 128>("row-major, DMA + 60 MFMA", A, W, sink);
  run<0, 0, 80, 128>("row-major, DMA + 80 MFMA (the GEMM's ratio)", A, W, sink);
  run<0, 0, 120, 128>("row-major, DMA + 120 MFMA", A, W, sink);
  run<1, 0, 80, 128>("K-tile-major, DMA + 80 MFMA", A, W, sink);
  run<0, 2, 80, 128>("row-major, nt policy, DMA + 80 MFMA", A, W, sink);
  run<0, 0, 80, 64>("row-major, 4 stages of BK=32, DMA + 80 MFMA", A, W, sink);
  printf("---- full K-step (DMA + fragment reads + MFMA)\n");
  run2<0, 0, 0>("as the product kernel (frag at use, DMA 1/band in ks0)", A, W, sink);
  run2<1, 0, 0>("A fragment 1 band ahead", A, W, sink);
  run2<2, 0, 0>("A fragment 2 bands ahead", A, W, sink);
  run2<0, 1, 0>("DMA pieces all before the MFMAs", A, W, sink);
  run2<1, 1, 0>("1 band ahead + DMA all before", A, W, sink);
  run2<1, 2, 0>("1 band ahead + DMA every other band", A, W, sink);
  run2<1, 0, 1>("1 band ahead + setprio around MFMAs", A, W, sink);
  run2<10, 0, 0>("all 10 A fragments of a K-half up front", A, W, sink);
  printf("---- K = 512 GEMM (K-sweeps 1) with an epilogue: [10240 x 2048] bf16 output = 42 MB\n");
  unsigned short* Y;
  hipMalloc(&Y, (size_t)N * O * 2);
  run2<0, 0, 0, 0>("no epilogue", A, W, sink, Y);
  run2<0, 0, 0, 1>("LDS transposition + bf16 conversion, NO stores", A, W, sink, Y);
  run2<0, 0, 0, 2>("LDS transposition + conversion + 16-byte row stores", A, W, sink, Y);
  run2<0, 0, 0, 3>("no LDS: conversion + stores in the accumulator layout", A, W, sink, Y);
  return 0;
}
