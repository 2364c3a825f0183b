// Developer probe: what bounds the K-loop of the 320 x 256 big-tile GEMM (csrc/linear.hip)?  Every CU streams the operand
// tiles of a [10240 x 512] x [512 -> 2048] GEMM through the 2-stage LDS ring by global_load_lds_dwordx4, exactly as the kernel
// does (72 pieces of 1 KiB per K-step, 9 per wave), with NO fragment reads and optionally the K-step's 80 MFMAs per wave on
// register operands.  Variants: operand rows at their natural 1-KiB stride (a piece = 8 rows x 128 B) vs. K-tile-major packed
// operands (a piece = 1 contiguous KiB); default vs. non-temporal cache policy; 2 x 72 KiB stages vs. 4 x 36 KiB (BK = 32).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate_probe.hip -o /tmp/dmap && /tmp/dmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
  return 0;
}
