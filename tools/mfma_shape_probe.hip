// Developer probe: MFMA throughput of the whole chip on REGISTER operands (no LDS, no memory), 8 waves per CU as in the GEMM
// kernels, for the two dense bf16 shapes of gfx950 - v_mfma_f32_16x16x32_bf16 (what csrc/linear.hip uses; 20 independent
// accumulators = 80 VGPRs) and v_mfma_f32_32x32x16_bf16 (5 accumulators = 80 VGPRs) - on constant and on pseudo-random operands.
// The question: is the K-loop's rate (1.29 PFLOP/s inside the loop on N(0,1) data) the matrix pipes' power-limited rate, and
// would the other instruction shape lift it?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o /tmp/mfmap && /tmp/mfmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ inline bf8 make_operand(uint32_t seed, bool random) {
  union { bf8 v; uint16_t u[8]; } r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t h = (seed * 2654435761u) ^ (i * 40503u + 12345u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    // random: sign + exponent around 1.0 + 7 random mantissa bits (values in +-[0.5, 2)); constant: 1.0
    r.u[i] = random ? (uint16_t)(((h & 1u) << 15) | ((126u + ((h >> 1) % 3u)) << 7) | ((h >> 8) & 0x7fu)) : (uint16_t)0x3f80;
  }
  return r.v;
}

// The same 16x16x32 loop with the GEMM's fragment traffic: LDSR ds_read_b128 per 20 MFMAs refresh the operands from a 64-KiB LDS
// image of pseudo-random bf16 (the 160 x 256 tile kernel reads 9 per 20 MFMAs, the 320 x 256 one 7).
template <int LDSR>
__global__ __launch_bounds__(512, 1) void probe_lds(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t t = blockIdx.x * 512 + threadIdx.x;
  for (int i = threadIdx.x; i < 65536 / 16; i += 512) reinterpret_cast<bf8*>(smem)[i] = make_operand(t * 131 + i, true);
  __syncthreads();
  bf8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = make_operand(t * 8 + i, true);
    b[i] = make_operand(t * 8 + 4 + i, true);
  }
  f32x4 acc[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t off = (threadIdx.x * 16) & 65535;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 20; ++i) {
      if (i < LDSR) {  // conflict-free: consecutive lanes read consecutive 16-byte slots
        const bf8 v = *reinterpret_cast<const bf8*>(smem + ((off + i * 8192) & 65535));
        if (i & 1) a[(i >> 1) & 3] = v; else b[(i >> 1) & 3] = v;
      }
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    off = (off + 1024) & 65535;
  }
  float out = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) out += acc[i][0] + acc[i][3];
  if (out == 123.456f) sink[t] = out;
}

template <int SHAPE>  // 16 or 32
__global__ __launch_bounds__(512, 1) void probe(int iters, int random, float* sink) {
  const uint32_t t = blockIdx.x * 512 + threadIdx.x;
  bf8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = make_operand(t * 8 + i, random);
    b[i] = make_operand(t * 8 + 4 + i, random);
  }
  float out = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 20; ++i) out += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)  // 10 MFMAs of 32768 flop = 20 of 16384
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + rep) & 3], b[(i + 2 * rep) & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) out += acc[i][0] + acc[i][15];
  }
  if (out == 123.456f) sink[t] = out;
}

int main() {
  float* sink;
  hipMalloc(&sink, 256 * 512 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;  // x 20 MFMAs (16x16x32) per wave: ~20 us at peak
  const double flop = 256.0 * 8 * iters * 20 * 16384.0;
  for (int shape : {16, 32})
    for (int random : {0, 1})
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) {
          if (shape == 16) hipLaunchKernelGGL(probe<16>, dim3(256), dim3(512), 0, 0, iters, random, sink);
          else hipLaunchKernelGGL(probe<32>, dim3(256), dim3(512), 0, 0, iters, random, sink);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("mfma %dx%d %s operands: %.1f us per launch, %.0f TFLOP/s\n", shape, shape, random ? "random  " : "constant", ms * 100.0, flop / (ms / 10 * 1e-3) / 1e12);
      }
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_lds<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_lds<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int ldsr : {0, 4, 9})
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int k = 0; k < 10; ++k) {
        if (ldsr == 0) hipLaunchKernelGGL(probe_lds<0>, dim3(256), dim3(512), 65536, 0, iters, sink);
        else if (ldsr == 4) hipLaunchKernelGGL(probe_lds<4>, dim3(256), dim3(512), 65536, 0, iters, sink);
        else hipLaunchKernelGGL(probe_lds<9>, dim3(256), dim3(512), 65536, 0, iters, sink);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("mfma 16x16 random operands + %d ds_read_b128 per 20 MFMAs: %.1f us per launch, %.0f TFLOP/s\n", ldsr, ms * 100.0, flop / (ms / 10 * 1e-3) / 1e12);
    }
  return 0;
}
