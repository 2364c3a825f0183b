// Developer probe: HBM write rate of the GEMM epilogue's store pattern (320x256 tiles, a wave instruction writes 8 rows x 128 B
// at a 4 KiB row stride) against a linear fill of the same 42 MB, and against 512-B row segments (4 rows x 512 B per instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern_probe.hip -o /tmp/spp && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 10240, O = 2048;

__global__ __launch_bounds__(512) void fill_linear(u32x4* out, long n16) {
  for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n16; i += (long)gridDim.x * 512) out[i] = u32x4{1, 2, 3, 4};
}
__global__ __launch_bounds__(512) void fill_tile128(unsigned short* out) {  // the epilogue's pattern
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3;
  const int m0 = (blockIdx.x / 8) * 320, n0 = (blockIdx.x % 8) * 256;
  for (int mi = 0; mi < 10; ++mi)
    for (int it = 0; it < 2; ++it) {
      const int row = m0 + wr * 160 + mi * 16 + it * 8 + (lane >> 3), col = n0 + wc * 64 + (lane & 7) * 8;
      *reinterpret_cast<u32x4*>(out + (long)row * O + col) = u32x4{1, 2, 3, 4};
    }
}
__global__ __launch_bounds__(512) void fill_tile512(unsigned short* out) {  // whole 512-B tile rows: 2 rows per wave instruction
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m0 = (blockIdx.x / 8) * 320, n0 = (blockIdx.x % 8) * 256;
  for (int i = 0; i < 20; ++i) {
    const int row = m0 + w * 40 + i * 2 + (lane >> 5), col = n0 + (lane & 31) * 8;
    *reinterpret_cast<u32x4*>(out + (long)row * O + col) = u32x4{1, 2, 3, 4};
  }
}
int main() {
  unsigned short* d;
  hipMalloc(&d, (size_t)N * O * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[3] = {"linear fill", "tile, 128-B row segments", "tile, 512-B row segments"};
  for (int k = 0; k < 3; ++k) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipEventRecord(e0, 0);
      if (k == 0) hipLaunchKernelGGL(fill_linear, dim3(256 * 4), dim3(512), 0, 0, (u32x4*)d, (long)N * O / 8);
      if (k == 1) hipLaunchKernelGGL(fill_tile128, dim3(256), dim3(512), 0, 0, d);
      if (k == 2) hipLaunchKernelGGL(fill_tile512, dim3(256), dim3(512), 0, 0, d);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 2 && ms < best) best = ms;
    }
    printf("%-28s %7.1f us  %5.2f TB/s\n", names[k], best * 1e3, (double)N * O * 2 / best * 1e-9);
  }
  return 0;
}
