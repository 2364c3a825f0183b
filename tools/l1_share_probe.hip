// Developer probe: is a CU's weight-stream rate bound by the bytes DELIVERED to its waves or by the bytes FILLED into its L1?
// 8 waves per CU stream a [segments x 64 KiB] image from L2 with 1-KiB global_load_dwordx4, 8 loads in flight per wave.
//   SHARE 0: every wave its own bytes (8 distinct streams: delivered = filled)
//   SHARE 1: wave w and wave w + 4 (same SIMD) read the SAME bytes at the same time (delivered = 2 x filled)
//   SHARE 2: same bytes, the second wave one 8-KiB batch behind the first
// If SHARE 1 moves the same DELIVERED bytes per microsecond as SHARE 0, two row panels per weight pass cost twice the stream; if it
// is ~2x faster per delivered byte, L1 hits are free and a second panel rides along.
//   hipcc --offload-arch=gfx950 -O3 tools/l1_share_probe.hip -o tools/bin/l1p && tools/bin/l1p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int SHARE>
__global__ __launch_bounds__(512, 1) void probe(const char* __restrict__ W, size_t image_bytes, int batches, unsigned* sink) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = SHARE ? (wave & 3) : wave;
  const int n_streams = SHARE ? 4 : 8;
  // stream s reads batches s, s + n_streams, ... (8 KiB each), wrapping inside the image
  u32x4 acc = {0, 0, 0, 0};
  u32x4 r[8];
  size_t off = (size_t)stream * 8192;
  if (SHARE == 2 && wave >= 4) off = (off + image_bytes - (size_t)n_streams * 8192) % image_bytes;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const u32x4*>(W + off + j * 1024 + lane * 16);
  for (int b = 1; b < batches; ++b) {
    off += (size_t)n_streams * 8192;
    if (off >= image_bytes) off -= image_bytes;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc ^= r[j];
      r[j] = *reinterpret_cast<const u32x4*>(W + off + j * 1024 + lane * 16);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) acc ^= r[j];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[threadIdx.x] = acc[0];
}

template <int SHARE>
static void run(const char* W, size_t image, int batches, unsigned* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<SHARE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHARE>), dim3(256), dim3(512), 100 * 1024, 0, W, image, batches, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double delivered = 8.0 * batches * 8192;  // per CU
  printf("share %d  image %.1f MiB: %.1f us, delivered %.2f MB per CU = %.1f GB/s per CU (filled %.1f GB/s)\n", SHARE, image / 1048576.0, best * 1e3,
         delivered / 1e6, delivered / (best * 1e-3) / 1e9, delivered / (SHARE ? 2 : 1) / (best * 1e-3) / 1e9);
}

int main() {
  char* W; unsigned* sink;
  const size_t cap = 16u << 20;
  hipMalloc(&W, cap); hipMalloc(&sink, 4096);
  hipMemset(W, 1, cap);
  for (size_t image : {size_t(2) << 20, size_t(6656) << 10}) {  // 2 MiB (well inside every L2) and the layer's 6.5 MiB
    for (int rep = 0; rep < 2; ++rep) {
      run<0>(W, image, 100, sink);
      run<1>(W, image, 100, sink);
      run<2>(W, image, 100, sink);
    }
  }
  return 0;
}
