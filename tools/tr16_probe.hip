// Developer probe: semantics of ds_read_b64_tr_b16 (LDS transpose read) on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
// Test 1: lane l reads at byte l*8 of a contiguous image            -> expect elem j = (l&15) + 16 j + 64 (l>>4).
// Test 2: 16-lane group g, lane i reads at g*512 + (i>>2)*128 + (i&3)*8 (rows of a 4x16 block at a 128-B stride)
//                                                                   -> expect elem j = g*256 + 64 j + i.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__global__ void probe(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)(unsigned)(size_t)(lds + l * 4));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)(unsigned)(size_t)(lds + g * 256 + (i >> 2) * 64 + (i & 3) * 4));
  for (int j = 0; j < 4; ++j) { out[l * 4 + j] = a[j]; out[256 + l * 4 + j] = b[j]; }
}

int main() {
  unsigned short *d, h[512];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      bad1 += h[l * 4 + j] != (l & 15) + 16 * j + 64 * (l >> 4);
      bad2 += h[256 + l * 4 + j] != (l >> 4) * 256 + 64 * j + (l & 15);
    }
  printf("test1 mismatches %d, test2 mismatches %d\n", bad1, bad2);
  for (int l = 0; l < 20; ++l) printf("lane %2d: %4d %4d %4d %4d | %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], h[256+l*4], h[256+l*4+1], h[256+l*4+2], h[256+l*4+3]);
  return 0;
}
