// Developer probe: what does it cost two CUs OF ONE XCD to swap a block of activations through that XCD's L2?  (The question behind
// a column-split of the layer chain over CU pairs: each CU would stream HALF of every weight - the chain is bound by the CU's L1
// path - and the pair would exchange panel halves three times per layer.)
//   workgroup b and b ^ 8 form a pair (both land on XCD b % 8); per iteration each side writes BYTES from registers to its slot,
//   raises a flag, waits for the partner's flag and reads the partner's slot back, checking every word.
//   MODE 0: memory-model clean - release store / acquire load at agent scope (the compiler's fences: buffer_wbl2 / buffer_inv)
//   MODE 1: same-XCD shortcut - plain stores, s_waitcnt, flag store; the reader polls and reads with sc1 loads (TCP bypass, L2 hit)
//   hipcc --offload-arch=gfx950 -O3 tools/pair_exchange_probe.hip -o tools/bin/pxp && tools/bin/pxp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ unsigned word_of(unsigned it, unsigned b, unsigned idx) { return (it * 2654435761u) ^ (b * 40503u + idx * 97u + 12345u); }

template <int MODE, int PER_THREAD /* 16-byte words per thread: 6 -> 48 KiB per side, 12 -> 96 KiB */>
__global__ __launch_bounds__(512, 1) void probe(unsigned* xbuf, unsigned* flags, unsigned long long* stats, unsigned* errs, unsigned* xcc, int reps) {
  extern __shared__ unsigned char smem[];
  const unsigned b = blockIdx.x, p = b ^ 8u, tid = threadIdx.x;
  constexpr unsigned kWords = 512u * PER_THREAD * 4u;  // dwords per slot
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (tid == 0) xcc[b] = id & 15u;
  unsigned long long t_write = 0, t_wait = 0, t_read = 0;
  unsigned bad = 0, timeouts = 0;
  unsigned* const mine = xbuf + (size_t)b * 2 * kWords;
  unsigned* const theirs = xbuf + (size_t)p * 2 * kWords;
  for (int it = 1; it <= reps; ++it) {
    unsigned* dst = mine + (it & 1) * kWords;
    const unsigned* src = theirs + (it & 1) * kWords;
    u32x4 v[PER_THREAD];
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      const unsigned idx = (j * 512 + tid) * 4;
      v[j] = u32x4{word_of(it, b, idx), word_of(it, b, idx + 1), word_of(it, b, idx + 2), word_of(it, b, idx + 3)};
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) *reinterpret_cast<u32x4*>(dst + (j * 512 + tid) * 4) = v[j];
    if constexpr (MODE == 0) {
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + b * 32, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + b * 32, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned spins = 0;
    if constexpr (MODE == 0) {
      while (__hip_atomic_load(flags + p * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(1);
    } else {
      while (__hip_atomic_load(flags + p * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(1);
    }
    if (spins >= (1u << 18)) { ++timeouts; break; }
    const unsigned long long t2 = __builtin_readcyclecounter();
    u32x4 r[PER_THREAD];
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < PER_THREAD; ++j) r[j] = *reinterpret_cast<const u32x4*>(src + (j * 512 + tid) * 4);
    } else {
#pragma unroll
      for (int j = 0; j < PER_THREAD; ++j) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[j]) : "v"(src + (j * 512 + tid) * 4) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < PER_THREAD; ++j) asm volatile("" : "+v"(r[j]));
    }
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      const unsigned idx = (j * 512 + tid) * 4;
      bad += (r[j][0] != word_of(it, p, idx)) + (r[j][1] != word_of(it, p, idx + 1)) + (r[j][2] != word_of(it, p, idx + 2)) + (r[j][3] != word_of(it, p, idx + 3));
    }
    const unsigned long long t3 = __builtin_readcyclecounter();
    t_write += t1 - t0; t_wait += t2 - t1; t_read += t3 - t2;
  }
  if (bad) atomicAdd(errs, bad);
  if (timeouts) atomicAdd(errs + 1, timeouts);
  if (tid == 0) { stats[b * 3] = t_write; stats[b * 3 + 1] = t_wait; stats[b * 3 + 2] = t_read; }
}

template <int MODE, int PT>
static void run(int grid, int reps) {
  unsigned *xbuf, *flags, *errs, *xcc; unsigned long long* stats;
  const size_t slot = 512ull * PT * 16;
  hipMalloc(&xbuf, (size_t)grid * 2 * slot); hipMalloc(&flags, grid * 32 * 4); hipMalloc(&errs, 8); hipMalloc(&xcc, grid * 4); hipMalloc(&stats, grid * 24);
  hipMemset(xbuf, 0, (size_t)grid * 2 * slot); hipMemset(flags, 0, grid * 32 * 4); hipMemset(errs, 0, 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, PT>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, PT>), dim3(grid), dim3(512), 100 * 1024, 0, xbuf, flags, stats, errs, xcc, reps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> st(grid * 3); std::vector<unsigned> xc(grid); unsigned er[2];
  hipMemcpy(st.data(), stats, grid * 24, hipMemcpyDeviceToHost); hipMemcpy(xc.data(), xcc, grid * 4, hipMemcpyDeviceToHost); hipMemcpy(er, errs, 8, hipMemcpyDeviceToHost);
  double w = 0, wt = 0, rd = 0; int mism = 0;
  for (int b = 0; b < grid; ++b) { w += st[b * 3]; wt += st[b * 3 + 1]; rd += st[b * 3 + 2]; mism += xc[b] != xc[b ^ 8]; }
  const double n = (double)grid * reps;
  printf("mode %d  %3zu KiB/side  grid %d: %.2f us per exchange (wall); cycles write+flag %.0f  wait %.0f  read+check %.0f; wrong words %u, timeouts %u, pairs on different XCDs %d, xcc of wg0..15:",
         MODE, slot / 1024, grid, ms * 1e3 / reps, w / n, wt / n, rd / n, er[0], er[1], mism / 2);
  for (int b = 0; b < 16 && b < grid; ++b) printf(" %u", xc[b]);
  printf("\n");
  hipFree(xbuf); hipFree(flags); hipFree(errs); hipFree(xcc); hipFree(stats);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 6>(256, reps); run<1, 6>(256, reps);
    run<0, 12>(256, reps); run<1, 12>(256, reps);
    run<1, 6>(208, reps);
  }
  return 0;
}
