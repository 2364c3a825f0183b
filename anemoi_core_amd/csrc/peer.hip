// Device-initiated exchange of node rows between the ranks of a model-parallel group (scope row e).
//
// Replaces, on the inference path, the per-layer halo all-to-all of the reference (distributed/primitives.py:422-460, call site
// layers/block.py:1159-1172), the needed-rows exchange of the mappers and the output all-gather (primitives.py:60-183): instead
// of a host-issued RCCL collective between two hipGraph segments, ONE kernel of the rank's own stream
//   1. stores the rows a peer needs straight into that peer's receive buffer (peer memory mapped through hipIpc: on one
//      node that is a store over xGMI, point to point, every link carrying only its own rows),
//   2. publishes them: system-scope release, then a flag (the channel's epoch) in the peer's flag block,
//   3. waits until the flags of all peers it receives from show the same epoch.
// The kernel is an ordinary graph node: a rank's whole sharded forward is one hipGraph, no host in the loop.
//
// Memory: two allocations per rank, exported once (anemoi_peer_export) and opened by every peer (anemoi_peer_open):
//   payload arena  - receive buffers; plain device memory by default (as RCCL writes into user buffers), consumed by LATER
//                    kernels of the receiving stream (kernel boundary = acquire);
//   flag block     - uint32 words, uncached fine-grained memory (polled while a peer writes).
// Protocol per channel (one per call site of a forward, created collectively): words [flags[P] | seq | ticket].
//   epoch = seq + 1 (read by every workgroup at entry; written back by the last workgroup at exit).  Rows are copied by all
//   workgroups, each then releases at system scope and takes a ticket; the LAST workgroup signals and waits.  Flags only ever
//   grow (compared with wrap-around), so nothing is reset between forwards; a receive buffer is reused once per forward, and
//   the forward-level barrier (anemoi_peer_exchange_rows with no rows, all peers signalled / expected) keeps a fast rank from
//   overwriting rows a slow rank has not consumed yet.
// A bounded spin (timeout in wall-clock ticks) turns a lost peer into an error word instead of a hung GPU.
#include <string.h>

#include "common.h"

namespace anemoi {

namespace {

constexpr int kPeerThreads = 256;

// table[6][P] (int64, device memory), built once per channel by the host side:
//   [0] remote_base   address (in THIS process) of the first byte this rank's rows occupy in peer p's receive buffer
//   [1] remote_flag   address of the flag word peer p polls for this rank
//   [2] send_begin    first row of the packed send order that goes to peer p
//   [3] send_count    number of rows for peer p
//   [4] signal        1: write the epoch to remote_flag[p] (rows were sent, or barrier)
//   [5] expect        1: wait for peer p's epoch in local_flags[p]
struct PeerArgs {
  const char* src;
  int64_t ld_src;             // bytes between source rows
  const int32_t* send_index;  // source row of packed row i (nullptr: i)
  const int64_t* table;
  int32_t P;
  int32_t row_chunks;  // 16-byte chunks per row
  int32_t total_rows;
  uint32_t* local_flags;  // [P] then seq, ticket
  uint32_t* status;
  uint64_t timeout_ticks;
  int32_t write_through;  // 1: rows leave with system-scope write-through stores (the release then has nothing to write back)
};

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(kPeerThreads) void peer_exchange_kernel(PeerArgs a) {
  __shared__ uint32_t s_epoch;
  __shared__ int s_last;
  uint32_t* seq = a.local_flags + a.P;
  uint32_t* ticket = a.local_flags + a.P + 1;
  if (threadIdx.x == 0) s_epoch = __hip_atomic_load(seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int64_t* remote_base = a.table;
  const int64_t* send_begin = a.table + 2 * a.P;
  const int64_t* send_count = a.table + 3 * a.P;
  const int64_t row_bytes = (int64_t)a.row_chunks * 16;
  const int64_t total = (int64_t)a.total_rows * a.row_chunks;
  for (int64_t c = (int64_t)blockIdx.x * kPeerThreads + threadIdx.x; c < total; c += (int64_t)gridDim.x * kPeerThreads) {
    const int row = (int)(c / a.row_chunks), col = (int)(c - (int64_t)row * a.row_chunks);
    int p = 0;
    while (row >= send_begin[p] + send_count[p]) ++p;  // P is small; begins ascend, empty peers are skipped by the test
    const int src_row = a.send_index ? a.send_index[row] : row;
    const uint4 v = *reinterpret_cast<const uint4*>(a.src + (int64_t)src_row * a.ld_src + (int64_t)col * 16);
    char* dstp = reinterpret_cast<char*>(remote_base[p]) + (row - send_begin[p]) * row_bytes + (int64_t)col * 16;
    if (a.write_through) {
      // 16-byte system-scope write-through store: the bytes go straight to their home (the peer's HBM over xGMI), nothing
      // stays dirty in this XCD's L2, so publishing needs a drained vmcnt instead of a buffer_wbl2 of the whole L2 (guide:
      // "sc1 payload -> vmcnt(0) -> sc1 flag" costs 1.7-1.9x a bare hand-off, the write-back form 2-2.5x and more under load)
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dstp), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
    } else {
      *reinterpret_cast<uint4*>(dstp) = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // publish this workgroup's stores at system scope BEFORE its ticket can be seen (guide: fence, then a drained vmcnt the
    // compiler cannot drop, then the flag / ticket)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!s_last) return;
  // ---- last workgroup: every workgroup's rows are released; signal the peers, wait for theirs
  const uint32_t epoch = s_epoch;
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other workgroups' releases, through their tickets
  const uint64_t t_released = wall_clock64();
  __syncthreads();
  if ((int)threadIdx.x < a.P) {
    const int p = threadIdx.x;
    const int64_t* remote_flag = a.table + a.P;
    const int64_t* signal = a.table + 4 * a.P;
    const int64_t* expect = a.table + 5 * a.P;
    if (signal[p]) __hip_atomic_store(reinterpret_cast<uint32_t*>(remote_flag[p]), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // fail fast: once a wait of this rank has timed out (status != 0) later exchanges do not wait again - a broken wire costs
    // ONE time-out, after which the host finds the status word (PeerWire.check) and falls back to the RCCL path
    if (expect[p] && ld_sys(a.status) == 0) {
      const uint64_t t0 = wall_clock64();
      while ((int32_t)(ld_sys(a.local_flags + p) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > a.timeout_ticks) {
          __hip_atomic_store(a.status, 0x80000000u | (uint32_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // pairs with the peers' releases; the rows themselves are read by LATER kernels
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(seq, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // diagnostics in the status block (words 2-4; exchanges of a stream are serialised, one writer): how many exchanges ran, the
    // 100 MHz ticks between "all my rows released" and "every expected peer's flag seen" summed over them, and the longest one -
    // what a first run on real xGMI links needs next to the time-out word (bench.py prints them per rank)
    const uint32_t dt = (uint32_t)(wall_clock64() - t_released);
    a.status[2] += 1u;
    a.status[3] += dt;
    if (dt > a.status[4]) a.status[4] = dt;
  }
}

}  // namespace
}  // namespace anemoi

using namespace anemoi;

static int hip_fail(const char* what, hipError_t e) {
  set_error("%s: %s", what, hipGetErrorString(e));
  (void)hipGetLastError();
  return ANEMOI_E_LAUNCH;
}

extern "C" int anemoi_peer_alloc(void** ptr, int64_t bytes, int32_t kind) {
  ANEMOI_REQUIRE(ptr != nullptr && bytes > 0 && kind >= 0 && kind <= 2, "peer_alloc: bad arguments (bytes=%lld kind=%d)", (long long)bytes, kind);
  hipError_t e;
  if (kind == ANEMOI_PEER_MEM_DEFAULT)
    e = hipMalloc(ptr, (size_t)bytes);
  else
    e = hipExtMallocWithFlags(ptr, (size_t)bytes, kind == ANEMOI_PEER_MEM_FINEGRAINED ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
  if (e != hipSuccess) return hip_fail("peer_alloc", e);
  e = hipMemset(*ptr, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) return hip_fail("peer_alloc: memset", e);
  return ANEMOI_OK;
}

extern "C" int anemoi_peer_free(void* ptr) {
  if (ptr == nullptr) return ANEMOI_OK;
  const hipError_t e = hipFree(ptr);
  return e == hipSuccess ? ANEMOI_OK : hip_fail("peer_free", e);
}

extern "C" int anemoi_peer_export(void* ptr, void* handle_out) {
  static_assert(sizeof(hipIpcMemHandle_t) == ANEMOI_PEER_HANDLE_BYTES, "handle size");
  ANEMOI_REQUIRE(ptr != nullptr && handle_out != nullptr, "peer_export: null argument");
  const hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), ptr);
  return e == hipSuccess ? ANEMOI_OK : hip_fail("peer_export (hipIpcGetMemHandle; is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", e);
}

extern "C" int anemoi_peer_open(const void* handle, void** ptr_out) {
  ANEMOI_REQUIRE(handle != nullptr && ptr_out != nullptr, "peer_open: null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  const hipError_t e = hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess);
  return e == hipSuccess ? ANEMOI_OK : hip_fail("peer_open (hipIpcOpenMemHandle)", e);
}

extern "C" int anemoi_peer_close(void* ptr) {
  if (ptr == nullptr) return ANEMOI_OK;
  const hipError_t e = hipIpcCloseMemHandle(ptr);
  return e == hipSuccess ? ANEMOI_OK : hip_fail("peer_close", e);
}

extern "C" int anemoi_peer_exchange_rows(const void* src, int64_t ld_src_bytes, const int32_t* send_index, const int64_t* table,
                                         int32_t n_peers, int32_t row_bytes, int32_t total_rows, uint32_t* local_flags,
                                         uint32_t* status, int64_t timeout_ticks, void* stream) {
  ANEMOI_REQUIRE(table != nullptr && local_flags != nullptr && status != nullptr && n_peers >= 1 && n_peers <= kPeerThreads,
                 "peer_exchange_rows: bad arguments (n_peers=%d)", n_peers);
  ANEMOI_REQUIRE(total_rows >= 0 && (total_rows == 0 || (src != nullptr && row_bytes > 0 && row_bytes % 16 == 0 && ld_src_bytes % 16 == 0 &&
                                                         (reinterpret_cast<uintptr_t>(src) & 15) == 0)),
                 "peer_exchange_rows: rows must be 16-byte multiples at 16-byte aligned addresses (row_bytes=%d ld=%lld)", row_bytes,
                 (long long)ld_src_bytes);
  static const int write_through = env_int(getenv("ANEMOI_PEER_WRITE_THROUGH"), 0, 0, 1);
  PeerArgs a{static_cast<const char*>(src), ld_src_bytes, send_index, table, n_peers, row_bytes / 16, total_rows, local_flags, status,
             (uint64_t)(timeout_ticks > 0 ? timeout_ticks : 0), write_through};
  // a few workgroups move <= 1 MB faster than one (stores in flight over the links); a barrier is one workgroup
  const int64_t chunks = (int64_t)total_rows * (row_bytes / 16);
  int grid = (int)((chunks + 4 * kPeerThreads - 1) / (4 * kPeerThreads));
  grid = grid < 1 ? 1 : (grid > 64 ? 64 : grid);
  hipLaunchKernelGGL(peer_exchange_kernel, dim3(grid), dim3(kPeerThreads), 0, as_stream(stream), a);
  return check_launch("peer_exchange_kernel");
}
