// All-reduce / all-gather for tensor-parallel decode over the xGMI mesh, written for MI355X.
//
// An 8-GPU MI355X node is a full mesh: every GPU has a direct xGMI link to each of its 7 peers. A
// ring (what a generic collective library runs) sends everything over one link per hop and pays
// 2(N-1) hops of latency; on a mesh every rank can instead read its peers' buffers directly, all
// links at once. Tensor-parallel decode all-reduces [batch, hidden] bf16 twice per layer, 16 KiB to
// a few MiB each: latency, not bandwidth. Two kernels:
//
//   one-shot  (small):  copy x into my buffer | flag barrier | out = sum over ranks of their buffers
//                       one barrier, (N-1) x bytes pulled per rank, every link busy.
//   two-shot  (large):  copy x in | barrier | reduce MY 1/N of the vector from all peers into my
//                       result area | barrier | gather the other N-1 reduced parts from their owners.
//                       2 (N-1)/N x bytes per rank: the bandwidth-optimal schedule on a full mesh.
//
// Buffers are allocated here (hipExtMallocWithFlags, uncached = fine-grained: coherent between
// devices inside a running kernel) and shared between the ranks' processes with hipIpc handles
// (parallel/xgmi.py exchanges them). Synchronisation is by flags in those buffers:
//   * block b of a rank only ever depends on block b of its peers (same partition of the data), so
//     the barrier is per block index: thread t < N stores the call's epoch into peer t's flag
//     [stage][b][my rank] and spins on its own flag [stage][b][t];
//   * the epoch is a per-block counter in local device memory, advanced by the kernel itself:
//     nothing in the launch arguments changes between calls, so the launch can sit in a hipGraph;
//   * two data slots alternate by epoch parity. A rank can run at most one call ahead of a peer
//     (it needs every peer's flag to pass a barrier), so writing slot (e+1)&1 never touches what a
//     peer still reads for call e, and passing the first barrier of call e+1 proves call e is over
//     everywhere before call e+2 reuses its slot. No trailing barrier is needed.
//   * every spin is bounded (a limit in seconds of the 100 MHz wall clock): a missing peer sets a status
//     word and the kernel ends with a wrong result instead of hanging the GPU. The epochs keep advancing, so
//     the protocol is in step again as soon as the late rank arrives. While the status word is set (nobody has
//     read and cleared it yet) later barriers wait for 1/16384 of the limit only: a dead peer costs one limit,
//     not one per collective of a 126-layer step;
//   * beside its epoch every rank publishes a TAG of the call (bytes and mode) in a third flag region, double
//     buffered by epoch parity like the data; a rank that passes the first barrier compares its peers' tags with
//     its own and sets status bit 2 on a difference: ranks whose schedulers have drifted apart (one launches
//     [T, hidden] for another T, or a gather against a sum) are caught instead of summing stale slot bytes.
// Sums run in rank order in fp32 on every rank, so all ranks hold bit-identical results.
//
// No reference counterpart (the reference has no GPU code).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

// the hash of the sources this binary was built from (build.py passes KLM_SRC_HASH and finds the marker to decide staleness)
#ifndef KLM_SRC_HASH
#define KLM_SRC_HASH "unhashed"
#endif
extern "C" __attribute__((used)) const char klm_src_hash_marker[] = "KLM_SRC_HASH=" KLM_SRC_HASH;

extern "C" {
int klm_xgmi_alloc(size_t bytes, void** ptr, void* handle64, int* fine_grained);
int klm_xgmi_open(const void* handle64, void** ptr);
int klm_xgmi_close(void* ptr);
int klm_xgmi_free(void* ptr);
int klm_xgmi_collective(void* const* data, const void* sig_table, void* my_sig, int rank, int world, const void* x,
                        void* out, size_t nbytes, int mode, int blocks, void* counters, void* status,
                        size_t slot_bytes, double spin_limit_s, void* stream);
int klm_xgmi_limits(int* max_ranks, int* max_blocks, int* sig_bytes, int* counter_bytes);
}

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // 16 bytes in registers (a plain vector: arrays of it stay out of scratch)

constexpr int MAXR = 8;            // ranks
constexpr int MAXB = 128;          // blocks per call
constexpr int NT = 512;            // threads per block

enum Mode { SUM_ONE_SHOT = 0, SUM_TWO_SHOT = 1, GATHER = 2 };

// Every rank's buffers as mapped in this process. Only indexed with compile-time constants in the kernels
// (a run-time index into a by-value kernel argument sends the struct through scratch memory).
struct Peers {
  char* data[MAXR];
  uint32_t* const* sig;      // device array of the ranks' flag areas (indexed per lane)
  char* my_data;
  uint32_t* my_sig;
};

__device__ __forceinline__ float lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}

struct Acc8 {
  float v[8];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  __device__ __forceinline__ void add(const u32x4& a) {
    v[0] += lo(a.x); v[1] += hi(a.x); v[2] += lo(a.y); v[3] += hi(a.y);
    v[4] += lo(a.z); v[5] += hi(a.z); v[6] += lo(a.w); v[7] += hi(a.w);
  }
  __device__ __forceinline__ u32x4 pack() const {
    return u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
  }
};

// Barrier between block b of every rank. All of this block's earlier stores are made visible to the
// system before the flags go out; all of its later loads see what the peers stored before theirs.
constexpr int ST_TAG_MISMATCH = 4;     // status bits: 1 << stage (1, 2) = a peer did not arrive; 4 = a peer ran a different call

// `tag` != 0: this is the call's first barrier; publish the tag with the flag and compare the peers' with it.
template <int N>
__device__ __forceinline__ void block_barrier(const Peers& p, int rank, int stage, uint32_t epoch, int* status, long long spin_limit,
                                              uint32_t tag = 0) {
  __threadfence_system();
  __syncthreads();
  const int t = threadIdx.x;
  if (t < N) {
    const int slot = (stage * MAXB + blockIdx.x) * MAXR;
    const int tslot = ((2 + (epoch & 1)) * MAXB + blockIdx.x) * MAXR;
    if (tag) __hip_atomic_store(p.sig[t] + tslot + rank, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // ordered by the release below
    __hip_atomic_store(p.sig[t] + slot + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = p.my_sig + slot + t;
    // an earlier barrier already gave up and nobody has cleared the word: do not pay the whole limit again
    const long long limit = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? (spin_limit >> 14) + 1 : spin_limit;
    const long long t0 = wall_clock64();
    bool arrived = true;
    while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      if (wall_clock64() - t0 > limit) {
        __hip_atomic_fetch_or(status, 1 << stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrived = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (tag && arrived && __hip_atomic_load(p.my_sig + tslot + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag)
      __hip_atomic_fetch_or(status, ST_TAG_MISMATCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  __threadfence_system();
}

// what a call is, in one nonzero word: its size in 16-byte units and its mode
__device__ __forceinline__ uint32_t call_tag(size_t n16, int mode) { return ((uint32_t)n16 << 2 | (uint32_t)mode) + 1u; }

// n16: 16-byte units of the vector. Block b owns the units i with (i / NT) % gridDim.x == b.
template <int N, int MODE>
__global__ __launch_bounds__(NT) void xgmi_one_shot(Peers p, int rank, const u32x4* x, u32x4* out,
                                                    size_t n16, uint32_t* counters, int* status, size_t slot_bytes, long long spin_limit) {
  const uint32_t epoch = counters[blockIdx.x] + 1;
  const size_t off = (epoch & 1) * slot_bytes;
  const size_t first = (size_t)blockIdx.x * NT + threadIdx.x, step = (size_t)gridDim.x * NT;
  u32x4* mine = reinterpret_cast<u32x4*>(p.my_data + off);
  for (size_t i = first; i < n16; i += step) mine[i] = x[i];
  block_barrier<N>(p, rank, 0, epoch, status, spin_limit, call_tag(n16, MODE));
  for (size_t i = first; i < n16; i += step) {
    u32x4 v[N];
#pragma unroll
    for (int r = 0; r < N; ++r) v[r] = reinterpret_cast<const u32x4*>(p.data[r] + off)[i];
    if (MODE == GATHER) {
#pragma unroll
      for (int r = 0; r < N; ++r) out[(size_t)r * n16 + i] = v[r];
    } else {
      Acc8 a;
      a.zero();
#pragma unroll
      for (int r = 0; r < N; ++r) a.add(v[r]);
      out[i] = a.pack();
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) counters[blockIdx.x] = epoch;
}

// Slot layout: [input: slot_bytes/2][reduced part of this rank: slot_bytes/2]. part16 = n16 / N.
template <int N>
__global__ __launch_bounds__(NT) void xgmi_two_shot(Peers p, int rank, const u32x4* x, u32x4* out,
                                                    size_t part16, uint32_t* counters, int* status, size_t slot_bytes, long long spin_limit) {
  const uint32_t epoch = counters[blockIdx.x] + 1;
  const size_t off = (epoch & 1) * slot_bytes, res = off + slot_bytes / 2;
  const size_t first = (size_t)blockIdx.x * NT + threadIdx.x, step = (size_t)gridDim.x * NT;
  u32x4* mine = reinterpret_cast<u32x4*>(p.my_data + off);
  for (int q = 0; q < N; ++q)
    for (size_t j = first; j < part16; j += step) mine[q * part16 + j] = x[q * part16 + j];
  block_barrier<N>(p, rank, 0, epoch, status, spin_limit, call_tag(part16 * N, SUM_TWO_SHOT));
  u32x4* my_res = reinterpret_cast<u32x4*>(p.my_data + res);
  for (size_t j = first; j < part16; j += step) {
    u32x4 v[N];
#pragma unroll
    for (int r = 0; r < N; ++r) v[r] = reinterpret_cast<const u32x4*>(p.data[r] + off)[rank * part16 + j];
    Acc8 a;
    a.zero();
#pragma unroll
    for (int r = 0; r < N; ++r) a.add(v[r]);
    const u32x4 s = a.pack();
    my_res[j] = s;
    out[rank * part16 + j] = s;
  }
  block_barrier<N>(p, rank, 1, epoch, status, spin_limit);
  for (size_t j = first; j < part16; j += step) {
    u32x4 v[N];
#pragma unroll
    for (int q = 0; q < N; ++q)               // one load per peer in flight: all links in use
      if (q != rank) v[q] = reinterpret_cast<const u32x4*>(p.data[q] + res)[j];
#pragma unroll
    for (int q = 0; q < N; ++q)
      if (q != rank) out[q * part16 + j] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) counters[blockIdx.x] = epoch;
}

template <int N>
int launch(const Peers& p, int rank, const void* x, void* out, size_t nbytes, int mode, int blocks, void* counters, void* status,
           size_t slot_bytes, long long spin, hipStream_t s) {
  const size_t n16 = nbytes / 16;
  auto* xi = static_cast<const u32x4*>(x);
  auto* oi = static_cast<u32x4*>(out);
  auto* c = static_cast<uint32_t*>(counters);
  auto* st = static_cast<int*>(status);
  if (mode == SUM_ONE_SHOT)
    hipLaunchKernelGGL((xgmi_one_shot<N, SUM_ONE_SHOT>), dim3(blocks), dim3(NT), 0, s, p, rank, xi, oi, n16, c, st, slot_bytes, spin);
  else if (mode == GATHER)
    hipLaunchKernelGGL((xgmi_one_shot<N, GATHER>), dim3(blocks), dim3(NT), 0, s, p, rank, xi, oi, n16, c, st, slot_bytes, spin);
  else
    hipLaunchKernelGGL((xgmi_two_shot<N>), dim3(blocks), dim3(NT), 0, s, p, rank, xi, oi, n16 / N, c, st, slot_bytes, spin);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

int klm_xgmi_limits(int* max_ranks, int* max_blocks, int* sig_bytes, int* counter_bytes) {
  *max_ranks = MAXR;
  *max_blocks = MAXB;
  *sig_bytes = 4 * MAXB * MAXR * (int)sizeof(uint32_t);      // two barrier stages + two tag regions (epoch parity)
  *counter_bytes = MAXB * (int)sizeof(uint32_t);
  return 0;
}

// Zeroed device memory another process can map, and its 64-byte IPC handle. Fine-grained (uncached) when the
// runtime grants it, else ordinary device memory (fine_grained = 0; still correct between processes on one GPU).
int klm_xgmi_alloc(size_t bytes, void** ptr, void* handle64, int* fine_grained) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  *fine_grained = 1;
  if (hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    *fine_grained = 0;
    if (hipMalloc(ptr, bytes) != hipSuccess) return -2;
  }
  if (hipMemset(*ptr, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -2;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, *ptr) != hipSuccess) {
    (void)hipGetLastError();
    if (!*fine_grained) return -3;
    (void)hipFree(*ptr);                      // some runtimes only export plain allocations
    *fine_grained = 0;
    if (hipMalloc(ptr, bytes) != hipSuccess) return -2;
    if (hipMemset(*ptr, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipIpcGetMemHandle(&h, *ptr) != hipSuccess) return -3;
  }
  memcpy(handle64, &h, 64);
  return 0;
}

int klm_xgmi_open(const void* handle64, void** ptr) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  return hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? 0 : -3;
}

int klm_xgmi_close(void* ptr) { return hipIpcCloseMemHandle(ptr) == hipSuccess ? 0 : -3; }
int klm_xgmi_free(void* ptr) { return hipFree(ptr) == hipSuccess ? 0 : -2; }

// data[r]: rank r's data buffer as mapped in THIS process (own buffer at index `rank`); sig_table: DEVICE array of
// the ranks' flag areas, likewise mapped here; my_sig = its entry `rank`.
// mode 0: sum, one shot (nbytes <= slot_bytes). 1: sum, two shot (nbytes <= slot_bytes / 2, nbytes % (16 world) == 0).
// 2: all-gather, out holds world x nbytes (nbytes <= slot_bytes). x and out are 16-byte aligned; x may equal out for sums.
int klm_xgmi_collective(void* const* data, const void* sig_table, void* my_sig, int rank, int world, const void* x, void* out,
                        size_t nbytes, int mode, int blocks, void* counters, void* status, size_t slot_bytes, double spin_limit_s,
                        void* stream) {
  if (world < 1 || world > MAXR || rank < 0 || rank >= world || blocks < 1 || blocks > MAXB) return -1;
  if (!x || !out || !counters || !status || !sig_table || !my_sig || nbytes == 0 || nbytes % 16) return -1;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) return -1;
  if (mode == SUM_TWO_SHOT ? (nbytes > slot_bytes / 2 || nbytes % (16 * (size_t)world)) : nbytes > slot_bytes) return -1;
  if (mode != SUM_ONE_SHOT && mode != SUM_TWO_SHOT && mode != GATHER) return -1;
  Peers p;
  for (int r = 0; r < MAXR; ++r) {
    p.data[r] = r < world ? static_cast<char*>(data[r]) : nullptr;
    if (r < world && !p.data[r]) return -1;
  }
  p.sig = static_cast<uint32_t* const*>(sig_table);
  p.my_data = p.data[rank];
  p.my_sig = static_cast<uint32_t*>(my_sig);
  auto s = static_cast<hipStream_t>(stream);
  const long long spin = (long long)((spin_limit_s > 0 ? spin_limit_s : 2.0) * 1e8);    // wall_clock64 runs at 100 MHz
  switch (world) {
    case 1: return launch<1>(p, rank, x, out, nbytes, mode, blocks, counters, status, slot_bytes, spin, s);
    case 2: return launch<2>(p, rank, x, out, nbytes, mode, blocks, counters, status, slot_bytes, spin, s);
    case 4: return launch<4>(p, rank, x, out, nbytes, mode, blocks, counters, status, slot_bytes, spin, s);
    case 8: return launch<8>(p, rank, x, out, nbytes, mode, blocks, counters, status, slot_bytes, spin, s);
    default: return -1;
  }
}
