// wqaa_peer.hip - the M = 1 output exchange of the column-parallel operator without a collective: every rank stores its
// [1, N/P] slice straight into the [1, N] row of every peer (device memory mapped through hipIpc*) and posts a step number;
// the same launch waits for the peers' posts.  Replaces the small-message all-gather after a decode GEMV (tens of
// microseconds of RCCL latency against ~7 us of kernel).  The reference has no multi-GPU path
// (SURVEY.md section 5); the sharding is this library's (bitblas_amd/parallel.py), the operator per rank is the reference's
// (bitblas/ops/general_matmul/__init__.py:605-658 forward).
//
// Memory: the windows are allocated UNCACHED (hipDeviceMallocUncached) so that a store is visible to the owning device
// without waiting for a kernel boundary of the writer; data then post are ordered by a system-scope release, the wait takes
// system-scope acquires.  One workgroup per peer: slices are KBs.  Every spin is bounded (100 MHz realtime clock).
#include "wqaa_common.h"

namespace wqaa {

struct PeerExchangeArgs {
  const uint4* src;            // this rank's slice (inside its own window)
  uint32_t n16;                // 16-byte pieces of the slice
  int world, rank;
  uint32_t step;               // what this exchange posts and waits for
  unsigned long long timeout_ticks;
  uint32_t* status;            // own device word: 0, or 1 + the peer whose post did not arrive in time
  uint4* dst[WQAA_PEER_MAX];            // peer p's window at this rank's columns (unused for p == rank)
  uint32_t* post[WQAA_PEER_MAX];        // peer p's flag word for this rank
  const uint32_t* flags;       // own flag words [world]
};

__global__ void __launch_bounds__(256) wq_peer_exchange_kernel(const PeerExchangeArgs a) {
  const int p = blockIdx.x;
  if (p != a.rank) {
    uint4* dst = a.dst[p];
    for (uint32_t i = threadIdx.x; i < a.n16; i += blockDim.x) dst[i] = a.src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(a.post[p], a.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // the block of this rank's own index waits for everybody else's post
  if (threadIdx.x < (unsigned)a.world && (int)threadIdx.x != a.rank) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      const uint32_t v = __hip_atomic_load(a.flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((int32_t)(v - a.step) >= 0) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
        __hip_atomic_store(a.status, 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}

}  // namespace wqaa

using namespace wqaa;

extern "C" {

int wqaa_peer_alloc(size_t bytes, void** ptr) {
  if (!ptr || bytes == 0) {
    set_error(WQAA_ERR_BAD_DESC, "wqaa_peer_alloc: null pointer / zero size");
    return WQAA_ERR_BAD_DESC;
  }
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
  if (e == hipSuccess) e = hipMemset(p, 0, bytes);
  // hipMemset of device memory is asynchronous: a peer's first remote store or flag post must not race a late clear
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "wqaa_peer_alloc(%zu): %s", bytes, hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  *ptr = p;
  return WQAA_OK;
}

int wqaa_peer_free(void* ptr) {
  if (ptr && hipFree(ptr) != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "wqaa_peer_free failed");
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

int wqaa_peer_export(const void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == WQAA_PEER_HANDLE_BYTES, "handle size");
  if (!ptr || !handle64) {
    set_error(WQAA_ERR_BAD_DESC, "wqaa_peer_export: null pointer");
    return WQAA_ERR_BAD_DESC;
  }
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, const_cast<void*>(ptr));
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 needed on dmabuf-only hosts)", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  memcpy(handle64, &h, sizeof(h));
  return WQAA_OK;
}

int wqaa_peer_open(const void* handle64, void** ptr) {
  if (!handle64 || !ptr) {
    set_error(WQAA_ERR_BAD_DESC, "wqaa_peer_open: null pointer");
    return WQAA_ERR_BAD_DESC;
  }
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  *ptr = p;
  return WQAA_OK;
}

int wqaa_peer_close(void* ptr) {
  if (ptr && hipIpcCloseMemHandle(ptr) != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "hipIpcCloseMemHandle failed");
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

int wqaa_peer_exchange(const wqaa_peer_exchange_desc* d, void* stream) {
  if (!d || d->world < 2 || d->world > WQAA_PEER_MAX || d->rank < 0 || d->rank >= d->world || !d->src || !d->flags || !d->status ||
      d->bytes == 0 || (d->bytes & 15) || (reinterpret_cast<uintptr_t>(d->src) & 15)) {
    set_error(WQAA_ERR_BAD_DESC, "wqaa_peer_exchange: world 2..%d, rank inside it, a 16-byte aligned slice of whole 16-byte pieces", WQAA_PEER_MAX);
    return WQAA_ERR_BAD_DESC;
  }
  PeerExchangeArgs a{};
  a.src = reinterpret_cast<const uint4*>(d->src);
  a.n16 = (uint32_t)(d->bytes / 16);
  a.world = d->world;
  a.rank = d->rank;
  a.step = d->step;
  a.timeout_ticks = (unsigned long long)(d->timeout_ms > 0 ? d->timeout_ms : 2000) * 100000ull;
  a.status = d->status;
  a.flags = d->flags;
  for (int p = 0; p < d->world; ++p) {
    if (p == d->rank) continue;
    if (!d->dst[p] || !d->post[p] || (reinterpret_cast<uintptr_t>(d->dst[p]) & 15)) {
      set_error(WQAA_ERR_BAD_DESC, "wqaa_peer_exchange: peer %d has no (16-byte aligned) destination / flag word", p);
      return WQAA_ERR_BAD_DESC;
    }
    a.dst[p] = reinterpret_cast<uint4*>(d->dst[p]);
    a.post[p] = d->post[p];
  }
  hipLaunchKernelGGL(wq_peer_exchange_kernel, dim3(d->world), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "wqaa_peer_exchange: %s", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

}  // extern "C"
