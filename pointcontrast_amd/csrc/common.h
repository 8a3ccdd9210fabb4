// Shared host/device helpers of libpcmi (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pcmi.h"

namespace pcmi {

void set_error(const char* fmt, ...);

#define PCMI_HIP_CHECK(expr)                                                             \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      pcmi::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return PCMI_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)

#define PCMI_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      pcmi::set_error(__VA_ARGS__);    \
      return (code);                   \
    }                                  \
  } while (0)

// kernel launches of the calling thread so far (every launch site ends in PCMI_LAUNCH_CHECK): what the executor's timing
// mode reads before and after an op to report launches per op (pcmi_net_timed_launches)
extern thread_local long g_launches;
#define PCMI_LAUNCH_CHECK()         \
  do {                              \
    ++pcmi::g_launches;             \
    PCMI_HIP_CHECK(hipGetLastError()); \
  } while (0)

static inline hipStream_t as_stream(pcmi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---- packed coordinate key ---------------------------------------------------------------
// (batch:10 | x:18 | y:18 | z:18), x/y/z biased by 2^17.  All-ones is the empty sentinel.
constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kCoordBias = 1 << 17;

__host__ __device__ inline uint64_t pack_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint32_t)b << 54) | ((uint64_t)(uint32_t)(x + kCoordBias) << 36) |
         ((uint64_t)(uint32_t)(y + kCoordBias) << 18) | (uint64_t)(uint32_t)(z + kCoordBias);
}

__host__ __device__ inline uint32_t hash_key(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

// ---- last-arriver hand-off between the workgroups of ONE launch -------------------------------------------------
// Every workgroup of a group publishes its partial result with plain stores and then calls arrive_last() on the
// group's counter; exactly one call (the last to arrive) returns true, and that workgroup then sees the stores of
// all the others.  Protocol as MI355X_MICROARCH.md prescribes (per-XCD L2s are not coherent, a CU's L1 is never
// refreshed by other CUs): plain stores -> __syncthreads -> lane-0 agent-scope RELEASE (buffer_wbl2) -> explicit
// s_waitcnt vmcnt(0) (the compiler may drop its own) -> relaxed agent-scope ticket; the last arriver does ONE
// agent-scope ACQUIRE (buffer_inv) -> __syncthreads -> plain loads.  The counter is left at zero (self-resetting), so
// a zero-initialised counter pool (stream_counters, internal.h) serves every later launch on the same stream.
// Nobody spins: correctness does not depend on which workgroups are resident.
#if defined(__HIPCC__)
__device__ inline bool arrive_last(unsigned* counter, unsigned expected, unsigned* s_flag /* __shared__ */) {
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = (t + 1u == expected) ? 1u : 0u;
    if (last) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0u;
}
#endif

// floor division by a positive power-of-two-free divisor (coordinates may be negative)
__host__ __device__ inline int floor_div(int a, int d) {
  int q = a / d;
  return (a % d != 0 && ((a < 0) != (d < 0))) ? q - 1 : q;
}

}  // namespace pcmi
