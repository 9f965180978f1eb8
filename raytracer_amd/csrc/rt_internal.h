// rt_internal.h — what rt_lib.hip and rt_multi.hip share beyond the public ABI (include/rt_abi.h).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace rti {

// Every call by which the library allocates device memory or creates / destroys a runtime object goes through the
// wrappers below, so that "this entry point allocates nothing" is a checkable statement: rt_stats.lib_alloc_calls
// reports the count, and tests/test_zz_no_alloc.py asserts that it does not move across rt_render_device_noalloc.
extern std::atomic<uint64_t> g_alloc_calls;

template <class T>
inline hipError_t dev_malloc(T** p, size_t bytes) {
  g_alloc_calls++;
  return hipMalloc(reinterpret_cast<void**>(p), bytes);
}
inline hipError_t dev_free(void* p) {
  g_alloc_calls++;
  return hipFree(p);
}
inline hipError_t stream_create(hipStream_t* s, unsigned flags) {
  g_alloc_calls++;
  return hipStreamCreateWithFlags(s, flags);
}
inline hipError_t stream_destroy(hipStream_t s) {
  g_alloc_calls++;
  return hipStreamDestroy(s);
}
inline hipError_t event_create(hipEvent_t* e, unsigned flags = hipEventDefault) {
  g_alloc_calls++;
  return hipEventCreateWithFlags(e, flags);
}
inline hipError_t event_destroy(hipEvent_t e) {
  g_alloc_calls++;
  return hipEventDestroy(e);
}

}  // namespace rti

// Whether the last fused render issued on `ctx` was cut short by the scheduler's watchdog (a wave found no work for
// seconds and gave up: tiles it skipped were never written).  The caller must have synchronised the render's stream.
// Returns RT_OK, RT_E_INTERNAL (message in rt_last_error) or RT_E_HIP.
struct rt_ctx;
int rt_internal_frame_status(rt_ctx* ctx);
