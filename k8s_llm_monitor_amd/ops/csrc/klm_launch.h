// Host-side launch helpers shared by the GEMM and attention launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "klm_ops.h"

// >64 KiB of dynamic LDS needs a per-function opt-in that HIP keeps PER DEVICE: one flag per (kernel, device), set
// under std::call_once, so a process that drives several GPUs, or two threads racing on their first call, are
// both fine. The launch must go to the current device (checked by klm_stream_on_current_device).
template <auto KERNEL>
static int klm_ensure_dyn_lds(int bytes) {
  constexpr int MAXDEV = 64;
  static std::once_flag once[MAXDEV];
  static hipError_t rc[MAXDEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return KLM_ERR_HIP;
  if (dev < 0 || dev >= MAXDEV)
    return hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? KLM_OK : KLM_ERR_HIP;
  std::call_once(once[dev], [&] { rc[dev] = hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
  return rc[dev] == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}
