// ggq_host.hpp -- host-side helpers shared by the translation units of libggq_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

namespace ggq {

// remember `e` for ggq_last_hip_error() on the calling thread; returns GGQ_ERR_HIP
int hip_fail(hipError_t e);

}  // namespace ggq
