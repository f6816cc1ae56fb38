// ggq_host.hpp -- host-side helpers shared by the translation units of libggq_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

namespace ggq {

// remember `e` for ggq_last_hip_error() on the calling thread; returns GGQ_ERR_HIP
int hip_fail(hipError_t e);

// Measurement knobs.  The SHIPPED library reads no environment variable: launch geometry is a function of the arguments only.  Lab builds
// (tools/build_variant.py -DGGQ_LAB ...) compile the lookup in, so that one variant library can be A/B-ed over a knob on one box:
// lab_int(name, lo, hi) = the integer value of environment variable `name` if set and within [lo, hi], else -1 (always -1 when shipped).
#ifdef GGQ_LAB
}  // namespace ggq
#include <cstdlib>
namespace ggq {
inline int lab_int(const char* name, int lo, int hi)      // header-only: a lab build may pass -DGGQ_LAB to ONE translation unit (tools/build_variant.py --unit)
{
    const char* e = getenv(name);
    if (!e || !*e) return -1;
    const int x = atoi(e);
    return (x >= lo && x <= hi) ? x : -1;
}
#else
constexpr int lab_int(const char*, int, int) { return -1; }
#endif

}  // namespace ggq
