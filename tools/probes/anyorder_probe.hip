// anyorder_probe.hip -- does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let a kernel start while its predecessor on
// the SAME stream is still running on gfx950?  hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; this measures it.
// A = one workgroup spinning ~300 us; B = a short kernel launched right behind it, (1) normally, (2) with hipExtAnyOrderLaunch.
// Prints B.start - A.end in microseconds (negative = B started before A ended = overlap).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>

__global__ void spin(uint64_t* stamps, uint64_t ticks)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t0; stamps[1] = wall_clock64(); }
}

int main()
{
    uint64_t* d;
    hipMalloc(&d, 4 * sizeof(uint64_t));
    hipStream_t s;
    hipStreamCreate(&s);
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const double us_per_tick = 1e3 / (double)rate_khz;
    const uint64_t long_ticks = (uint64_t)(300.0 / us_per_tick), short_ticks = (uint64_t)(2.0 / us_per_tick);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(d, 0, 4 * sizeof(uint64_t), s);
            hipStreamSynchronize(s);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, long_ticks);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d + 2, short_ticks);
            else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d + 2, short_ticks);
            hipError_t e = hipStreamSynchronize(s);
            uint64_t h[4];
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("ANYORDER mode=%s rep=%d err=%d  A ran %.1f us, B.start - A.end = %+.2f us\n", mode ? "any-order" : "in-order", rep, (int)e,
                   (double)(h[1] - h[0]) * us_per_tick, ((double)h[2] - (double)h[1]) * us_per_tick);
        }
    }
    return 0;
}
