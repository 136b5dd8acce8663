// Does hipExtAnyOrderLaunch drop the barrier bit on gfx950?  Two 200-us single-block kernels on ONE stream: serial = 400 us, overlapped = 200 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(long long cycles, int *out)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
    if (out) out[0] = 1;
}
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long ticks = 20000;     // s_memrealtime: 100 MHz -> 200 us
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, ticks, (int *)nullptr);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, ticks, (int *)nullptr);
            else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, (int *)nullptr);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 100LL, (int *)nullptr);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us\n", mode ? "second kernel any-order" : "plain", ms * 1e3f);
        }
    }
    return 0;
}
