// Training path of the ubdvss hot path on gfx950: fused loss, backward, Adam.
// Adam update (loss: loss.hip; backward + train step: backward.hip).
#include "common.h"

__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, size_t n, float lr_t, float b1, float b2, float eps, float gscale)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

extern "C" int ubd_adam_step(float *params, const float *grads, float *m, float *v, size_t count, int t, float lr,
                             float beta1, float beta2, float eps, float grad_scale, void *stream)
{
    UBD_REQUIRE(params && grads && m && v && t >= 1, "ubd_adam_step: bad argument");
    // Keras 2.2 Adam: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    int grid = (int)((count + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, count,
                       (float)lr_t, beta1, beta2, eps, grad_scale);
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}


// ---- pipelining helper of ModelRunner (no counterpart in the reference: it runs predict and postprocess serially) --------
// One wave that idles for `microseconds` (s_sleep on the constant 100 MHz clock) and exits: enqueued on the forward stream in
// front of the next batch's first kernel, it gives the postprocess of the previous batch -- launched on a second stream at
// the same moment -- the head start it needs to get its 32 whole-CU blocks placed.  Without it the first stem kernel
// (16 384 small blocks) keeps every CU partly occupied, the postprocess front end only starts once that kernel drains
// (~70 us) and then holds 32 CUs right under the PERSISTENT second stem kernel, whose 96 blocks for those CUs wait for it:
// 83 -> 118 us (kernel timeline in DESIGN.md).
__global__ void stream_delay_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int ubd_stream_delay(void *stream, int microseconds)
{
    UBD_REQUIRE(microseconds >= 0 && microseconds <= 1000, "ubd_stream_delay: 0..1000 us, got %d", microseconds);
    if (microseconds == 0) return 0;
    hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
    UBD_CHECK_HIP(hipGetLastError());
    return 0;
}
