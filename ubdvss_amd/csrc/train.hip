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


