// Fused Adam over ONE flat buffer per model (parameters, gradients and both moments are
// each a single contiguous allocation; the nn.Parameters are views into it), so the
// whole optimizer step of G or D is one launch: 16 B read + 12 B written per parameter.
#include "gmt_common.h"
#include "adam_elem.h"

namespace {
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float b1, float b2, float eps,
                            float step_size, float bc2_sqrt, float grad_scale)
{
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        gmt_adam::adam_elem(pp.x, gg.x * grad_scale, mm.x, vv.x, b1, b2, eps, step_size, bc2_sqrt);
        gmt_adam::adam_elem(pp.y, gg.y * grad_scale, mm.y, vv.y, b1, b2, eps, step_size, bc2_sqrt);
        gmt_adam::adam_elem(pp.z, gg.z * grad_scale, mm.z, vv.z, b1, b2, eps, step_size, bc2_sqrt);
        gmt_adam::adam_elem(pp.w, gg.w * grad_scale, mm.w, vv.w, b1, b2, eps, step_size, bc2_sqrt);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        gmt_adam::adam_elem(p[i], g[i] * grad_scale, m[i], v[i], b1, b2, eps, step_size, bc2_sqrt);
}
}  // namespace

extern "C" int gmt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                             float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                             hipStream_t stream)
{
    GMT_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 1 && step >= 1);
    GMT_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(gmt_grid_1d((n + 3) / 4, 256)), dim3(256), 0, stream,
                       param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
