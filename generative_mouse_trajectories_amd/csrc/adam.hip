// Fused Adam over ONE flat buffer per model (parameters, gradients and both moments are
// each a single contiguous allocation; the nn.Parameters are views into it), so the
// whole optimizer step of G or D is one launch: 16 B read + 12 B written per parameter.
#include "gmt_common.h"
#include "adam_elem.h"

namespace {
// Each thread owns ADAM_U float4 of every array per trip and requests all 4 x ADAM_U of them before the first is
// used: 256 B in flight per lane instead of 64 (HBM-bound: 28 B per parameter, nothing to compute).
constexpr int ADAM_U = 4;

__global__ __launch_bounds__(256)
void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                 float* __restrict__ v, long long n, float b1, float b2, float eps,
                 float step_size, float bc2_sqrt, float grad_scale)
{
    const long long n4 = n >> 2;
    const long long tile = (long long)blockDim.x * ADAM_U;          // float4 per block and trip
    for (long long base = (long long)blockIdx.x * tile; base < n4; base += (long long)gridDim.x * tile) {
        float4 pp[ADAM_U], gg[ADAM_U], mm[ADAM_U], vv[ADAM_U];
        long long idx[ADAM_U];
#pragma unroll
        for (int u = 0; u < ADAM_U; ++u) {
            const long long i = base + (long long)u * blockDim.x + threadIdx.x;
            idx[u] = i < n4 ? i : -1;
            const long long j = i < n4 ? i : n4 - 1;                  // clamped: no branch around the loads
            pp[u] = reinterpret_cast<float4*>(p)[j];
            gg[u] = reinterpret_cast<const float4*>(g)[j];
            mm[u] = reinterpret_cast<float4*>(m)[j];
            vv[u] = reinterpret_cast<float4*>(v)[j];
        }
#pragma unroll
        for (int u = 0; u < ADAM_U; ++u) {
            gmt_adam::adam_elem(pp[u].x, gg[u].x * grad_scale, mm[u].x, vv[u].x, b1, b2, eps, step_size, bc2_sqrt);
            gmt_adam::adam_elem(pp[u].y, gg[u].y * grad_scale, mm[u].y, vv[u].y, b1, b2, eps, step_size, bc2_sqrt);
            gmt_adam::adam_elem(pp[u].z, gg[u].z * grad_scale, mm[u].z, vv[u].z, b1, b2, eps, step_size, bc2_sqrt);
            gmt_adam::adam_elem(pp[u].w, gg[u].w * grad_scale, mm[u].w, vv[u].w, b1, b2, eps, step_size, bc2_sqrt);
            if (idx[u] >= 0) {
                reinterpret_cast<float4*>(p)[idx[u]] = pp[u];
                reinterpret_cast<float4*>(m)[idx[u]] = mm[u];
                reinterpret_cast<float4*>(v)[idx[u]] = vv[u];
            }
        }
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x)
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
    hipLaunchKernelGGL(adam_kernel, dim3(gmt_grid_1d((n + 3) / 4, 256 * ADAM_U, 256 * 16)), dim3(256), 0, stream,
                       param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
