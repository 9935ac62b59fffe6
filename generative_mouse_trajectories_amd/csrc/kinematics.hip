// Kinematic features of RecurrentGAN.py:315-389 ("mode A"), forward and backward,
// fp32, time-major layout: pos [L][B][2], dts [L][B], feat [L][B][9].
// One thread per (t, b) point; consecutive lanes walk b, so every load and store of a
// wave is one contiguous run (12 B in / 36 B out per point forward).  HBM-bound.
#include "gmt_common.h"
#include "kinematics_point.h"

namespace {

__global__ void kin_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ dts,
                               float* __restrict__ feat, int L, int B, float eps)
{
    const long long n = (long long)L * B;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x)
        gmt_kin::kin_fwd_point(idx, pos, dts, feat, L, B, eps);
}

__global__ void kin_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ dts,
                               const float* __restrict__ gfeat, float* __restrict__ gpos,
                               float* __restrict__ gdts, int L, int B, float eps)
{
    const long long n = (long long)L * B;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x)
        gmt_kin::kin_bwd_point(idx, pos, dts, gfeat, gpos, gdts, L, B, eps);
}

}  // namespace
extern "C" int gmt_kinematics_fwd(const float* pos, const float* dts, float* feat,
                                  int L, int B, float eps, hipStream_t stream)
{
    GMT_CHECK_ARG(pos && dts && feat && L >= 1 && B >= 1);
    const long long n = (long long)L * B;
    hipLaunchKernelGGL(kin_fwd_kernel, dim3(gmt_grid_1d(n, 256)), dim3(256), 0, stream, pos, dts, feat, L, B, eps);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_kinematics_bwd(const float* pos, const float* dts, const float* gfeat,
                                  float* gpos, float* gdts, int L, int B, float eps, hipStream_t stream)
{
    GMT_CHECK_ARG(pos && dts && gfeat && gpos && gdts && L >= 1 && B >= 1);
    const long long n = (long long)L * B;
    hipLaunchKernelGGL(kin_bwd_kernel, dim3(gmt_grid_1d(n, 256)), dim3(256), 0, stream,
                       pos, dts, gfeat, gpos, gdts, L, B, eps);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
