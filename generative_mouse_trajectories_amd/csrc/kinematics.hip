// Kinematic features of RecurrentGAN.py:315-389 ("mode A"), forward and backward,
// fp32, time-major layout: pos [L][B][2], dts [L][B], feat [L][B][9].
// One thread per trajectory walking a chunk of timesteps; consecutive lanes walk b, so every load
// and store of a wave is one contiguous run (12 B in / 36 B out per point forward).  HBM-bound.
#include "gmt_common.h"
#include "kinematics_point.h"

namespace {

constexpr int KB = 256;        // trajectories per block
#ifndef GMT_KIN_KCH
#define GMT_KIN_KCH 8
#endif
constexpr int KCH = GMT_KIN_KCH;  // consecutive timesteps per block

// Block = 256 consecutive trajectories x KCH consecutive timesteps.  A thread keeps its trajectory and walks t,
// so the 3-step look-back (look-ahead in the backward) re-reads lines this CU fetched one trip earlier instead
// of lines 12 B * B apart in HBM, and the 9 features of the block's 256 points at one t - one contiguous run of
// 2304 floats - go through LDS and leave (arrive) as float4.
__global__ __launch_bounds__(KB)
void kin_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ dts,
                    float* __restrict__ feat, int L, int B, float eps)
{
    __shared__ float4 stage4[KB * 9 / 4];
    float* stage = reinterpret_cast<float*>(stage4);
    const int b0 = blockIdx.x * KB, b = b0 + threadIdx.x;
    const int nb = (B - b0 < KB) ? B - b0 : KB;                  // trajectories of this block
    const int t0 = blockIdx.y * KCH, t1 = (t0 + KCH < L) ? t0 + KCH : L;
    for (int t = t0; t < t1; ++t) {
        const long long base = (long long)t * B + b0;
        if (b < B) {
            float o[9];
            gmt_kin::kin_fwd_vals(base + threadIdx.x, pos, dts, o, L, B, eps);
#pragma unroll
            for (int c = 0; c < 9; ++c) stage[threadIdx.x * 9 + c] = o[c];      // stride 9: conflict-free
        }
        __syncthreads();
        const int cnt = nb * 9;
        float* dst = feat + base * 9;
        if (((base * 9) & 3) == 0) {
            for (int i = threadIdx.x; i * 4 < cnt; i += KB) {
                if (i * 4 + 4 <= cnt) reinterpret_cast<float4*>(dst)[i] = stage4[i];
                else for (int e = i * 4; e < cnt; ++e) dst[e] = stage[e];
            }
        } else {
            for (int e = threadIdx.x; e < cnt; e += KB) dst[e] = stage[e];
        }
        __syncthreads();
    }
}

// Backward: the feature gradients of steps s .. s+3 are four such runs; they sit in a ring of four LDS buffers,
// one new run per step.
__global__ __launch_bounds__(KB)
void kin_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ dts,
                    const float* __restrict__ gfeat, float* __restrict__ gpos,
                    float* __restrict__ gdts, int L, int B, float eps)
{
    __shared__ float4 ring4[4][KB * 9 / 4];
    const int b0 = blockIdx.x * KB, b = b0 + threadIdx.x;
    const int nb = (B - b0 < KB) ? B - b0 : KB;
    const int s0 = blockIdx.y * KCH, s1 = (s0 + KCH < L) ? s0 + KCH : L;
    auto stage_run = [&](int t) {                                // gradients of step t -> ring[t & 3] (t <= L-1)
        const long long base = (long long)t * B + b0;
        const int cnt = nb * 9;
        const float* src = gfeat + base * 9;
        float* st = reinterpret_cast<float*>(ring4[t & 3]);
        if (((base * 9) & 3) == 0) {
            for (int i = threadIdx.x; i * 4 < cnt; i += KB) {
                if (i * 4 + 4 <= cnt) ring4[t & 3][i] = reinterpret_cast<const float4*>(src)[i];
                else for (int e = i * 4; e < cnt; ++e) st[e] = src[e];
            }
        } else {
            for (int e = threadIdx.x; e < cnt; e += KB) st[e] = src[e];
        }
    };
    for (int t = s0; t < s0 + 3 && t < L; ++t) stage_run(t);
    for (int s = s0; s < s1; ++s) {
        if (s + 3 < L) stage_run(s + 3);
        __syncthreads();
        if (b < B) {
            const long long idx = (long long)s * B + b;
            const float* ring = reinterpret_cast<const float*>(ring4);
            const int tx = threadIdx.x;
            float gx, gy, gd;
            gmt_kin::kin_bwd_vals(idx, pos, dts, [&](int k, int c) { return ring[((s + k) & 3) * (KB * 9) + tx * 9 + c]; },
                                  gx, gy, gd, L, B, eps);
            reinterpret_cast<float2*>(gpos)[idx] = make_float2(gx, gy);
            gdts[idx] = gd;
        }
        __syncthreads();                                         // ring[s & 3] is overwritten by the next trip
    }
}

}  // namespace
extern "C" int gmt_kinematics_fwd(const float* pos, const float* dts, float* feat,
                                  int L, int B, float eps, hipStream_t stream)
{
    GMT_CHECK_ARG(pos && dts && feat && L >= 1 && B >= 1);
    hipLaunchKernelGGL(kin_fwd_kernel, dim3((B + KB - 1) / KB, (L + KCH - 1) / KCH), dim3(KB), 0, stream, pos, dts, feat, L, B, eps);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_kinematics_bwd(const float* pos, const float* dts, const float* gfeat,
                                  float* gpos, float* gdts, int L, int B, float eps, hipStream_t stream)
{
    GMT_CHECK_ARG(pos && dts && gfeat && gpos && gdts && L >= 1 && B >= 1);
    hipLaunchKernelGGL(kin_bwd_kernel, dim3((B + KB - 1) / KB, (L + KCH - 1) / KCH), dim3(KB), 0, stream,
                       pos, dts, gfeat, gpos, gdts, L, B, eps);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
