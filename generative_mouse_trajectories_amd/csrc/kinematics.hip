// Kinematic features of RecurrentGAN.py:315-389 ("mode A"), forward and backward,
// fp32, time-major layout: pos [L][B][2], dts [L][B], feat [L][B][9].
// One thread per trajectory walking a chunk of timesteps; consecutive lanes walk b, so every load
// and store of a wave is one contiguous run (12 B in / 36 B out per point forward).  HBM-bound.
#include "gmt_common.h"
#include "kinematics_point.h"

namespace {

constexpr int KB = 256;        // trajectories per block (4 waves, each with its own 64 trajectories)
#ifndef GMT_KIN_KCH
#define GMT_KIN_KCH 8
#endif

// Block = 256 consecutive trajectories x `kch` consecutive timesteps.  A thread keeps its trajectory and walks t,
// so the 3-step look-back (look-ahead in the backward) re-reads lines this CU fetched one trip earlier.  The 9
// features of a WAVE's 64 points at one t are one contiguous run of 576 floats in memory: they go through a
// wave-private LDS region (stride-9 per-lane accesses are conflict-free) and leave / arrive as float4 - no
// workgroup barrier anywhere, a wave only orders its own LDS accesses.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// JVP = true: the forward-mode derivative instead (kin_jvp_vals: feature tangents for the position tangents phat).
template <bool JVP>
__global__ __launch_bounds__(KB)
void kin_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ dts, const float* __restrict__ phat,
                    float* __restrict__ feat, int L, int B, float eps, int kch)
{
    __shared__ float4 stage4[KB / 64][64 * 9 / 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* stage = reinterpret_cast<float*>(stage4[wave]);
    const int bw = blockIdx.x * KB + wave * 64;                   // first trajectory of this wave
    if (bw >= B) return;
    const int b = bw + lane;
    const int nbw = (B - bw < 64) ? B - bw : 64;                  // trajectories of this wave
    const int t0 = blockIdx.y * kch, t1 = (t0 + kch < L) ? t0 + kch : L;
    for (int t = t0; t < t1; ++t) {
        const long long base = (long long)t * B + bw;
        if (lane < nbw) {
            float o[9];
            if constexpr (JVP) gmt_kin::kin_jvp_vals(base + lane, pos, dts, phat, o, L, B, eps);
            else gmt_kin::kin_fwd_vals(base + lane, pos, dts, o, L, B, eps);
#pragma unroll
            for (int c = 0; c < 9; ++c) stage[lane * 9 + c] = o[c];
        }
        wave_lds_fence();
        const int cnt = nbw * 9;
        float* dst = feat + base * 9;
        if (((base * 9) & 3) == 0) {
            for (int i = lane; i * 4 < cnt; i += 64) {
                if (i * 4 + 4 <= cnt) reinterpret_cast<float4*>(dst)[i] = stage4[wave][i];
                else for (int e = i * 4; e < cnt; ++e) dst[e] = stage[e];
            }
        } else {
            for (int e = lane; e < cnt; e += 64) dst[e] = stage[e];
        }
        wave_lds_fence();
    }
    (void)b;
}

// Backward: the feature gradients of steps s .. s+3 are four such runs; they sit in a wave-private ring of four LDS
// regions, one new run per step, requested from HBM one step before it is staged.
__global__ __launch_bounds__(KB)
void kin_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ dts,
                    const float* __restrict__ gfeat, float* __restrict__ gpos,
                    float* __restrict__ gdts, int L, int B, float eps, int kch)
{
    __shared__ float4 ring4[KB / 64][4][64 * 9 / 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bw = blockIdx.x * KB + wave * 64;
    if (bw >= B) return;
    const int nbw = (B - bw < 64) ? B - bw : 64;
    const int cnt = nbw * 9;
    const int s0 = blockIdx.y * kch, s1 = (s0 + kch < L) ? s0 + kch : L;
    const bool vec = ((((long long)B * 9) & 3) == 0) && (((long long)bw * 9 & 3) == 0) && (cnt & 3) == 0;   // every run float4-aligned
    float4 pre[3];
    auto fetch_run = [&](int t) {                                // gradients of step t -> registers (t <= L-1)
        const float4* src = reinterpret_cast<const float4*>(gfeat + ((long long)t * B + bw) * 9);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = lane + 64 * j;
            pre[j] = (i * 4 < cnt) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_run = [&](int t) {                                // registers -> ring[t & 3]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = lane + 64 * j;
            if (i * 4 < cnt) ring4[wave][t & 3][i] = pre[j];
        }
    };
    auto stage_scalar = [&](int t) {                             // unaligned shapes: one float at a time
        const float* src = gfeat + ((long long)t * B + bw) * 9;
        float* st = reinterpret_cast<float*>(ring4[wave][t & 3]);
        for (int e = lane; e < cnt; e += 64) st[e] = src[e];
    };
    for (int t = s0; t < s0 + 3 && t < L; ++t) {
        if (vec) { fetch_run(t); store_run(t); } else stage_scalar(t);
    }
    if (vec && s0 + 3 < L) fetch_run(s0 + 3);
    for (int s = s0; s < s1; ++s) {
        if (s + 3 < L) {
            if (vec) {
                store_run(s + 3);
                if (s + 4 < L) fetch_run(s + 4);                 // in flight under this step's arithmetic
            } else {
                stage_scalar(s + 3);
            }
        }
        wave_lds_fence();
        if (lane < nbw) {
            const long long idx = (long long)s * B + bw + lane;
            const float* ring = reinterpret_cast<const float*>(ring4[wave]);
            float gx, gy, gd;
            gmt_kin::kin_bwd_vals(idx, pos, dts, [&](int k, int c) { return ring[((s + k) & 3) * (64 * 9) + lane * 9 + c]; },
                                  gx, gy, gd, L, B, eps);
            reinterpret_cast<float2*>(gpos)[idx] = make_float2(gx, gy);
            gdts[idx] = gd;
        }
        wave_lds_fence();                                        // ring[s & 3] is overwritten by the next trip
    }
}

// timesteps per block: long chunks (3-step halo = 9 % at 32) when there are enough trajectories to fill the chip
static int kin_chunk(int L, int B)
{
    const long long blocks32 = (long long)((B + KB - 1) / KB) * ((L + 31) / 32);
    return blocks32 >= 1024 ? 32 : GMT_KIN_KCH;
}

}  // namespace
extern "C" int gmt_kinematics_fwd(const float* pos, const float* dts, float* feat,
                                  int L, int B, float eps, hipStream_t stream)
{
    GMT_CHECK_ARG(pos && dts && feat && L >= 1 && B >= 1);
    const int kch = kin_chunk(L, B);
    hipLaunchKernelGGL(kin_fwd_kernel<false>, dim3((B + KB - 1) / KB, (L + kch - 1) / kch), dim3(KB), 0, stream, pos, dts, nullptr, feat, L, B, eps, kch);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_kinematics_jvp(const float* pos, const float* dts, const float* phat, float* fhat,
                                  int L, int B, float eps, hipStream_t stream)
{
    GMT_CHECK_ARG(pos && dts && phat && fhat && L >= 1 && B >= 1);
    const int kch = kin_chunk(L, B);
    hipLaunchKernelGGL(kin_fwd_kernel<true>, dim3((B + KB - 1) / KB, (L + kch - 1) / kch), dim3(KB), 0, stream, pos, dts, phat, fhat, L, B, eps, kch);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

namespace {
// out[i] = g[i] * (act[i] > 0 ? 1 : slope): the LeakyReLU derivative as a mask (float4 lanes, n % 4 == 0)
__global__ __launch_bounds__(256)
void lrelu_mask_mul_kernel(const float4* __restrict__ g, const float4* __restrict__ act, float4* __restrict__ out, float slope, long long n4)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = act[i], v = g[i];
        out[i] = make_float4(v.x * (a.x > 0.f ? 1.f : slope), v.y * (a.y > 0.f ? 1.f : slope),
                             v.z * (a.z > 0.f ? 1.f : slope), v.w * (a.w > 0.f ? 1.f : slope));
    }
}
}  // namespace

extern "C" int gmt_lrelu_mask_mul(const float* g, const float* act, float* out, float slope, long long n, hipStream_t stream)
{
    GMT_CHECK_ARG(g && act && out && n >= 4 && (n % 4) == 0);
    GMT_CHECK_ARG((((uintptr_t)g | (uintptr_t)act | (uintptr_t)out) & 15) == 0);
    hipLaunchKernelGGL(lrelu_mask_mul_kernel, dim3(gmt_grid_1d(n / 4, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(g), reinterpret_cast<const float4*>(act), reinterpret_cast<float4*>(out), slope, n / 4);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_kinematics_bwd(const float* pos, const float* dts, const float* gfeat,
                                  float* gpos, float* gdts, int L, int B, float eps, hipStream_t stream)
{
    GMT_CHECK_ARG(pos && dts && gfeat && gpos && gdts && L >= 1 && B >= 1);
    const int kch = kin_chunk(L, B);
    hipLaunchKernelGGL(kin_bwd_kernel, dim3((B + KB - 1) / KB, (L + kch - 1) / kch), dim3(KB), 0, stream,
                       pos, dts, gfeat, gpos, gdts, L, B, eps, kch);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
