// Shared helpers for the gfx950 kernels of libgmt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gmt_hip.h"

#define GMT_CHECK_ARG(cond) do { if (!(cond)) return GMT_ERR_BAD_ARG; } while (0)
#define GMT_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return GMT_ERR_HIP; } while (0)

static inline int gmt_grid_1d(long long n, int block, int cap = 256 * 8) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32,
// 1 ulp each): |abs err| ~ 1.5e-7, far inside the 1e-4 parity budget, and ~6x fewer
// VALU issue slots than the libm forms - the gate math shares the SIMD with the MFMAs.
__device__ __forceinline__ float gmt_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float gmt_tanh(float x) {
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f;
}

__device__ __forceinline__ float gmt_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double gmt_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
