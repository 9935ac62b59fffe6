// Spectral normalisation of the critic's Linear weights (torch.nn.utils.spectral_norm as used at
// RG.py:713-736): per training-mode forward one power iteration
//   v <- normalize(W^T u),  u <- normalize(W v),  sigma = u . (W v),  W_sn = W / sigma
// and in the backward  dW = dW_sn / sigma - (sum dW_sn (.) W) / sigma^2 * u v^T.
// As ATen ops this is ~14 + 8 launches of a few microseconds per layer and call (5 layers, 3-4 critic
// calls per iteration: ~330 launches); here every layer of a call is one workgroup (16 waves) of ONE launch.
// The matrices are tiny (<= 128 x 256), so the kernel is latency-bound by design: W is read three times
// from L2, all reductions are fixed-order (deterministic).
#include "gmt_common.h"

namespace {

struct sn_fwd_table { gmt_sn_fwd_layer l[GMT_SN_MAX_LAYERS]; };
struct sn_bwd_table { gmt_sn_bwd_layer l[GMT_SN_MAX_LAYERS]; };

constexpr int SN_NT = 1024;                            // one workgroup of 16 waves per layer

// sum over the threads of the workgroup, result in every thread (fixed order: deterministic)
__device__ __forceinline__ float block_sum(float v, float* red)
{
    v = gmt_wave_sum(v);
    __syncthreads();                                    // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SN_NT / 64; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(SN_NT)
void sn_fwd_kernel(sn_fwd_table tab, float eps, int power_iterate)
{
    const gmt_sn_fwd_layer L = tab.l[blockIdx.x];
    const int O = L.out_features, I = L.in_features;
    extern __shared__ float sm[];
    float* sv = sm;                                     // [I]
    float* su = sm + I;                                 // [O]
    float* swv = sm + I + O;                            // [O]  W v
    float* part = sm + I + 2 * O;                       // [G][I] partial column sums
    __shared__ float red[SN_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ W = L.w_orig;

    for (int i = tid; i < O; i += SN_NT) su[i] = L.u[i];
    for (int j = tid; j < I; j += SN_NT) sv[j] = L.v[j];
    __syncthreads();
    if (power_iterate) {
        // v = normalize(W^T u): the threads form G groups of IR >= I columns; group g sums rows g, g+G, ...
        // (neighbouring threads read neighbouring floats), then the G partial sums are added in order
        int IR = 1;
        while (IR < I && IR < SN_NT) IR <<= 1;
        const int G = SN_NT / IR;
        const int g = tid / IR;
        for (int j = tid % IR; j < I; j += IR) {
            float acc = 0.f;
            for (int i = g; i < O; i += G) acc += W[(size_t)i * I + j] * su[i];
            part[g * I + j] = acc;
        }
        __syncthreads();
        float ss = 0.f;
        for (int j = tid; j < I; j += SN_NT) {
            float acc = 0.f;
            for (int q = 0; q < G; ++q) acc += part[q * I + j];
            sv[j] = acc;
            ss += acc * acc;
        }
        const float nv = fmaxf(sqrtf(block_sum(ss, red)), eps);
        for (int j = tid; j < I; j += SN_NT) sv[j] = sv[j] / nv;
        __syncthreads();
    }
    // W v: one wave per row
    for (int i = wave; i < O; i += SN_NT / 64) {
        float acc = 0.f;
        for (int j = lane; j < I; j += 64) acc += W[(size_t)i * I + j] * sv[j];
        acc = gmt_wave_sum(acc);
        if (lane == 0) swv[i] = acc;
    }
    __syncthreads();
    if (power_iterate) {
        float ss = 0.f;
        for (int i = tid; i < O; i += SN_NT) ss += swv[i] * swv[i];
        const float nu = fmaxf(sqrtf(block_sum(ss, red)), eps);
        for (int i = tid; i < O; i += SN_NT) su[i] = swv[i] / nu;
        __syncthreads();
        for (int i = tid; i < O; i += SN_NT) L.u[i] = su[i];
        for (int j = tid; j < I; j += SN_NT) L.v[j] = sv[j];
    }
    if (L.uv_saved) {                                   // the u, v this call's sigma was formed with (for the backward)
        for (int i = tid; i < O; i += SN_NT) L.uv_saved[i] = su[i];
        for (int j = tid; j < I; j += SN_NT) L.uv_saved[O + j] = sv[j];
    }
    float sp = 0.f;
    for (int i = tid; i < O; i += SN_NT) sp += su[i] * swv[i];
    const float sigma = block_sum(sp, red);
    if (tid == 0 && L.sigma) L.sigma[0] = sigma;
    const size_t n = (size_t)O * I;
    for (size_t k = tid; k < n; k += SN_NT) L.w_out[k] = W[k] / sigma;
}

__global__ __launch_bounds__(SN_NT)
void sn_bwd_kernel(sn_bwd_table tab)
{
    const gmt_sn_bwd_layer L = tab.l[blockIdx.x];
    const int O = L.out_features, I = L.in_features;
    __shared__ float red[SN_NT / 64];
    const int tid = threadIdx.x;
    const size_t n = (size_t)O * I;
    float sp = 0.f;
    for (size_t k = tid; k < n; k += SN_NT) sp += L.d_w_sn[k] * L.w_orig[k];
    const float s = block_sum(sp, red);
    const float sigma = L.sigma[0];
    const float c = s / (sigma * sigma);
    const float* u = L.uv_saved;
    const float* v = L.uv_saved + O;
    // thread -> (row group, column): no per-element division
    int IR = 1;
    while (IR < I && IR < SN_NT) IR <<= 1;
    const int G = SN_NT / IR;
    for (int j = tid % IR; j < I; j += IR) {
        const float cv = c * v[j];
        for (int i = tid / IR; i < O; i += G) {
            const size_t k = (size_t)i * I + j;
            L.d_w_orig[k] = L.d_w_sn[k] / sigma - cv * u[i];
        }
    }
}

}  // namespace

extern "C" int gmt_spectral_norm_fwd(const gmt_sn_fwd_layer* layers, int n_layers, float eps, int power_iterate,
                                     hipStream_t stream)
{
    GMT_CHECK_ARG(layers && n_layers >= 1 && n_layers <= GMT_SN_MAX_LAYERS && eps > 0.f);
    sn_fwd_table tab;
    size_t lds = 0;
    for (int k = 0; k < n_layers; ++k) {
        const gmt_sn_fwd_layer& l = layers[k];
        GMT_CHECK_ARG(l.w_orig && l.u && l.v && l.w_out && l.out_features >= 1 && l.in_features >= 1);
        int ir = 1;
        while (ir < l.in_features && ir < SN_NT) ir <<= 1;
        const size_t need = ((size_t)l.in_features * (1 + SN_NT / ir) + 2 * (size_t)l.out_features) * sizeof(float);
        if (need > lds) lds = need;
        tab.l[k] = l;
    }
    if (lds > 64 * 1024) return GMT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sn_fwd_kernel, dim3(n_layers), dim3(SN_NT), lds, stream, tab, eps, power_iterate);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_spectral_norm_bwd(const gmt_sn_bwd_layer* layers, int n_layers, hipStream_t stream)
{
    GMT_CHECK_ARG(layers && n_layers >= 1 && n_layers <= GMT_SN_MAX_LAYERS);
    sn_bwd_table tab;
    for (int k = 0; k < n_layers; ++k) {
        const gmt_sn_bwd_layer& l = layers[k];
        GMT_CHECK_ARG(l.d_w_sn && l.w_orig && l.uv_saved && l.sigma && l.d_w_orig && l.out_features >= 1 &&
                      l.in_features >= 1);
        tab.l[k] = l;
    }
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(n_layers), dim3(SN_NT), 0, stream, tab);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
