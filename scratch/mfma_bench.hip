// micro-benchmark: what fp32 MFMA rate does one wave per SIMD reach (a) from registers only,
// (b) with LDS A reads, (c) with weights streamed from L2 (1-deep / 2-deep prefetch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDA 33
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float4* __restrict__ w, float* out, int iters, int K)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < K * LDA; i += 256) lds[i] = 0.001f * (i & 15);
    __syncthreads();
    f32x16 acc[4] = {};
    const int hi = lane >> 5, lo = lane & 31;
    const float* ap = lds + hi * LDA + lo;
    const int ns4 = K >> 3;
    const float4* wp = w + (size_t)wave * ns4 * 4 * 64;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // registers only
            float a = 0.5f + lane, b = 0.25f;
#pragma unroll 1
            for (int s4 = 0; s4 < ns4; ++s4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
            }
        } else if (MODE == 1) {     // LDS A reads, register B
            float b = 0.25f;
#pragma unroll 1
            for (int s4 = 0; s4 < ns4; ++s4) {
                const float* aq = ap + 8 * s4 * LDA;
                float a0 = aq[0], a1 = aq[2 * LDA], a2 = aq[4 * LDA], a3 = aq[6 * LDA];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b, acc[g], 0, 0, 0);
                }
            }
        } else if (MODE == 2) {     // streamed weights, 1-deep prefetch
            float4 wn[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wn[g] = wp[g * 64 + lane];
#pragma unroll 1
            for (int s4 = 0; s4 < ns4; ++s4) {
                float4 wc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) wc[g] = wn[g];
                if (s4 + 1 < ns4) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) wn[g] = wp[((s4 + 1) * 4 + g) * 64 + lane];
                }
                const float* aq = ap + 8 * s4 * LDA;
                float a0 = aq[0], a1 = aq[2 * LDA], a2 = aq[4 * LDA], a3 = aq[6 * LDA];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wc[g].x, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wc[g].y, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, wc[g].z, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, wc[g].w, acc[g], 0, 0, 0);
                }
            }
        } else if (MODE == 3) {     // streamed weights, 2-deep prefetch (manual 2x unroll)
            float4 w0[4], w1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { w0[g] = wp[g * 64 + lane]; w1[g] = wp[(4 + g) * 64 + lane]; }
#pragma unroll 1
            for (int s4 = 0; s4 < ns4; s4 += 2) {
                float4 wc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) wc[g] = w0[g];
                if (s4 + 2 < ns4) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) w0[g] = wp[((s4 + 2) * 4 + g) * 64 + lane];
                }
                const float* aq = ap + 8 * s4 * LDA;
                float a0 = aq[0], a1 = aq[2 * LDA], a2 = aq[4 * LDA], a3 = aq[6 * LDA];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wc[g].x, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wc[g].y, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, wc[g].z, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, wc[g].w, acc[g], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) wc[g] = w1[g];
                if (s4 + 3 < ns4) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) w1[g] = wp[((s4 + 3) * 4 + g) * 64 + lane];
                }
                aq += 8 * LDA;
                a0 = aq[0]; a1 = aq[2 * LDA]; a2 = aq[4 * LDA]; a3 = aq[6 * LDA];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wc[g].x, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wc[g].y, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, wc[g].z, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, wc[g].w, acc[g], 0, 0, 0);
                }
            }
        } else if (MODE == 4) {     // ping-pong register buffers, no copies: distance = 16 MFMAs
            float4 w0[4], w1[4];
            float a0[4], a1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) w0[g] = wp[g * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) a0[j] = ap[2 * j * LDA];
#pragma unroll 1
            for (int s4 = 0; s4 < ns4; s4 += 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) w1[g] = wp[((s4 + 1) * 4 + g) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) a1[j] = ap[(8 * (s4 + 1) + 2 * j) * LDA];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], w0[g].x, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], w0[g].y, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[2], w0[g].z, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[3], w0[g].w, acc[g], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int sn = (s4 + 2 < ns4) ? s4 + 2 : 0;      // harmless reload on the last trip
#pragma unroll
                for (int g = 0; g < 4; ++g) w0[g] = wp[(sn * 4 + g) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) a0[j] = ap[(8 * sn + 2 * j) * LDA];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], w1[g].x, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], w1[g].y, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[2], w1[g].z, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[3], w1[g].w, acc[g], 0, 0, 0);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[g][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, const float4* w, float* out, int grid, int iters, int K)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    size_t lds = (size_t)K * LDA * 4;
    k<MODE><<<grid, 256, lds>>>(w, out, 4, K);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<MODE><<<grid, 256, lds>>>(w, out, iters, K);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double nm = (double)iters * (K / 2) * 4;          // MFMAs per wave
    double tf = nm * 4096.0 * 4 * grid / (ms * 1e-3) / 1e12;
    printf("%-28s grid %4d K %4d: %8.3f ms  %6.1f ns/MFMA/wave  %7.1f TF/s\n", name, grid, K, ms, ms * 1e6 / nm, tf);
}

int main()
{
    const int K = 256;
    float4* w; float* out;
    CHECK(hipMalloc(&w, (size_t)4 * (K / 8) * 4 * 64 * 16));
    CHECK(hipMemset(w, 0, (size_t)4 * (K / 8) * 4 * 64 * 16));
    CHECK(hipMalloc(&out, 1024 * 256 * 4));
    for (int grid : {256, 512}) {
        run<0>("registers only", w, out, grid, 400, K);
        run<1>("LDS A", w, out, grid, 400, K);
        run<2>("stream 1-deep", w, out, grid, 400, K);
        run<3>("stream 2-deep", w, out, grid, 400, K);
        run<4>("stream ping-pong", w, out, grid, 400, K);
    }
    return 0;
}
