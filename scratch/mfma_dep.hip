#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int NA>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[NA] = {};
    float a = 0.5f + lane, b = 0.25f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16 / NA; ++j)
#pragma unroll
            for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < NA; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[g][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NA> void run(float* out)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    k<NA><<<256, 256>>>(out, 10); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<NA><<<256, 256>>>(out, iters); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%d independent accumulators: %.1f ns/MFMA/wave\n", NA, ms * 1e6 / (iters * 16.0));
}
int main() { float* out; CHECK(hipMalloc(&out, 256 * 256 * 4)); run<1>(out); run<2>(out); run<4>(out); return 0; }
