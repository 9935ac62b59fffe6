// LDS-DMA throughput of one CU from L2-hot operands: how fast can a workgroup fill (256 + 128) x 64-k bf16 stage images?
//   variants: source-address XOR swizzle on / off; 4 or 8 waves; 128-byte or 64-byte rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>   // 0: swizzled 128-B rows, 1: linear 128-B rows, 2: swizzled 64-B rows, 3: register staging (global_load + ds_write), swizzled dest
__global__ __launch_bounds__(512) void fill(const char* __restrict__ src, long long ld, int ntiles, int* sink)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int rows = 384;
    for (int t = 0; t < ntiles; ++t) {
        char* dst = lds + (t & 1) * 49152;
        const char* g = src + (long long)(t & 7) * 128;          // 8 K-tiles, all L2-hot
        if (MODE == 0 || MODE == 1) {
            for (int r0 = wave * 8; r0 < rows; r0 += nw * 8) {
                const int row = r0 + (lane >> 3);
                const int chunk = MODE == 0 ? ((lane & 7) ^ ((row >> 1) & 7)) : (lane & 7);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (long long)row * ld + chunk * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + r0 * 128), 16, 0, 0);
            }
        } else if (MODE == 2) {
            for (int half = 0; half < 2; ++half)
                for (int r0 = wave * 16; r0 < rows; r0 += nw * 16) {
                    const int row = r0 + (lane >> 2);
                    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (long long)row * ld + half * 64 + chunk * 16),
                                                     (__attribute__((address_space(3))) void*)(dst + half * 24576 + r0 * 64), 16, 0, 0);
                }
        } else {
            for (int r0 = wave * 8; r0 < rows; r0 += nw * 8) {
                const int row = r0 + (lane >> 3);
                const float4 v = *reinterpret_cast<const float4*>(g + (long long)row * ld + (lane & 7) * 16);
                *reinterpret_cast<float4*>(dst + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4)) = v;
            }
        }
        if (t & 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(int*)lds;
}
int main()
{
    char* src; int* sink;
    const long long ld = 2048;
    CK(hipMalloc(&src, 384 * ld * 64)); CK(hipMemset(src, 1, 384 * ld * 64)); CK(hipMalloc(&sink, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ntiles = 2000;
    const char* names[4] = {"LDS-DMA, swizzled source, 128-B rows", "LDS-DMA, linear source, 128-B rows", "LDS-DMA, swizzled source, 64-B rows", "global_load + ds_write_b128"};
    for (int mode = 0; mode < 4; ++mode)
        for (int threads = 256; threads <= 512; threads *= 2)
            for (int grid = 1; grid <= 256; grid *= 256) {
                auto k = mode == 0 ? fill<0> : mode == 1 ? fill<1> : mode == 2 ? fill<2> : fill<3>;
                CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 98304, 0, src, ld, ntiles, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                printf("%-40s %d waves, %3d workgroups: %.3f us per 48 KiB tile = %.1f GB/s per CU\n", names[mode], threads / 64, grid,
                       ms * 1e3 / ntiles, 49152.0 / (ms * 1e-3 / ntiles) / 1e9);
            }
    return 0;
}
