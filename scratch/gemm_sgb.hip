// Experiment: the 256 x 256 bf16 GEMM tile on 4 waves (one per SIMD, 128 x 128 each) with the K-step's LDS reads and DMA pieces
// interleaved between its 16 MFMAs by sched_group_barrier, against the library's loop.  Timing only (no epilogue of interest).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I generative_mouse_trajectories_amd/csrc scratch/gemm_sgb.hip -o scratch/gemm_sgb
#include "bf16_tile.h"
#include <stdio.h>
#include <vector>
using namespace gmt_bf16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int VARIANT>
__device__ __forceinline__ void loop_sgb(const __bf16* __restrict__ Pg, long long ldp, const __bf16* __restrict__ Qg, long long ldq,
                                         int nk, char* lds, f32x16 (&acc)[4][4], int wave, int lane)
{
    constexpr int STAGE = 512 * 128;
    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    auto piece = [&](int t, char* b, int p) {          // 16 pieces per thread and tile: rows [32 p', +32) of P (p < 8) or Q
        const bool isq = p >= 8;
        const int j = p & 7;
        const int row = j * 32 + wave * 8 + (lane >> 3);
        const int chunk = gmt_bf16_maps::stage_src_chunk(row, lane & 7);
        const __bf16* src = (isq ? Qg + (long long)row * ldq : Pg + (long long)row * ldp) + (long long)t * 64 + chunk * 8;
        char* dst = b + (isq ? 256 * 128 : 0) + (j * 32 + wave * 8) * 128;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < 16; ++p) piece(0, lds, p);
    bf16x8 pf[2][4], qf[2][4];
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* cur = lds + (t & 1) * STAGE;
        char* nxt = lds + ((t + 1) & 1) * STAGE;
        const bool fill = t + 1 < nk;
        const char* pt = cur + (wp * 128 + lo) * 128;
        const char* qt = cur + 256 * 128 + (wq * 128 + lo) * 128;
        auto read_frags = [&](int s, int set) {
            const int off = gmt_bf16_maps::stage_addr(lo, 2 * s + hi) - lo * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[set][i] = *reinterpret_cast<const bf16x8*>(pt + i * 32 * 128 + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) qf[set][j] = *reinterpret_cast<const bf16x8*>(qt + j * 32 * 128 + off);
        };
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) read_frags(s + 1, (s + 1) & 1);
            if (fill && s < 3) {
#pragma unroll
                for (int p = s * 16 / 3; p < (s + 1) * 16 / 3; ++p) piece(t + 1, nxt, p);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[s & 1][i], qf[s & 1][j], acc[i][j], 0, 0, 0);
            if (VARIANT == 1) {
                // 16 MFMAs with one LDS read behind each of the first 8 and one DMA piece behind each of the next 5-6
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            } else {
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

__global__ __launch_bounds__(512) void gemm8_kernel(const __bf16* A, long long lda, const __bf16* W, long long ldw, __bf16* C, long long ldc,
                                                    int n_p, long long n_q, int nk)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tp; long long tq;
    tile_of_block(blockIdx.x, gridDim.x, n_p, n_q, tp, tq);
    f32x16 acc[4][2];
    zero_acc<4, 2>(acc);
    mainloop<4, 2, 4>(W + (long long)tp * 256 * ldw, ldw, A + tq * 256 * lda, lda, nk, lds, acc, wave, lane);
    const int wp = wave >> 2, wq = wave & 3, lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long long m = tq * 256 + wq * 64 + j * 32 + lo;
                const int n = tp * 256 + wp * 128 + i * 32 + 8 * g + 4 * hi;
                *reinterpret_cast<bf16x4*>(C + m * ldc + n) = pack4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            }
}

template <int VARIANT>
__global__ __launch_bounds__(256) void gemm_kernel(const __bf16* A, long long lda, const __bf16* W, long long ldw, __bf16* C, long long ldc,
                                                   int n_p, long long n_q, int nk)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tp; long long tq;
    tile_of_block(blockIdx.x, gridDim.x, n_p, n_q, tp, tq);
    f32x16 acc[4][4];
    zero_acc<4, 4>(acc);
    if (VARIANT == 2) mainloop<4, 4, 2>(W + (long long)tp * 256 * ldw, ldw, A + tq * 256 * lda, lda, nk, lds, acc, wave, lane);
    else loop_sgb<VARIANT>(W + (long long)tp * 256 * ldw, ldw, A + tq * 256 * lda, lda, nk, lds, acc, wave, lane);
    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long long m = tq * 256 + wq * 128 + j * 32 + lo;
                const int n = tp * 256 + wp * 128 + i * 32 + 8 * g + 4 * hi;
                *reinterpret_cast<bf16x4*>(C + m * ldc + n) = pack4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            }
}

int main()
{
    const long long M = 262144; const int N = 4096;
    __bf16 *A, *W, *C;
    CK(hipMalloc(&A, M * 4096 * 2)); CK(hipMalloc(&W, (size_t)N * 4096 * 2)); CK(hipMalloc(&C, M * N * 2));
    {   // random operands (zeros would flatter the clock)
        std::vector<unsigned short> h(1 << 24);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 3) & 0x8000u)); }   // +-[0.5, 2)
        for (size_t off = 0; off < (size_t)M * 4096 * 2; off += h.size() * 2) CK(hipMemcpy((char*)A + off, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (size_t off = 0; off < (size_t)N * 4096 * 2; off += h.size() * 2) CK(hipMemcpy((char*)W + off, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int K : {1024, 4096}) {
        for (int v = 0; v < 4; ++v) {
            auto k = v == 0 ? gemm_kernel<0> : v == 1 ? gemm_kernel<1> : v == 2 ? gemm_kernel<2> : gemm8_kernel;
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            const int n_p = N / 256; const long long n_q = M / 256;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k, dim3((unsigned)(n_p * n_q)), dim3(v == 3 ? 512 : 256), 131072, 0, A, (long long)K, W, (long long)K, C, (long long)N, n_p, n_q, K / 64);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const char* names[4] = {"4 waves, pieces + reads after fences", "4 waves, sched_group_barrier interleave", "4 waves, library loop (burst)", "8 waves, library loop (production)"};
            printf("K=%d  %-42s %.3f ms = %.0f TFLOP/s\n", K, names[v], ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
