// Weight-gradient contraction of the recurrent layers:  dW[n][i] = sum_r dA[r][n] * X[r][i]
// over R = L*B row-steps (K ~ 5e5, N = 4H = 512, I = 128 / 256) - the "plain GEMM over time"
// that autograd runs for RecurrentGAN.py:1021 / 1109.  Its shape (tiny output, huge K) is
// the worst case of a tiled library GEMM, so it is a split-K fp32-MFMA kernel here:
//   * a workgroup owns a 256 (n) x 128 (i) output tile for one slab of rows and keeps it in
//     128 accumulator registers per lane (4 waves x [64 x 128]);
//   * both operands are row-major with the contraction index r as the ROW, which is exactly
//     the k-major order the MFMA fragments want: staging is a straight float4 copy
//     global -> registers -> LDS (no transpose), issued one chunk ahead (T14 split);
//   * partial tiles go to a [KSPLIT][N][I] slab buffer and a second pass sums them in a fixed
//     order (deterministic; float atomics would make gradients run-to-run different);
//   * the column sums of dA (= bias gradient) are accumulated by the staging threads for free.
#include "lstm_tile.h"

namespace {

constexpr int WG_TN = 256, WG_TI = 128, WG_RC = 16;     // tile n, tile i, rows per chunk
constexpr int WG_LDA = WG_TN + 4, WG_LDX = WG_TI + 4;   // LDS pitches (floats): +4 keeps float4 alignment

__global__ __launch_bounds__(256, 2)
void wgrad_kernel(const float* __restrict__ dA, long long lda, const float* __restrict__ X, long long ldx,
                  float* __restrict__ slab, float* __restrict__ db_slab, int N, int I, long long R, int ksplit)
{
    __shared__ float sA[2][WG_RC][WG_LDA];
    __shared__ float sX[2][WG_RC][WG_LDX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    const int n_tiles = N / WG_TN, i_tiles = I / WG_TI;
    int bid = blockIdx.x;
    const int it = bid % i_tiles; bid /= i_tiles;
    const int nt = bid % n_tiles; bid /= n_tiles;
    const int ks = bid;
    const long long n_chunks = R / WG_RC;
    const long long cpk = (n_chunks + ksplit - 1) / ksplit;
    const long long c0 = (long long)ks * cpk;
    long long c1 = c0 + cpk; if (c1 > n_chunks) c1 = n_chunks;

    // staging map: A chunk = 16 rows x 64 float4, X chunk = 16 rows x 32 float4
    const int a_c4 = tid & 63, a_r0 = tid >> 6;         // rows a_r0 + 4j, j = 0..3
    const int x_c4 = tid & 31, x_r0 = tid >> 5;         // rows x_r0 + 8j, j = 0..1
    const float* gA = dA + (long long)nt * WG_TN + 4 * a_c4;
    const float* gX = X + (long long)it * WG_TI + 4 * x_c4;
    float4 ra[WG_RC / 4], rx[WG_RC / 8];
    float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto g_load = [&](long long c) {
        const long long r = c * WG_RC;
#pragma unroll
        for (int j = 0; j < WG_RC / 4; ++j) ra[j] = *reinterpret_cast<const float4*>(gA + (r + a_r0 + 4 * j) * lda);
#pragma unroll
        for (int j = 0; j < WG_RC / 8; ++j) rx[j] = *reinterpret_cast<const float4*>(gX + (r + x_r0 + 8 * j) * ldx);
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WG_RC / 4; ++j) {
            *reinterpret_cast<float4*>(&sA[buf][a_r0 + 4 * j][4 * a_c4]) = ra[j];
            colsum.x += ra[j].x; colsum.y += ra[j].y; colsum.z += ra[j].z; colsum.w += ra[j].w;
        }
#pragma unroll
        for (int j = 0; j < WG_RC / 8; ++j) *reinterpret_cast<float4*>(&sX[buf][x_r0 + 8 * j][4 * x_c4]) = rx[j];
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    int cur = 0;
    if (c0 < c1) { g_load(c0); s_store(0); }
    __syncthreads();
    for (long long c = c0; c < c1; ++c) {
        if (c + 1 < c1) g_load(c + 1);
        const float* pa = &sA[cur][hi][wave * 64 + lo];
        const float* px = &sX[cur][hi][lo];
#pragma unroll
        for (int s = 0; s < WG_RC / 2; ++s) {
            const float a0 = pa[2 * s * WG_LDA], a1 = pa[2 * s * WG_LDA + 32];
            float b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = px[2 * s * WG_LDX + 32 * q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[q], acc[0][q], 0, 0, 0);
                acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[q], acc[1][q], 0, 0, 0);
            }
        }
        if (c + 1 < c1) s_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // partial tile -> slab[ks][n][i]
    float* out = slab + ((long long)ks * N + (long long)nt * WG_TN + wave * 64) * I + (long long)it * WG_TI + lo;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(long long)(a * 32 + gmt_acc_row(r, lane)) * I + 32 * q] = acc[a][q][r];
    // bias gradient: column sums of this slab of dA (only the it == 0 workgroups write)
    if (db_slab && it == 0) {
        __syncthreads();
        float* red = &sA[0][0][0];                      // reuse: [4][256]
        red[a_r0 * WG_TN + 4 * a_c4 + 0] = colsum.x; red[a_r0 * WG_TN + 4 * a_c4 + 1] = colsum.y;
        red[a_r0 * WG_TN + 4 * a_c4 + 2] = colsum.z; red[a_r0 * WG_TN + 4 * a_c4 + 3] = colsum.w;
        __syncthreads();
        const float v = red[tid] + red[WG_TN + tid] + red[2 * WG_TN + tid] + red[3 * WG_TN + tid];
        db_slab[(long long)ks * N + (long long)nt * WG_TN + tid] = v;
    }
}

// out[j] (+)= sum_k slab[k][j]   (fixed order); float4 lanes
__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long long n, int ksplit,
                                   int accumulate)
{
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 s = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < ksplit; ++k) {
            const float4 v = reinterpret_cast<const float4*>(slab + (long long)k * n)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

}  // namespace

extern "C" int gmt_wgrad_workspace_floats(int N, int I, long long R, int* ksplit_out, long long* floats_out)
{
    GMT_CHECK_ARG(N >= WG_TN && (N % WG_TN) == 0 && I >= WG_TI && (I % WG_TI) == 0 && R >= WG_RC && (R % WG_RC) == 0);
    const int tiles = (N / WG_TN) * (I / WG_TI);
    int ks = (2 * 256 + tiles - 1) / tiles;             // ~2 workgroups per CU
    const long long n_chunks = R / WG_RC;
    if (ks > n_chunks) ks = (int)n_chunks;
    if (ks < 1) ks = 1;
    if (ksplit_out) *ksplit_out = ks;
    if (floats_out) *floats_out = (long long)ks * N * I + (long long)ks * N;
    return GMT_OK;
}

extern "C" int gmt_wgrad(const float* dA, long long lda, const float* X, long long ldx, long long R, int N, int I,
                         float* dW, float* db, int accumulate, float* workspace, long long workspace_floats,
                         hipStream_t stream)
{
    GMT_CHECK_ARG(dA && X && dW && workspace);
    GMT_CHECK_ARG((lda % 4) == 0 && (ldx % 4) == 0 && (((uintptr_t)dA | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)workspace) & 15) == 0);
    int ks = 0; long long need = 0;
    const int rc = gmt_wgrad_workspace_floats(N, I, R, &ks, &need);
    if (rc != GMT_OK) return rc;
    if (workspace_floats < need) return GMT_ERR_WORKSPACE;
    float* slab = workspace;
    float* db_slab = db ? workspace + (long long)ks * N * I : nullptr;
    const int grid = ks * (N / WG_TN) * (I / WG_TI);
    hipLaunchKernelGGL(wgrad_kernel, dim3(grid), dim3(256), 0, stream, dA, lda, X, ldx, slab, db_slab, N, I, R, ks);
    GMT_LAUNCH_CHECK();
    const long long n = (long long)N * I;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(gmt_grid_1d(n / 4, 256)), dim3(256), 0, stream, slab, dW, n, ks, accumulate);
    GMT_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(gmt_grid_1d(N / 4, 256)), dim3(256), 0, stream, db_slab, db,
                           (long long)N, ks, accumulate);
        GMT_LAUNCH_CHECK();
    }
    return GMT_OK;
}
