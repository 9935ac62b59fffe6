// Dense bf16 GEMMs of the config-3 path (H = 512): the contractions that do NOT sit on the recurrence and are therefore
// hoisted out of the time loop and run over all L*B row-steps at once -
//   pre  = x W_ih^T + b       (nn.LSTM's input half of the gates, RG.py:720-727, both directions in one call),
//   e    = lrelu(h1 W2^T + b) (feature encoder, RG.py:712-717),
//   dX   = dA W_ih            (autograd's mirror of the above; optional LeakyReLU-derivative epilogue).
// C[m][n] = sum_k A[m][k] W[n][k]; tile / staging / lane maps: bf16_tile.h.
#include "bf16_tile.h"

namespace {

using namespace gmt_bf16;

constexpr int C_PITCH = TP * 2 + 16;                           // bytes per row of the staged output tile
constexpr int GEMM_LDS = TQ * C_PITCH > LDS_BYTES ? TQ * C_PITCH : LDS_BYTES;

// EPI: 0 = +bias (or nothing), 1 = lrelu(+bias), 2 = * lrelu'(act)
template <int EPI>
__global__ __launch_bounds__(NTHREADS)
void gemm_nt_kernel(const __bf16* __restrict__ A, long long lda, const __bf16* __restrict__ W, long long ldw,
                    __bf16* __restrict__ C, long long ldc, const float* __restrict__ bias,
                    const __bf16* __restrict__ act, long long ldact, float slope, int n_p, long long n_q, int nk)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tp; long long tq;
    tile_of_block(blockIdx.x, gridDim.x, n_p, n_q, tp, tq);
    f32x16 acc[4][4];
    zero_acc(acc);
    mainloop<4, 4>(W + (long long)tp * TP * ldw, ldw, A + tq * TQ * lda, lda, nk, lds, acc, wave, lane);

    // Epilogue through LDS: a lane's natural store is 8 bytes of one row (16 with its hi partner) - a tile would leave as
    // 8 192 partial-line requests.  The tile is written to LDS as C[m][n] (pitch 528 B) and leaves as full 512-byte rows.
    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wp * 128 + i * 32 + 8 * g + 4 * hi;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *reinterpret_cast<const f32x4*>(bias + tp * TP + nl);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ml = wq * 128 + j * 32 + lo;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bv[e];
                if (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
                }
                *reinterpret_cast<bf16x4*>(lds + ml * C_PITCH + nl * 2) = pack4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    __syncthreads();
    const int tid = threadIdx.x;
    const long long m0 = tq * TQ;
    const int n0 = tp * TP + (tid & 31) * 8;
#pragma unroll 4
    for (int it = 0; it < TQ / 8; ++it) {
        const int row = it * 8 + (tid >> 5);
        bf16x8 v = *reinterpret_cast<const bf16x8*>(lds + row * C_PITCH + (tid & 31) * 16);
        if (EPI == 2) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(act + (m0 + row) * ldact + n0);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)a[e] > 0.f ? v[e] : (__bf16)(slope * (float)v[e]);
        }
        *reinterpret_cast<bf16x8*>(C + (m0 + row) * ldc + n0) = v;
    }
}

template <int EPI>
int launch_nt(const __bf16* A, long long lda, const __bf16* W, long long ldw, __bf16* C, long long ldc, const float* bias,
              const __bf16* act, long long ldact, float slope, long long M, int N, int K, hipStream_t stream)
{
    auto k = gemm_nt_kernel<EPI>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS) != hipSuccess)
        return GMT_ERR_HIP;
    const int n_p = N / TP;
    const long long n_q = M / TQ;
    const long long nblocks = n_p * n_q;
    if (nblocks > 0x7fffffffLL) return GMT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(NTHREADS), GEMM_LDS, stream, A, lda, W, ldw, C, ldc, bias, act, ldact,
                       slope, n_p, n_q, K / BK);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

}  // namespace

extern "C" int gmt_bf16_gemm_nt(const void* A, long long lda, const void* W, long long ldw, void* C, long long ldc,
                                const float* bias, const void* act, long long ldact, int epilogue, float slope,
                                long long M, int N, int K, hipStream_t stream)
{
    GMT_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0);
    GMT_CHECK_ARG(M % TQ == 0 && N % TP == 0 && K % BK == 0);
    GMT_CHECK_ARG(lda >= K && ldw >= K && ldc >= N && (lda % 8) == 0 && (ldw % 8) == 0 && (ldc % 8) == 0);
    GMT_CHECK_ARG(epilogue >= 0 && epilogue <= 2 && (epilogue != 2 || (act && ldact >= N && (ldact % 8) == 0)));
    const __bf16* a = (const __bf16*)A; const __bf16* w = (const __bf16*)W; __bf16* c = (__bf16*)C;
    const __bf16* ac = (const __bf16*)act;
    switch (epilogue) {
        case 0: return launch_nt<0>(a, lda, w, ldw, c, ldc, bias, ac, ldact, slope, M, N, K, stream);
        case 1: return launch_nt<1>(a, lda, w, ldw, c, ldc, bias, ac, ldact, slope, M, N, K, stream);
        default: return launch_nt<2>(a, lda, w, ldw, c, ldc, bias, ac, ldact, slope, M, N, K, stream);
    }
}
