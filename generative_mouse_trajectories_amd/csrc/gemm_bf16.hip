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

// EPI: 0 = +bias (or nothing), 1 = lrelu(+bias), 2 = * lrelu'(act).  8 waves (2 along the weights x 4 along the rows, two per
// SIMD so that one computes while the other waits at the K-tile barrier), each 128 x 64 of the 256 x 256 tile.
constexpr int GEMM_THREADS = 512;

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS)
void gemm_nt_kernel(const __bf16* __restrict__ A, long long lda, const __bf16* __restrict__ W, long long ldw,
                    __bf16* __restrict__ C, long long ldc, const float* __restrict__ bias,
                    const __bf16* __restrict__ act, long long ldact, float slope, int n_p, long long n_q, int nk)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tp; long long tq;
    tile_of_block(blockIdx.x, gridDim.x, n_p, n_q, tp, tq);
    f32x16 acc[4][2];
    zero_acc<4, 2>(acc);
    mainloop<4, 2, 4>(W + (long long)tp * TP * ldw, ldw, A + tq * TQ * lda, lda, nk, lds, acc, wave, lane);

    // Epilogue through LDS: a lane's natural store is 8 bytes of one row (16 with its hi partner) - a tile would leave as
    // 8 192 partial-line requests.  The tile is written to LDS as C[m][n] (pitch 528 B) and leaves as full 512-byte rows.
    const int wp = wave >> 2, wq = wave & 3, lo = lane & 31, hi = lane >> 5;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wp * 128 + i * 32 + 8 * g + 4 * hi;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *reinterpret_cast<const f32x4*>(bias + tp * TP + nl);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ml = wq * 64 + j * 32 + lo;
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
    for (int it = 0; it < TQ / 16; ++it) {
        const int row = it * 16 + (tid >> 5);
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
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(GEMM_THREADS), GEMM_LDS, stream, A, lda, W, ldw, C, ldc, bias, act, ldact,
                       slope, n_p, n_q, K / BK);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

// ---- weight gradients: dW[n][i] = sum_r dA[r][n] X[r][i] over the L*B row-steps ("TN": the contraction index is the
// slow one of both operands).  The [64 r][256 col] tiles go to LDS row-major by LDS-DMA; the MFMA fragments (8 consecutive
// r of one column per lane) come out of gfx950's transposing LDS read, ds_read_b64_tr_b16 (4 r x 16 columns per 16-lane
// group; cdna_hip_programming.md T10).  16-byte chunk c of row r sits at c ^ ((r & 3) << 2), so the four rows a 32-lane
// half reads (64 B each) fall on different banks.  Split over row ranges (slabs), fp32 partial sums, reduced by the caller:
// deterministic, no float atomics.
constexpr int WG_ROWB = 512;                                   // bytes per LDS row: 256 bf16 columns
constexpr int WG_STAGE = 2 * BK * WG_ROWB;                     // dA tile + X tile: 64 KiB
constexpr int WG_LDS = 2 * WG_STAGE;

__device__ __forceinline__ void stage_rows(const __bf16* __restrict__ g, long long ld, char* lds_tile, int wave, int lane)
{
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {                          // 8 rows (4 waves x 2) per round
        const int row = j * 8 + wave * 2 + (lane >> 5);
        const int chunk = gmt_bf16_maps::wg_src_chunk(row, lane & 31);
        const __bf16* src = g + (long long)row * ld + chunk * 8;
        char* dst = lds_tile + (j * 8 + wave * 2) * WG_ROWB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// fragment of column block cb (32 columns) for k-step s (16 rows): lane (lo, hi) gets rows 16 s + 8 hi + 0..7 of column lo
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int col0, int s, int lane)
{
    s16x4 h[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int row, col;                                           // the row and first of 4 columns this lane addresses
        gmt_bf16_maps::wg_tr_lane(lane, col0, s, half, row, col);
        const char* a = tile + gmt_bf16_maps::wg_addr(row, col);
        h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
    }
    s16x8 v = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(NTHREADS)
void wgrad_tn_kernel(const __bf16* __restrict__ dA, long long lda, const __bf16* __restrict__ X, long long ldx,
                     float* __restrict__ slabs, long long M, int N, int I, int n_p, int n_q, long long rows_per_slab)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // slabs of one row range on one XCD (blockIdx % 8): its dA / X rows are shared by the tiles in that L2
    const int tiles = n_p * n_q;
    const int slab = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int tp = tile % n_p, tq = tile / n_p;
    const long long r0 = slab * rows_per_slab;
    long long r1 = r0 + rows_per_slab;
    if (r1 > M) r1 = M;
    const int nk = r1 > r0 ? (int)((r1 - r0) / BK) : 0;
    const __bf16* Ag = dA + r0 * lda + (long long)tp * 256;
    const __bf16* Xg = X + r0 * ldx + (long long)tq * 256;

    f32x16 acc[4][4];
    zero_acc<4, 4>(acc);
    const int wp = wave >> 1, wq = wave & 1;
    if (nk > 0) {
        stage_rows(Ag, lda, lds, wave, lane);
        stage_rows(Xg, ldx, lds + BK * WG_ROWB, wave, lane);
    }
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* cur = lds + (t & 1) * WG_STAGE;
        if (t + 1 < nk) {
            char* nxt = lds + ((t + 1) & 1) * WG_STAGE;
            stage_rows(Ag + (long long)(t + 1) * BK * lda, lda, nxt, wave, lane);
            stage_rows(Xg + (long long)(t + 1) * BK * ldx, ldx, nxt + BK * WG_ROWB, wave, lane);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 pf[4], qf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[i] = tr_frag(cur, wp * 128 + i * 32, s, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) qf[j] = tr_frag(cur + BK * WG_ROWB, wq * 128 + j * 32, s, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
        }
    }
    const int lo = lane & 31, hi = lane >> 5;
    float* out = slabs + (long long)slab * N * I;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = tq * 256 + wq * 128 + j * 32 + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = tp * 256 + wp * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                out[(long long)n * I + col] = acc[i][j][r];
            }
        }
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

extern "C" int gmt_bf16_wgrad_tn(const void* dA, long long lda, const void* X, long long ldx, float* slabs,
                                 long long M, int N, int I, int n_slabs, hipStream_t stream)
{
    GMT_CHECK_ARG(dA && X && slabs && M > 0 && N > 0 && I > 0 && n_slabs >= 1);
    GMT_CHECK_ARG(M % BK == 0 && N % 256 == 0 && I % 256 == 0 && lda >= N && ldx >= I && (lda % 8) == 0 && (ldx % 8) == 0);
    if (hipFuncSetAttribute((const void*)wgrad_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS) != hipSuccess)
        return GMT_ERR_HIP;
    const int n_p = N / 256, n_q = I / 256;
    long long rows_per_slab = ((M / BK + n_slabs - 1) / n_slabs) * BK;
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3((unsigned)(n_p * n_q * n_slabs)), dim3(NTHREADS), WG_LDS, stream,
                       (const __bf16*)dA, lda, (const __bf16*)X, ldx, slabs, M, N, I, n_p, n_q, rows_per_slab);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
