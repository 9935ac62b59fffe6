// Input-gradient contraction of a recurrent layer:  dX[r][n] = sum_d sum_k dA[d][r][k] * W_d[k][n]
// over R = L*B row-steps (tall and skinny: R ~ 5e5, K = 4H = 512, N = I = 128 / 256) - what autograd computes
// for the input of torch.nn.LSTM (RG.py:772) and of the feature-encoder Linears (RG.py:712-717, 761) in
// RG.py:1021 / 1109.  The same kernel does the encoder's  d(pre-activation) = (dY W) * LeakyReLU'(act)
// (optional epilogue), so no separate element-wise pass exists.
//
// Structure = the streamed input contraction of lstm_fwd_kernel turned into a kernel of its own:
//   * a workgroup walks 32-row tiles (grid-stride); the tile of dA (32 x K) is staged TRANSPOSED in LDS
//     ([k][33], the MFMA A operand of lstm_tile.h), the next tile's rows are already in flight in registers
//     while the current one is contracted;
//   * the weights stream from L2 in fragment-major order (gmt_pack_weight; all workgroups read the same
//     <= 1 MB), wave w owns columns g * 32 NWD + 32 w + [0, 32), g < NG;
//   * both directions accumulate into the same registers (no second pass over dX), fp32 MFMA 32x32x2;
//   * two workgroups per CU when LDS allows: one's staging hides under the other's MFMAs.
#include "lstm_tile.h"

namespace {

template <int NWD, int NG, int K>
__global__ __launch_bounds__(NWD * 64)
void dx_gemm_kernel(const float* __restrict__ dA, const float* __restrict__ w_pk, float* __restrict__ dX,
                    const float* __restrict__ act, float slope, long long R, int ndir, long long n_tiles)
{
    constexpr int NT = NWD * 64;
    constexpr int N = 32 * NWD * NG;
    constexpr int KQ = K / 4;                           // float4 per row
    constexpr int XS = (GMT_TM * KQ) / NT;              // float4 per thread and tile
    static_assert((GMT_TM * KQ) % NT == 0, "staging map");
    extern __shared__ float smem[];
    float* aT = smem;                                   // [K][LDA]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lo = lane & 31;

    float4 xs[XS];
    // item i = tid + j * NT of a tile: row i / KQ, float4 column i % KQ
    auto a_load = [&](long long tile, int d) {
        const float4* src = reinterpret_cast<const float4*>(dA + ((size_t)d * R + (size_t)tile * GMT_TM) * K);
#pragma unroll
        for (int j = 0; j < XS; ++j) xs[j] = (src + j * NT)[tid];
    };
    auto a_store = [&]() {
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            const int it = tid + j * NT;
            float* d = aT + (size_t)(4 * (it % KQ)) * GMT_LDA + it / KQ;
            d[0] = xs[j].x; d[GMT_LDA] = xs[j].y; d[2 * GMT_LDA] = xs[j].z; d[3 * GMT_LDA] = xs[j].w;
        }
    };

    f32x16 acc[NG];
    auto zero = [&]() {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    };
    zero();
    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;
    a_load(tile, 0);
    while (tile < n_tiles) {
        for (int d = 0; d < ndir; ++d) {
            a_store();
            __syncthreads();
            // the rows contracted next (the next direction of this tile, or the next tile; the last trip re-reads
            // its own: no branch around the loads)
            long long tn = tile; int dn = d + 1;
            if (dn == ndir) { dn = 0; tn = tile + gridDim.x; if (tn >= n_tiles) { tn = tile; dn = d; } }
            a_load(tn, dn);
            __builtin_amdgcn_sched_barrier(0);
            const float4* w = reinterpret_cast<const float4*>(w_pk) + ((size_t)d * NWD + wave) * (K / 8) * NG * 64;
            gmt_mma_stream<NG>(acc, aT, w, K, lane);
            __syncthreads();
        }
        // epilogue: lane holds C[row = acc_row(r)][col = g * 32 NWD + 32 wave + lo]
        const size_t r0 = (size_t)tile * GMT_TM + 4 * hi;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int col = g * 32 * NWD + 32 * wave + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t o = (r0 + (r & 3) + 8 * (r >> 2)) * N + col;
                float v = acc[g][r];
                if (act) v *= (act[o] > 0.f) ? 1.f : slope;
                dX[o] = v;
            }
        }
        zero();
        tile += gridDim.x;
    }
}

template <int NWD, int NG, int K>
int launch_dx(const float* dA, const float* w_pk, float* dX, const float* act, float slope, long long R, int ndir,
              hipStream_t stream)
{
    const size_t lds = (size_t)K * GMT_LDA * sizeof(float);
    if (lds > 160 * 1024 - 256) return GMT_ERR_UNSUPPORTED;
    auto k = dx_gemm_kernel<NWD, NG, K>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GMT_ERR_HIP;
    const long long n_tiles = R / GMT_TM;
    const int per_cu = (2 * lds <= 150 * 1024) ? 2 : 1;
    long long grid = 256LL * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NWD * 64), lds, stream, dA, w_pk, dX, act, slope, R, ndir, n_tiles);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

// K in {N, 2N, 4N}: the encoder's square Linear and the LSTM layers' I = 2H and I = H inputs (K = 4H)
template <int NWD, int NG>
int launch_dx_k(int K, const float* dA, const float* w_pk, float* dX, const float* act, float slope, long long R, int ndir,
                hipStream_t stream)
{
    constexpr int N = 32 * NWD * NG;
    if (K == N) return launch_dx<NWD, NG, N>(dA, w_pk, dX, act, slope, R, ndir, stream);
    if (K == 2 * N) return launch_dx<NWD, NG, 2 * N>(dA, w_pk, dX, act, slope, R, ndir, stream);
    if constexpr (N <= 256) {
        if (K == 4 * N) return launch_dx<NWD, NG, 4 * N>(dA, w_pk, dX, act, slope, R, ndir, stream);
    }
    return GMT_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int gmt_dx_gemm(const float* dA, const float* w_pk, float* dX, const float* act, float slope,
                           long long R, int K, int N, int ndir, hipStream_t stream)
{
    GMT_CHECK_ARG(dA && w_pk && dX && R >= GMT_TM && (R % GMT_TM) == 0 && ndir >= 1 && ndir <= 2);
    GMT_CHECK_ARG((((uintptr_t)dA | (uintptr_t)w_pk) & 15) == 0);
    switch (N) {
        case 32: return launch_dx_k<1, 1>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        case 64: return launch_dx_k<2, 1>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        case 128: return launch_dx_k<4, 1>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        case 256: return launch_dx_k<4, 2>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        case 512: return launch_dx_k<4, 4>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        default: return GMT_ERR_UNSUPPORTED;
    }
}
