// Input projection of an LSTM layer for ALL timesteps at once:
//   pregates[d][r][4u + g] = bias[d][g H + u] + sum_k x[r][k] * W_ih_d[g H + u][k]      r < R = L*B row-steps
// - the time-parallel half of the gate computation of torch.nn.LSTM (RG.py:720-727, called at RG.py:772).
//
// The persistent layer kernels fuse this contraction into their time loop (lstm_fwd_kernel IX > 0), which is
// right when the batch fills the machine: the MFMA work is the same wherever it runs.  The gradient-penalty
// sub-batch (RG.py:969-989: a quarter of the batch) fills a quarter of the CUs, and there every FLOP taken OUT of
// the time loop runs on the whole machine instead: this kernel is lstm_fwd_kernel's x-part as a grid over 32-row
// tiles (same fragment-major W_ih, same accumulator layout, both directions from one staged tile), and the
// persistent kernels then take `pregates` (IX = 0) and keep only the recurrent h W_hh^T part in the loop.
#include "lstm_tile.h"

namespace {

template <int NW, int IX>
__global__ __launch_bounds__(NW * 64)
void xproj_kernel(const float* __restrict__ x, const float* __restrict__ w_ih_pk, const float* __restrict__ bias,
                  float* __restrict__ pre, long long R, int ndir, long long n_tiles)
{
    constexpr int H = NW * 32, I = IX * H, NT = NW * 64;
    constexpr int KQ = I / 4;                           // float4 per row
    constexpr int XS = (GMT_TM * KQ) / NT;              // float4 per thread and tile (= 4 IX)
    static_assert((GMT_TM * KQ) % NT == 0 && NT % KQ == 0, "staging map");
    extern __shared__ float smem[];
    float* aT = smem;                                   // [I][LDA]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    const int col = wave * 32 + lo;

    float4 xs[XS];
    auto a_load = [&](long long tile) {
        const float4* src = reinterpret_cast<const float4*>(x + (size_t)tile * GMT_TM * I);
#pragma unroll
        for (int j = 0; j < XS; ++j) xs[j] = (src + j * NT)[tid];
    };
    auto a_store = [&]() {
        float* d0 = aT + (size_t)(4 * (tid % KQ)) * GMT_LDA + tid / KQ;
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            float* d = d0 + j * (NT / KQ);
            d[0] = xs[j].x; d[GMT_LDA] = xs[j].y; d[2 * GMT_LDA] = xs[j].z; d[3 * GMT_LDA] = xs[j].w;
        }
    };
    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;
    a_load(tile);
    while (tile < n_tiles) {
        a_store();
        __syncthreads();
        const long long tn = tile + gridDim.x < n_tiles ? tile + gridDim.x : tile;   // last trip re-reads its own tile
        a_load(tn);
        __builtin_amdgcn_sched_barrier(0);
        for (int d = 0; d < ndir; ++d) {
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float b = bias ? bias[(size_t)d * 4 * H + g * H + col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = b;
            }
            const float4* w = reinterpret_cast<const float4*>(w_ih_pk) + ((size_t)d * NW + wave) * (I / 8) * 4 * 64;
            gmt_mma_stream<4>(acc, aT, w, I, lane);
            float* o = pre + (((size_t)d * R + (size_t)tile * GMT_TM + 4 * hi) * (4 * H)) + 4 * col;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gmt_st4(o + (size_t)((r & 3) + 8 * (r >> 2)) * (4 * H), acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        }
        __syncthreads();
        tile += gridDim.x;
    }
}

template <int NW, int IX>
int launch_xproj(const float* x, const float* w_ih_pk, const float* bias, float* pre, long long R, int ndir, hipStream_t stream)
{
    constexpr int I = IX * NW * 32;
    const size_t lds = (size_t)I * GMT_LDA * sizeof(float);
    auto k = xproj_kernel<NW, IX>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GMT_ERR_HIP;
    const long long n_tiles = R / GMT_TM;
    long long grid = 512;                               // two workgroups per CU: one's staging under the other's MFMAs
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NW * 64), lds, stream, x, w_ih_pk, bias, pre, R, ndir, n_tiles);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

}  // namespace

extern "C" int gmt_lstm_xproj(const float* x, const float* w_ih_pk, const float* bias, float* pregates,
                              long long R, int I, int H, int ndir, hipStream_t stream)
{
    GMT_CHECK_ARG(x && w_ih_pk && pregates && R >= GMT_TM && (R % GMT_TM) == 0 && (ndir == 1 || ndir == 2));
    GMT_CHECK_ARG((((uintptr_t)x | (uintptr_t)w_ih_pk | (uintptr_t)pregates) & 15) == 0);
    GMT_CHECK_ARG(I == H || I == 2 * H);
#define GMT_CASE(NW)                                                                                             \
    case NW * 32:                                                                                                \
        return I == H ? launch_xproj<NW, 1>(x, w_ih_pk, bias, pregates, R, ndir, stream)                         \
                      : launch_xproj<NW, 2>(x, w_ih_pk, bias, pregates, R, ndir, stream);
    switch (H) {
        GMT_CASE(1) GMT_CASE(2) GMT_CASE(4) GMT_CASE(8)
        default: return GMT_ERR_UNSUPPORTED;
    }
#undef GMT_CASE
}
