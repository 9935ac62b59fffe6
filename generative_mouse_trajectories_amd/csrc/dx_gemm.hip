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

// ---------------------------------------------------------------------------------------------------------
// The hot shapes (N a multiple of 128, K <= 512: the H = 128 layers and the encoder) keep their weight block
// RESIDENT: a workgroup owns one 128-column block of one direction's W at a time - 4 waves x 32 columns x K rows =
// K/2 VGPRs a lane, loaded once per pass - and walks its tiles with nothing but dA flowing: global -> registers
// (one float4 ahead of each group of MFMAs) -> LDS (double-buffered, written between the MFMAs of the previous tile)
// -> MFMA.  No weight traffic and no vmcnt stall inside a tile.  Directions are PASSES of the same workgroup over the
// same tiles: pass d > 0 starts its accumulators from the partial dX the workgroup itself stored in pass d - 1
// (prefetched a tile ahead), so there is still one launch and no cross-workgroup dependency.  One wave per SIMD (the
// weights fill the register file), like lstm_bwd_kernel.
//
// Round 3: the WEIGHTS are the MFMA's A operand (A[m = output column][k] - the same register contents as the B fragment
// B[k][n] of the other form) and the dA tile its B operand, B[k][n = row]: the contraction yields dX TRANSPOSED in the
// accumulators, lane = row, four consecutive registers = four consecutive output columns.  What that buys with one wave
// per SIMD, where every issue slot shows: the tile sits in LDS ROW-major (pitch K + 2 floats: the 64 lanes of an operand
// read - 32 rows x 2 k - fall into 64 different banks), so staging is two ds_write_b64 per float4 instead of four
// transposing scalar writes; the epilogue's stores, the partial sums of the previous pass and the activation values
// move as float4 (4 instead of 16 vector-memory instructions each per tile).
template <int K, int N>
__global__ __launch_bounds__(256, 1)
void dx_gemm_res_kernel(const float* __restrict__ dA, const float* __restrict__ w_pk, float* __restrict__ dX,
                        const float* __restrict__ act, float slope, long long R, int ndir, long long n_tiles)
{
    constexpr int NT = 256, KS = K / 2, KQ = K / 4;
    constexpr int XS = (GMT_TM * KQ) / NT;              // float4 of dA per thread and tile (K = 512: 16)
    constexpr int NGRP = KS / 8;                        // MFMA groups per tile (8 k-steps each)
    constexpr int HALF = NGRP / 2;
    constexpr int LDR = K + 2;                          // LDS row pitch (floats)
    static_assert(XS <= HALF, "one staged float4 per group");
    static_assert(NT % KQ == 0, "staging map: whole rows per pass of the threads");
    constexpr int n_nb = N / 128;
    extern __shared__ float smem[];                     // [2][32][LDR]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    // The n_nb column blocks of one worker read the SAME dA tiles.  Workgroups go to the 8 XCDs round-robin
    // (blockIdx % 8), each XCD with an L2 of its own: the blocks of a worker are therefore given indices 8 apart - same
    // XCD, dispatched back to back - so that the second reader of a tile finds it in that L2 (round 3 counters: with
    // neighbouring indices the fabric carried dA twice, 4.88 GB for 2.16 GB of dA at K512 / N256).
    const long long workers = gridDim.x / n_nb;
    const bool xcd_pairs = (n_nb > 1) && (workers % 8 == 0);
    const int nb = xcd_pairs ? (int)((blockIdx.x / 8) % n_nb) : (int)(blockIdx.x % n_nb);
    const long long worker = xcd_pairs ? (long long)(blockIdx.x / (8 * n_nb)) * 8 + blockIdx.x % 8 : blockIdx.x / n_nb;
    if (worker >= n_tiles) return;
    const long long last_tile = worker + ((n_tiles - 1 - worker) / workers) * workers;   // this worker's final tile
    // addressing: one uniform base per tile (scalar registers) + ONE per-lane offset per buffer, laundered at each use
    // so that `offset + constant` is recomputed instead of being kept in a register per constant
    const unsigned c_off = (unsigned)(lo * N + nb * 128 + wave * 32 + 4 * hi);            // within a tile of dX / act: row lo, + 8 q
    const unsigned s_off = (unsigned)((tid / KQ) * LDR + 4 * (tid % KQ));                 // within an LDS buffer

    float4 xs[XS];
    float4 cin[4], actv[4];
    for (int d = 0; d < ndir; ++d) {
        float wr[1][KS];
        gmt_load_resident<1, KS>(wr, reinterpret_cast<const float4*>(w_pk) + ((size_t)d * (N / 32) + nb * 4 + wave) * (K / 8) * 64, lane);
        const float* Ad = dA + (size_t)d * R * K;
        const bool first = (d == 0), final = (d == ndir - 1);
        auto a_load1 = [&](long long tile, int j) {
            const float4* src = reinterpret_cast<const float4*>(Ad + (size_t)tile * GMT_TM * K) + j * NT;   // uniform
            xs[j] = src[tid];
        };
        auto a_store1 = [&](float* buf, int j) {
            // item tid + j NT: row (tid + j NT) / KQ, float4 column (tid + j NT) % KQ; rows are 8-byte aligned
            float2* p = reinterpret_cast<float2*>(buf + gmt_opaque_u(s_off) + j * (NT / KQ) * LDR);
            p[0] = make_float2(xs[j].x, xs[j].y);
            p[1] = make_float2(xs[j].z, xs[j].w);
        };
        // prologue: tile 0 -> LDS[0], tile 1 -> registers, partial sums of tile 0
        long long tile = worker;
#pragma unroll
        for (int j = 0; j < XS; ++j) a_load1(tile, j);
#pragma unroll
        for (int j = 0; j < XS; ++j) a_store1(smem, j);
        {
            const long long t1 = tile + workers <= last_tile ? tile + workers : last_tile;
#pragma unroll
            for (int j = 0; j < XS; ++j) a_load1(t1, j);
        }
        {
            const float* cb = dX + (size_t)tile * GMT_TM * N;
#pragma unroll
            for (int q = 0; q < 4; ++q) cin[q] = first ? make_float4(0.f, 0.f, 0.f, 0.f) : gmt_ld4(cb + c_off + 8 * q);
        }
        __syncthreads();
        int cur = 0;
        for (; tile <= last_tile; tile += workers) {
            const long long t1 = tile + workers <= last_tile ? tile + workers : last_tile;
            const long long t2 = t1 + workers <= last_tile ? t1 + workers : last_tile;
            const float* a_cur = smem + (size_t)cur * GMT_TM * LDR;
            float* a_nxt = smem + (size_t)(cur ^ 1) * GMT_TM * LDR;
            // accumulator register r = 4 q + e holds dX[row lo][column 8 q + 4 hi + e] of this wave's 32 columns
            f32x16 acc, alt;
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[4 * q] = cin[q].x; acc[4 * q + 1] = cin[q].y; acc[4 * q + 2] = cin[q].z; acc[4 * q + 3] = cin[q].w; }
#pragma unroll
            for (int r = 0; r < 16; ++r) alt[r] = 0.f;
            float* c0 = dX + (size_t)tile * GMT_TM * N;                // uniform bases
            const float* c1 = dX + (size_t)t1 * GMT_TM * N;
            const float* ab = act ? act + (size_t)tile * GMT_TM * N : dX;
            const float* bp = a_cur + lo * LDR + hi;                   // B operand: dA[row lo][k = 2 ks + hi]
            float b[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[0][j] = bp[2 * j];
#pragma unroll
            for (int grp = 0; grp < NGRP; ++grp) {
                if (grp + 1 < NGRP) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[(grp + 1) & 1][j] = bp[2 * ((grp + 1) * 8 + j)];
                }
                // first half of the groups: tile t1 (in registers since the previous trip) -> the other LDS buffer,
                // and this tile's activation values; second half: tile t2 -> registers, tile t1's partial sums
                if (grp < HALF) {
#pragma unroll
                    for (int j = 0; j < XS; ++j)
                        if ((j * HALF) / XS == grp) a_store1(a_nxt, j);
                    if (final && act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if ((q * HALF) / 4 == grp) actv[q] = gmt_ld4(ab + gmt_opaque_u(c_off) + 8 * q);
                    }
                } else {
                    const int g2 = grp - HALF;
#pragma unroll
                    for (int j = 0; j < XS; ++j)
                        if ((j * HALF) / XS == g2) a_load1(t2, j);
                    if (!first) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if ((q * HALF) / 4 == g2) cin[q] = gmt_ld4(c1 + gmt_opaque_u(c_off) + 8 * q);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j & 1) alt = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[0][grp * 8 + j], b[grp & 1][j], alt, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[0][grp * 8 + j], b[grp & 1][j], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                float* cw = c0 + gmt_opaque_u(c_off);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(acc[4 * q] + alt[4 * q], acc[4 * q + 1] + alt[4 * q + 1],
                                           acc[4 * q + 2] + alt[4 * q + 2], acc[4 * q + 3] + alt[4 * q + 3]);
                    if (final && act) {
                        v.x *= (actv[q].x > 0.f) ? 1.f : slope; v.y *= (actv[q].y > 0.f) ? 1.f : slope;
                        v.z *= (actv[q].z > 0.f) ? 1.f : slope; v.w *= (actv[q].w > 0.f) ? 1.f : slope;
                    }
                    *reinterpret_cast<float4*>(cw + 8 * q) = v;
                }
            }
            __syncthreads();                            // every wave is done reading a_cur and writing a_nxt
            cur ^= 1;
        }
    }
}

template <int K, int N>
int launch_dx_res(const float* dA, const float* w_pk, float* dX, const float* act, float slope, long long R,
                  int ndir, hipStream_t stream)
{
    const size_t lds = (size_t)2 * GMT_TM * (K + 2) * sizeof(float);
    auto k = dx_gemm_res_kernel<K, N>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GMT_ERR_HIP;
    const long long n_tiles = R / GMT_TM;
    constexpr int n_nb = N / 128;
    long long workers = 256 / n_nb;
    if (workers > n_tiles) workers = n_tiles;
    hipLaunchKernelGGL(k, dim3((unsigned)(workers * n_nb)), dim3(256), lds, stream, dA, w_pk, dX, act, slope, R, ndir, n_tiles);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
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

extern "C" int gmt_dx_gemm_resident(int K, int N)
{
    // (K, N) = (4H, H), (4H, 2H) of the H = 128 / 64 / 32 layers where N % 128 == 0, and the encoder's (H, H)
    return ((K == 512 && (N == 128 || N == 256)) || (K == 256 && N == 128) || (K == 128 && N == 128) ||
            (K == 256 && N == 256)) ? 1 : 0;
}

extern "C" int gmt_dx_gemm(const float* dA, const float* w_pk, float* dX, const float* act, float slope,
                           long long R, int K, int N, int ndir, hipStream_t stream)
{
    GMT_CHECK_ARG(dA && w_pk && dX && R >= GMT_TM && (R % GMT_TM) == 0 && ndir >= 1 && ndir <= 2);
    GMT_CHECK_ARG((((uintptr_t)dA | (uintptr_t)w_pk) & 15) == 0);
    if (gmt_dx_gemm_resident(K, N)) {
        if (K == 512 && N == 128) return launch_dx_res<512, 128>(dA, w_pk, dX, act, slope, R, ndir, stream);
        if (K == 512 && N == 256) return launch_dx_res<512, 256>(dA, w_pk, dX, act, slope, R, ndir, stream);
        if (K == 256 && N == 128) return launch_dx_res<256, 128>(dA, w_pk, dX, act, slope, R, ndir, stream);
        if (K == 256 && N == 256) return launch_dx_res<256, 256>(dA, w_pk, dX, act, slope, R, ndir, stream);
        return launch_dx_res<128, 128>(dA, w_pk, dX, act, slope, R, ndir, stream);
    }
    switch (N) {
        case 32: return launch_dx_k<1, 1>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        case 64: return launch_dx_k<2, 1>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        case 128: return launch_dx_k<4, 1>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        case 256: return launch_dx_k<4, 2>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        case 512: return launch_dx_k<4, 4>(K, dA, w_pk, dX, act, slope, R, ndir, stream);
        default: return GMT_ERR_UNSUPPORTED;
    }
}
