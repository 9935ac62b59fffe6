// The two small dense stacks of the step, each direction one launch:
//   * generator condition encoder  (Generator._compute_condition RG.py:469-484, condition_encoder RG.py:440-447) plus
//     the time-invariant gate term g0 = cond . W_ih_l0[:, 3:3+H]^T + b_ih_l0 + b_hh_l0 that the rollout kernels take;
//   * critic output head            (Discriminator.output_layers RG.py:730-737, applied at RG.py:781).
// As ATen ops these are ~25 launches per call (cat, three Linears, two LayerNorms, LeakyReLUs, dropout, a transpose into
// the rollout's unit-major gate order) and as many again under autograd; negligible FLOPs (82 k /
// 180 k per trajectory), so the design target is launch count and latency, not the matrix pipe:
// a workgroup owns 32 rows and ALL columns of every layer (wave w = columns [32 w, 32 w + 32), lstm_tile.h decomposition),
// activations stay in LDS between layers, row-major with an ODD pitch - one layout serves both as the A operand of
// v_mfma_f32_32x32x2_f32 (lane = row: pitch odd -> 32 banks) and for row-wise LayerNorm statistics (lanes = columns).
// Weights are read as they lie in memory, row-major, one dword per lane and k-step (they are L2-resident and tiny; no
// packing launches, any row pitch / column offset - W_ih_l0's condition columns start at column 3 of a 3 + H + 4 pitch).
// The weight gradients are NOT computed here: the backward kernels write the adjoints of the Linear pre-activations
// and the caller contracts them on the split-K kernel (csrc/wgrad.hip), which is deterministic and already tuned.
#include "lstm_tile.h"

namespace {

__device__ __forceinline__ float mlp_lrelu(float v, float s) { return v > 0.f ? v : s * v; }

// acc[j] += A . W_j^T :  A = LDS row-major [32][lda]; tile j's column (lane & 31) is the weight row wrow[j] (this lane's),
// K contraction values from wrow[j][0] (any K: k-steps beyond it contribute zeros); k-step ks contracts
// k = 2 ks + (lane >> 5).  Four k-steps' operands are requested before their MFMAs are issued.
template <int NT_>
__device__ __forceinline__ void mlp_mma_wT(f32x16 (&acc)[NT_], const float* __restrict__ a_lds, int lda,
                                           const float* const (&wrow)[NT_], int K, int lane)
{
    const int hi = lane >> 5, lo = lane & 31;
    const float* ap = a_lds + lo * lda + hi;
    const int nks = (K + 1) >> 1;
    for (int ks0 = 0; ks0 < nks; ks0 += 4) {
        float a[4], b[4][NT_];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 2 * (ks0 + u) + hi;
            const bool in = k < K;
            const float av = ap[in ? 2 * (ks0 + u) : 0];
            a[u] = in ? av : 0.f;
#pragma unroll
            for (int j = 0; j < NT_; ++j) {
                const float v = wrow[j][in ? k : 0];
                b[u][j] = in ? v : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NT_; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][j], acc[j], 0, 0, 0);
    }
}

// acc[j] += A . W[:, tile j] :  B[k][n] = w[rowof(k) * ldw + n],  n = wcol offset (this lane's column of tile 0) + tstride j.
// K even.
template <int NT_, class RowOf>
__device__ __forceinline__ void mlp_mma_w(f32x16 (&acc)[NT_], const float* __restrict__ a_lds, int lda,
                                          const float* __restrict__ wcol, long long ldw, int tstride, int K, int lane, RowOf rowof)
{
    const int hi = lane >> 5, lo = lane & 31;
    const float* ap = a_lds + lo * lda + hi;
#pragma unroll 8
    for (int ks = 0; ks < K / 2; ++ks) {
        const float a = ap[2 * ks];
        const float* wr = wcol + (long long)rowof(2 * ks + hi) * ldw;
        float b[NT_];
#pragma unroll
        for (int j = 0; j < NT_; ++j) b[j] = wr[tstride * j];
#pragma unroll
        for (int j = 0; j < NT_; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
    }
}

// sum over the NP consecutive lanes that share a row (NP a power of two <= 32, groups aligned)
template <int NP>
__device__ __forceinline__ float mlp_part_sum(float v)
{
#pragma unroll
    for (int o = NP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// accumulator tile (32 rows x this wave's 32 columns) -> LDS row-major
__device__ __forceinline__ void mlp_acc_to_lds(const f32x16& acc, float* buf, int ld, int col, int lane)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[gmt_acc_row(r, lane) * ld + col] = acc[r];
}

// Row-wise LayerNorm + LeakyReLU of y [32][H+1] (LDS) -> act [32][H+1] (LDS), thread (row, part) owns 16 columns.
// Rows >= rows_valid leave as zeros.  Saves xhat / rstd / the activation to global when the pointers are given.
template <int NW>
__device__ __forceinline__ void mlp_layer_norm(const float* ybuf, float* abuf, const float* __restrict__ g,
                                               const float* __restrict__ b, float slope, float eps, int row0, int rows_valid,
                                               float* __restrict__ xhat_out, float* __restrict__ rstd_out,
                                               float* __restrict__ act_out, int tid)
{
    constexpr int H = NW * 32, NP = 2 * NW, LD = H + 1;
    const int row = tid / NP, part = tid % NP, c0 = 16 * part;
    const bool valid = row0 + row < rows_valid;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { v[j] = ybuf[row * LD + c0 + j]; s += v[j]; }
    const float mean = mlp_part_sum<NP>(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { v[j] -= mean; q += v[j] * v[j]; }
    const float var = mlp_part_sum<NP>(q) * (1.0f / H);
    const float rstd = valid ? 1.0f / sqrtf(var + eps) : 0.f;
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        v[j] *= rstd;                                                   // xhat (0 on padding rows)
        a[j] = valid ? mlp_lrelu(v[j] * g[c0 + j] + b[c0 + j], slope) : 0.f;
        abuf[row * LD + c0 + j] = a[j];
    }
    const size_t o = (size_t)(row0 + row) * H + c0;
    if (xhat_out) {
#pragma unroll
        for (int j = 0; j < 16; j += 4) gmt_st4(xhat_out + o + j, v[j], v[j + 1], v[j + 2], v[j + 3]);
        if (part == 0) rstd_out[row0 + row] = rstd;
    }
    if (act_out) {
#pragma unroll
        for (int j = 0; j < 16; j += 4) gmt_st4(act_out + o + j, a[j], a[j + 1], a[j + 2], a[j + 3]);
    }
}

// Backward of the same row-wise LayerNorm: dz [32][H+1] (LDS) = adjoint of the LayerNorm OUTPUT (LeakyReLU' applied),
// overwritten in place by the adjoint of its INPUT (each thread reads and writes only its own 16 slots);
// dx = rstd (dz g - mean(dz g) - xhat mean(dz g xhat)).  Also written to dy_out (global).
template <int NW>
__device__ __forceinline__ void mlp_layer_norm_bwd(float* dzbuf, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                   const float* __restrict__ g, int row0, float* __restrict__ dy_out, int tid)
{
    constexpr int H = NW * 32, NP = 2 * NW, LD = H + 1;
    const int row = tid / NP, part = tid % NP, c0 = 16 * part;
    const size_t o = (size_t)(row0 + row) * H + c0;
    float d[16], xh[16];
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        const float4 t = gmt_ld4(xhat + o + j);
        xh[j] = t.x; xh[j + 1] = t.y; xh[j + 2] = t.z; xh[j + 3] = t.w;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        d[j] = dzbuf[row * LD + c0 + j] * g[c0 + j];
        s1 += d[j];
        s2 += d[j] * xh[j];
    }
    const float m1 = mlp_part_sum<NP>(s1) * (1.0f / H), m2 = mlp_part_sum<NP>(s2) * (1.0f / H);
    const float rs = rstd[row0 + row];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        d[j] = rs * (d[j] - m1 - xh[j] * m2);
        dzbuf[row * LD + c0 + j] = d[j];
    }
#pragma unroll
    for (int j = 0; j < 16; j += 4) gmt_st4(dy_out + o + j, d[j], d[j + 1], d[j + 2], d[j + 3]);
}

// per-tile partial LayerNorm parameter gradients: d g[c] = sum_rows dz xhat, d b[c] = sum_rows dz  (dz before mlp_layer_norm_bwd)
template <int NW>
__device__ __forceinline__ void mlp_ln_param_partials(const float* dzbuf, const float* __restrict__ xhat, int row0,
                                                      float* __restrict__ dg_out, float* __restrict__ db_out, int tid)
{
    constexpr int H = NW * 32, LD = H + 1;
    if (tid < H) {
        float sg = 0.f, sb = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const float dz = dzbuf[r * LD + tid];
            sg += dz * xhat[(size_t)(row0 + r) * H + tid];
            sb += dz;
        }
        dg_out[tid] = sg;
        db_out[tid] = sb;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// condition encoder
// ------------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64)
void cond_fwd_kernel(const gmt_cond_fwd_args p)
{
    constexpr int H = NW * 32, NT = NW * 64, LD = H + 1;
    extern __shared__ float smem[];
    const int K1 = 6 + p.latent;
    const int ld0 = (K1 + 1) | 1;                       // odd, >= K1 + 1 (the pad column stays zero)
    float* cin = smem;                                  // [32][ld0]
    float* ybuf = cin + 32 * ld0;                       // [32][LD]  pre-LayerNorm
    float* abuf = ybuf + 32 * LD;                       // [32][LD]  activation = next layer's A operand
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 31;
    const int col = wave * 32 + lo;
    const int row0 = blockIdx.x * 32;

    // ---- the 6 + latent inputs of the first Linear (RG.py:469-484)
    for (int i = tid; i < 32 * ld0; i += NT) cin[i] = 0.f;
    __syncthreads();
    if (tid < 32 && row0 + tid < p.B) {
        const size_t r = (size_t)(row0 + tid);
        const float sx = p.start[r * p.ld_start], sy = p.start[r * p.ld_start + 1];
        const float ex = p.end[r * p.ld_end], ey = p.end[r * p.ld_end + 1];
        const float dx = ex - sx, dy = ey - sy;
        float* c = cin + tid * ld0;
        c[0] = sx; c[1] = sy; c[2] = ex; c[3] = ey;
        c[4] = sqrtf(dx * dx + dy * dy + p.dist_eps);
        c[5] = atan2f(dy, dx);
    }
    for (int i = tid; i < 32 * p.latent; i += NT) {
        const int r = i / p.latent, c = i % p.latent;
        if (row0 + r < p.B) cin[r * ld0 + 6 + c] = p.z[(size_t)(row0 + r) * p.ld_z + c];
    }
    __syncthreads();
    if (p.c_in) {
        for (int i = tid; i < 32 * p.c_pitch; i += NT) {
            const int r = i / p.c_pitch, c = i % p.c_pitch;
            p.c_in[(size_t)(row0 + r) * p.c_pitch + c] = c < K1 ? cin[r * ld0 + c] : 0.f;
        }
    }
    // ---- Linear 1 + LayerNorm + LeakyReLU
    {
        f32x16 acc[1];
        const float bv = p.b1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = bv;
        const float* const wrow[1] = {p.w1 + (size_t)col * K1};
        mlp_mma_wT<1>(acc, cin, ld0, wrow, K1, lane);
        mlp_acc_to_lds(acc[0], ybuf, LD, col, lane);
    }
    __syncthreads();
    mlp_layer_norm<NW>(ybuf, abuf, p.ln1_g, p.ln1_b, p.slope, p.ln_eps, row0, p.B, p.xhat1, p.rstd1, p.a1, tid);
    __syncthreads();
    // ---- Linear 2 + LayerNorm + LeakyReLU
    {
        f32x16 acc[1];
        const float bv = p.b2[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = bv;
        const float* const wrow[1] = {p.w2 + (size_t)col * H};
        mlp_mma_wT<1>(acc, abuf, LD, wrow, H, lane);
        mlp_acc_to_lds(acc[0], ybuf, LD, col, lane);
    }
    __syncthreads();                                    // (every wave is past its reads of abuf = a1 here)
    mlp_layer_norm<NW>(ybuf, abuf, p.ln2_g, p.ln2_b, p.slope, p.ln_eps, row0, p.B, p.xhat2, p.rstd2, p.cond, tid);
    if (!p.w_g) return;
    __syncthreads();
    // ---- g0 = cond . W_g^T + b_ih + b_hh : the four gates of this wave's 32 units
    {
        f32x16 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float bv = p.bias_a[g * H + col] + p.bias_b[g * H + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = bv;
        }
        const float* const wrow[4] = {p.w_g + (size_t)(0 * H + col) * p.ld_wg, p.w_g + (size_t)(1 * H + col) * p.ld_wg,
                                      p.w_g + (size_t)(2 * H + col) * p.ld_wg, p.w_g + (size_t)(3 * H + col) * p.ld_wg};
        mlp_mma_wT<4>(acc, abuf, LD, wrow, H, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + gmt_acc_row(r, lane);
            const bool valid = row < p.B;
            float* o = p.g0 + (size_t)row * (4 * H);
            const float v0 = valid ? acc[0][r] : 0.f, v1 = valid ? acc[1][r] : 0.f;
            const float v2 = valid ? acc[2][r] : 0.f, v3 = valid ? acc[3][r] : 0.f;
            if (p.g0_unit_major) gmt_st4(o + 4 * col, v0, v1, v2, v3);
            else { o[col] = v0; o[H + col] = v1; o[2 * H + col] = v2; o[3 * H + col] = v3; }
        }
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64)
void cond_bwd_kernel(const gmt_cond_bwd_args p)
{
    constexpr int H = NW * 32, NT = NW * 64, LD = H + 1;
    constexpr int CH = (4 * H < 256) ? 4 * H : 256;     // columns of d_g0 staged at a time
    extern __shared__ float smem[];
    float* dg = smem;                                   // [32][CH + 1]
    float* dz = dg + 32 * (CH + 1);                     // [32][LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 31;
    const int col = wave * 32 + lo;
    const int row0 = blockIdx.x * 32;
    float* part = p.ln_part + (size_t)blockIdx.x * 4 * H;

    // ---- d cond = d_cond + d_g0 . W_g
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc[0][r] = p.d_cond ? p.d_cond[(size_t)(row0 + gmt_acc_row(r, lane)) * H + col] : 0.f;
    if (p.d_g0) {
        for (int c0 = 0; c0 < 4 * H; c0 += CH) {
            __syncthreads();
            for (int i = tid; i < 32 * (CH / 4); i += NT) {
                const int r = i / (CH / 4), q = i % (CH / 4);
                float* src = p.d_g0 + (size_t)(row0 + r) * (4 * H) + c0 + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row0 + r < p.B) v = gmt_ld4(src);
                else *reinterpret_cast<float4*>(src) = v;
                float* d = dg + r * (CH + 1) + 4 * q;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            __syncthreads();
            const int um = p.g0_unit_major;
            mlp_mma_w<1>(acc, dg, CH + 1, p.w_g + col, p.ld_wg, 0, CH, lane,
                         [=](int k) { const int n = c0 + k; return um ? (n & 3) * H + (n >> 2) : n; });
        }
    }
    // ---- through LeakyReLU and LayerNorm 2
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = gmt_acc_row(r, lane);
        const float c = p.cond[(size_t)(row0 + row) * H + col];
        // (padding rows: whatever adjoint the caller's buffers hold there must not reach a parameter gradient)
        dz[row * LD + col] = row0 + row < p.B ? acc[0][r] * (c > 0.f ? 1.0f : p.slope) : 0.f;
    }
    __syncthreads();
    mlp_ln_param_partials<NW>(dz, p.xhat2, row0, part, part + H, tid);
    __syncthreads();
    mlp_layer_norm_bwd<NW>(dz, p.xhat2, p.rstd2, p.ln2_g, row0, p.dy2, tid);
    __syncthreads();
    // ---- d a1 = dy2 . W2, through LeakyReLU and LayerNorm 1
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
    mlp_mma_w<1>(acc, dz, LD, p.w2 + col, H, 0, H, lane, [](int k) { return k; });
    __syncthreads();                                    // every wave has read dy2 from dz
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = gmt_acc_row(r, lane);
        const float a = p.a1[(size_t)(row0 + row) * H + col];
        dz[row * LD + col] = acc[0][r] * (a > 0.f ? 1.0f : p.slope);
    }
    __syncthreads();
    mlp_ln_param_partials<NW>(dz, p.xhat1, row0, part + 2 * H, part + 3 * H, tid);
    __syncthreads();
    mlp_layer_norm_bwd<NW>(dz, p.xhat1, p.rstd1, p.ln1_g, row0, p.dy1, tid);
}

// ------------------------------------------------------------------------------------------------------------------
// critic output head
// ------------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64)
void head_fwd_kernel(const gmt_head_fwd_args p)
{
    constexpr int H = NW * 32, NT = NW * 64, LD = H + 1, N2 = H / 2, LD2 = N2 + 1;
    constexpr int CH = (2 * H < 256) ? 2 * H : 256;     // columns of x staged at a time
    extern __shared__ float smem[];
    float* xb = smem;                                   // [32][CH + 1]
    float* a1b = xb + 32 * (CH + 1);                    // [32][LD]
    float* a2b = a1b + 32 * LD;                         // [32][LD2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 31;
    const int col = wave * 32 + lo;
    const int row0 = blockIdx.x * 32;
    const uint32_t thresh24 = (uint32_t)(p.dropout_p * 16777216.0f);
    const float keep_scale = 1.0f / (1.0f - p.dropout_p);

    // mask mode (p.m1 given): the adjoint of the head's backward pass - the same three contractions with the saved
    // activations' LeakyReLU' / dropout factors in place of the non-linearities and no biases (second backward of the
    // gradient penalty, RG.py:829-835)
    const bool mm = p.m1 != nullptr;
    // ---- Linear(2H -> H) + LeakyReLU + Dropout
    {
        f32x16 acc[1];
        const float bv = mm ? 0.f : p.b1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = bv;
        for (int c0 = 0; c0 < 2 * H; c0 += CH) {
            __syncthreads();
            for (int i = tid; i < 32 * (CH / 4); i += NT) {
                const int r = i / (CH / 4), q = i % (CH / 4);
                const float4 v = gmt_ld4(p.x + (size_t)(row0 + r) * (2 * H) + c0 + 4 * q);
                float* d = xb + r * (CH + 1) + 4 * q;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            __syncthreads();
            const float* const wrow[1] = {p.w1 + (size_t)col * (2 * H) + c0};
            mlp_mma_wT<1>(acc, xb, CH + 1, wrow, CH, lane);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = gmt_acc_row(r, lane);
            float v = mm ? acc[0][r] * (p.m1[(size_t)(row0 + row) * H + col] > 0.f ? 1.0f : p.slope) : mlp_lrelu(acc[0][r], p.slope);
            if (thresh24) v = gmt_keep((uint32_t)(row0 + row) * (uint32_t)H + (uint32_t)col, p.seed, thresh24) ? v * keep_scale : 0.f;
            a1b[row * LD + col] = v;
            if (p.a1) p.a1[(size_t)(row0 + row) * H + col] = v;
        }
    }
    __syncthreads();
    // ---- Linear(H -> H/2) + LeakyReLU: the waves whose columns exist
    if (32 * wave < N2) {
        const bool in = col < N2;
        f32x16 acc[1];
        const float bv = (in && !mm) ? p.b2[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = bv;
        const float* const wrow[1] = {p.w2 + (size_t)(in ? col : N2 - 1) * H};
        mlp_mma_wT<1>(acc, a1b, LD, wrow, H, lane);
        if (in) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = gmt_acc_row(r, lane);
                const float v = mm ? acc[0][r] * (p.m2[(size_t)(row0 + row) * p.a2_pitch + col] > 0.f ? 1.0f : p.slope)
                                   : mlp_lrelu(acc[0][r], p.slope);
                a2b[row * LD2 + col] = v;
                if (p.a2) p.a2[(size_t)(row0 + row) * p.a2_pitch + col] = v;
            }
        }
    }
    if (p.a2 && p.a2_pitch > N2) {
        const int np = p.a2_pitch - N2;
        for (int i = tid; i < 32 * np; i += NT) p.a2[(size_t)(row0 + i / np) * p.a2_pitch + N2 + i % np] = 0.f;
    }
    __syncthreads();
    // ---- Linear(H/2 -> 1)
    if (tid < 32) {
        float s = mm ? 0.f : p.b3[0];
#pragma unroll 8
        for (int j = 0; j < N2; ++j) s += a2b[tid * LD2 + j] * p.w3[j];
        p.score[row0 + tid] = s;
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64)
void head_bwd_kernel(const gmt_head_bwd_args p)
{
    constexpr int H = NW * 32, NT = NW * 64, LD = H + 1, N2 = H / 2, LD2 = N2 + 1;
    extern __shared__ float smem[];
    float* d2b = smem;                                  // [32][LD2]
    float* d1b = d2b + 32 * LD2;                        // [32][LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 31;
    const int col = wave * 32 + lo;
    const int row0 = blockIdx.x * 32;
    const uint32_t thresh24 = (uint32_t)(p.dropout_p * 16777216.0f);
    const float keep_scale = 1.0f / (1.0f - p.dropout_p);

    // ---- dy2 = d_score w3 LeakyReLU'(a2)
    for (int i = tid; i < 32 * N2; i += NT) {
        const int r = i / N2, n = i % N2;
        const float a = p.a2[(size_t)(row0 + r) * p.a2_pitch + n];
        const float v = p.d_score[row0 + r] * p.w3[n] * (a > 0.f ? 1.0f : p.slope);
        d2b[r * LD2 + n] = v;
        p.dy2[(size_t)(row0 + r) * N2 + n] = v;
    }
    if (tid < 32) gmt_st4(p.ds4 + (size_t)(row0 + tid) * 4, p.d_score[row0 + tid], 0.f, 0.f, 0.f);
    __syncthreads();
    // ---- dy1 = (dy2 . W2) mask LeakyReLU'(a1)
    {
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        mlp_mma_w<1>(acc, d2b, LD2, p.w2 + col, H, 0, N2, lane, [](int k) { return k; });
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = gmt_acc_row(r, lane);
            const float a = p.a1[(size_t)(row0 + row) * H + col];
            float v = acc[0][r] * (a > 0.f ? 1.0f : p.slope);
            if (thresh24) v = gmt_keep((uint32_t)(row0 + row) * (uint32_t)H + (uint32_t)col, p.seed, thresh24) ? v * keep_scale : 0.f;
            d1b[row * LD + col] = v;
            p.dy1[(size_t)(row0 + row) * H + col] = v;
        }
    }
    __syncthreads();
    // ---- dx = dy1 . W1 : this wave's columns col and H + col
    {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        mlp_mma_w<2>(acc, d1b, LD, p.w1 + col, 2 * H, H, H, lane, [](int k) { return k; });
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* o = p.dx + (size_t)(row0 + gmt_acc_row(r, lane)) * (2 * H) + col;
            o[0] = acc[0][r];
            o[H] = acc[1][r];
        }
    }
}

template <class K, class A>
int mlp_launch(K kernel, const A& args, int n_tiles, int threads, size_t lds, hipStream_t stream)
{
    if (lds > 160 * 1024 - 512) return GMT_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GMT_ERR_HIP;
    hipLaunchKernelGGL(kernel, dim3((unsigned)n_tiles), dim3(threads), lds, stream, args);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

#define MLP_DISPATCH(H, CALL)                                                                                    \
    switch (H) {                                                                                                 \
        case 32: { constexpr int NW = 1; CALL; }                                                                 \
        case 64: { constexpr int NW = 2; CALL; }                                                                 \
        case 128: { constexpr int NW = 4; CALL; }                                                                \
        case 256: { constexpr int NW = 8; CALL; }                                                                \
        case 512: { constexpr int NW = 16; CALL; }                                                               \
        default: return GMT_ERR_UNSUPPORTED;                                                                     \
    }

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int gmt_cond_encoder_fwd(const gmt_cond_fwd_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && a->start && a->end && a->z && a->w1 && a->b1 && a->ln1_g && a->ln1_b && a->w2 && a->b2 && a->ln2_g &&
                  a->ln2_b && a->cond);
    GMT_CHECK_ARG(a->B >= 1 && a->Bp >= a->B && (a->Bp % 32) == 0 && a->latent >= 0 && a->ld_start >= 2 && a->ld_end >= 2 &&
                  a->ld_z >= a->latent);
    GMT_CHECK_ARG(!a->w_g || (a->g0 && a->bias_a && a->bias_b && a->ld_wg >= a->H && al16(a->g0)));
    const bool save = a->c_in || a->xhat1 || a->rstd1 || a->a1 || a->xhat2 || a->rstd2;
    if (save) GMT_CHECK_ARG(a->c_in && a->xhat1 && a->rstd1 && a->a1 && a->xhat2 && a->rstd2 && a->c_pitch >= 6 + a->latent &&
                            al16(a->xhat1) && al16(a->a1) && al16(a->xhat2));
    GMT_CHECK_ARG(al16(a->cond));
    const int K1 = 6 + a->latent;
    const size_t lds = (size_t)32 * (((K1 + 1) | 1) + 2 * (a->H + 1)) * sizeof(float);
    MLP_DISPATCH(a->H, return mlp_launch(cond_fwd_kernel<NW>, *a, a->Bp / 32, NW * 64, lds, stream))
}

extern "C" int gmt_cond_encoder_bwd(const gmt_cond_bwd_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && (a->d_g0 || a->d_cond) && a->cond && a->xhat2 && a->rstd2 && a->ln2_g && a->w2 && a->a1 && a->xhat1 &&
                  a->rstd1 && a->ln1_g && a->dy2 && a->dy1 && a->ln_part);
    GMT_CHECK_ARG(!a->d_g0 || (a->w_g && a->ld_wg >= a->H && al16(a->d_g0)));
    GMT_CHECK_ARG(a->B >= 1 && a->Bp >= a->B && (a->Bp % 32) == 0);
    GMT_CHECK_ARG(al16(a->xhat1) && al16(a->xhat2) && al16(a->dy1) && al16(a->dy2));
    const int H = a->H, CH = (4 * H < 256) ? 4 * H : 256;
    const size_t lds = (size_t)32 * ((CH + 1) + (H + 1)) * sizeof(float);
    MLP_DISPATCH(a->H, return mlp_launch(cond_bwd_kernel<NW>, *a, a->Bp / 32, NW * 64, lds, stream))
}

extern "C" int gmt_critic_head_fwd(const gmt_head_fwd_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && a->x && a->w1 && a->w2 && a->w3 && a->score && al16(a->x));
    GMT_CHECK_ARG((a->m1 != nullptr) == (a->m2 != nullptr));
    GMT_CHECK_ARG(a->m1 || (a->b1 && a->b2 && a->b3));
    GMT_CHECK_ARG(a->Bp >= 32 && (a->Bp % 32) == 0 && a->dropout_p >= 0.f && a->dropout_p < 1.f);
    GMT_CHECK_ARG(!a->a2 || a->a2_pitch >= a->H / 2);
    GMT_CHECK_ARG((long long)a->Bp * a->H < (1ll << 32));              // 32-bit dropout counter
    const int H = a->H, CH = (2 * H < 256) ? 2 * H : 256;
    const size_t lds = (size_t)32 * ((CH + 1) + (H + 1) + (H / 2 + 1)) * sizeof(float);
    MLP_DISPATCH(a->H, return mlp_launch(head_fwd_kernel<NW>, *a, a->Bp / 32, NW * 64, lds, stream))
}

extern "C" int gmt_critic_head_bwd(const gmt_head_bwd_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && a->d_score && a->a1 && a->a2 && a->w1 && a->w2 && a->w3 && a->dx && a->dy1 && a->dy2 && a->ds4);
    GMT_CHECK_ARG(a->Bp >= 32 && (a->Bp % 32) == 0 && a->dropout_p >= 0.f && a->dropout_p < 1.f && a->a2_pitch >= a->H / 2);
    GMT_CHECK_ARG(al16(a->ds4) && (long long)a->Bp * a->H < (1ll << 32));
    const int H = a->H;
    const size_t lds = (size_t)32 * ((H + 1) + (H / 2 + 1)) * sizeof(float);
    MLP_DISPATCH(a->H, return mlp_launch(head_bwd_kernel<NW>, *a, a->Bp / 32, NW * 64, lds, stream))
}
