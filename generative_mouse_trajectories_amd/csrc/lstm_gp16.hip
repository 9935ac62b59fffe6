// 16-row-tile variants of the three recurrent sweeps of the gradient-penalty pass (compute_gradient_penalty,
// RecurrentGAN.py:790-842: critic forward on the interpolates, first backward under create_graph=True, second-order
// sweep; the final backward reuses the first's kernel) for H = 128.
//
// Why: the penalty runs on a quarter of the batch (RG.py:969-989) - 1 024 rows at B = 4096 - and a persistent
// workgroup owns its rows for the whole sequence, so with the 32-row tiles of lstm_seq.hip / lstm_gp.hip the pass
// occupies 64 of the 256 CUs and its time is the per-workgroup step latency.  Here a workgroup owns 16 rows:
// twice the workgroups (128 CUs), half the matrix work and half the cell arithmetic per step and workgroup.
//   * contraction on v_mfma_f32_16x16x4_f32 (exact fp32, same k-ordered fmaf chain as the 32x32x2 form):
//       A: lane l holds A[row l & 15][k = l >> 4],  B: lane l holds B[k = l >> 4][col l & 15],
//       C/D: lane l, register r holds C[row 4 (l >> 4) + r][col l & 15];
//   * wave w owns hidden units [32w, 32w + 32) as two 16-unit halves; its 4 gates x 2 halves = 8 accumulator
//     tiles keep the four gates of one (row, unit) in one lane and register, like the 32-row kernels;
//   * W_hh (4H x H fp32 = 256 KB) is resident in the register file for the whole sequence - 256 VGPRs a lane -
//     read once from the nn.Parameter's own row-major storage (no packed copy);
//   * the input contractions are hoisted out of the loop (gmt_lstm_xproj): the kernels take x W_ih^T (+ b) /
//     xhat W_ih^T for all timesteps as an input and keep only the recurrent part;
//   * the A operand (h_{t-1}, or d(pre-activation) in the backward) sits in LDS as [k][16 rows]: reads of a k-step
//     are conflict-free, a lane's four rows of one column are one ds_write_b128.
// Buffers, packed-sequence masks and formulas are those of lstm_fwd_kernel / lstm_bwd_kernel / lstm_bwd2_kernel.
#include "lstm_tile.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H16 = 128;            // hidden units
constexpr int R16 = 16;             // rows per workgroup

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct Tile16 {
    int lane, wave, q, n, dir, row0;
    __device__ __forceinline__ int row(int r) const { return 4 * q + r; }                 // row within the tile
    __device__ __forceinline__ int unit(int hh) const { return 32 * wave + 16 * hh + n; } // hidden unit
};

__device__ __forceinline__ Tile16 tile16(int n_tiles)
{
    Tile16 t;
    t.lane = threadIdx.x & 63;
    t.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    t.q = t.lane >> 4; t.n = t.lane & 15;
    t.dir = blockIdx.x / n_tiles;
    t.row0 = (blockIdx.x % n_tiles) * R16;
    return t;
}

// lengths of the tile's rows -> s_len, their max / min -> tmax / tmin
__device__ __forceinline__ void tile_lengths(const int* __restrict__ lengths, int row0, int L, int* s_len, int* s_mm, int& tmax, int& tmin)
{
    if (threadIdx.x == 0) { s_mm[0] = 0; s_mm[1] = L; }
    __syncthreads();
    if (threadIdx.x < R16) {
        int ln = lengths ? lengths[row0 + threadIdx.x] : L;
        ln = ln < 1 ? 1 : (ln > L ? L : ln);           // RG.py:766 clamp(min=1)
        s_len[threadIdx.x] = ln;
        atomicMax(&s_mm[0], ln);
        atomicMin(&s_mm[1], ln);
    }
    __syncthreads();
    tmax = __builtin_amdgcn_readfirstlane(s_mm[0]);
    tmin = __builtin_amdgcn_readfirstlane(s_mm[1]);
}

// forward-shaped recurrent weights: tile (gate g, half hh), k-step s: W[g H + unit(hh)][4 s + q]
__device__ __forceinline__ void load_w_fwd(float (&wr)[4][2][32], const float* __restrict__ W, const Tile16& t)
{
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const float* wrow = W + (size_t)(g * H16 + t.unit(hh)) * H16 + t.q;
#pragma unroll
            for (int s = 0; s < 32; ++s) wr[g][hh][s] = wrow[4 * s];
        }
}

// acc[g][hh] += hT (LDS [k][16]) . W_hh^T, W resident
__device__ __forceinline__ void mma_fwd(f32x4 (&acc)[4][2], const float* __restrict__ hT, const float (&wr)[4][2][32], const Tile16& t)
{
    const float* ap = hT + t.q * R16 + t.n;
    float a[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[0][j] = ap[64 * j];
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        if (grp + 1 < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[(grp + 1) & 1][j] = ap[64 * (4 * (grp + 1) + j)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) acc[g][hh] = mfma16(a[grp & 1][j], wr[g][hh][4 * grp + j], acc[g][hh]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 1)
void lstm16_fwd_kernel(const float* pregates /* [ndir][L][B][4H], may alias `gates` */, const float* __restrict__ w_hh,
                       const int* __restrict__ lengths, float* gates, float* __restrict__ c_all, float* __restrict__ h_all,
                       float* __restrict__ h_final, int L, int B, int ndir, int n_tiles)
{
    constexpr int H = H16;
    __shared__ __attribute__((aligned(16))) float hT[H * R16];
    __shared__ int s_len[R16];
    __shared__ int s_mm[2];
    const Tile16 t_ = tile16(n_tiles);
    int tmax, tmin;
    tile_lengths(lengths, t_.row0, L, s_len, s_mm, tmax, tmin);
    for (int i = threadIdx.x; i < H * R16; i += 256) hT[i] = 0.0f;
    float wr[4][2][32];
    load_w_fwd(wr, w_hh + (size_t)t_.dir * 4 * H * H, t_);
    __syncthreads();

    float c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = 0.f;
    const size_t lane_row = (size_t)t_.row0 + 4 * t_.q;
    float4 pv[8];
    auto p_load = [&](int t) {
        const float* pp = pregates + (((size_t)t_.dir * L + t) * B + lane_row) * (4 * H);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[hh * 4 + r] = gmt_ld4(pp + (size_t)r * (4 * H) + 4 * t_.unit(hh));
    };
    if (tmax > 0) p_load(t_.dir ? tmax - 1 : 0);

    for (int step = 0; step < tmax; ++step) {
        const int t = t_.dir ? (tmax - 1 - step) : step;
        f32x4 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) acc[g][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_fwd(acc, hT, wr, t_);
        float pre[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pre[0][j] = acc[0][j >> 2][j & 3] + pv[j].x; pre[1][j] = acc[1][j >> 2][j & 3] + pv[j].y;
            pre[2][j] = acc[2][j >> 2][j & 3] + pv[j].z; pre[3][j] = acc[3][j >> 2][j & 3] + pv[j].w;
        }
        __builtin_amdgcn_sched_barrier(0);
        p_load((step + 1 < tmax) ? (t_.dir ? t - 1 : t + 1) : t);          // unconditional: the last step re-reads its own
        __syncthreads();                                                    // every wave is done reading hT
        const size_t rs = ((size_t)t_.dir * L + t) * B + lane_row;
        const bool masked = !(t < tmin);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int u = t_.unit(hh);
            float4* hslot = reinterpret_cast<float4*>(hT + u * R16 + 4 * t_.q);
            float hn[4];
            float4 hold = make_float4(0.f, 0.f, 0.f, 0.f);
            if (masked) hold = *hslot;
            const float ho[4] = {hold.x, hold.y, hold.z, hold.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = hh * 4 + r;
                const bool act = !masked || (t < s_len[4 * t_.q + r]);
                const float ig = gmt_sigmoid(pre[0][j]), fg = gmt_sigmoid(pre[1][j]);
                const float gg = gmt_tanh(pre[2][j]), og = gmt_sigmoid(pre[3][j]);
                const float cn = fg * c[j] + ig * gg;
                const float hv = og * gmt_tanh(cn);
                c[j] = act ? cn : c[j];
                hn[r] = act ? hv : ho[r];
                gmt_st4(gates + (rs + r) * (4 * H) + 4 * u, act ? ig : 0.f, act ? fg : 0.f, act ? gg : 0.f, act ? og : 0.f);
                c_all[(rs + r) * H + u] = act ? cn : 0.f;
                h_all[((size_t)t * B + lane_row + r) * (ndir * H) + t_.dir * H + u] = act ? hv : 0.f;
            }
            *hslot = make_float4(hn[0], hn[1], hn[2], hn[3]);
        }
        __syncthreads();                                                    // the new h is in place
    }
    for (int t = tmax; t < L; ++t)                                          // padded tail of the whole tile
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) h_all[((size_t)t * B + lane_row + r) * (ndir * H) + t_.dir * H + t_.unit(hh)] = 0.f;
    if (h_final) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int u = t_.unit(hh);
#pragma unroll
            for (int r = 0; r < 4; ++r) h_final[(lane_row + r) * (ndir * H) + t_.dir * H + u] = hT[u * R16 + 4 * t_.q + r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Out of place (da_out); TOTALS: also the per-step totals dct / dht (first backward under create_graph);
// EXTRA: adjoints of the saved gates / cells as extra inputs (the backward after the second-order sweep).
template <bool HAS_DY, bool EXTRA, bool TOTALS>
__global__ __launch_bounds__(256, 1)
void lstm16_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_all, const float* __restrict__ w_hh,
                       const int* __restrict__ lengths, const float* __restrict__ dh_all, const float* __restrict__ dh_final,
                       const float* __restrict__ dgates_extra, const float* __restrict__ dc_extra,
                       float* __restrict__ da_out, float* __restrict__ dct_out, float* __restrict__ dht_out,
                       int L, int B, int ndir, int n_tiles)
{
    constexpr int H = H16;
    __shared__ __attribute__((aligned(16))) float daT[4 * H * R16];        // [column 4 u + g][row]
    __shared__ int s_len[R16];
    __shared__ int s_mm[2];
    const Tile16 t_ = tile16(n_tiles);
    int tmax, tmin;
    tile_lengths(lengths, t_.row0, L, s_len, s_mm, tmax, tmin);
    // dh_prev[row][u] = sum over columns (4 unit_k + gate) of da: k-step s holds columns 4 s + q, i.e. unit_k = s, gate = q
    float wr[2][H];
    {
        const float* W = w_hh + (size_t)t_.dir * 4 * H * H;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int s = 0; s < H; ++s) wr[hh][s] = W[(size_t)(t_.q * H + s) * H + t_.unit(hh)];
    }
    const size_t lane_row = (size_t)t_.row0 + 4 * t_.q;
    float dh[8], dc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dh[j] = dh_final ? dh_final[(lane_row + (j & 3)) * (ndir * H) + t_.dir * H + t_.unit(j >> 2)] : 0.f;
        dc[j] = 0.f;
    }
    // the saved activations (and extra adjoints) of a step are requested one step ahead, right before the MFMAs of the
    // step before it, so their HBM latency hides under the matrix work
    float4 pg[8], pge[EXTRA ? 8 : 1];
    float pcn[8], pcp[8], pdy[8], pce[EXTRA ? 8 : 1];
    auto fetch = [&](int t) {
        const int tp = t_.dir ? t + 1 : t - 1;
        const bool tp_in = (tp >= 0) && (tp < L);
        const size_t rs = ((size_t)t_.dir * L + t) * B + lane_row;
        const size_t rsp = ((size_t)t_.dir * L + (tp_in ? tp : t)) * B + lane_row;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = j & 3, u = t_.unit(j >> 2);
            pg[j] = gmt_ld4(gates + (rs + r) * (4 * H) + 4 * u);
            pcn[j] = c_all[(rs + r) * H + u];
            pcp[j] = c_all[(rsp + r) * H + u];
            pdy[j] = HAS_DY ? dh_all[((size_t)t * B + lane_row + r) * (ndir * H) + t_.dir * H + u] : 0.f;
            if constexpr (EXTRA) {
                pge[j] = gmt_ld4(dgates_extra + (rs + r) * (4 * H) + 4 * u);
                pce[j] = dc_extra[(rs + r) * H + u];
            }
        }
    };
    if (tmax > 0) fetch(t_.dir ? 0 : tmax - 1);
    for (int step = tmax - 1; step >= 0; --step) {
        const int t = t_.dir ? (tmax - 1 - step) : step;
        const int tp = t_.dir ? t + 1 : t - 1;                              // the step the forward pass visited before t
        const bool tp_in = (tp >= 0) && (tp < L);
        const size_t rs = ((size_t)t_.dir * L + t) * B + lane_row;
        f32x4 acc[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int u = t_.unit(hh);
            float dav[4][4];                                                // [gate][r]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = hh * 4 + r;
                const int ln = s_len[4 * t_.q + r];
                const bool act = t < ln;
                const bool has_prev = t_.dir ? (tp < ln) : (tp >= 0);
                const float ig = pg[j].x, fg = pg[j].y, gg = pg[j].z, og = pg[j].w;
                const float cn = pcn[j];
                const float cp = (has_prev && tp_in) ? pcp[j] : 0.f;
                const float dht = dh[j] + pdy[j];
                const float tc = gmt_tanh(cn);
                float dog = dht * tc;
                float dct = dc[j] + dht * og * (1.f - tc * tc);
                if constexpr (EXTRA) dct += pce[j];
                float dig = dct * gg, dgg = dct * ig, dfg = dct * cp;
                if constexpr (EXTRA) { dig += pge[j].x; dfg += pge[j].y; dgg += pge[j].z; dog += pge[j].w; }
                dc[j] = act ? dct * fg : dc[j];
                dav[0][r] = act ? dig * ig * (1.f - ig) : 0.f;
                dav[1][r] = act ? dfg * fg * (1.f - fg) : 0.f;
                dav[2][r] = act ? dgg * (1.f - gg * gg) : 0.f;
                dav[3][r] = act ? dog * og * (1.f - og) : 0.f;
                acc[hh][r] = act ? 0.f : dh[j];            // live row: dh_prev = da W_hh only; frozen row: state passes through
                gmt_st4(da_out + (rs + r) * (4 * H) + 4 * u, dav[0][r], dav[1][r], dav[2][r], dav[3][r]);
                if constexpr (TOTALS) {
                    dct_out[(rs + r) * H + u] = act ? dct : 0.f;
                    dht_out[(rs + r) * H + u] = act ? dht : 0.f;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(daT + (4 * u + g) * R16 + 4 * t_.q) = make_float4(dav[g][0], dav[g][1], dav[g][2], dav[g][3]);
        }
        __syncthreads();
        fetch((step > 0) ? (t_.dir ? t + 1 : t - 1) : t);                   // next step's operands (the last re-reads its own)
        {
            const float* ap = daT + t_.q * R16 + t_.n;
            float a[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[0][j] = ap[64 * j];
#pragma unroll
            for (int grp = 0; grp < H / 8; ++grp) {
                if (grp + 1 < H / 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[(grp + 1) & 1][j] = ap[64 * (8 * (grp + 1) + j)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[0] = mfma16(a[grp & 1][j], wr[0][8 * grp + j], acc[0]);
                    acc[1] = mfma16(a[grp & 1][j], wr[1][8 * grp + j], acc[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) dh[j] = acc[j >> 2][j & 3];
        __syncthreads();
    }
    for (int t = tmax; t < L; ++t) {                                        // padding of the whole tile: zero d(pre-activation)
        const size_t rs = ((size_t)t_.dir * L + t) * B + lane_row;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) gmt_st4(da_out + (rs + r) * (4 * H) + 4 * t_.unit(hh), 0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ second order
__global__ __launch_bounds__(256, 1)
void lstm16_bwd2_kernel(const float* __restrict__ pre /* [ndir][L][B][4H] = xhat W_ih^T */, const float* __restrict__ w_hh,
                        const int* __restrict__ lengths, const float* __restrict__ gates, const float* __restrict__ c_all,
                        const float* __restrict__ dct_all, const float* __restrict__ dht_all,
                        float* __restrict__ hhat_all, float* __restrict__ hhat_final,
                        float* __restrict__ g_gates, float* __restrict__ g_c, int L, int B, int ndir, int n_tiles)
{
    constexpr int H = H16;
    __shared__ __attribute__((aligned(16))) float hT[H * R16];
    __shared__ int s_len[R16];
    __shared__ int s_mm[2];
    const Tile16 t_ = tile16(n_tiles);
    int tmax, tmin;
    tile_lengths(lengths, t_.row0, L, s_len, s_mm, tmax, tmin);
    for (int i = threadIdx.x; i < H * R16; i += 256) hT[i] = 0.0f;
    float wr[4][2][32];
    load_w_fwd(wr, w_hh + (size_t)t_.dir * 4 * H * H, t_);
    __syncthreads();
    float chat[8], gc_pend[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { chat[j] = 0.f; gc_pend[j] = 0.f; }
    const size_t lane_row = (size_t)t_.row0 + 4 * t_.q;
    for (int step = 0; step < tmax; ++step) {
        const int t = t_.dir ? (tmax - 1 - step) : step;
        const int tp = t_.dir ? t + 1 : t - 1;                              // step visited just before t by this sweep
        const bool tp_in = (tp >= 0) && (tp < L);
        const size_t rs = ((size_t)t_.dir * L + t) * B + lane_row;
        const size_t rsp = ((size_t)t_.dir * L + (tp < 0 ? 0 : (tp >= L ? L - 1 : tp))) * B + lane_row;
        float4 q4[8], p4[8];
        float v_cn[8], v_cp[8], v_dct[8], v_dht[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                                       // requested ahead of the MFMAs
            const int r = j & 3, u = t_.unit(j >> 2);
            q4[j] = gmt_ld4(gates + (rs + r) * (4 * H) + 4 * u);
            p4[j] = gmt_ld4(pre + (rs + r) * (4 * H) + 4 * u);
            v_cn[j] = c_all[(rs + r) * H + u];
            v_cp[j] = c_all[(rsp + r) * H + u];
            v_dct[j] = dct_all[(rs + r) * H + u];
            v_dht[j] = dht_all[(rs + r) * H + u];
        }
        f32x4 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) acc[g][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_fwd(acc, hT, wr, t_);
        __syncthreads();                                                    // every wave is done reading hT
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int u = t_.unit(hh);
            float4* hslot = reinterpret_cast<float4*>(hT + u * R16 + 4 * t_.q);
            const float4 hold = *hslot;
            const float ho[4] = {hold.x, hold.y, hold.z, hold.w};
            float hn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = hh * 4 + r;
                const int ln = s_len[4 * t_.q + r];
                const bool act = t < ln;
                const bool has_prev = tp_in && (t_.dir ? (tp < ln) : true);
                const float ig = q4[j].x, fg = q4[j].y, gg = q4[j].z, og = q4[j].w;
                const float cn = v_cn[j];
                const float cp = has_prev ? v_cp[j] : 0.f;
                const float dct = v_dct[j], dht = v_dht[j];
                const float tc = gmt_tanh(cn);
                const float omt = 1.f - tc * tc;
                const float di = dct * gg, dg = dct * ig, df = dct * cp, dov = dht * tc;
                const float a_i = acc[0][hh][r] + p4[j].x, a_f = acc[1][hh][r] + p4[j].y;
                const float a_g = acc[2][hh][r] + p4[j].z, a_o = acc[3][hh][r] + p4[j].w;
                const float adi = a_i * ig * (1.f - ig);
                const float adf = a_f * fg * (1.f - fg);
                const float adg = a_g * (1.f - gg * gg);
                const float ado = a_o * og * (1.f - og);
                const float cprev_hat = chat[j];
                const float adct = adi * gg + adg * ig + adf * cp + cprev_hat * fg;
                const float o_gi = a_i * di * (1.f - 2.f * ig) + adg * dct;
                const float o_gf = a_f * df * (1.f - 2.f * fg) + cprev_hat * dct;
                const float o_gg = a_g * dg * (-2.f * gg) + adi * dct;
                const float o_go = a_o * dov * (1.f - 2.f * og) + adct * dht * omt;
                const float atc = adct * dht * og * (-2.f * tc) + ado * dht;
                const float o_gc = atc * omt;
                const float hnew = adct * og * omt + ado * tc;
                chat[j] = act ? adct : chat[j];
                hn[r] = act ? hnew : ho[r];
                gmt_st4(g_gates + (rs + r) * (4 * H) + 4 * u, act ? o_gi : 0.f, act ? o_gf : 0.f, act ? o_gg : 0.f, act ? o_go : 0.f);
                // the step before this one (if the sweep had one) now has its complete g_c
                if (step > 0) g_c[(rsp + r) * H + u] = gc_pend[j] + ((act && has_prev) ? adf * dct : 0.f);
                gc_pend[j] = act ? o_gc : 0.f;
                if (hhat_all) hhat_all[((size_t)t * B + lane_row + r) * (ndir * H) + t_.dir * H + u] = act ? hnew : 0.f;
            }
            *hslot = make_float4(hn[0], hn[1], hn[2], hn[3]);
        }
        if (step + 1 == tmax) {                                             // the last step of the sweep has no successor
#pragma unroll
            for (int j = 0; j < 8; ++j) g_c[(rs + (j & 3)) * H + t_.unit(j >> 2)] = gc_pend[j];
        }
        __syncthreads();
    }
    for (int t = tmax; t < L; ++t) {                                        // padded tail of the tile
        const size_t rs = ((size_t)t_.dir * L + t) * B + lane_row;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = t_.unit(hh);
                gmt_st4(g_gates + (rs + r) * (4 * H) + 4 * u, 0.f, 0.f, 0.f, 0.f);
                g_c[(rs + r) * H + u] = 0.f;
                if (hhat_all) hhat_all[((size_t)t * B + lane_row + r) * (ndir * H) + t_.dir * H + u] = 0.f;
            }
    }
    if (hhat_final) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int u = t_.unit(hh);
#pragma unroll
            for (int r = 0; r < 4; ++r) hhat_final[(lane_row + r) * (ndir * H) + t_.dir * H + u] = hT[u * R16 + 4 * t_.q + r];
        }
    }
}

}  // namespace

extern "C" int gmt_lstm16_supported(int H, int B) { return (H == H16 && B >= R16 && (B % R16) == 0) ? 1 : 0; }

extern "C" int gmt_lstm16_layer_fwd(const float* pregates, const float* w_hh, const int* lengths, float* gates, float* c_all,
                                    float* h_all, float* h_final, int L, int B, int H, int ndir, hipStream_t stream)
{
    GMT_CHECK_ARG(pregates && w_hh && gates && c_all && h_all && L >= 1 && (ndir == 1 || ndir == 2));
    if (!gmt_lstm16_supported(H, B)) return GMT_ERR_UNSUPPORTED;
    const int n_tiles = B / R16;
    hipLaunchKernelGGL(lstm16_fwd_kernel, dim3(n_tiles * ndir), dim3(256), 0, stream, pregates, w_hh, lengths, gates, c_all,
                       h_all, h_final, L, B, ndir, n_tiles);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_lstm16_layer_bwd(const float* gates, const float* c_all, const float* w_hh, const int* lengths,
                                    const float* dh_all, const float* dh_final, const float* dgates_extra, const float* dc_extra,
                                    float* da_out, float* dct_out, float* dht_out, int L, int B, int H, int ndir,
                                    hipStream_t stream)
{
    GMT_CHECK_ARG(gates && c_all && w_hh && da_out && L >= 1 && (ndir == 1 || ndir == 2));
    GMT_CHECK_ARG((dgates_extra != nullptr) == (dc_extra != nullptr) && (dct_out != nullptr) == (dht_out != nullptr));
    if (!gmt_lstm16_supported(H, B)) return GMT_ERR_UNSUPPORTED;
    const int n_tiles = B / R16;
    const bool dy = dh_all != nullptr, extra = dgates_extra != nullptr, totals = dct_out != nullptr;
#define GMT_GO(DYV, EXV, TOV)                                                                                         \
    hipLaunchKernelGGL((lstm16_bwd_kernel<DYV, EXV, TOV>), dim3(n_tiles * ndir), dim3(256), 0, stream, gates, c_all, w_hh,  \
                       lengths, dh_all, dh_final, dgates_extra, dc_extra, da_out, dct_out, dht_out, L, B, ndir, n_tiles)
    if (dy) { if (extra) { if (totals) GMT_GO(true, true, true); else GMT_GO(true, true, false); }
              else       { if (totals) GMT_GO(true, false, true); else GMT_GO(true, false, false); } }
    else    { if (extra) { if (totals) GMT_GO(false, true, true); else GMT_GO(false, true, false); }
              else       { if (totals) GMT_GO(false, false, true); else GMT_GO(false, false, false); } }
#undef GMT_GO
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_lstm16_layer_bwd2(const float* pre, const float* w_hh, const int* lengths, const float* gates,
                                     const float* c_all, const float* dct_all, const float* dht_all, float* hhat_all,
                                     float* hhat_final, float* g_gates, float* g_c, int L, int B, int H, int ndir,
                                     hipStream_t stream)
{
    GMT_CHECK_ARG(pre && w_hh && gates && c_all && dct_all && dht_all && g_gates && g_c && L >= 1 && (ndir == 1 || ndir == 2));
    if (!gmt_lstm16_supported(H, B)) return GMT_ERR_UNSUPPORTED;
    const int n_tiles = B / R16;
    hipLaunchKernelGGL(lstm16_bwd2_kernel, dim3(n_tiles * ndir), dim3(256), 0, stream, pre, w_hh, lengths, gates, c_all,
                       dct_all, dht_all, hhat_all, hhat_final, g_gates, g_c, L, B, ndir, n_tiles);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
