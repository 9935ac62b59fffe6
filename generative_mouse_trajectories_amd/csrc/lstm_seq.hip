// Persistent (whole-sequence) LSTM layer kernels for the discriminator's packed,
// bidirectional LSTM - RecurrentGAN.py:720-727 called at RG.py:764-778, i.e.
// torch.nn.LSTM over pack_padded_sequence(lengths, enforce_sorted=False).
//
// One launch runs one layer, both directions, all timesteps.  Workgroup = (direction,
// tile of 32 trajectories); see lstm_tile.h for the wave/MFMA/LDS decomposition.
// Packed-sequence semantics are a per-row length mask: a row's state advances at step t
// only when t < len; the reverse direction walks t = L-1 .. 0 and therefore starts each
// row at its own last valid step with zero state.  Outputs at padded steps are zero.
//
// Forward saves the activated gates (i,f,g,o; unit-major columns, see lstm_tile.h), c and h of
// every step for the backward sweep; backward overwrites the gate buffer in place with
// d(pre-activation), which the host then contracts with x / h_prev for the weight gradients
// (csrc/wgrad.hip) and with W_ih for the input gradient.
#include "lstm_tile.h"
#include <type_traits>

// k-steps of W_hh kept in LDS rather than registers by the H = 128 fused forward kernel
#ifndef GMT_FWD_KL
#define GMT_FWD_KL 16
#endif
// the same for W_hh^T in the H = 128 backward kernel
#ifndef GMT_BWD_KL
#define GMT_BWD_KL 64
#endif

namespace {

// ------------------------------------------------------------------ weight packing
__global__ void pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int k_valid, int n_valid,
                            int n_waves, int n_gates, int gate_stride, long long sk, long long sn)
{
    const long long total = (long long)n_waves * (K / 8) * n_gates * 64 * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        long long q = i >> 8;
        const int g = (int)(q % n_gates); q /= n_gates;
        const int s4 = (int)(q % (K / 8));
        const int w = (int)(q / (K / 8));
        const int k = 2 * (4 * s4 + e) + (lane >> 5);
        const int n = g * gate_stride + 32 * w + (lane & 31);
        dst[i] = (k < k_valid && n < n_valid) ? src[(long long)k * sk + (long long)n * sn] : 0.0f;
    }
}

// ------------------------------------------------------------------ forward
// NW waves (H = 32 NW).  RES: recurrent weights resident in registers.  IX = I / H:
//   IX = 0  the caller precomputed x W_ih^T + b into `pregates` (read from HBM);
//   IX = 1,2 the input contraction runs inside the kernel, W_ih streamed from L2.
// FULL: gates, c_all and h_all are all given (the training call): stores without pointer tests.
// B must be a multiple of 32 (the host pads the batch), so there are no row guards.
template <int NW, bool RES, int IX, bool FULL>
__global__ __launch_bounds__(NW * 64, 1)
void lstm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pregates,
                     const float* __restrict__ w_ih_pk, const float* __restrict__ w_hh_pk,
                     const float* __restrict__ bias, const int* __restrict__ lengths,
                     float* __restrict__ gates, float* __restrict__ c_all, float* __restrict__ h_all,
                     float* __restrict__ h_final, int L, int B, int ndir, int n_tiles,
                     float x_drop_p, uint32_t x_drop_seed)
{
    constexpr int H = NW * 32;
    constexpr int KS = H / 2;
    constexpr int I = (IX > 0 ? IX : 1) * H;
    // inverted dropout on the layer input (the inter-layer dropout of nn.LSTM, RG.py:640): element
    // (t, row, k) of x is kept iff gmt_keep16((t*B + row)*I + k, seed)
    const uint32_t x_thresh16 = (uint32_t)(x_drop_p * 65536.0f);
    const float x_keep_scale = 1.0f / (1.0f - x_drop_p);
    constexpr int NT = NW * 64;
    extern __shared__ float smem[];
    // at H = 128 the last KL k-steps of W_hh live in LDS, not registers
    constexpr int KL = (RES && NW == 4) ? GMT_FWD_KL : 0;   // (IX = 0: the prefetched pregates need the registers)
    constexpr int KR = KS - KL;
    float* hT = smem;                                  // [H][LDA]  h_{t-1}, transposed (MFMA A operand)
    float* xT = smem + H * GMT_LDA;                    // [I][LDA]  x_t, transposed (IX > 0 only)
    float4* wl = reinterpret_cast<float4*>(smem + (H + (IX > 0 ? I : 0)) * GMT_LDA)
                 + (size_t)(threadIdx.x >> 6) * (KL / 4) * 4 * 64;   // [NW][KL/4][4][64] float4
    __shared__ int s_len[GMT_TM];
    __shared__ int s_maxlen, s_minlen;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x / n_tiles;
    const int row0 = (blockIdx.x % n_tiles) * GMT_TM;
    const int hi = lane >> 5, lo = lane & 31;
    const int col = wave * 32 + lo;                    // hidden unit of this lane

    if (tid == 0) { s_maxlen = 0; s_minlen = L; }
    __syncthreads();
    if (tid < GMT_TM) {
        int ln = lengths ? lengths[row0 + tid] : L;
        ln = ln < 1 ? 1 : (ln > L ? L : ln);           // RG.py:766 clamp(min=1)
        s_len[tid] = ln;
        atomicMax(&s_maxlen, ln);
        atomicMin(&s_minlen, ln);
    }
    for (int i = tid; i < H * GMT_LDA; i += NT) hT[i] = 0.0f;
    __syncthreads();
    const int tmax = __builtin_amdgcn_readfirstlane(s_maxlen);   // steps [tmax, L) are padding for the whole tile
    const int tmin = __builtin_amdgcn_readfirstlane(s_minlen);   // steps [0, tmin) have every row active

    const float4* whh = reinterpret_cast<const float4*>(w_hh_pk) + ((size_t)dir * NW + wave) * (H / 8) * 4 * 64;
    const float4* wih = IX > 0 ? reinterpret_cast<const float4*>(w_ih_pk) + ((size_t)dir * NW + wave) * (I / 8) * 4 * 64
                               : nullptr;
    float wr[4][RES ? KR : 1];
    if constexpr (RES) {
        gmt_load_resident<4, KR>(wr, whh, lane);
        for (int i = lane; i < (KL / 4) * 4 * 64; i += 64) wl[i] = whh[(KR / 4) * 4 * 64 + i];   // visible after the next barrier
    }

    float bz[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (IX > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bz[g] = bias[(size_t)dir * 4 * H + g * H + col];
    }

    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;

    // Step order (two barriers a step, single LDS buffers):
    //   h-part   acc = bias + h_{t-1} W_hh^T   registers / LDS only: the global stores of the previous
    //                                          cell update drain under it, nothing waits on them
    //   x_store  x_t (registers) -> xT         every wave is past the previous x-part (barrier A)
    //   barrier B
    //   x-part   acc += x_t W_ih^T             W_ih streamed from L2
    //   x_load   x_{t+1} -> registers          lands under the cell update and the next h-part
    //   cell     gates, c, h -> hT, HBM        every wave is past the h-part (barrier B)
    //   barrier A
    constexpr int XQ = I / 4;                          // float4 per row
    constexpr int XS = (GMT_TM * XQ) / NT;             // float4 per thread (= 4 * IX)
    float4 xs[IX > 0 ? XS : 1];
    auto x_load = [&](int t) {
        const float4* src = reinterpret_cast<const float4*>(x + ((size_t)t * B + row0) * I);
#pragma unroll
        for (int j = 0; j < XS; ++j) xs[j] = (src + j * NT)[tid];   // uniform base + one lane offset
    };
    auto x_store = [&](int t) {
        // item tid + j*NT is row tid/XQ + j*(NT/XQ), float4 column tid%XQ: one LDS address, the
        // j-th float4 sits j*(NT/XQ) floats further on
        static_assert(NT % XQ == 0, "x tile staging assumes NT is a multiple of I/4");
        float* d0 = xT + (size_t)(4 * (tid % XQ)) * GMT_LDA + tid / XQ;
        if (x_thresh16) {                              // uniform: the dropped copy of x exists only in LDS
            const uint32_t i0 = (uint32_t)(((size_t)t * B + row0) * I) + 4u * (uint32_t)tid;
#pragma unroll
            for (int j = 0; j < XS; ++j) gmt_drop4(xs[j], i0 + (uint32_t)(j * NT * 4), x_drop_seed, x_thresh16, x_keep_scale);
        }
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            float* d = d0 + j * (NT / XQ);
            d[0] = xs[j].x; d[GMT_LDA] = xs[j].y; d[2 * GMT_LDA] = xs[j].z; d[3 * GMT_LDA] = xs[j].w;
        }
    };

    if constexpr (IX > 0) {
        if (tmax > 0) x_load(dir ? tmax - 1 : 0);
    }

    const size_t lane_row = (size_t)row0 + 4 * hi;     // first row of this lane's accumulator rows
    // IX = 0: the hoisted x W_ih^T + b of this lane's 16 (row, unit) pairs, one float4 (i,f,g,o) each; requested a
    // whole step ahead (right after the previous step consumed its own), added to the recurrent part after the MFMAs
    float4 pv[IX == 0 ? 16 : 1];
    auto p_load = [&](int t) {
        const float* pp = pregates + (((size_t)dir * L + t) * B + lane_row) * (4 * H) + 4 * col;
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] = gmt_ld4(pp + (size_t)((r & 3) + 8 * (r >> 2)) * (4 * H));
    };
    if constexpr (IX == 0) {
        if (tmax > 0) p_load(dir ? tmax - 1 : 0);
    }
    float* hs = hT + col * GMT_LDA + 4 * hi;           // this lane's 16 state slots (updated in place)
    GMT_PH_DECL;
    for (int step = 0; step < tmax; ++step) {
        const int t = dir ? (tmax - 1 - step) : step;
        GMT_PH(5);
        const size_t rs = ((size_t)dir * L + t) * B + lane_row;
        f32x16 acc[4];
        if constexpr (IX > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = bz[g];
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        }
        if constexpr (RES) gmt_mma_resident<4, KS, KR>(acc, hT, wr, lane, wl);
        else gmt_mma_stream<4>(acc, hT, whh, H, lane);
        GMT_PH(1);
        if constexpr (IX > 0) x_store(t);
        GMT_PH(3);
        __syncthreads();                               // B
        GMT_PH(4);
        if constexpr (IX > 0) {
            gmt_mma_stream<4>(acc, xT, wih, I, lane);
            // unconditional (the last step re-reads its own tile): a branch here splits the staging
            // registers' live ranges and the compiler then waits for the loads on the spot
            const int tn = (step + 1 < tmax) ? (dir ? t - 1 : t + 1) : t;
            x_load(tn);
            GMT_PH(0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[0][r] += pv[r].x; acc[1][r] += pv[r].y; acc[2][r] += pv[r].z; acc[3][r] += pv[r].w;
            }
            __builtin_amdgcn_sched_barrier(0);
            p_load((step + 1 < tmax) ? (dir ? t - 1 : t + 1) : t);     // unconditional, like x_load
        }

        float* gp = gates ? gates + rs * (4 * H) + 4 * col : nullptr;
        float* cp = c_all ? c_all + rs * H + col : nullptr;
        float* hp = h_all ? h_all + ((size_t)t * B + lane_row) * (ndir * H) + dir * H + col : nullptr;
        // the cell update; MASKED = some row of the tile is already past its length at step t
        // (branch-free selects), otherwise no masks, no reads of the old state
        auto cell = [&](auto masked) {
            constexpr bool MASKED = decltype(masked)::value;
            unsigned am = 0xffffu;
            if constexpr (MASKED) {
                am = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) am |= (unsigned)(t < s_len[(r & 3) + 8 * (r >> 2) + 4 * hi]) << r;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (r & 3) + 8 * (r >> 2);     // row within the tile, minus 4*hi
                const float ig = gmt_sigmoid(acc[0][r]);
                const float fg = gmt_sigmoid(acc[1][r]);
                const float gg = gmt_tanh(acc[2][r]);
                const float og = gmt_sigmoid(acc[3][r]);
                const float cn = fg * c[r] + ig * gg;
                const float hv = og * gmt_tanh(cn);
                float* q = gp + (size_t)mr * (4 * H);
                if constexpr (MASKED) {
                    const bool act = (am >> r) & 1u;
                    c[r] = act ? cn : c[r];
                    hs[mr] = act ? hv : hs[mr];
                    if (FULL || gp) gmt_st4(q, act ? ig : 0.f, act ? fg : 0.f, act ? gg : 0.f, act ? og : 0.f);
                    if (FULL || cp) cp[(size_t)mr * H] = act ? cn : 0.f;
                    if (FULL || hp) hp[(size_t)mr * (ndir * H)] = act ? hv : 0.f;
                } else {
                    c[r] = cn;
                    hs[mr] = hv;
                    if (FULL || gp) gmt_st4(q, ig, fg, gg, og);                                // FULL: no per-row branches
                    if (FULL || cp) cp[(size_t)mr * H] = cn;
                    if (FULL || hp) hp[(size_t)mr * (ndir * H)] = hv;
                }
            }
        };
        if (t < tmin) cell(std::false_type{});
        else cell(std::true_type{});
        GMT_PH(2);
        __syncthreads();                               // A
    }
    GMT_PH_DUMP;

    // padded tail of the whole tile: outputs are zero (pad_packed_sequence)
    if (h_all) {
        for (int t = tmax; t < L; ++t) {
            float* hp = h_all + ((size_t)t * B + lane_row) * (ndir * H) + dir * H + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) hp[(size_t)((r & 3) + 8 * (r >> 2)) * (ndir * H)] = 0.f;
        }
    }
    if (h_final) {
        const float* hl = hs;
        float* hf = h_final + lane_row * (ndir * H) + dir * H + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2);
            hf[(size_t)mr * (ndir * H)] = hl[mr];
        }
    }
}

// ------------------------------------------------------------------ backward
// Reverse sweep of one layer.  State: dh, dc of the wave's 32 units (accumulator layout).
// Per step: d(pre-activation) from the saved gates -> written over the gate buffer and to
// LDS (transposed) -> dh_prev = da W_hh as an MFMA contraction over K = 4H with W_hh
// resident in registers (or streamed).
// PLAIN: the ordinary training call (no extra adjoints, no extra outputs, d(pre-activation) in place):
// no per-row pointer tests in the element-wise phase.
// DY: 0 no dh_all (the top layer, whose only adjoint is dh_final); 1 dh_all is given; 2 dh_all is given and is
// the adjoint of the DROPPED output (inter-layer dropout fused into the consumer): the keep mask is applied here.
// DY 3 / 4 = 1 / 2 with dh_all given as PAIRS [L][B][ndir*H][2] whose two slots are added here: the partial input
// gradients the two directions of the layer above wrote with FX (below).
// EXTRA (gradient penalty, second backward): dgates_extra and dc_extra are both given and are prefetched
// with the other operands; given singly they are read at use.
// FX (H = 128, plain call): the layer's INPUT gradient is contracted in the same time loop - after step t's
// dh_prev = da_t W_hh, the tile of da_t that already sits in LDS meets W_ih (streamed from L2 in fragment-major order,
// like the forward kernel's x-part): dx_t = da_t W_ih, 2 x 256 more MFMAs per wave and step at I = 2H.  The element-wise
// phase of a step (~9 k cycles that nothing hides with one wave per SIMD) is then amortised over three times the matrix
// work, d(pre-activation) is not read back from HBM by a separate contraction, and that contraction's launch is gone.
// The two directions run in different workgroups with opposite time orders, so each writes its own slot of a pair
// ([L*B][I][2], slot = direction); the consumer - the sweep of the layer below, DY 3 / 4 - adds the two at use.
template <int NW, bool RES, bool PLAIN, int DY, bool EXTRA, int FX = 0>
__global__ __launch_bounds__(NW * 64, 1)
void lstm_bwd_kernel(float* __restrict__ gates /* in: i,f,g,o   out: da */, const float* __restrict__ c_all,
                     const float* __restrict__ w_hh_pkT, const int* __restrict__ lengths,
                     const float* __restrict__ dh_all /* [L][B][ndir*H] or null */,
                     const float* __restrict__ dh_final /* [B][ndir*H] or null */,
                     const float* __restrict__ dgates_extra /* [ndir][L][B][4H] or null: extra adjoint of the activated gates */,
                     const float* __restrict__ dc_extra /* [ndir][L][B][H] or null */,
                     float* __restrict__ da_out /* null: overwrite `gates` in place */,
                     float* __restrict__ dct_out /* [ndir][L][B][H] or null: total cell adjoint (double backward) */,
                     float* __restrict__ dht_out /* [ndir][L][B][H] or null: total hidden adjoint (double backward) */,
                     int L, int B, int ndir, int n_tiles, float dy_drop_p, uint32_t dy_drop_seed,
                     const float* __restrict__ w_ih_pkT /* FX: [ndir] blocks, K = 4H (gate-major), N = FX * 128 */,
                     float* __restrict__ dx_pairs /* FX: [L*B][FX * 128][2] */)
{
    constexpr int H = NW * 32;
    constexpr int KS = 2 * H;                          // k-steps over K = 4H
    constexpr bool HAS_DY = DY > 0;
    constexpr bool PAIRS = DY >= 3, DROP = (DY == 2 || DY == 4);
    static_assert(FX == 0 || (PLAIN && NW == 4), "fused input gradient: plain H = 128 call");
    const uint32_t dy_thresh16 = (uint32_t)(dy_drop_p * 65536.0f);
    const float dy_keep_scale = 1.0f / (1.0f - dy_drop_p);
    // at H = 128 the last KL k-steps of W_hh^T live in LDS, not registers (the prefetch registers need the room)
    constexpr int KL = (RES && NW == 4) ? GMT_BWD_KL : 0;
    constexpr int KR = KS - KL;
    extern __shared__ float smem[];
    float* daT = smem;                                 // [4H][LDA]
    float4* wl = reinterpret_cast<float4*>(smem + 4 * H * GMT_LDA) + (size_t)(threadIdx.x >> 6) * (KL / 4) * 64;
    __shared__ int s_len[GMT_TM];
    __shared__ int s_maxlen, s_minlen;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x / n_tiles;
    const int row0 = (blockIdx.x % n_tiles) * GMT_TM;
    const int hi = lane >> 5, lo = lane & 31;
    const int col = wave * 32 + lo;

    if (tid == 0) { s_maxlen = 0; s_minlen = L; }
    __syncthreads();
    if (tid < GMT_TM) {
        int ln = lengths ? lengths[row0 + tid] : L;
        ln = ln < 1 ? 1 : (ln > L ? L : ln);
        s_len[tid] = ln;
        atomicMax(&s_maxlen, ln);
        atomicMin(&s_minlen, ln);
    }
    __syncthreads();
    const int tmax = __builtin_amdgcn_readfirstlane(s_maxlen);
    const int tmin = __builtin_amdgcn_readfirstlane(s_minlen);

    const float4* wT = reinterpret_cast<const float4*>(w_hh_pkT) + ((size_t)dir * NW + wave) * (4 * H / 8) * 64;
    float wr[1][RES ? KR : 1];
    if constexpr (RES) {
        gmt_load_resident<1, KR>(wr, wT, lane);
        for (int i = lane; i < (KL / 4) * 64; i += 64) wl[i] = wT[(KR / 4) * 64 + i];   // visible after the first barrier of the sweep
    }

    const size_t lane_row = (size_t)row0 + 4 * hi;
    float dh[16], dc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = (r & 3) + 8 * (r >> 2);
        dh[r] = dh_final ? dh_final[(lane_row + mr) * (ndir * H) + dir * H + col] : 0.f;
        dc[r] = 0.f;
    }

    // Software pipeline: the saved activations of the NEXT step are fetched into registers during
    // the MFMA phase - one accumulator row (6 loads) ahead of each group of 16 MFMAs, so neither
    // their issue slots nor their HBM latency are exposed.  c_t is carried from the previous
    // step's c_prev fetch.
    float pg[4][16], pcp[16], pdy[16], cn_c[16];
    float pdy2[PAIRS ? 16 : 1];
    float pge[EXTRA ? 4 : 1][16], pce[16];             // EXTRA: adjoints of the saved gates / cells of the next step
    auto prefetch_row = [&](int t, int r) {
        const int tp = dir ? t + 1 : t - 1;
        const bool tp_ok = (tp >= 0) && (tp < L);
        const size_t rs = ((size_t)dir * L + t) * B + lane_row;
        const int mr = (r & 3) + 8 * (r >> 2);
        const float4 q = gmt_ld4(gates + (rs + mr) * (4 * H) + 4 * col);
        pg[0][r] = q.x; pg[1][r] = q.y; pg[2][r] = q.z; pg[3][r] = q.w;
        // raw (tp outside [0, L): c_t itself, unused - selected away at use; touching a loaded value
        // here would wait for it on the spot)
        pcp[r] = c_all[(((size_t)dir * L + (tp_ok ? tp : t)) * B + lane_row + mr) * H + col];
        if constexpr (PAIRS) {
            const float2 q2 = *reinterpret_cast<const float2*>(dh_all + 2 * (((size_t)t * B + lane_row + mr) * (ndir * H) + dir * H + col));
            pdy[r] = q2.x; pdy2[r] = q2.y;
        } else if constexpr (HAS_DY) pdy[r] = dh_all[((size_t)t * B + lane_row + mr) * (ndir * H) + dir * H + col];
        else pdy[r] = 0.f;
        if constexpr (EXTRA) {
            const float4 e = gmt_ld4(dgates_extra + (rs + mr) * (4 * H) + 4 * col);
            pge[0][r] = e.x; pge[1][r] = e.y; pge[2][r] = e.z; pge[3][r] = e.w;
            pce[r] = dc_extra[(rs + mr) * H + col];
        }
    };
    if (tmax > 0) {
        const int t0 = dir ? 0 : tmax - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) prefetch_row(t0, r);
        const float* cn_p = c_all + (((size_t)dir * L + t0) * B + lane_row) * H + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) cn_c[r] = cn_p[(size_t)((r & 3) + 8 * (r >> 2)) * H];
    }

    // forward visited t in order (dir 0: 0..tmax-1, dir 1: tmax-1..0); walk it backwards
    GMT_PH_DECL;
    for (int step = tmax - 1; step >= 0; --step) {
        const int t = dir ? (tmax - 1 - step) : step;
        const int tp = dir ? t + 1 : t - 1;            // the step the forward pass visited before t
        GMT_PH(5);
        const size_t rs = ((size_t)dir * L + t) * B + lane_row;
        float* wp = ((!PLAIN && da_out) ? da_out : gates) + rs * (4 * H) + 4 * col;
        float* dct_p = (!PLAIN && dct_out) ? dct_out + rs * H + col : nullptr;
        float* dht_p = (!PLAIN && dht_out) ? dht_out + rs * H + col : nullptr;
        const float* ge_p = (!PLAIN && dgates_extra) ? dgates_extra + rs * (4 * H) + 4 * col : nullptr;
        const float* ce_p = (!PLAIN && dc_extra) ? dc_extra + rs * H + col : nullptr;
        float* dl = daT + col * GMT_LDA + 4 * hi;      // this lane's slots of the transposed da tile
        f32x16 acc[1];
        // MASKED = some row of the tile is past its length at t, or has no earlier step: selects,
        // no branches; otherwise every row is live and has a predecessor (or tp < 0 for all)
        const bool tp_in = (tp >= 0) && (tp < L);      // uniform: the forward pass had an earlier step at all
        auto sweep = [&](auto masked) {
            constexpr bool MASKED = decltype(masked)::value;
            unsigned am = 0xffffu, pm = 0xffffu;
            if constexpr (MASKED) {
                am = 0; pm = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ln = s_len[(r & 3) + 8 * (r >> 2) + 4 * hi];
                    am |= (unsigned)(t < ln) << r;
                    pm |= (unsigned)(dir ? (tp < ln) : (tp >= 0)) << r;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (r & 3) + 8 * (r >> 2);
                const bool act = !MASKED || ((am >> r) & 1u);
                const bool has_prev = !MASKED || ((pm >> r) & 1u);
                const float ig = pg[0][r], fg = pg[1][r], gg = pg[2][r], og = pg[3][r];
                const float cn = cn_c[r];
                const float cp = (has_prev && tp_in) ? pcp[r] : 0.f;
                float dyv = pdy[r];
                if constexpr (PAIRS) dyv += pdy2[r];
                float dht = dh[r] + dyv;
                if constexpr (DROP) {                  // same element index as the consumer's x staging
                    const uint32_t i = (uint32_t)(((size_t)t * B + lane_row + mr) * (ndir * H)) + dir * H + col;
                    dht = dh[r] + (gmt_keep16(i, dy_drop_seed, dy_thresh16) ? dyv * dy_keep_scale : 0.f);
                }
                const float tc = gmt_tanh(cn);
                float dog = dht * tc;
                float dct = dc[r] + dht * og * (1.f - tc * tc);
                if constexpr (EXTRA) dct += pce[r];
                else if (!PLAIN && ce_p) dct += ce_p[(size_t)mr * H];
                float dig = dct * gg, dgg = dct * ig, dfg = dct * cp;
                if constexpr (EXTRA) {
                    dig += pge[0][r]; dfg += pge[1][r]; dgg += pge[2][r]; dog += pge[3][r];
                } else if (!PLAIN && ge_p) {
                    const float4 ep = gmt_ld4(ge_p + (size_t)mr * (4 * H));
                    dig += ep.x; dfg += ep.y; dgg += ep.z; dog += ep.w;
                }
                dc[r] = act ? dct * fg : dc[r];
                const float dai = act ? dig * ig * (1.f - ig) : 0.f;
                const float daf = act ? dfg * fg * (1.f - fg) : 0.f;
                const float dag = act ? dgg * (1.f - gg * gg) : 0.f;
                const float dao = act ? dog * og * (1.f - og) : 0.f;
                acc[0][r] = act ? 0.f : dh[r];         // live row: dh_prev = da W_hh only; frozen row: state passes through
                cn_c[r] = pcp[r];                      // c of the step processed next
                // d(pre-activation) replaces the gates (zero at padded steps: the weight-gradient GEMM reads it)
                gmt_st4(wp + (size_t)mr * (4 * H), dai, daf, dag, dao);
                if (!PLAIN && dct_p) dct_p[(size_t)mr * H] = act ? dct : 0.f;
                if (!PLAIN && dht_p) dht_p[(size_t)mr * H] = act ? dht : 0.f;
                dl[0 * H * GMT_LDA + mr] = dai;
                dl[1 * H * GMT_LDA + mr] = daf;
                dl[2 * H * GMT_LDA + mr] = dag;
                dl[3 * H * GMT_LDA + mr] = dao;
            }
        };
        if (dir ? (t + 1 < tmin) : (t < tmin)) sweep(std::false_type{});
        else sweep(std::true_type{});
        GMT_PH(0);
        __syncthreads();
        GMT_PH(1);
        // next step's operands (the last step re-reads its own: no branch, no split live ranges)
        const int tn = (step > 0) ? (dir ? t + 1 : t - 1) : t;
        if constexpr (RES) {
            gmt_mma_resident<1, KS, KR>(acc, daT, wr, lane, wl,
                                        [&](int grp) {
                                            // KS/8 groups a step: spread the 16 rows evenly over them
#pragma unroll
                                            for (int r = 0; r < 16; ++r)
                                                if ((r * (KS / 8)) / 16 == grp) prefetch_row(tn, r);
                                        });
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) prefetch_row(tn, r);
            GMT_PH(2);
            gmt_mma_stream<1>(acc, daT, wT, 4 * H, lane);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = acc[0][r];
        GMT_PH(3);
        if constexpr (FX > 0) {
            // dx_t = da_t W_ih: this wave's columns g * 128 + col, g < FX (da of padded steps / rows is zero, so is dx)
            f32x16 accx[FX];
#pragma unroll
            for (int g = 0; g < FX; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) accx[g][r] = 0.f;
            const float4* wx = reinterpret_cast<const float4*>(w_ih_pkT) + ((size_t)dir * NW + wave) * (4 * H / 8) * FX * 64;
            gmt_mma_stream<FX>(accx, daT, wx, 4 * H, lane);
            float* xo = dx_pairs + 2 * (((size_t)t * B + lane_row) * (FX * 128) + col) + dir;
#pragma unroll
            for (int g = 0; g < FX; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    xo[2 * ((size_t)((r & 3) + 8 * (r >> 2)) * (FX * 128) + g * 128)] = accx[g][r];
        }
        __syncthreads();
        GMT_PH(4);
    }
    GMT_PH_DUMP;
    // steps [tmax, L) of the tile are padding: zero their d(pre-activation)
    for (int t = tmax; t < L; ++t) {
        float* gp = ((!PLAIN && da_out) ? da_out : gates) + (((size_t)dir * L + t) * B + lane_row) * (4 * H) + 4 * col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gmt_st4(gp + (size_t)((r & 3) + 8 * (r >> 2)) * (4 * H), 0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (FX > 0) {
            float* xo = dx_pairs + 2 * (((size_t)t * B + lane_row) * (FX * 128) + col) + dir;
#pragma unroll
            for (int g = 0; g < FX; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) xo[2 * ((size_t)((r & 3) + 8 * (r >> 2)) * (FX * 128) + g * 128)] = 0.f;
        }
    }
}

template <int NW>
int launch_fwd(const float* x, const float* pregates, const float* w_ih_pk, const float* w_hh_pk, const float* bias,
               const int* lengths, float* gates, float* c_all, float* h_all, float* h_final,
               int L, int B, int I, int ndir, float x_drop_p, uint32_t x_drop_seed, hipStream_t stream)
{
    constexpr int H = NW * 32;
    constexpr bool RES = (NW <= 4);
    const int n_tiles = B / GMT_TM;
    const dim3 grid(n_tiles * ndir), block(NW * 64);
    const bool full = gates && c_all && h_all;         // the training call: branch-free stores
#define GMT_GO(IXV, LDSF)                                                                                         \
    {                                                                                                             \
        const size_t lds = (size_t)(LDSF) * GMT_LDA * sizeof(float)                                               \
                           + ((RES && NW == 4) ? (size_t)NW * GMT_FWD_KL * 4 * 64 * sizeof(float) : 0);             \
        if (lds > 160 * 1024 - 256) return GMT_ERR_UNSUPPORTED;                                                   \
        auto k = full ? lstm_fwd_kernel<NW, RES, IXV, true> : lstm_fwd_kernel<NW, RES, IXV, false>;               \
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return GMT_ERR_HIP;                                                                                   \
        hipLaunchKernelGGL(k, grid, block, lds, stream, x, pregates, w_ih_pk, w_hh_pk, bias, lengths, gates,      \
                           c_all, h_all, h_final, L, B, ndir, n_tiles, x_drop_p, x_drop_seed);                    \
    }
    if (!x) GMT_GO(0, H)
    else if (I == H) GMT_GO(1, H + H)
    else if (I == 2 * H) GMT_GO(2, H + 2 * H)
    else return GMT_ERR_UNSUPPORTED;
#undef GMT_GO
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

template <int NW>
int launch_bwd(float* gates, const float* c_all, const float* w_hh_pkT, const int* lengths, const float* dh_all,
               const float* dh_final, const float* dgates_extra, const float* dc_extra,
               float* da_out, float* dct_out, float* dht_out, int L, int B, int ndir,
               float dy_drop_p, uint32_t dy_drop_seed, hipStream_t stream,
               int dh_pairs = 0, const float* w_ih_pkT = nullptr, float* dx_pairs = nullptr, int dx_n = 0)
{
    constexpr int H = NW * 32;
    constexpr bool RES = (NW <= 4);
    const int n_tiles = B / GMT_TM;
    const size_t lds = (size_t)(4 * H) * GMT_LDA * sizeof(float) + ((RES && NW == 4) ? (size_t)NW * GMT_BWD_KL * 64 * sizeof(float) : 0);
    if (lds > 160 * 1024 - 256) return GMT_ERR_UNSUPPORTED;
    const bool plain = !dgates_extra && !dc_extra && !da_out && !dct_out && !dht_out;
    const bool extra = dgates_extra && dc_extra;
    const bool drop = dh_all && dy_drop_p > 0.f;
    if (drop && !plain) return GMT_ERR_UNSUPPORTED;     // the fused mask exists for the ordinary training call only
    auto k = plain ? (drop ? lstm_bwd_kernel<NW, RES, true, 2, false>
                           : dh_all ? lstm_bwd_kernel<NW, RES, true, 1, false> : lstm_bwd_kernel<NW, RES, true, 0, false>)
           : extra ? (dh_all ? lstm_bwd_kernel<NW, RES, false, 1, true> : lstm_bwd_kernel<NW, RES, false, 0, true>)
                   : (dh_all ? lstm_bwd_kernel<NW, RES, false, 1, false> : lstm_bwd_kernel<NW, RES, false, 0, false>);
    if (dh_pairs || dx_pairs) {
        // the fused-input-gradient pair of calls (gmt_lstm_layer_bwd_fx): producer = top layer (no dh_all) at H = 128
        // writing I = 2H columns, consumer = the layer below reading the pairs
        if constexpr (NW == 4) {
            if (!plain) return GMT_ERR_UNSUPPORTED;
            if (dx_pairs) {
                if (dh_all || dx_n != 2 * H || !w_ih_pkT) return GMT_ERR_UNSUPPORTED;
                k = lstm_bwd_kernel<NW, RES, true, 0, false, 2>;
            } else {
                if (!dh_all) return GMT_ERR_BAD_ARG;
                k = drop ? lstm_bwd_kernel<NW, RES, true, 4, false> : lstm_bwd_kernel<NW, RES, true, 3, false>;
            }
        } else {
            return GMT_ERR_UNSUPPORTED;
        }
    }
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GMT_ERR_HIP;
    hipLaunchKernelGGL(k, dim3(n_tiles * ndir), dim3(NW * 64), lds, stream, gates, c_all, w_hh_pkT, lengths,
                       dh_all, dh_final, dgates_extra, dc_extra, da_out, dct_out, dht_out, L, B, ndir, n_tiles,
                       dy_drop_p, dy_drop_seed, w_ih_pkT, dx_pairs);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

}  // namespace

extern "C" int gmt_pack_weight(const float* src, float* dst, int K, int k_valid, int n_valid, int n_waves, int n_gates,
                               int gate_stride, long long stride_k, long long stride_n, hipStream_t stream)
{
    GMT_CHECK_ARG(src && dst && K >= 8 && (K % 8) == 0 && k_valid >= 1 && k_valid <= K && n_valid >= 1 &&
                  n_waves >= 1 && n_gates >= 1);
    const long long total = (long long)n_waves * (K / 8) * n_gates * 256;
    hipLaunchKernelGGL(pack_kernel, dim3(gmt_grid_1d(total, 256)), dim3(256), 0, stream, src, dst, K, k_valid,
                       n_valid, n_waves, n_gates, gate_stride, stride_k, stride_n);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_lstm_layer_fwd(const float* x, const float* pregates, const float* w_ih_pk, const float* w_hh_pk,
                                  const float* bias, const int* lengths, float* gates, float* c_all, float* h_all,
                                  float* h_final, int L, int B, int I, int H, int ndir,
                                  float x_dropout_p, unsigned x_dropout_seed, hipStream_t stream)
{
    GMT_CHECK_ARG(x_dropout_p >= 0.f && x_dropout_p < 1.f && (x_dropout_p == 0.f || x));
    GMT_CHECK_ARG(w_hh_pk && L >= 1 && B >= GMT_TM && (B % GMT_TM) == 0 && (ndir == 1 || ndir == 2));
    GMT_CHECK_ARG((x != nullptr) != (pregates != nullptr));
    if (x) GMT_CHECK_ARG(w_ih_pk && bias && (I == H || I == 2 * H));
    switch (H) {
#define GMT_CASE(NW) case NW * 32: return launch_fwd<NW>(x, pregates, w_ih_pk, w_hh_pk, bias, lengths, gates, c_all, \
                                                         h_all, h_final, L, B, I, ndir, x_dropout_p, x_dropout_seed, stream);
        GMT_CASE(1) GMT_CASE(2) GMT_CASE(4) GMT_CASE(8)
#undef GMT_CASE
        default: return GMT_ERR_UNSUPPORTED;
    }
}

extern "C" int gmt_lstm_layer_bwd_fx(float* gates, const float* c_all, const float* w_hh_pkT, const int* lengths,
                                     const float* dh_pairs, const float* dh_final, const float* w_ih_pkT, float* dx_pairs,
                                     int L, int B, int H, int I, int ndir, float dy_dropout_p, unsigned dy_dropout_seed,
                                     hipStream_t stream)
{
    GMT_CHECK_ARG(dy_dropout_p >= 0.f && dy_dropout_p < 1.f);
    GMT_CHECK_ARG(gates && c_all && w_hh_pkT && L >= 1 && B >= GMT_TM && (B % GMT_TM) == 0 && (ndir == 1 || ndir == 2));
    GMT_CHECK_ARG((dh_pairs != nullptr) != (dx_pairs != nullptr));      // consumer or producer
    GMT_CHECK_ARG(!dx_pairs || (w_ih_pkT && (((uintptr_t)w_ih_pkT | (uintptr_t)dx_pairs) & 15) == 0));
    GMT_CHECK_ARG(!dh_pairs || (((uintptr_t)dh_pairs) & 7) == 0);
    if (H != 128) return GMT_ERR_UNSUPPORTED;
    return launch_bwd<4>(gates, c_all, w_hh_pkT, lengths, dh_pairs, dh_final, nullptr, nullptr, nullptr, nullptr, nullptr,
                         L, B, ndir, dy_dropout_p, dy_dropout_seed, stream, dh_pairs ? 1 : 0, w_ih_pkT, dx_pairs, I);
}

extern "C" int gmt_lstm_layer_bwd(float* gates, const float* c_all, const float* w_hh_pkT, const int* lengths,
                                  const float* dh_all, const float* dh_final, const float* dgates_extra,
                                  const float* dc_extra, float* da_out, float* dct_out, float* dht_out,
                                  int L, int B, int H, int ndir, float dy_dropout_p, unsigned dy_dropout_seed,
                                  hipStream_t stream)
{
    GMT_CHECK_ARG(dy_dropout_p >= 0.f && dy_dropout_p < 1.f);
    GMT_CHECK_ARG(gates && c_all && w_hh_pkT && L >= 1 && B >= GMT_TM && (B % GMT_TM) == 0 && (ndir == 1 || ndir == 2));
    switch (H) {
#define GMT_CASE(NW) case NW * 32: return launch_bwd<NW>(gates, c_all, w_hh_pkT, lengths, dh_all, dh_final, \
                                                         dgates_extra, dc_extra, da_out, dct_out, dht_out, L, B, ndir, \
                                                         dy_dropout_p, dy_dropout_seed, stream);
        GMT_CASE(1) GMT_CASE(2) GMT_CASE(4) GMT_CASE(8)
#undef GMT_CASE
        default: return GMT_ERR_UNSUPPORTED;
    }
}
