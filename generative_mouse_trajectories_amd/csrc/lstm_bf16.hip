// The recurrence of nn.LSTM (RG.py:720-727, 764-772) at hidden sizes whose recurrent weights do not fit a CU (config 3,
// H = 512: W_hh is 2 MB in bf16): ONE LAUNCH PER TIMESTEP.  The kernel boundary is the grid-wide hand-off of h_t (every
// workgroup of step t+1 needs all of h_t); a launch is 256-512 workgroups x ~20 us, so the ~2 us boundary is cheap, and
// no workgroup ever waits on another inside a kernel.  Both directions of a bidirectional layer share a launch
// (direction 0 processes t = step, direction 1 t = L-1-step).
//
//   forward  step:  a = pre[t] + h_prev W_hh^T   (pre = x W_ih^T + b, hoisted: gemm_bf16.hip)   -> i,f,g,o,c,h
//   backward step:  dh = dh_out[t] + dA[t_next] W_hh + ...  -> cell backward -> dA[t] (over the saved gates, in place)
//
// Layouts (time-major, B % 256 == 0, H % 128 == 0):
//   gates [L][B][ndir*4H] bf16, column of (dir d, unit u, gate g) = d*4H + (u/32)*128 + g*32 + u%32  ("slab" layout: the
//          four gate blocks of 32 units are adjacent, so a wave's 128 weight rows are the 4 gates of its 32 units and its
//          four accumulator blocks hold i,f,g,o of the same (row, unit) in the same lane and register);
//   c_all [L][B][ndir*H] fp32, h_all [L][B][ndir*H] bf16 (column d*H + u);
//   w_hh  [ndir][4H (slab order)][H] bf16;  w_hhT [ndir][H][4H (slab order)] bf16.
// Ragged batches (lengths[b] < L): direction 0 freezes a row's state from t = len on (its h_all rows then repeat the
// last valid h, so h_all[L-1] is the final state), direction 1 holds zero state until t = len-1 (final state =
// h_all[0]); the backward sweep writes dA = 0 for those row-steps.
#include "bf16_tile.h"
#include "row_groups.h"
#include "lstm_tile.h"      // the counter-based dropout mask shared with the fp32 kernels (gmt_keep_pair_hash)

namespace {

using namespace gmt_bf16;

// Forward step: 256 (gate columns = 64 units x 4 gates) x 128 (rows) tiles, 32-deep K stages (48 KiB) and an epilogue that
// handles one accumulator block column (64 rows) at a time in 64 KiB of LDS: TWO workgroups per CU, so that one's epilogue -
// the step is as much HBM traffic as matrix work: 15 KB per row-step-direction - runs under the other's MFMAs.
// Epilogue images: pre-activations / gates [64][512 B], cell state [64][256 B], h [64][128 B], dropped h [64][128 B].
constexpr int FWD_TQ = 128;
constexpr int FWD_G_OFF = 0, FWD_C_OFF = 64 * 512, FWD_H_OFF = FWD_C_OFF + 64 * 256, FWD_HD_OFF = FWD_H_OFF + 64 * 128;
constexpr int FWD_NST = 3;                                     // ring slots of the contraction: 3 x 24 KiB
constexpr int FWD_LDS = FWD_NST * 24 * 1024;                   // 72 KiB (>= the epilogue's 64 KiB of images): two workgroups per CU
static_assert(FWD_LDS >= FWD_HD_OFF + 64 * 128, "epilogue images");

// keep-and-scale of 4 consecutive elements starting at idx (idx % 4 == 0) of the inter-layer dropout mask (RG.py:725): the
// same function of (seed, element index in [L][B][ndir*H]) as the fp32 path's gmt_keep16, so ops.dropout_keep_mask replays it
__device__ __forceinline__ void drop4(float (&v)[4], uint32_t idx, uint32_t seed, uint32_t thresh16, float scale)
{
    const uint32_t h0 = gmt_keep_pair_hash(idx >> 1, seed), h1 = gmt_keep_pair_hash((idx >> 1) + 1u, seed);
    v[0] = (h0 & 0xffffu) >= thresh16 ? v[0] * scale : 0.f;
    v[1] = (h0 >> 16) >= thresh16 ? v[1] * scale : 0.f;
    v[2] = (h1 & 0xffffu) >= thresh16 ? v[2] * scale : 0.f;
    v[3] = (h1 >> 16) >= thresh16 ? v[3] * scale : 0.f;
}

template <bool RAGGED>
__global__ __launch_bounds__(NTHREADS, 2)
void lstm_step_fwd_kernel(__bf16* __restrict__ G, float* __restrict__ Call, __bf16* __restrict__ Hall,
                          __bf16* __restrict__ Hdrop, const __bf16* __restrict__ Whh, const int* __restrict__ lengths,
                          int L, int B, int H, int ndir, int step, int mt0, int n_mt, int n_pt, float drop_p, uint32_t drop_seed)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int d, pt, mt;
    patch_of_block(blockIdx.x, gridDim.x, n_mt, n_pt, d, pt, mt);
    mt += mt0;                                                  // this launch covers row tiles [mt0, mt0 + n_mt)
    const int t = d == 0 ? step : L - 1 - step;
    const int tprev = d == 0 ? t - 1 : t + 1;
    const bool has_prev = step > 0;
    const long long ldg = (long long)ndir * 4 * H, ldh = (long long)ndir * H;
    const long long row0 = (long long)mt * FWD_TQ;
    // this tile's corner in the [B][..] planes of timestep t (and of the forward pass's predecessor)
    char* Gt = (char*)(G + ((long long)t * B + row0) * ldg + (long long)d * 4 * H + pt * TP);
    char* Ct = (char*)(Call + ((long long)t * B + row0) * ldh + (long long)d * H + pt * (TP / 4));
    char* Ht = (char*)(Hall + ((long long)t * B + row0) * ldh + (long long)d * H + pt * (TP / 4));
    char* Hdt = Hdrop ? (char*)(Hdrop + ((long long)t * B + row0) * ldh + (long long)d * H + pt * (TP / 4)) : nullptr;
    const uint32_t thresh16 = (uint32_t)(drop_p * 65536.0f);
    const float keep_scale = 1.0f / (1.0f - drop_p);
    const char* Cp = (const char*)(Call + ((long long)tprev * B + row0) * ldh + (long long)d * H + pt * (TP / 4));
    const __bf16* Hp = Hall + ((long long)tprev * B + row0) * ldh + (long long)d * H;   // full-width rows

    f32x16 acc[4][2];
    zero_acc<4, 2>(acc);
    GMT_PHASE(0);
    mainloop_ring<4, 2, 32, FWD_NST>(Whh + ((long long)d * 4 * H + (long long)pt * TP) * H, H, Hp, ldh, has_prev ? H / 32 : 0, lds, acc, wave, lane);
    GMT_PHASE(1);

    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    using GI = RowImage<512>;
    using CI = RowImage<256>;
    using HI = RowImage<128>;
    const int rl = wq * 32 + lo;                                // image row of this lane in both halves
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        auto rowmap = [j](int r) { return (r >> 5) * 64 + j * 32 + (r & 31); };   // image row -> tile row
        const int brow = wq * 64 + j * 32 + lo;
        __syncthreads();                                        // the LDS is free (mainloop / previous half done)
        GI::dma_in(Gt, ldg * 2, lds + FWD_G_OFF, 64, rowmap, wave, lane);
        if (has_prev) CI::dma_in(Cp, ldh * 4, lds + FWD_C_OFF, 64, rowmap, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GMT_PHASE(2 + 4 * j);
        bool valid = true;
        if (RAGGED) valid = t < lengths[row0 + brow];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int ucol = wp * 32 + 8 * g4 + 4 * hi;         // first of this lane's 4 units, within the tile's 64
            bf16x4 pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre[g] = *reinterpret_cast<const bf16x4*>(lds + FWD_G_OFF + GI::addr(rl, (wp * 128 + g * 32 + 8 * g4 + 4 * hi) * 2));
            f32x4 cp = {0.f, 0.f, 0.f, 0.f};
            if (has_prev) cp = *reinterpret_cast<const f32x4*>(lds + FWD_C_OFF + CI::addr(rl, ucol * 4));
            float cn[4], hn[4], gi[4], gf[4], gg[4], go[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                gi[e] = gmt_sigmoid(acc[0][j][r] + (float)pre[0][e]);
                gf[e] = gmt_sigmoid(acc[1][j][r] + (float)pre[1][e]);
                gg[e] = gmt_tanh(acc[2][j][r] + (float)pre[2][e]);
                go[e] = gmt_sigmoid(acc[3][j][r] + (float)pre[3][e]);
                cn[e] = gf[e] * cp[e] + gi[e] * gg[e];
                hn[e] = go[e] * gmt_tanh(cn[e]);
            }
            if (RAGGED) {
                if (!valid) {                                   // frozen (direction 0) / not started (direction 1)
                    bf16x4 hp = pack4(0.f, 0.f, 0.f, 0.f);
                    if (d == 0 && has_prev)
                        hp = *reinterpret_cast<const bf16x4*>(Hp + (long long)brow * ldh + pt * (TP / 4) + ucol);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cn[e] = d == 0 ? cp[e] : 0.f;
                        hn[e] = (float)hp[e];
                    }
                }
            }
            if (!RAGGED || valid) {                             // (the pre-activations of a skipped row-step stay as they are)
                *reinterpret_cast<bf16x4*>(lds + FWD_G_OFF + GI::addr(rl, (wp * 128 + 0 * 32 + 8 * g4 + 4 * hi) * 2)) = pack4(gi[0], gi[1], gi[2], gi[3]);
                *reinterpret_cast<bf16x4*>(lds + FWD_G_OFF + GI::addr(rl, (wp * 128 + 1 * 32 + 8 * g4 + 4 * hi) * 2)) = pack4(gf[0], gf[1], gf[2], gf[3]);
                *reinterpret_cast<bf16x4*>(lds + FWD_G_OFF + GI::addr(rl, (wp * 128 + 2 * 32 + 8 * g4 + 4 * hi) * 2)) = pack4(gg[0], gg[1], gg[2], gg[3]);
                *reinterpret_cast<bf16x4*>(lds + FWD_G_OFF + GI::addr(rl, (wp * 128 + 3 * 32 + 8 * g4 + 4 * hi) * 2)) = pack4(go[0], go[1], go[2], go[3]);
            }
            f32x4 cv = {cn[0], cn[1], cn[2], cn[3]};
            *reinterpret_cast<f32x4*>(lds + FWD_C_OFF + CI::addr(rl, ucol * 4)) = cv;
            *reinterpret_cast<bf16x4*>(lds + FWD_H_OFF + HI::addr(rl, ucol * 2)) = pack4(hn[0], hn[1], hn[2], hn[3]);
            if (Hdt) {                                          // the next layer's input: h after the inter-layer dropout
                const long long row = (long long)t * B + row0 + brow;
                drop4(hn, (uint32_t)(row * ldh + d * H + pt * (TP / 4) + ucol), drop_seed, thresh16, keep_scale);
                *reinterpret_cast<bf16x4*>(lds + FWD_HD_OFF + HI::addr(rl, ucol * 2)) = pack4(hn[0], hn[1], hn[2], hn[3]);
            }
        }
        __syncthreads();
        GMT_PHASE(3 + 4 * j);
        GI::copy_out(Gt, ldg * 2, lds + FWD_G_OFF, 64, rowmap, wave, lane);
        CI::copy_out(Ct, ldh * 4, lds + FWD_C_OFF, 64, rowmap, wave, lane);
        HI::copy_out(Ht, ldh * 2, lds + FWD_H_OFF, 64, rowmap, wave, lane);
        if (Hdt) HI::copy_out(Hdt, ldh * 2, lds + FWD_HD_OFF, 64, rowmap, wave, lane);
        GMT_PHASE(4 + 4 * j);
#ifdef GMT_PHASE_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GMT_PHASE(5 + 4 * j);
#endif
    }
}

// Backward step: 128 (units) x 128 (rows) tiles (K = 4H is long and the output narrow, so the smaller tile is what fills
// the chip: ndir H/128 B/128 workgroups, two per CU).  The epilogue handles one accumulator block of every wave at a time
// (a "quarter": 64 rows x 2 slabs), with the saved gates - overwritten by d(pre-activation) - and the two cell-state
// planes moving through LDS as whole rows; the carried dc (and the thin dh inputs) are read and written directly.
constexpr int GP2_TQ = 64, GP_BWD_LDS = 3 * (128 + 64) * 128, GP_BWD2_LDS = 3 * (256 + 64) * 128;   // the penalty's step kernels: 72 / 120 KiB rings
constexpr int BWD_G_OFF = 0, BWD_C_OFF = 64 * 512, BWD_CP_OFF = BWD_C_OFF + 64 * 256;      // 32 + 16 + 16 KiB
constexpr int BWD_LDS = 72 * 1024;                             // >= the contraction's ring (3 x 24 KiB with 64-row tiles) and the epilogue images

// the backward steps' contraction over K columns: 128 units x 64 rows in 64-deep slots, x 128 rows in 32-deep ones
template <int NQB>
__device__ __forceinline__ void bwd_contract(const __bf16* __restrict__ Pg, long long ldp, const __bf16* __restrict__ Qg,
                                             long long ldq, int K, char* lds, f32x16 (&acc)[2][NQB], int wave, int lane)
{
    if constexpr (NQB == 1) mainloop_ring<2, 1, 64, 3>(Pg, ldp, Qg, ldq, K / 64, lds, acc, wave, lane);
    else mainloop_ring<2, 2, 32, 3>(Pg, ldp, Qg, ldq, K / 32, lds, acc, wave, lane);
}

template <bool RAGGED, int NQB>
__global__ __launch_bounds__(NTHREADS, 2)
void lstm_step_bwd_kernel(__bf16* __restrict__ G, const float* __restrict__ Call, const __bf16* __restrict__ dHout,
                          const float* __restrict__ dh_fin, float* __restrict__ dc_carry, float* __restrict__ dh_carry,
                          const __bf16* __restrict__ WhhT, const int* __restrict__ lengths,
                          int L, int B, int H, int ndir, int step, int mt0, int n_mt, int n_pt, float drop_p, uint32_t drop_seed)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using T = Tile<2, NQB>;                                    // 128 units x 128 or 64 rows (the latter when B is small)
    const uint32_t thresh16 = (uint32_t)(drop_p * 65536.0f);
    const float keep_scale = 1.0f / (1.0f - drop_p);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int d, pt, mt;
    patch_of_block(blockIdx.x, gridDim.x, n_mt, n_pt, d, pt, mt);
    mt += mt0;
    const int t = d == 0 ? L - 1 - step : step;
    const int tnext = d == 0 ? t + 1 : t - 1;                  // processed by the previous launch of this sweep
    const int tprev = d == 0 ? t - 1 : t + 1;                  // the forward pass's predecessor of t
    const bool has_next = step > 0, has_prev = step < L - 1;
    const long long ldg = (long long)ndir * 4 * H, ldh = (long long)ndir * H;
    const long long row0 = (long long)mt * T::TQ;
    // tile corners: rows row0.., units 128 pt.. (4 slabs = 512 gate columns) of direction d
    char* Gt = (char*)(G + ((long long)t * B + row0) * ldg + (long long)d * 4 * H + pt * 512);
    const __bf16* Gn = G + ((long long)tnext * B + row0) * ldg + (long long)d * 4 * H;
    const char* Ct = (const char*)(Call + ((long long)t * B + row0) * ldh + (long long)d * H + pt * 128);
    const char* Cp = (const char*)(Call + ((long long)tprev * B + row0) * ldh + (long long)d * H + pt * 128);
    const __bf16* dHt = dHout ? dHout + ((long long)t * B + row0) * ldh + (long long)d * H + pt * 128 : nullptr;
    float* dcc = dc_carry + ((long long)d * B + row0) * H + pt * 128;
    float* dhc = dh_carry + ((long long)d * B + row0) * H + pt * 128;
    const float* dhf = dh_fin + ((long long)d * B + row0) * H + pt * 128;

    f32x16 acc[2][NQB];
    zero_acc<2, NQB>(acc);
    bwd_contract<NQB>(WhhT + ((long long)d * H + (long long)pt * T::TP) * 4 * H, 4LL * H, Gn, ldg, has_next ? 4 * H : 0,
                      lds, acc, wave, lane);

    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    using GI = RowImage<512>;
    using CI = RowImage<256>;
    const int rl = wq * 32 + lo;                               // image row of this lane in every quarter
#pragma unroll
    for (int j = 0; j < NQB; ++j) {
        auto rowmap = [j](int r) { return (r >> 5) * (32 * NQB) + j * 32 + (r & 31); };
        const int brow = wq * (32 * NQB) + j * 32 + lo;                // this lane's row within the tile
        bool valid = true;
        if (RAGGED) valid = t < lengths[row0 + brow];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // image columns: [wave-row 0: slab i][wave-row 1: slab 2 + i]
            auto gcol = [i](int c) { return (((c >> 4) * 2 + i) * 128) * 2 + (c & 15) * 16; };
            auto ccol = [i](int c) { return (((c >> 3) * 2 + i) * 32) * 4 + (c & 7) * 16; };
            __syncthreads();
            GI::dma_in(Gt, ldg * 2, lds + BWD_G_OFF, 64, rowmap, gcol, wave, lane);
            CI::dma_in(Ct, ldh * 4, lds + BWD_C_OFF, 64, rowmap, ccol, wave, lane);
            if (has_prev) CI::dma_in(Cp, ldh * 4, lds + BWD_CP_OFF, 64, rowmap, ccol, wave, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int slab_u = (wp * 2 + i) * 32;              // first unit of this wave's slab, within the tile's 128
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int u = slab_u + 8 * g4 + 4 * hi;
                const long long so = (long long)brow * ldh + u;
                const long long co = (long long)brow * H + u;
                float dh[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) dh[e] = acc[i][j][4 * g4 + e];
                if (dHt) {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>(dHt + so);
                    float dv[4] = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                    // dh_out is the gradient of the DROPPED output (the layer above read h through the mask)
                    if (thresh16) drop4(dv, (uint32_t)(((long long)t * B + row0 + brow) * ldh + d * H + pt * 128 + u), drop_seed, thresh16, keep_scale);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh[e] += dv[e];
                }
                if (step == 0) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(dhf + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh[e] += v[e];
                }
                f32x4 dc = {0.f, 0.f, 0.f, 0.f};
                if (has_next) dc = *reinterpret_cast<const f32x4*>(dcc + co);
                const int gbyte = wp * 256 + (8 * g4 + 4 * hi) * 2;           // + gate * 64
                if (RAGGED) {
                    if (d == 0 && has_next) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(dhc + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dh[e] += v[e];
                    }
                    if (!valid) {
                        // frozen (direction 0) or not yet started (direction 1): no cell here; direction 0 passes dh and dc on
                        const bf16x4 z = pack4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int g = 0; g < 4; ++g) *reinterpret_cast<bf16x4*>(lds + BWD_G_OFF + GI::addr(rl, gbyte + g * 64)) = z;
                        f32x4 dv = {dh[0], dh[1], dh[2], dh[3]};
                        if (d == 0) *reinterpret_cast<f32x4*>(dhc + co) = dv;
                        if (!has_next) *reinterpret_cast<f32x4*>(dcc + co) = dc;
                        continue;
                    }
                }
                bf16x4 gt[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) gt[g] = *reinterpret_cast<const bf16x4*>(lds + BWD_G_OFF + GI::addr(rl, gbyte + g * 64));
                const int cbyte = wp * 128 + (8 * g4 + 4 * hi) * 4;
                const f32x4 c = *reinterpret_cast<const f32x4*>(lds + BWD_C_OFF + CI::addr(rl, cbyte));
                f32x4 cp = {0.f, 0.f, 0.f, 0.f};
                if (has_prev) cp = *reinterpret_cast<const f32x4*>(lds + BWD_CP_OFF + CI::addr(rl, cbyte));
                float da[4][4];
                f32x4 dcn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gi = (float)gt[0][e], gf = (float)gt[1][e], gg = (float)gt[2][e], go = (float)gt[3][e];
                    const float tc = gmt_tanh(c[e]);
                    const float dct = dc[e] + dh[e] * go * (1.f - tc * tc);
                    da[0][e] = dct * gg * gi * (1.f - gi);
                    da[1][e] = dct * cp[e] * gf * (1.f - gf);
                    da[2][e] = dct * gi * (1.f - gg * gg);
                    da[3][e] = dh[e] * tc * go * (1.f - go);
                    dcn[e] = dct * gf;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<bf16x4*>(lds + BWD_G_OFF + GI::addr(rl, gbyte + g * 64)) = pack4(da[g][0], da[g][1], da[g][2], da[g][3]);
                *reinterpret_cast<f32x4*>(dcc + co) = dcn;
                if (RAGGED && d == 0) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(dhc + co) = z;
                }
            }
            __syncthreads();
            GI::copy_out(Gt, ldg * 2, lds + BWD_G_OFF, 64, rowmap, gcol, wave, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Gradient penalty (compute_gradient_penalty, RG.py:790-842): torch.autograd.grad(..., create_graph=True) differentiates
// THROUGH the critic's backward pass.  Two more step kernels, for full-length batches (the penalty runs on ceil(0.25 B) rows
// every 4th D step - 0.4 of 55 GFLOP per trained trajectory - so they trade the LDS-staged epilogues for plain accesses):
//   lstm_step_bwd_gp_kernel  the backward step, out of place, that also keeps the per-step totals dct / dht (the second-order
//                            sweep needs them) and/or takes the adjoints of the saved gates / cell states as extra inputs;
//   lstm_step_bwd2_kernel    the second-order sweep.  The first backward sweep B maps (dY, dh_final; i,f,g,o,c; W) -> dX and is
//                            linear in dY, so its adjoint is a forward-in-time recurrence of the forward layer's shape,
//                            ahat_t = xhat_t W_ih^T + hhat_{t-1} W_hh^T (the xhat term hoisted into a GEMM), followed by the
//                            element-wise second-derivative block (same formulas as csrc/lstm_gp.hip).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS, 2)
void lstm_step_bwd_gp_kernel(const __bf16* __restrict__ G, __bf16* __restrict__ dA, const float* __restrict__ Call,
                             const __bf16* __restrict__ dHout, const float* __restrict__ dh_fin,
                             const float* __restrict__ gg_in, const float* __restrict__ gc_in,
                             float* __restrict__ dct_all, float* __restrict__ dht_all, float* __restrict__ dc_carry,
                             const __bf16* __restrict__ WhhT, int L, int B, int H, int ndir, int step, int n_mt, int n_pt)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using T = Tile<2, 1>;                                      // 128 units x 64 rows (64 x 64 tiles - 256 workgroups - are no faster)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int d, pt, mt;
    patch_of_block(blockIdx.x, gridDim.x, n_mt, n_pt, d, pt, mt);
    const int t = d == 0 ? L - 1 - step : step;
    const int tnext = d == 0 ? t + 1 : t - 1;
    const int tprev = d == 0 ? t - 1 : t + 1;
    const bool has_next = step > 0, has_prev = step < L - 1;
    const long long ldg = (long long)ndir * 4 * H, ldh = (long long)ndir * H;
    const long long row0 = (long long)mt * T::TQ;
    f32x16 acc[2][1];
    zero_acc<2, 1>(acc);
    mainloop_ring<2, 1, 64, 3>(WhhT + ((long long)d * H + (long long)pt * T::TP) * 4 * H, 4LL * H,
                               dA + ((long long)tnext * B + row0) * ldg + (long long)d * 4 * H, ldg, has_next ? 4 * H / BK : 0, lds, acc, wave, lane);
    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    constexpr int j = 0;
    {
        const long long b = row0 + wq * 32 + lo;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int un = pt * 128 + wp * 64 + i * 32 + 8 * g4 + 4 * hi;       // unit within the direction
                const long long go = ((long long)t * B + b) * ldg + (long long)d * 4 * H + (un / 32) * 128 + (un % 32);
                const long long so = ((long long)t * B + b) * ldh + (long long)d * H + un;
                const long long sp = ((long long)tprev * B + b) * ldh + (long long)d * H + un;
                const long long co = ((long long)d * B + b) * H + un;
                float dh[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) dh[e] = acc[i][j][4 * g4 + e];
                if (dHout) {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>(dHout + so);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh[e] += (float)v[e];
                }
                if (step == 0) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(dh_fin + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh[e] += v[e];
                }
                f32x4 dc = {0.f, 0.f, 0.f, 0.f};
                if (has_next) dc = *reinterpret_cast<const f32x4*>(dc_carry + co);
                if (gc_in) {                                     // adjoint of the saved c_t (second-order pass)
                    const f32x4 v = *reinterpret_cast<const f32x4*>(gc_in + so);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dc[e] += v[e];
                }
                bf16x4 gt[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) gt[g] = *reinterpret_cast<const bf16x4*>(G + go + g * 32);
                const f32x4 c = *reinterpret_cast<const f32x4*>(Call + so);
                f32x4 cp = {0.f, 0.f, 0.f, 0.f};
                if (has_prev) cp = *reinterpret_cast<const f32x4*>(Call + sp);
                f32x4 xg[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (gg_in) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) xg[g] = *reinterpret_cast<const f32x4*>(gg_in + go + g * 32);
                }
                float da[4][4];
                f32x4 dcn, dctv, dhtv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gi = (float)gt[0][e], gf = (float)gt[1][e], gg = (float)gt[2][e], gov = (float)gt[3][e];
                    const float tc = gmt_tanh(c[e]);
                    const float dct = dc[e] + dh[e] * gov * (1.f - tc * tc);
                    da[0][e] = (dct * gg + xg[0][e]) * gi * (1.f - gi);
                    da[1][e] = (dct * cp[e] + xg[1][e]) * gf * (1.f - gf);
                    da[2][e] = (dct * gi + xg[2][e]) * (1.f - gg * gg);
                    da[3][e] = (dh[e] * tc + xg[3][e]) * gov * (1.f - gov);
                    dcn[e] = dct * gf;
                    dctv[e] = dct; dhtv[e] = dh[e];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<bf16x4*>(dA + go + g * 32) = pack4(da[g][0], da[g][1], da[g][2], da[g][3]);
                *reinterpret_cast<f32x4*>(dc_carry + co) = dcn;
                if (dct_all) {
                    *reinterpret_cast<f32x4*>(dct_all + so) = dctv;
                    *reinterpret_cast<f32x4*>(dht_all + so) = dhtv;
                }
            }
    }
}

__global__ __launch_bounds__(NTHREADS)
void lstm_step_bwd2_kernel(const __bf16* __restrict__ PH, const __bf16* __restrict__ G, const float* __restrict__ Call,
                           const float* __restrict__ dct_all, const float* __restrict__ dht_all,
                           __bf16* __restrict__ hhat_all, float* __restrict__ hhat_fin, float* __restrict__ gg_out,
                           float* __restrict__ gc_out, float* __restrict__ chat_carry, const __bf16* __restrict__ Whh,
                           int L, int B, int H, int ndir, int step, int n_mt, int n_pt)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int d, pt, mt;
    patch_of_block(blockIdx.x, gridDim.x, n_mt, n_pt, d, pt, mt);
    // the adjoint sweep runs in the FORWARD order of the direction: its step s is the first backward sweep's step L-1-s
    const int t = d == 0 ? step : L - 1 - step;
    const int tprev = d == 0 ? t - 1 : t + 1;                  // previous step of this sweep = the forward pass's predecessor
    const bool has_prev = step > 0;
    const long long ldg = (long long)ndir * 4 * H, ldh = (long long)ndir * H;
    const long long row0 = (long long)mt * GP2_TQ;              // 256 gate columns x 64 rows
    f32x16 acc[4][1];
    zero_acc<4, 1>(acc);
    mainloop_ring<4, 1, 64, 3>(Whh + ((long long)d * 4 * H + (long long)pt * TP) * H, H,
                               hhat_all + ((long long)tprev * B + row0) * ldh + (long long)d * H, ldh, has_prev ? H / BK : 0, lds, acc, wave, lane);
    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    constexpr int j = 0;
    {
        const long long b = row0 + wq * 32 + lo;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int un = pt * 64 + wp * 32 + 8 * g4 + 4 * hi;
            const long long go = ((long long)t * B + b) * ldg + (long long)d * 4 * H + (un / 32) * 128 + (un % 32);
            const long long so = ((long long)t * B + b) * ldh + (long long)d * H + un;
            const long long sp = ((long long)tprev * B + b) * ldh + (long long)d * H + un;
            const long long co = ((long long)d * B + b) * H + un;
            bf16x4 gt[4], ph[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                gt[g] = *reinterpret_cast<const bf16x4*>(G + go + g * 32);
                ph[g] = *reinterpret_cast<const bf16x4*>(PH + go + g * 32);
            }
            const f32x4 cn = *reinterpret_cast<const f32x4*>(Call + so);
            f32x4 cpv = {0.f, 0.f, 0.f, 0.f}, chp = {0.f, 0.f, 0.f, 0.f};
            if (has_prev) {
                cpv = *reinterpret_cast<const f32x4*>(Call + sp);
                chp = *reinterpret_cast<const f32x4*>(chat_carry + co);
            }
            const f32x4 dctv = *reinterpret_cast<const f32x4*>(dct_all + so);
            const f32x4 dhtv = *reinterpret_cast<const f32x4*>(dht_all + so);
            f32x4 ogi, ogf, ogg, ogo, ogc, hnew, chn, gcp;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                const float ig = (float)gt[0][e], fg = (float)gt[1][e], gg = (float)gt[2][e], og = (float)gt[3][e];
                const float cp = cpv[e], dct = dctv[e], dht = dhtv[e];
                const float tc = gmt_tanh(cn[e]);
                const float omt = 1.f - tc * tc;
                const float di = dct * gg, dg = dct * ig, df = dct * cp, dov = dht * tc;
                const float a_i = acc[0][j][r] + (float)ph[0][e], a_f = acc[1][j][r] + (float)ph[1][e];
                const float a_g = acc[2][j][r] + (float)ph[2][e], a_o = acc[3][j][r] + (float)ph[3][e];
                const float adi = a_i * ig * (1.f - ig);
                const float adf = a_f * fg * (1.f - fg);
                const float adg = a_g * (1.f - gg * gg);
                const float ado = a_o * og * (1.f - og);
                const float adct = adi * gg + adg * ig + adf * cp + chp[e] * fg;
                ogi[e] = a_i * di * (1.f - 2.f * ig) + adg * dct;
                ogf[e] = a_f * df * (1.f - 2.f * fg) + chp[e] * dct;
                ogg[e] = a_g * dg * (-2.f * gg) + adi * dct;
                ogo[e] = a_o * dov * (1.f - 2.f * og) + adct * dht * omt;
                const float atc = adct * dht * og * (-2.f * tc) + ado * dht;
                ogc[e] = atc * omt;
                hnew[e] = adct * og * omt + ado * tc;
                chn[e] = adct;
                gcp[e] = adf * dct;                              // what c_{t-1} receives through the forget-gate adjoint
            }
            *reinterpret_cast<f32x4*>(gg_out + go + 0 * 32) = ogi;
            *reinterpret_cast<f32x4*>(gg_out + go + 1 * 32) = ogf;
            *reinterpret_cast<f32x4*>(gg_out + go + 2 * 32) = ogg;
            *reinterpret_cast<f32x4*>(gg_out + go + 3 * 32) = ogo;
            *reinterpret_cast<f32x4*>(gc_out + so) = ogc;
            if (has_prev) {                                      // the previous launch wrote its own part of g_c[t_prev]
                f32x4 v = *reinterpret_cast<const f32x4*>(gc_out + sp);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += gcp[e];
                *reinterpret_cast<f32x4*>(gc_out + sp) = v;
            }
            *reinterpret_cast<f32x4*>(chat_carry + co) = chn;
            *reinterpret_cast<bf16x4*>(hhat_all + so) = pack4(hnew[0], hnew[1], hnew[2], hnew[3]);
            if (step == L - 1) *reinterpret_cast<f32x4*>(hhat_fin + co) = hnew;
        }
    }
}

}  // namespace

extern "C" int gmt_bf16_lstm_layer_fwd(void* gates, float* c_all, void* h_all, void* h_drop, const void* w_hh,
                                       const int* lengths, int ragged, int L, int B, int H, int ndir, float drop_p,
                                       unsigned int drop_seed, hipStream_t stream)
{
    GMT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (h_drop == nullptr) == (drop_p == 0.f));
    GMT_CHECK_ARG(drop_p == 0.f || (long long)L * B * ndir * H < 0xffffffffLL);   // the mask's element counter is 32 bits (no mask, no bound)
    GMT_CHECK_ARG(gates && c_all && h_all && w_hh && L >= 1 && (ndir == 1 || ndir == 2));
    GMT_CHECK_ARG(B > 0 && B % 256 == 0 && H > 0 && (4 * H) % TP == 0 && H % BK == 0 && (!ragged || lengths));
    const int n_mt = B / FWD_TQ, n_pt = 4 * H / TP;
    auto k = ragged ? lstm_step_fwd_kernel<true> : lstm_step_fwd_kernel<false>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS) != hipSuccess)
        return GMT_ERR_HIP;
    const int ng = row_groups(B, FWD_TQ), g_mt = n_mt / ng;
    const int rc = run_row_groups(ng, L, stream, [&](int g, int step, hipStream_t st) {
        hipLaunchKernelGGL(k, dim3(ndir * g_mt * n_pt), dim3(NTHREADS), FWD_LDS, st, (__bf16*)gates, c_all, (__bf16*)h_all,
                           (__bf16*)h_drop, (const __bf16*)w_hh, lengths, L, B, H, ndir, step, g * g_mt, g_mt, n_pt, drop_p, drop_seed);
    });
    if (rc != GMT_OK) return rc;
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_bf16_lstm_layer_bwd(void* gates, const float* c_all, const void* dh_out, const float* dh_fin,
                                       float* dc_carry, float* dh_carry, const void* w_hhT, const int* lengths,
                                       int ragged, int L, int B, int H, int ndir, float drop_p, unsigned int drop_seed,
                                       hipStream_t stream)
{
    GMT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || dh_out));
    GMT_CHECK_ARG(gates && c_all && dh_fin && dc_carry && w_hhT && L >= 1 && (ndir == 1 || ndir == 2));
    GMT_CHECK_ARG(B > 0 && B % 256 == 0 && H > 0 && H % 128 == 0 && (!ragged || (lengths && dh_carry)));
    // 128-row tiles when they fill the chip twice over (two workgroups per CU), 64-row tiles otherwise (B = 4096: 512 workgroups)
    const int n_pt = H / 128;
    const bool small = ndir * (B / 128) * n_pt < 512;
    const int n_mt = B / (small ? 64 : 128);
    auto k = small ? (ragged ? lstm_step_bwd_kernel<true, 1> : lstm_step_bwd_kernel<false, 1>)
                   : (ragged ? lstm_step_bwd_kernel<true, 2> : lstm_step_bwd_kernel<false, 2>);
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS) != hipSuccess)
        return GMT_ERR_HIP;
    const int ng = row_groups(B, small ? 64 : 128), g_mt = n_mt / ng;
    const int rc = run_row_groups(ng, L, stream, [&](int g, int step, hipStream_t st) {
        hipLaunchKernelGGL(k, dim3(ndir * g_mt * n_pt), dim3(NTHREADS), BWD_LDS, st, (__bf16*)gates, c_all,
                           (const __bf16*)dh_out, dh_fin, dc_carry, dh_carry, (const __bf16*)w_hhT, lengths, L, B, H, ndir, step,
                           g * g_mt, g_mt, n_pt, drop_p, drop_seed);
    });
    if (rc != GMT_OK) return rc;
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_bf16_lstm_layer_bwd_gp(const void* gates, void* dA, const float* c_all, const void* dh_out, const float* dh_fin,
                                          const float* gg_in, const float* gc_in, float* dct_all, float* dht_all,
                                          float* dc_carry, const void* w_hhT, int L, int B, int H, int ndir, hipStream_t stream)
{
    using T = Tile<2, 1>;
    GMT_CHECK_ARG(gates && dA && c_all && dh_fin && dc_carry && w_hhT && L >= 1 && (ndir == 1 || ndir == 2));
    GMT_CHECK_ARG(B > 0 && B % 256 == 0 && H > 0 && H % T::TP == 0 && (dct_all == nullptr) == (dht_all == nullptr));
    const int n_mt = B / T::TQ, n_pt = H / T::TP;
    if (hipFuncSetAttribute((const void*)lstm_step_bwd_gp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GP_BWD_LDS) != hipSuccess)
        return GMT_ERR_HIP;
    for (int step = 0; step < L; ++step)
        hipLaunchKernelGGL(lstm_step_bwd_gp_kernel, dim3(ndir * n_mt * n_pt), dim3(NTHREADS), GP_BWD_LDS, stream,
                           (const __bf16*)gates, (__bf16*)dA, c_all, (const __bf16*)dh_out, dh_fin, gg_in, gc_in, dct_all, dht_all,
                           dc_carry, (const __bf16*)w_hhT, L, B, H, ndir, step, n_mt, n_pt);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_bf16_lstm_layer_bwd2(const void* pre_hat, const void* gates, const float* c_all, const float* dct_all,
                                        const float* dht_all, void* hhat_all, float* hhat_fin, float* g_gates, float* g_c,
                                        float* chat_carry, const void* w_hh, int L, int B, int H, int ndir, hipStream_t stream)
{
    GMT_CHECK_ARG(pre_hat && gates && c_all && dct_all && dht_all && hhat_all && hhat_fin && g_gates && g_c && chat_carry && w_hh);
    GMT_CHECK_ARG(L >= 1 && (ndir == 1 || ndir == 2) && B > 0 && B % 256 == 0 && H > 0 && (4 * H) % TP == 0 && H % BK == 0);
    const int n_mt = B / GP2_TQ, n_pt = 4 * H / TP;
    if (hipFuncSetAttribute((const void*)lstm_step_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GP_BWD2_LDS) != hipSuccess)
        return GMT_ERR_HIP;
    for (int step = 0; step < L; ++step)
        hipLaunchKernelGGL(lstm_step_bwd2_kernel, dim3(ndir * n_mt * n_pt), dim3(NTHREADS), GP_BWD2_LDS, stream,
                           (const __bf16*)pre_hat, (const __bf16*)gates, c_all, dct_all, dht_all, (__bf16*)hhat_all, hhat_fin,
                           g_gates, g_c, chat_carry, (const __bf16*)w_hh, L, B, H, ndir, step, n_mt, n_pt);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_bf16_set_row_groups(int n)
{
    if (n < 1 || n > MAX_GROUPS) return GMT_ERR_BAD_ARG;
    row_groups_setting().store(n);
    return GMT_OK;
}
