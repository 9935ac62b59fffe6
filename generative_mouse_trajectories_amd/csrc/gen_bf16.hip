// The generator's autoregressive rollout (Generator.forward / generate, RG.py:508-690) at H = 512 in bf16 operands / fp32
// accumulation - the no-grad form (sampling, and the fake batch of a D step).  Built from the blocks of the bf16 critic
// (bf16_tile.h): the recurrent weights do not fit a CU at this width, so a timestep is three launches -
//   layer 0:  a = cond_pre + [h0_{t-1} | u_t] [W_hh_l0 | W_u]^T          K = H + 64   -> c0, h0_t
//   layer 1:  a = bias    + [h0_t | h1_{t-1}] [W_ih_l1 | W_hh_l1]^T      K = 2H       -> c1, h1_t
//   head:     y = lrelu(h1_t W1^T + b1), o = y W2^T + b2; (dx, dy, softplus(o2) + 0.001) -> out_seq[t]; position update,
//             teacher forcing, and the next step's 7 input features u_{t+1} = (cur, end - pos, |.|, atan2) (RG.py:561-595)
// - with the kernel boundary as the grid-wide hand-off, like the critic's recurrence.  The concatenated operands live in
// ping-pong buffers x0 [2][B][H+64] and x1 [2][B][2H] that the epilogues write into directly, so every contraction reads
// ONE K-contiguous operand.  The 7 input features ride in the spare slots of layer 0's last K tile as bf16 hi + lo pairs
// (slots 0-6 u_hi, 7-13 u_lo, both against W_u): they span 1e-3 .. 1e0 and steer the trajectory, 8 bits are not enough.
// The condition term cond W_cond^T + b is constant over time and computed once by the caller (gmt_bf16_gemm_nt).
#include "bf16_tile.h"
#include "row_groups.h"
#include "lstm_tile.h"      // gmt_keep: the generator's counter-based inter-layer dropout mask, as in the fp32 rollout

namespace {

using namespace gmt_bf16;

constexpr int G_TQ = 128;                                      // rows per workgroup of the cell kernels
constexpr int G_NST = 3, G_BK = 64;                            // ring slots of the contractions; k per slot
constexpr int G_LDS = G_NST * (256 + 128) * G_BK * 2;          // 144 KiB: one workgroup per CU
// the cell kernel's epilogue images, all 128 rows at once: pre-activations / gates [128][512 B], c [128][256 B], h, dropped h
constexpr int G_PRE_OFF = 0, G_C_OFF = 128 * 512, G_H_OFF = G_C_OFF + 128 * 256, G_HD_OFF = G_H_OFF + 128 * 128;
static_assert(G_HD_OFF + 128 * 128 <= G_LDS, "epilogue images");
constexpr int G_CELL_THREADS = 512;                            // 8 waves = 2 (gate slabs) x 4 (32-row groups): two per SIMD

// One LSTM layer step over all rows.  X: [B][ldx] bf16 (K = 32 nk32 columns used); W: [4H slab order][32 nk32] bf16.
// pre: [B][4H] bf16 plane (slab order) or NULL (then bias[4H]); c_prev -> c_out [B][H] fp32; h -> h0 (and h1 if given).
// The contraction is bounded by what a CU's vector-memory path delivers to the LDS (64 B/clk: 48 KiB per 64-deep K-tile
// against 1 024 cycles of MFMAs) and an LDS-DMA instruction holds its wave's issue for 60-180 cycles, so the tile is worked
// by two waves per SIMD: one wave's DMA issue and fragment waits run under the other's MFMAs.
__global__ __launch_bounds__(G_CELL_THREADS)
void gen_cell_kernel(const __bf16* __restrict__ W, const __bf16* __restrict__ X, long long ldx, int nk32,
                     const __bf16* __restrict__ pre, const float* __restrict__ bias,
                     const float* __restrict__ c_prev, float* __restrict__ c_out,
                     __bf16* __restrict__ h0, long long ldh0, __bf16* __restrict__ h1, long long ldh1,
                     __bf16* __restrict__ gates_out, int H, int mt0, int n_mt, int n_pt, float drop_p, uint32_t drop_seed, long long drop_row0)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int d, pt, mt;
    patch_of_block(blockIdx.x, gridDim.x, n_mt, n_pt, d, pt, mt);
    const long long row0 = (long long)(mt0 + mt) * G_TQ;         // this launch covers row tiles [mt0, mt0 + n_mt)
    const long long K = 32LL * nk32;

    f32x16 acc[4][1];
    zero_acc<4, 1>(acc);
    GMT_PHASE(0);
    mainloop_ring<4, 1, G_BK, G_NST, 4>(W + (long long)pt * TP * K, K, X + row0 * ldx, ldx, nk32 * 32 / G_BK, lds, acc, wave, lane);
    GMT_PHASE(1);

    const int wp = wave >> 2, wq = wave & 3, lo = lane & 31, hi = lane >> 5;
    using GI = RowImage<512>;
    using CI = RowImage<256>;
    using HI = RowImage<128>;
    const int rl = wq * 32 + lo;                                 // this lane's row within the tile (= image row)
    const char* Pt = pre ? (const char*)(pre + row0 * 4 * H + pt * TP) : nullptr;
    const char* Cp = (const char*)(c_prev + row0 * H + pt * (TP / 4));
    char* Co = (char*)(c_out + row0 * H + pt * (TP / 4));
    char* Go = gates_out ? (char*)(gates_out + row0 * 4 * H + pt * TP) : nullptr;     // training: the activated gates, for BPTT
    // nn.LSTM's inter-layer dropout (RG.py:451-457): the copy of h that feeds the next layer (h1) goes through the mask
    // keep(seed, (t*B + row)*H + unit) - ops.dropout_keep_mask(T, B, H, p, seed) replays it; the recurrent copy (h0) does not
    const uint32_t thresh24 = (uint32_t)(drop_p * 16777216.0f);
    const float keep_scale = 1.0f / (1.0f - drop_p);
    auto rowmap = [](int r) { return r; };
    __syncthreads();                                             // the ring's last tile is read by everybody
    if (Pt) GI::dma_in(Pt, 4LL * H * 2, lds + G_PRE_OFF, 128, rowmap, wave, lane, 8);
    CI::dma_in(Cp, (long long)H * 4, lds + G_C_OFF, 128, rowmap, wave, lane, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GMT_PHASE(2);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int ucol = wp * 32 + 8 * g4 + 4 * hi;
        float a[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (Pt) {
                const bf16x4 p = *reinterpret_cast<const bf16x4*>(lds + G_PRE_OFF + GI::addr(rl, (wp * 128 + g * 32 + 8 * g4 + 4 * hi) * 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) a[g][e] = acc[g][0][4 * g4 + e] + (float)p[e];
            } else {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + pt * TP + wp * 128 + g * 32 + 8 * g4 + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[g][e] = acc[g][0][4 * g4 + e] + b[e];
            }
        }
        const f32x4 cp = *reinterpret_cast<const f32x4*>(lds + G_C_OFF + CI::addr(rl, ucol * 4));
        f32x4 cn;
        float hn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[0][e] = gmt_sigmoid(a[0][e]); a[1][e] = gmt_sigmoid(a[1][e]);
            a[2][e] = gmt_tanh(a[2][e]);    a[3][e] = gmt_sigmoid(a[3][e]);
            cn[e] = a[1][e] * cp[e] + a[0][e] * a[2][e];
            hn[e] = a[3][e] * gmt_tanh(cn[e]);
        }
        if (Go) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<bf16x4*>(lds + G_PRE_OFF + GI::addr(rl, (wp * 128 + g * 32 + 8 * g4 + 4 * hi) * 2)) =
                    pack4(a[g][0], a[g][1], a[g][2], a[g][3]);
        }
        *reinterpret_cast<f32x4*>(lds + G_C_OFF + CI::addr(rl, ucol * 4)) = cn;
        *reinterpret_cast<bf16x4*>(lds + G_H_OFF + HI::addr(rl, ucol * 2)) = pack4(hn[0], hn[1], hn[2], hn[3]);
        if (h1 && thresh24) {
            const uint32_t idx = (uint32_t)((drop_row0 + row0 + rl) * H + pt * (TP / 4) + ucol);
#pragma unroll
            for (int e = 0; e < 4; ++e) hn[e] = gmt_keep(idx + e, drop_seed, thresh24) ? hn[e] * keep_scale : 0.f;
            *reinterpret_cast<bf16x4*>(lds + G_HD_OFF + HI::addr(rl, ucol * 2)) = pack4(hn[0], hn[1], hn[2], hn[3]);
        }
    }
    __syncthreads();
    GMT_PHASE(3);
    CI::copy_out(Co, (long long)H * 4, lds + G_C_OFF, 128, rowmap, wave, lane, 8);
    if (Go) GI::copy_out(Go, 4LL * H * 2, lds + G_PRE_OFF, 128, rowmap, wave, lane, 8);
    HI::copy_out((char*)(h0 + row0 * ldh0 + pt * (TP / 4)), ldh0 * 2, lds + G_H_OFF, 128, rowmap, wave, lane, 8);
    if (h1) HI::copy_out((char*)(h1 + row0 * ldh1 + pt * (TP / 4)), ldh1 * 2, lds + (thresh24 ? G_HD_OFF : G_H_OFF), 128, rowmap, wave, lane, 8);
    GMT_PHASE(4);
#ifdef GMT_PHASE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GMT_PHASE(5);
#endif
}

// Head + step bookkeeping.  HH = H/2 must be 256 (one weight tile): the three outputs of a row are sums over all HH units.
struct head_args {
    const __bf16* wh1; const float* bh1; const float* wh2; const float* bh2;
    const __bf16* h; long long ldh;                  // h1_t rows
    const float* ends; const float* target; const unsigned char* tf_mask;
    float* out_seq; float* pos; float* tf_pos;
    __bf16* u_next; long long ldu;                   // &x0_next[0][H]: 14 slots written per row
    __bf16* y_out; float* oraw_out;                  // training: [B][HH] head activation, [B] pre-softplus dt logit (or NULL)
    int t, B, H, blk0;                               // blk0: first row tile of this launch (row groups)
};

// NQB = 1: 64-row tiles (B / 64 workgroups - the launch is small, so the smaller tile is what spreads it over the chip);
// NQB = 2: 128-row tiles for batches that fill the chip either way.  One workgroup per CU (the ring takes the LDS), so the
// epilogue may use the whole register file.
template <int NQB>
__global__ __launch_bounds__(NTHREADS)
void gen_head_kernel(head_args a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int TQr = 64 * NQB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long row0 = (long long)(a.blk0 + blockIdx.x) * TQr;
    const int HH = a.H / 2;

    f32x16 acc[4][NQB];
    zero_acc<4, NQB>(acc);
    GMT_PHASE(10);
    mainloop_ring<4, NQB, G_BK, G_NST>(a.wh1, a.H, a.h + row0 * a.ldh, a.ldh, a.H / G_BK, lds, acc, wave, lane);
    GMT_PHASE(11);
    __syncthreads();
    float* w2s = reinterpret_cast<float*>(lds);               // [3][HH]
    float* bs = w2s + 3 * 256;                                // [HH]
    float* red = bs + 256;                                    // [2][TQr][4]
    for (int i = threadIdx.x; i < 3 * HH; i += NTHREADS) w2s[i] = a.wh2[i];
    for (int i = threadIdx.x; i < HH; i += NTHREADS) bs[i] = a.bh1[i];
    __syncthreads();
    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    float o[NQB][3];
#pragma unroll
    for (int j = 0; j < NQB; ++j) o[j][0] = o[j][1] = o[j][2] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int u = wp * 128 + i * 32 + 8 * g4 + 4 * hi;
            const f32x4 b = *reinterpret_cast<const f32x4*>(bs + u);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(w2s + u);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(w2s + HH + u);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(w2s + 2 * HH + u);
#pragma unroll
            for (int j = 0; j < NQB; ++j) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[i][j][4 * g4 + e] + b[e];
                    y[e] = v > 0.f ? v : 0.2f * v;
                }
                if (a.y_out)
                    *reinterpret_cast<bf16x4*>(a.y_out + (row0 + wq * (32 * NQB) + j * 32 + lo) * HH + u) = pack4(y[0], y[1], y[2], y[3]);
                o[j][0] += y[0] * w0[0] + y[1] * w0[1] + y[2] * w0[2] + y[3] * w0[3];
                o[j][1] += y[0] * w1[0] + y[1] * w1[1] + y[2] * w1[2] + y[3] * w1[3];
                o[j][2] += y[0] * w2[0] + y[1] * w2[1] + y[2] * w2[2] + y[3] * w2[3];
            }
        }
#pragma unroll
    for (int j = 0; j < NQB; ++j) {
#pragma unroll
        for (int k = 0; k < 3; ++k) o[j][k] += __shfl_xor(o[j][k], 32, 64);
        if (hi == 0) {
            float* r = red + (wp * TQr + wq * (32 * NQB) + j * 32 + lo) * 4;
            r[0] = o[j][0]; r[1] = o[j][1]; r[2] = o[j][2];
        }
    }
    __syncthreads();
    GMT_PHASE(12);
    if (threadIdx.x < TQr) {
        const int rr = threadIdx.x;
        const long long row = row0 + rr;
        const float o0 = red[rr * 4 + 0] + red[(TQr + rr) * 4 + 0] + a.bh2[0];
        const float o1 = red[rr * 4 + 1] + red[(TQr + rr) * 4 + 1] + a.bh2[1];
        const float o2 = red[rr * 4 + 2] + red[(TQr + rr) * 4 + 2] + a.bh2[2];
        const float dt = (o2 > 20.f ? o2 : log1pf(expf(o2))) + 0.001f;                 // F.softplus, RG.py:577-580
        float* os = a.out_seq + ((long long)a.t * a.B + row) * 3;
        os[0] = o0; os[1] = o1; os[2] = dt;
        if (a.oraw_out) a.oraw_out[row] = o2;
        float px = a.pos[row * 2] + o0, py = a.pos[row * 2 + 1] + o1;                  // RG.py:585
        float c0 = o0, c1 = o1, c2 = dt;
        if (a.target) {
            const float* tg = a.target + ((long long)a.t * a.B + row) * 3;
            const float tx = a.tf_pos[row * 2] + tg[0], ty = a.tf_pos[row * 2 + 1] + tg[1];   // RG.py:591-593, running form
            a.tf_pos[row * 2] = tx; a.tf_pos[row * 2 + 1] = ty;
            if (a.tf_mask && a.tf_mask[a.t]) { c0 = tg[0]; c1 = tg[1]; c2 = tg[2]; px = tx; py = ty; }
        }
        a.pos[row * 2] = px; a.pos[row * 2 + 1] = py;
        const float rx = a.ends[row * 2] - px, ry = a.ends[row * 2 + 1] - py;          // RG.py:502-504
        const float u[7] = {c0, c1, c2, rx, ry, sqrtf(rx * rx + ry * ry + 1e-8f), atan2f(ry, rx)};
        __bf16* un = a.u_next + row * a.ldu;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const __bf16 h = (__bf16)u[s];
            un[s] = h;
            un[7 + s] = (__bf16)(u[s] - (float)h);
        }
    }
    GMT_PHASE(13);
}

// ---------------------------------------------------------------------------------------------------------------------
// BPTT of the rollout (autograd over RG.py:559-595), three launches per timestep in reverse order:
//   book:   du_{t+1} = da0_{t+1} W_u (what step t+1's 7 inputs received) -> the gradients of the fed-back output and of the
//           running position; total d out_t; head backward up to d(ypre_t)                                   (one wave a row)
//   layer 1: dh1_t = d(ypre_t) W1 + da1_{t+1} W_hh_l1 -> cell backward -> da1_t        (two contractions into one accumulator)
//   layer 0: dh0_t = da1_t W_ih_l1 + da0_{t+1} W_hh_l0 -> cell backward -> da0_t
// The saved gates are overwritten in place by d(pre-activation); the weight gradients are dense GEMMs over all T*B
// row-steps afterwards (gmt_bf16_wgrad_tn against the saved operands x0 / x1).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GB_G_OFF = 0, GB_C_OFF = 64 * 512, GB_CP_OFF = GB_C_OFF + 64 * 256;          // as the critic's backward step
constexpr int GB_LDS = 72 * 1024;                              // >= the contraction's ring (3 x 24 KiB with 64-row tiles) and the epilogue images

template <int NQB>
__device__ __forceinline__ void gb_contract(const __bf16* __restrict__ Pg, long long ldp, const __bf16* __restrict__ Qg,
                                            long long ldq, int K, char* lds, f32x16 (&acc)[2][NQB], int wave, int lane)
{
    if constexpr (NQB == 1) mainloop_ring<2, 1, 64, 3>(Pg, ldp, Qg, ldq, K / 64, lds, acc, wave, lane);
    else mainloop_ring<2, 2, 32, 3>(Pg, ldp, Qg, ldq, K / 32, lds, acc, wave, lane);
}

template <int NQB>
__global__ __launch_bounds__(NTHREADS, 2)
void gen_bwd_cell_kernel(const __bf16* __restrict__ P1, const __bf16* __restrict__ Q1, long long ldq1, int nk1,
                         const __bf16* __restrict__ P2, const __bf16* __restrict__ Q2, long long ldq2, int nk2,
                         __bf16* __restrict__ gates, const float* __restrict__ c_t, const float* __restrict__ c_prev,
                         float* __restrict__ dc_carry, int has_next, int H, int mt0, int n_mt, int n_pt,
                         float drop_p, uint32_t drop_seed, long long drop_row0)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using T = Tile<2, NQB>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int d, pt, mt;
    patch_of_block(blockIdx.x, gridDim.x, n_mt, n_pt, d, pt, mt);
    const long long row0 = (long long)(mt0 + mt) * T::TQ;
    f32x16 acc[2][NQB];
    zero_acc<2, NQB>(acc);
    gb_contract<NQB>(P1 + (long long)pt * T::TP * (nk1 * BK), (long long)nk1 * BK, Q1 + row0 * ldq1, ldq1, nk1 * BK, lds, acc, wave, lane);
    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    const uint32_t thresh24 = (uint32_t)(drop_p * 16777216.0f);
    if (thresh24) {
        // the first contraction is the gradient of the DROPPED copy of h (what the layer above read): through the mask
        const float keep_scale = 1.0f / (1.0f - drop_p);
#pragma unroll
        for (int j = 0; j < NQB; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int u = pt * 128 + wp * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const uint32_t idx = (uint32_t)((drop_row0 + row0 + wq * (32 * NQB) + j * 32 + lo) * H + u);
                    acc[i][j][r] = gmt_keep(idx, drop_seed, thresh24) ? acc[i][j][r] * keep_scale : 0.f;
                }
    }
    if (nk2 > 0) {
        __syncthreads();                                        // the first contraction's last tile is read by everybody
        gb_contract<NQB>(P2 + (long long)pt * T::TP * (nk2 * BK), (long long)nk2 * BK, Q2 + row0 * ldq2, ldq2, nk2 * BK, lds, acc, wave, lane);
    }
    char* Gt = (char*)(gates + row0 * 4 * H + pt * 512);
    const char* Ct = (const char*)(c_t + row0 * H + pt * 128);
    const char* Cp = (const char*)(c_prev + row0 * H + pt * 128);
    float* dcc = dc_carry + row0 * H + pt * 128;

    using GI = RowImage<512>;
    using CI = RowImage<256>;
    const int rl = wq * 32 + lo;
#pragma unroll
    for (int j = 0; j < NQB; ++j) {
        auto rowmap = [j](int r) { return (r >> 5) * (32 * NQB) + j * 32 + (r & 31); };
        const int brow = wq * (32 * NQB) + j * 32 + lo;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            auto gcol = [i](int c) { return (((c >> 4) * 2 + i) * 128) * 2 + (c & 15) * 16; };
            auto ccol = [i](int c) { return (((c >> 3) * 2 + i) * 32) * 4 + (c & 7) * 16; };
            __syncthreads();
            GI::dma_in(Gt, 4LL * H * 2, lds + GB_G_OFF, 64, rowmap, gcol, wave, lane);
            CI::dma_in(Ct, (long long)H * 4, lds + GB_C_OFF, 64, rowmap, ccol, wave, lane);
            CI::dma_in(Cp, (long long)H * 4, lds + GB_CP_OFF, 64, rowmap, ccol, wave, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int slab_u = (wp * 2 + i) * 32;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int u = slab_u + 8 * g4 + 4 * hi;
                const long long co = (long long)brow * H + u;
                f32x4 dc = {0.f, 0.f, 0.f, 0.f};
                if (has_next) dc = *reinterpret_cast<const f32x4*>(dcc + co);
                const int gbyte = wp * 256 + (8 * g4 + 4 * hi) * 2;
                bf16x4 gt[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) gt[g] = *reinterpret_cast<const bf16x4*>(lds + GB_G_OFF + GI::addr(rl, gbyte + g * 64));
                const int cbyte = wp * 128 + (8 * g4 + 4 * hi) * 4;
                const f32x4 c = *reinterpret_cast<const f32x4*>(lds + GB_C_OFF + CI::addr(rl, cbyte));
                const f32x4 cp = *reinterpret_cast<const f32x4*>(lds + GB_CP_OFF + CI::addr(rl, cbyte));
                float da[4][4];
                f32x4 dcn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dh = acc[i][j][4 * g4 + e];
                    const float gi = (float)gt[0][e], gf = (float)gt[1][e], gg = (float)gt[2][e], go = (float)gt[3][e];
                    const float tc = gmt_tanh(c[e]);
                    const float dct = dc[e] + dh * go * (1.f - tc * tc);
                    da[0][e] = dct * gg * gi * (1.f - gi);
                    da[1][e] = dct * cp[e] * gf * (1.f - gf);
                    da[2][e] = dct * gi * (1.f - gg * gg);
                    da[3][e] = dh * tc * go * (1.f - go);
                    dcn[e] = dct * gf;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<bf16x4*>(lds + GB_G_OFF + GI::addr(rl, gbyte + g * 64)) = pack4(da[g][0], da[g][1], da[g][2], da[g][3]);
                *reinterpret_cast<f32x4*>(dcc + co) = dcn;
            }
            __syncthreads();
            GI::copy_out(Gt, 4LL * H * 2, lds + GB_G_OFF, 64, rowmap, gcol, wave, lane);
        }
    }
}

// Bookkeeping + head backward of step t (t = -1: only the gradient of the first input).  One wave per row, 16 rows a block.
struct book_args {
    const __bf16* da0_next;     // [B][4H] d(pre-activation) of layer 0 at step t+1, or NULL (t = T-1)
    const __bf16* w_u;          // [4H][8] bf16: W_u in slab row order, 8th column zero
    const __bf16* u_next;       // &x0[t+1][0][H]: the 7 inputs of step t+1 as hi (0-6) and lo (7-13) parts, row stride ldu
    long long ldu;
    const float* d_out;         // [B][3] external gradient of out_t (NULL for t = -1)
    const __bf16* y; const float* oraw; const float* wh2;       // [B][HH], [B], [3][HH]
    float* p_carry;             // [B][2]: gradient of the position before step t+1 that flows on to step t+2 (in: P_{t+2}, out: P_{t+1})
    __bf16* dypre; float* d_o;  // [B][HH] bf16, [B][3] fp32 out
    float* d_in0;               // [B][3] out (t = -1 only)
    int tf_t, tf_next;          // teacher-forcing flags of steps t and t+1 (host copies of the mask)
    int t, B, H, blk0;          // blk0: first 16-row block of this launch (row groups)
};

__global__ __launch_bounds__(NTHREADS)
void gen_bwd_book_kernel(book_args a)
{
    __shared__ __attribute__((aligned(16))) __bf16 wus[2048 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int HH = a.H / 2, G4 = 4 * a.H;
    if (a.da0_next)
        for (int i = threadIdx.x; i < G4; i += NTHREADS)
            *reinterpret_cast<bf16x8*>(wus + i * 8) = *reinterpret_cast<const bf16x8*>(a.w_u + (long long)i * 8);
    __syncthreads();
    // du_{t+1} = da0_{t+1} W_u for the wave's 4 rows at once: all 16 row segments are requested before any is used (one exposed
    // memory latency instead of four), and every W_u row read from the LDS serves the 4 rows
    float du4[4][7];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int s2 = 0; s2 < 7; ++s2) du4[rr][s2] = 0.f;
    if (a.da0_next) {
        constexpr int NPASS = 4;                                // 4H / 512 with H = 512 (checked by the entry point)
        const long long rowb = (long long)(a.blk0 + blockIdx.x) * 16 + wave * 4;
        bf16x8 dv[4][NPASS];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int c = 0; c < NPASS; ++c)
                dv[rr][c] = *reinterpret_cast<const bf16x8*>(a.da0_next + (rowb + rr) * G4 + c * 512 + lane * 8);
#pragma unroll
        for (int c = 0; c < NPASS; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bf16x8 w = *reinterpret_cast<const bf16x8*>(wus + (c * 512 + lane * 8 + e) * 8);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float x = (float)dv[rr][c][e];
#pragma unroll
                    for (int s2 = 0; s2 < 7; ++s2) du4[rr][s2] += x * (float)w[s2];
                }
            }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int s2 = 0; s2 < 7; ++s2) du4[rr][s2] = gmt_wave_sum(du4[rr][s2]);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const long long row = (long long)(a.blk0 + blockIdx.x) * 16 + wave * 4 + rr;
        float du[7];
#pragma unroll
        for (int s2 = 0; s2 < 7; ++s2) du[s2] = du4[rr][s2];
        // gradient of the position before step t+1: through the dynamic features of step t+1, plus what flows on (RG.py:502-504, 585)
        float px = 0.f, py = 0.f;
        if (a.da0_next) {
            const __bf16* un = a.u_next + row * a.ldu;
            const float rx = (float)un[3] + (float)un[10], ry = (float)un[4] + (float)un[11], dist = (float)un[5] + (float)un[12];
            const float r2 = rx * rx + ry * ry;
            const float inv = r2 > 0.f ? 1.f / r2 : 0.f;
            const float drx = du[3] + du[5] * rx / dist - du[6] * ry * inv;
            const float dry = du[4] + du[5] * ry / dist + du[6] * rx * inv;
            px = -drx; py = -dry;
            if (!a.tf_next) { px += a.p_carry[row * 2]; py += a.p_carry[row * 2 + 1]; }
        }
        if (a.t < 0) {
            if (lane == 0) { a.d_in0[row * 3] = du[0]; a.d_in0[row * 3 + 1] = du[1]; a.d_in0[row * 3 + 2] = du[2]; }
            continue;
        }
        if (lane == 0) { a.p_carry[row * 2] = px; a.p_carry[row * 2 + 1] = py; }
        float g0 = a.d_out[row * 3], g1 = a.d_out[row * 3 + 1], g2 = a.d_out[row * 3 + 2];
        if (!a.tf_t) { g0 += du[0] + px; g1 += du[1] + py; g2 += du[2]; }          // cur_{t+1} = out_t, pos_{t+1} = pos_t + out_t[:2]
        const float o2 = a.oraw[row];
        const float do2 = g2 * (o2 > 20.f ? 1.f : 1.f / (1.f + expf(-o2)));        // softplus'
        if (lane == 0) { a.d_o[row * 3] = g0; a.d_o[row * 3 + 1] = g1; a.d_o[row * 3 + 2] = do2; }
        const int u0 = lane * 4;                                                    // HH = 256: 4 units a lane
        const bf16x4 yv = *reinterpret_cast<const bf16x4*>(a.y + row * HH + u0);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.wh2 + u0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(a.wh2 + HH + u0);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(a.wh2 + 2 * HH + u0);
        float dp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dy = g0 * w0[e] + g1 * w1[e] + do2 * w2[e];
            dp[e] = (float)yv[e] > 0.f ? dy : 0.2f * dy;
        }
        *reinterpret_cast<bf16x4*>(a.dypre + row * HH + u0) = pack4(dp[0], dp[1], dp[2], dp[3]);
    }
}

}  // namespace

extern "C" int gmt_bf16_gen_rollout_fwd(const gmt_bf16_gen_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && a->cond_pre && a->w0 && a->w1 && a->bias1 && a->wh1 && a->bh1 && a->wh2 && a->bh2 && a->ends && a->out_seq);
    GMT_CHECK_ARG(a->x0 && a->x1 && a->c0 && a->c1 && a->pos && a->T >= 1);
    GMT_CHECK_ARG(a->H == 512 && a->B > 0 && a->B % 256 == 0 && (!a->tf_mask || a->target));
    GMT_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f && (a->dropout_p == 0.f || (long long)a->T * a->B * a->H < 0xffffffffLL));
    const bool train = a->gates0 != nullptr;        // training: every step keeps its state (x0 / x1 / c0 / c1 have T+1 slots)
    GMT_CHECK_ARG(!train || (a->gates1 && a->y_all && a->oraw_all));
    const int H = a->H, B = a->B, K0 = H + 64, K1 = 2 * H;
    const long long ld0 = train ? GMT_BF16_GEN_X0_TRAIN_WIDTH : K0;
    const int n_mt = B / G_TQ, n_pt = 4 * H / TP;
    const bool head_small = B / 128 < 512;                     // 64-row head tiles unless 128-row ones fill the chip twice over
    auto head = head_small ? gen_head_kernel<1> : gen_head_kernel<2>;
    const int head_lds = G_NST * (256 + (head_small ? 64 : 128)) * G_BK * 2;
    if (hipFuncSetAttribute((const void*)gen_cell_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)head, hipFuncAttributeMaxDynamicSharedMemorySize, head_lds) != hipSuccess)
        return GMT_ERR_HIP;
    __bf16* x0 = (__bf16*)a->x0; __bf16* x1 = (__bf16*)a->x1;
    const int ng = row_groups(B, G_TQ), g_mt = n_mt / ng;
    const int rc = run_row_groups(ng, a->T, stream, [&](int g, int t, hipStream_t st) {
        const size_t cur = train ? t : (t & 1), nxt = train ? t + 1 : ((t & 1) ^ 1);
        __bf16* x0c = x0 + cur * B * ld0; __bf16* x0n = x0 + nxt * B * ld0;
        __bf16* x1c = x1 + cur * B * K1; __bf16* x1n = x1 + nxt * B * K1;
        // state slots: c[cur] = state before step t, c[nxt] = after (ping-pong when not training)
        __bf16* g0 = train ? (__bf16*)a->gates0 + (size_t)t * B * 4 * H : nullptr;
        __bf16* g1 = train ? (__bf16*)a->gates1 + (size_t)t * B * 4 * H : nullptr;
        // layer 0: reads x0[cur] = [h0_{t-1} | u_t]; h0_t -> x0[nxt][:, :H] (next step's recurrence) and x1[cur][:, :H]
        hipLaunchKernelGGL(gen_cell_kernel, dim3(g_mt * n_pt), dim3(G_CELL_THREADS), G_LDS, st, (const __bf16*)a->w0, x0c, ld0,
                           K0 / 32, (const __bf16*)a->cond_pre, (const float*)nullptr, a->c0 + cur * B * H,
                           a->c0 + nxt * B * H, x0n, ld0, x1c, (long long)K1, g0, H, g * g_mt, g_mt, n_pt, a->dropout_p, a->seed, (long long)t * B);
        // layer 1: reads x1[cur] = [h0_t | h1_{t-1}]; h1_t -> x1[nxt][:, H:]
        hipLaunchKernelGGL(gen_cell_kernel, dim3(g_mt * n_pt), dim3(G_CELL_THREADS), G_LDS, st, (const __bf16*)a->w1, x1c, (long long)K1,
                           K1 / 32, (const __bf16*)nullptr, a->bias1, a->c1 + cur * B * H, a->c1 + nxt * B * H,
                           x1n + H, (long long)K1, (__bf16*)nullptr, 0LL, g1, H, g * g_mt, g_mt, n_pt, 0.f, 0u, 0LL);
        head_args h;
        h.wh1 = (const __bf16*)a->wh1; h.bh1 = a->bh1; h.wh2 = a->wh2; h.bh2 = a->bh2;
        h.h = x1n + H; h.ldh = K1;
        h.ends = a->ends; h.target = a->target; h.tf_mask = a->tf_mask;
        h.out_seq = a->out_seq; h.pos = a->pos; h.tf_pos = a->pos + (size_t)B * 2;
        h.u_next = x0n + H; h.ldu = ld0;
        h.y_out = train ? (__bf16*)a->y_all + (size_t)t * B * (H / 2) : nullptr;
        h.oraw_out = train ? a->oraw_all + (size_t)t * B : nullptr;
        const int hb = head_small ? 2 * g_mt : g_mt;            // head tiles per group
        h.t = t; h.B = B; h.H = H; h.blk0 = g * hb;
        hipLaunchKernelGGL(head, dim3(hb), dim3(NTHREADS), head_lds, st, h);
    });
    if (rc != GMT_OK) return rc;
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_bf16_gen_rollout_bwd(const gmt_bf16_gen_bwd_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && a->gates0 && a->gates1 && a->c0 && a->c1 && a->x0 && a->y_all && a->oraw_all && a->d_out);
    GMT_CHECK_ARG(a->w1T && a->whh1T && a->wih1T && a->whh0T && a->w_u && a->wh2 && a->dypre_all && a->d_o && a->d_in0);
    GMT_CHECK_ARG(a->dc0 && a->dc1 && a->p_carry && a->tf_mask_host && a->H == 512 && a->B > 0 && a->B % 256 == 0 && a->T >= 1);
    const int H = a->H, B = a->B, T = a->T, HH = H / 2;
    const int n_pt = H / 128;
    const bool small = (B / 128) * n_pt < 512;                 // 64-row tiles unless 128-row ones fill the chip twice over
    const int n_mt = B / (small ? 64 : 128);
    auto cell = small ? gen_bwd_cell_kernel<1> : gen_bwd_cell_kernel<2>;
    if (hipFuncSetAttribute((const void*)cell, hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS) != hipSuccess)
        return GMT_ERR_HIP;
    const size_t ld0 = GMT_BF16_GEN_X0_TRAIN_WIDTH;
    __bf16* g0 = (__bf16*)a->gates0; __bf16* g1 = (__bf16*)a->gates1;
    const __bf16* x0 = (const __bf16*)a->x0;
    const int tile_rows = small ? 64 : 128;
    const int ng = row_groups(B, tile_rows), g_mt = n_mt / ng;
    const int rc = run_row_groups(ng, T + 1, stream, [&](int g, int step, hipStream_t st) {
        const int t = T - 1 - step;
        const bool has_next = t < T - 1;
        book_args b;
        b.da0_next = has_next ? g0 + (size_t)(t + 1) * B * 4 * H : nullptr;
        b.w_u = (const __bf16*)a->w_u;
        b.u_next = x0 + (size_t)(t + 1) * B * ld0 + H; b.ldu = (long long)ld0;
        b.d_out = t >= 0 ? a->d_out + (size_t)t * B * 3 : nullptr;
        b.y = t >= 0 ? (const __bf16*)a->y_all + (size_t)t * B * HH : nullptr;
        b.oraw = t >= 0 ? a->oraw_all + (size_t)t * B : nullptr;
        b.wh2 = a->wh2; b.p_carry = a->p_carry;
        b.dypre = t >= 0 ? (__bf16*)a->dypre_all + (size_t)t * B * HH : nullptr;
        b.d_o = t >= 0 ? a->d_o + (size_t)t * B * 3 : nullptr;
        b.d_in0 = a->d_in0;
        b.tf_t = t >= 0 ? a->tf_mask_host[t] : 0;
        b.tf_next = has_next ? a->tf_mask_host[t + 1] : 0;
        b.t = t; b.B = B; b.H = H; b.blk0 = g * (B / ng / 16);
        hipLaunchKernelGGL(gen_bwd_book_kernel, dim3(B / ng / 16), dim3(NTHREADS), 0, st, b);
        if (t < 0) return;
        // layer 1: dh1_t = d(ypre_t) W1 + da1_{t+1} W_hh_l1
        hipLaunchKernelGGL(cell, dim3(g_mt * n_pt), dim3(NTHREADS), GB_LDS, st,
                           (const __bf16*)a->w1T, (const __bf16*)b.dypre, (long long)HH, HH / BK,
                           (const __bf16*)a->whh1T, (const __bf16*)(g1 + (size_t)(has_next ? t + 1 : t) * B * 4 * H), 4LL * H,
                           has_next ? 4 * H / BK : 0, g1 + (size_t)t * B * 4 * H, a->c1 + (size_t)(t + 1) * B * H,
                           a->c1 + (size_t)t * B * H, a->dc1, (int)has_next, H, g * g_mt, g_mt, n_pt, 0.f, 0u, 0LL);
        // layer 0: dh0_t = da1_t W_ih_l1 + da0_{t+1} W_hh_l0
        hipLaunchKernelGGL(cell, dim3(g_mt * n_pt), dim3(NTHREADS), GB_LDS, st,
                           (const __bf16*)a->wih1T, (const __bf16*)(g1 + (size_t)t * B * 4 * H), 4LL * H, 4 * H / BK,
                           (const __bf16*)a->whh0T, (const __bf16*)(g0 + (size_t)(has_next ? t + 1 : t) * B * 4 * H), 4LL * H,
                           has_next ? 4 * H / BK : 0, g0 + (size_t)t * B * 4 * H, a->c0 + (size_t)(t + 1) * B * H,
                           a->c0 + (size_t)t * B * H, a->dc0, (int)has_next, H, g * g_mt, g_mt, n_pt, a->dropout_p, a->seed, (long long)t * B);
    });
    if (rc != GMT_OK) return rc;
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
