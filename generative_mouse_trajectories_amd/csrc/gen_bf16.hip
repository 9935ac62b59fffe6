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

namespace {

using namespace gmt_bf16;

constexpr int G_TQ = 128;                                      // rows per workgroup (two workgroups per CU)
constexpr int G_PRE_OFF = 0, G_C_OFF = 64 * 512, G_H_OFF = G_C_OFF + 64 * 256, G_LDS = G_H_OFF + 64 * 128;   // 56 KiB

// One LSTM layer step over all rows.  X: [B][ldx] bf16 (K = 32 nk32 columns used); W: [4H slab order][32 nk32] bf16.
// pre: [B][4H] bf16 plane (slab order) or NULL (then bias[4H]); c_prev -> c_out [B][H] fp32; h -> h0 (and h1 if given).
__global__ __launch_bounds__(NTHREADS, 2)
void gen_cell_kernel(const __bf16* __restrict__ W, const __bf16* __restrict__ X, long long ldx, int nk32,
                     const __bf16* __restrict__ pre, const float* __restrict__ bias,
                     const float* __restrict__ c_prev, float* __restrict__ c_out,
                     __bf16* __restrict__ h0, long long ldh0, __bf16* __restrict__ h1, long long ldh1,
                     int H, int n_mt, int n_pt)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int d, pt, mt;
    patch_of_block(blockIdx.x, gridDim.x, n_mt, n_pt, d, pt, mt);
    const long long row0 = (long long)mt * G_TQ;
    const long long K = 32LL * nk32;

    f32x16 acc[4][2];
    zero_acc<4, 2>(acc);
    mainloop32<4, 2>(W + (long long)pt * TP * K, K, X + row0 * ldx, ldx, nk32, lds, acc, wave, lane);

    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
    using GI = RowImage<512>;
    using CI = RowImage<256>;
    using HI = RowImage<128>;
    const int rl = wq * 32 + lo;
    const char* Pt = pre ? (const char*)(pre + row0 * 4 * H + pt * TP) : nullptr;
    const char* Cp = (const char*)(c_prev + row0 * H + pt * (TP / 4));
    char* Co = (char*)(c_out + row0 * H + pt * (TP / 4));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        auto rowmap = [j](int r) { return (r >> 5) * 64 + j * 32 + (r & 31); };
        __syncthreads();
        if (Pt) GI::dma_in(Pt, 4LL * H * 2, lds + G_PRE_OFF, 64, rowmap, wave, lane);
        CI::dma_in(Cp, (long long)H * 4, lds + G_C_OFF, 64, rowmap, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int ucol = wp * 32 + 8 * g4 + 4 * hi;
            float a[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (Pt) {
                    const bf16x4 p = *reinterpret_cast<const bf16x4*>(lds + G_PRE_OFF + GI::addr(rl, (wp * 128 + g * 32 + 8 * g4 + 4 * hi) * 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[g][e] = acc[g][j][4 * g4 + e] + (float)p[e];
                } else {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + pt * TP + wp * 128 + g * 32 + 8 * g4 + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[g][e] = acc[g][j][4 * g4 + e] + b[e];
                }
            }
            const f32x4 cp = *reinterpret_cast<const f32x4*>(lds + G_C_OFF + CI::addr(rl, ucol * 4));
            f32x4 cn;
            float hn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cn[e] = gmt_sigmoid(a[1][e]) * cp[e] + gmt_sigmoid(a[0][e]) * gmt_tanh(a[2][e]);
                hn[e] = gmt_sigmoid(a[3][e]) * gmt_tanh(cn[e]);
            }
            *reinterpret_cast<f32x4*>(lds + G_C_OFF + CI::addr(rl, ucol * 4)) = cn;
            *reinterpret_cast<bf16x4*>(lds + G_H_OFF + HI::addr(rl, ucol * 2)) = pack4(hn[0], hn[1], hn[2], hn[3]);
        }
        __syncthreads();
        CI::copy_out(Co, (long long)H * 4, lds + G_C_OFF, 64, rowmap, wave, lane);
        HI::copy_out((char*)(h0 + row0 * ldh0 + pt * (TP / 4)), ldh0 * 2, lds + G_H_OFF, 64, rowmap, wave, lane);
        if (h1) HI::copy_out((char*)(h1 + row0 * ldh1 + pt * (TP / 4)), ldh1 * 2, lds + G_H_OFF, 64, rowmap, wave, lane);
    }
}

// Head + step bookkeeping.  HH = H/2 must be 256 (one weight tile): the three outputs of a row are sums over all HH units.
struct head_args {
    const __bf16* wh1; const float* bh1; const float* wh2; const float* bh2;
    const __bf16* h; long long ldh;                  // h1_t rows
    const float* ends; const float* target; const unsigned char* tf_mask;
    float* out_seq; float* pos; float* tf_pos;
    __bf16* u_next; long long ldu;                   // &x0_next[0][H]: 14 slots written per row
    int t, B, H;
};

__global__ __launch_bounds__(NTHREADS, 2)
void gen_head_kernel(head_args a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long row0 = (long long)blockIdx.x * G_TQ;
    const int HH = a.H / 2;

    f32x16 acc[4][2];
    zero_acc<4, 2>(acc);
    mainloop32<4, 2>(a.wh1, a.H, a.h + row0 * a.ldh, a.ldh, a.H / 32, lds, acc, wave, lane);
    __syncthreads();
    float* w2s = reinterpret_cast<float*>(lds);               // [3][HH]
    float* red = w2s + 3 * 256;                               // [2][128][4]
    for (int i = threadIdx.x; i < 3 * HH; i += NTHREADS) w2s[i] = a.wh2[i];
    __syncthreads();
    const int wp = wave >> 1, wq = wave & 1, lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int u = wp * 128 + i * 32 + 8 * g4 + 4 * hi;
                const f32x4 b = *reinterpret_cast<const f32x4*>(a.bh1 + u);
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[i][j][4 * g4 + e] + b[e];
                    y[e] = v > 0.f ? v : 0.2f * v;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(w2s + k * HH + u);
                    o[k] += y[0] * w[0] + y[1] * w[1] + y[2] * w[2] + y[3] * w[3];
                }
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] += __shfl_xor(o[k], 32, 64);
        if (hi == 0) {
            float* r = red + ((wp * 128) + wq * 64 + j * 32 + lo) * 4;
            r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
        }
    }
    __syncthreads();
    if (threadIdx.x < G_TQ) {
        const int rr = threadIdx.x;
        const long long row = row0 + rr;
        const float o0 = red[rr * 4 + 0] + red[(128 + rr) * 4 + 0] + a.bh2[0];
        const float o1 = red[rr * 4 + 1] + red[(128 + rr) * 4 + 1] + a.bh2[1];
        const float o2 = red[rr * 4 + 2] + red[(128 + rr) * 4 + 2] + a.bh2[2];
        const float dt = (o2 > 20.f ? o2 : log1pf(expf(o2))) + 0.001f;                 // F.softplus, RG.py:577-580
        float* os = a.out_seq + ((long long)a.t * a.B + row) * 3;
        os[0] = o0; os[1] = o1; os[2] = dt;
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
}

}  // namespace

extern "C" int gmt_bf16_gen_rollout_fwd(const gmt_bf16_gen_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && a->cond_pre && a->w0 && a->w1 && a->bias1 && a->wh1 && a->bh1 && a->wh2 && a->bh2 && a->ends && a->out_seq);
    GMT_CHECK_ARG(a->x0 && a->x1 && a->c0 && a->c1 && a->pos && a->T >= 1);
    GMT_CHECK_ARG(a->H == 512 && a->B > 0 && a->B % 256 == 0 && (!a->tf_mask || a->target));
    const int H = a->H, B = a->B, K0 = H + 64, K1 = 2 * H;
    const int n_mt = B / G_TQ, n_pt = 4 * H / TP;
    if (hipFuncSetAttribute((const void*)gen_cell_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)gen_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS) != hipSuccess)
        return GMT_ERR_HIP;
    __bf16* x0 = (__bf16*)a->x0; __bf16* x1 = (__bf16*)a->x1;
    for (int t = 0; t < a->T; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        __bf16* x0c = x0 + (size_t)cur * B * K0; __bf16* x0n = x0 + (size_t)nxt * B * K0;
        __bf16* x1c = x1 + (size_t)cur * B * K1; __bf16* x1n = x1 + (size_t)nxt * B * K1;
        // layer 0: reads x0[cur] = [h0_{t-1} | u_t]; h0_t -> x0[nxt][:, :H] (next step's recurrence) and x1[cur][:, :H]
        hipLaunchKernelGGL(gen_cell_kernel, dim3(n_mt * n_pt), dim3(NTHREADS), G_LDS, stream, (const __bf16*)a->w0, x0c, (long long)K0,
                           K0 / 32, (const __bf16*)a->cond_pre, (const float*)nullptr, a->c0 + (size_t)nxt * B * H,
                           a->c0 + (size_t)cur * B * H, x0n, (long long)K0, x1c, (long long)K1, H, n_mt, n_pt);
        // layer 1: reads x1[cur] = [h0_t | h1_{t-1}]; h1_t -> x1[nxt][:, H:]
        hipLaunchKernelGGL(gen_cell_kernel, dim3(n_mt * n_pt), dim3(NTHREADS), G_LDS, stream, (const __bf16*)a->w1, x1c, (long long)K1,
                           K1 / 32, (const __bf16*)nullptr, a->bias1, a->c1 + (size_t)nxt * B * H, a->c1 + (size_t)cur * B * H,
                           x1n + H, (long long)K1, (__bf16*)nullptr, 0LL, H, n_mt, n_pt);
        head_args h;
        h.wh1 = (const __bf16*)a->wh1; h.bh1 = a->bh1; h.wh2 = a->wh2; h.bh2 = a->bh2;
        h.h = x1n + H; h.ldh = K1;
        h.ends = a->ends; h.target = a->target; h.tf_mask = a->tf_mask;
        h.out_seq = a->out_seq; h.pos = a->pos; h.tf_pos = a->pos + (size_t)B * 2;
        h.u_next = x0n + H; h.ldu = K0;
        h.t = t; h.B = B; h.H = H;
        hipLaunchKernelGGL(gen_head_kernel, dim3(B / G_TQ), dim3(NTHREADS), G_LDS, stream, h);
    }
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
