// Second-order ("double backward") sweep of the packed BiLSTM layer, needed by the WGAN-GP
// gradient penalty - compute_gradient_penalty, RecurrentGAN.py:790-842, where
// torch.autograd.grad(..., create_graph=True) differentiates THROUGH the LSTM's backward.
//
// The first backward sweep B (lstm_bwd_kernel) maps (dY, dh_final; i,f,g,o,c; W) -> dX.
// Given xhat = adjoint of dX this kernel computes the adjoints of B's inputs:
//   hhat_all  = adjoint of dY            (also the xhat of the layer above)
//   hhat_final= adjoint of dh_final
//   g_gates   = adjoint of the saved activated gates i,f,g,o
//   g_c       = adjoint of the saved cell states
// (the weight adjoints are GEMMs over the saved da / xhat / hhat buffers, done by the host).
// B is linear in dY, so its adjoint is a forward-in-time recurrence with exactly the shape
// of the forward layer:  ahat_t = xhat_t W_ih^T + hhat_{t-1} W_hh^T  (same fragment-major
// weights, W_hh resident in registers), followed by an element-wise second-derivative
// block.  Same workgroup / wave / MFMA decomposition as lstm_fwd_kernel (lstm_tile.h).
#include "lstm_tile.h"

namespace {

template <int NW, bool RES, int IX>
__global__ __launch_bounds__(NW * 64, 1)
void lstm_bwd2_kernel(const float* __restrict__ xhat /* IX = 0: [ndir][L][B][4H] = xhat W_ih^T, hoisted (gmt_lstm_xproj) */,
                      const float* __restrict__ w_ih_pk,
                      const float* __restrict__ w_hh_pk, const int* __restrict__ lengths,
                      const float* __restrict__ gates, const float* __restrict__ c_all,
                      const float* __restrict__ dct_all, const float* __restrict__ dht_all,
                      float* __restrict__ hhat_all, float* __restrict__ hhat_final,
                      float* __restrict__ g_gates, float* __restrict__ g_c,
                      int L, int B, int ndir, int n_tiles)
{
    constexpr int H = NW * 32;
    constexpr int KS = H / 2;
    constexpr int I = IX * H;                          // IX = 0: the input contraction was hoisted out of the loop
    constexpr int NT = NW * 64;
    extern __shared__ float smem[];
    float* hT = smem;                                  // [2][H][LDA]  hhat state
    float* xT = smem + 2 * H * GMT_LDA;                // [2][I][LDA]
    // at H = 128 the last KL k-steps of W_hh live in LDS, not registers (as in lstm_fwd_kernel)
    constexpr int KL = (RES && NW == 4) ? (IX == 2 ? 8 : IX == 1 ? 16 : 24) : 0;
    constexpr int KR = KS - KL;
    float4* wl = reinterpret_cast<float4*>(smem + (2 * H + 2 * I) * GMT_LDA) + (size_t)(threadIdx.x >> 6) * (KL / 4) * 4 * 64;
    __shared__ int s_len[GMT_TM];
    __shared__ int s_maxlen;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x / n_tiles;
    const int row0 = (blockIdx.x % n_tiles) * GMT_TM;
    const int hi = lane >> 5, lo = lane & 31;
    const int col = wave * 32 + lo;

    if (tid == 0) s_maxlen = 0;
    __syncthreads();
    if (tid < GMT_TM) {
        int ln = lengths ? lengths[row0 + tid] : L;
        ln = ln < 1 ? 1 : (ln > L ? L : ln);
        s_len[tid] = ln;
        atomicMax(&s_maxlen, ln);
    }
    for (int i = tid; i < 2 * H * GMT_LDA; i += NT) hT[i] = 0.0f;
    __syncthreads();
    const int tmax = s_maxlen;

    const float4* whh = reinterpret_cast<const float4*>(w_hh_pk) + ((size_t)dir * NW + wave) * (H / 8) * 4 * 64;
    const float4* wih = IX > 0 ? reinterpret_cast<const float4*>(w_ih_pk) + ((size_t)dir * NW + wave) * (I / 8) * 4 * 64 : nullptr;
    float wr[4][RES ? KR : 1];
    if constexpr (RES) {
        gmt_load_resident<4, KR>(wr, whh, lane);
        for (int i = lane; i < (KL / 4) * 4 * 64; i += 64) wl[i] = whh[(KR / 4) * 4 * 64 + i];   // visible after the next barrier
    }

    float chat[16];                                    // adjoint of the incoming cell adjoint (carried)
    float gc_pend[16];                                 // g_c of the step before, still missing this step's term
#pragma unroll
    for (int r = 0; r < 16; ++r) { chat[r] = 0.f; gc_pend[r] = 0.f; }

    constexpr int XQ = IX > 0 ? I / 4 : 1;
    constexpr int XS = IX > 0 ? (GMT_TM * XQ) / NT : 1;
    float4 xs[XS];
    auto x_load = [&](int t) {
        if constexpr (IX > 0) {
            const float4* src = reinterpret_cast<const float4*>(xhat + ((size_t)t * B + row0) * I);
#pragma unroll
            for (int j = 0; j < XS; ++j) xs[j] = src[tid + j * NT];
        }
    };
    auto x_store = [&](int buf) {
        if constexpr (IX > 0) {
#pragma unroll
            for (int j = 0; j < XS; ++j) {
                const int it = tid + j * NT;
                const int rr = it / XQ, c4 = it % XQ;
                float* d = xT + ((size_t)buf * I + 4 * c4) * GMT_LDA + rr;
                d[0] = xs[j].x; d[GMT_LDA] = xs[j].y; d[2 * GMT_LDA] = xs[j].z; d[3 * GMT_LDA] = xs[j].w;
            }
        }
    };

    int cur = 0;
    if (tmax > 0) { x_load(dir ? tmax - 1 : 0); x_store(0); }
    __syncthreads();

    const size_t lane_row = (size_t)row0 + 4 * hi;
    for (int step = 0; step < tmax; ++step) {
        const int t = dir ? (tmax - 1 - step) : step;
        const int tp = dir ? t + 1 : t - 1;            // step visited just before t by this sweep
        const size_t rs = ((size_t)dir * L + t) * B + lane_row;
        const size_t rsp = ((size_t)dir * L + (tp < 0 ? 0 : (tp >= L ? L - 1 : tp))) * B + lane_row;
        f32x16 acc[4];
        if (step + 1 < tmax) x_load(dir ? t - 1 : t + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        if constexpr (IX > 0) gmt_mma_stream<4>(acc, xT + (size_t)cur * I * GMT_LDA, wih, I, lane);
        if constexpr (RES) gmt_mma_resident<4, KS, KR>(acc, hT + (size_t)cur * H * GMT_LDA, wr, lane, wl);
        else gmt_mma_stream<4>(acc, hT + (size_t)cur * H * GMT_LDA, whh, H, lane);

        const float* ho = hT + (size_t)cur * H * GMT_LDA + col * GMT_LDA + 4 * hi;
        float* hn = hT + (size_t)(cur ^ 1) * H * GMT_LDA + col * GMT_LDA + 4 * hi;
        const float* gp = gates + rs * (4 * H) + 4 * col;
        const float* pre_p = xhat + rs * (4 * H) + 4 * col;       // IX = 0 only
        const float* cn_p = c_all + rs * H + col;
        const float* cp_p = c_all + rsp * H + col;
        const float* dct_p = dct_all + rs * H + col;
        const float* dht_p = dht_all + rs * H + col;
        float* gg_p = g_gates + rs * (4 * H) + 4 * col;
        float* gcp_p = g_c + rsp * H + col;
        float* hh_p = hhat_all ? hhat_all + ((size_t)t * B + lane_row) * (ndir * H) + dir * H + col : nullptr;
        const bool tp_in = (tp >= 0) && (tp < L);
        constexpr int RB = (NW >= 4) ? 4 : 8;          // rows per batch (register budget at H >= 128)
        // RB rows at a time: all their operands are requested before the first is used (one HBM round trip
        // per batch instead of one per row), and the row math is selects, not branches.  g_c of a step also
        // receives a term from the step after it (c_{t} enters f-gate adjoint of t+1): it is kept in
        // registers for one step (gc_pend) and stored then, instead of a read-modify-write.
#pragma unroll
        for (int half = 0; half < 16 / RB; ++half) {
            float4 q4[RB], p4[IX == 0 ? RB : 1];
            float v_cn[RB], v_cp[RB], v_dct[RB], v_dht[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = half * RB + j, mr = (r & 3) + 8 * (r >> 2);
                q4[j] = gmt_ld4(gp + (size_t)mr * (4 * H));
                if constexpr (IX == 0) p4[j] = gmt_ld4(pre_p + (size_t)mr * (4 * H));
                v_cn[j] = cn_p[(size_t)mr * H];
                v_cp[j] = cp_p[(size_t)mr * H];
                v_dct[j] = dct_p[(size_t)mr * H];
                v_dht[j] = dht_p[(size_t)mr * H];
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = half * RB + j, mr = (r & 3) + 8 * (r >> 2);
                const int ln = s_len[mr + 4 * hi];
                const bool act = t < ln;
                const bool has_prev = tp_in && (dir ? (tp < ln) : true);
                const float ig = q4[j].x, fg = q4[j].y, gg = q4[j].z, og = q4[j].w;
                const float cn = v_cn[j];
                const float cp = has_prev ? v_cp[j] : 0.f;
                const float dct = v_dct[j], dht = v_dht[j];
                const float tc = gmt_tanh(cn);
                const float omt = 1.f - tc * tc;
                // intermediates of the first backward sweep
                const float di = dct * gg, dg = dct * ig, df = dct * cp, dov = dht * tc;
                float a_i = acc[0][r], a_f = acc[1][r], a_g = acc[2][r], a_o = acc[3][r];
                if constexpr (IX == 0) { a_i += p4[j].x; a_f += p4[j].y; a_g += p4[j].z; a_o += p4[j].w; }
                const float adi = a_i * ig * (1.f - ig);
                const float adf = a_f * fg * (1.f - fg);
                const float adg = a_g * (1.f - gg * gg);
                const float ado = a_o * og * (1.f - og);
                const float cprev_hat = chat[r];
                const float adct = adi * gg + adg * ig + adf * cp + cprev_hat * fg;
                const float o_gi = a_i * di * (1.f - 2.f * ig) + adg * dct;
                const float o_gf = a_f * df * (1.f - 2.f * fg) + cprev_hat * dct;
                const float o_gg = a_g * dg * (-2.f * gg) + adi * dct;
                const float o_go = a_o * dov * (1.f - 2.f * og) + adct * dht * omt;
                const float atc = adct * dht * og * (-2.f * tc) + ado * dht;
                const float o_gc = atc * omt;
                const float hnew = adct * og * omt + ado * tc;
                chat[r] = act ? adct : chat[r];
                hn[mr] = act ? hnew : ho[mr];
                gmt_st4(gg_p + (size_t)mr * (4 * H), act ? o_gi : 0.f, act ? o_gf : 0.f, act ? o_gg : 0.f, act ? o_go : 0.f);
                // the step before this one (if the sweep had one) now has its complete g_c
                if (step > 0) gcp_p[(size_t)mr * H] = gc_pend[r] + ((act && has_prev) ? adf * dct : 0.f);
                gc_pend[r] = act ? o_gc : 0.f;
                if (hh_p) hh_p[(size_t)mr * (ndir * H)] = act ? hnew : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);         // keep the next batch's loads from being hoisted up here
        }
        if (step + 1 == tmax) {                        // the last step of the sweep has no successor
            float* gc_p = g_c + rs * H + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) gc_p[(size_t)((r & 3) + 8 * (r >> 2)) * H] = gc_pend[r];
        }
        if (step + 1 < tmax) x_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // padded tail of the tile
    for (int t = tmax; t < L; ++t) {
        const size_t rs = ((size_t)dir * L + t) * B + lane_row;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2);
            gmt_st4(g_gates + (rs + mr) * (4 * H) + 4 * col, 0.f, 0.f, 0.f, 0.f);
            g_c[(rs + mr) * H + col] = 0.f;
            if (hhat_all) hhat_all[((size_t)t * B + lane_row + mr) * (ndir * H) + dir * H + col] = 0.f;
        }
    }
    if (hhat_final) {
        const float* hl = hT + (size_t)cur * H * GMT_LDA + col * GMT_LDA + 4 * hi;
        float* hf = hhat_final + lane_row * (ndir * H) + dir * H + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2);
            hf[(size_t)mr * (ndir * H)] = hl[mr];
        }
    }
}

template <int NW>
int launch_bwd2(const float* xhat, const float* w_ih_pk, const float* w_hh_pk, const int* lengths, const float* gates,
                const float* c_all, const float* dct_all, const float* dht_all, float* hhat_all, float* hhat_final,
                float* g_gates, float* g_c, int L, int B, int I, int ndir, hipStream_t stream)
{
    constexpr int H = NW * 32;
    constexpr bool RES = (NW <= 4);
    const int n_tiles = B / GMT_TM;
    const dim3 grid(n_tiles * ndir), block(NW * 64);
#define GMT_GO(IXV)                                                                                               \
    {                                                                                                             \
        const size_t lds = (size_t)(2 * H + 2 * IXV * H) * GMT_LDA * sizeof(float)                                \
                           + ((RES && NW == 4) ? (size_t)NW * (IXV == 2 ? 8 : IXV == 1 ? 16 : 24) * 4 * 64 * sizeof(float) : 0); \
        if (lds > 160 * 1024 - 256) return GMT_ERR_UNSUPPORTED;                                                   \
        auto k = lstm_bwd2_kernel<NW, RES, IXV>;                                                                  \
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return GMT_ERR_HIP;                                                                                   \
        hipLaunchKernelGGL(k, grid, block, lds, stream, xhat, w_ih_pk, w_hh_pk, lengths, gates, c_all, dct_all,   \
                           dht_all, hhat_all, hhat_final, g_gates, g_c, L, B, ndir, n_tiles);                     \
    }
    if (!w_ih_pk) GMT_GO(0)
    else if (I == H) GMT_GO(1)
    else if (I == 2 * H) GMT_GO(2)
    else return GMT_ERR_UNSUPPORTED;
#undef GMT_GO
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

}  // namespace

extern "C" int gmt_lstm_layer_bwd2(const float* xhat, const float* w_ih_pk, const float* w_hh_pk, const int* lengths,
                                   const float* gates, const float* c_all, const float* dct_all, const float* dht_all,
                                   float* hhat_all, float* hhat_final, float* g_gates, float* g_c,
                                   int L, int B, int I, int H, int ndir, hipStream_t stream)
{
    GMT_CHECK_ARG(xhat && w_hh_pk && gates && c_all && dct_all && dht_all && g_gates && g_c);
    GMT_CHECK_ARG(L >= 1 && B >= GMT_TM && (B % GMT_TM) == 0 && (ndir == 1 || ndir == 2));
    switch (H) {
        case 32: return launch_bwd2<1>(xhat, w_ih_pk, w_hh_pk, lengths, gates, c_all, dct_all, dht_all, hhat_all,
                                       hhat_final, g_gates, g_c, L, B, I, ndir, stream);
        case 64: return launch_bwd2<2>(xhat, w_ih_pk, w_hh_pk, lengths, gates, c_all, dct_all, dht_all, hhat_all,
                                       hhat_final, g_gates, g_c, L, B, I, ndir, stream);
        case 128: return launch_bwd2<4>(xhat, w_ih_pk, w_hh_pk, lengths, gates, c_all, dct_all, dht_all, hhat_all,
                                        hhat_final, g_gates, g_c, L, B, I, ndir, stream);
        case 256: return launch_bwd2<8>(xhat, w_ih_pk, w_hh_pk, lengths, gates, c_all, dct_all, dht_all, hhat_all,
                                        hhat_final, g_gates, g_c, L, B, I, ndir, stream);
        default: return GMT_ERR_UNSUPPORTED;
    }
}
