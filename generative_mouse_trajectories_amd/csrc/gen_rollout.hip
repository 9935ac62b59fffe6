// Persistent autoregressive generator rollout - Generator.forward, RecurrentGAN.py:508-604:
// the whole T-step loop (dynamic features -> 2-layer LSTM cell -> output head ->
// softplus -> position update -> teacher-forcing select) is ONE launch, forward, and one
// launch, backward (BPTT incl. the position / next-input feedback paths).
//
// Workgroup = tile of 32 trajectories for all T steps (lstm_tile.h decomposition).
//   * the time-invariant part of the layer-0 input contraction, condition . W_ih[:, 3:3+H]^T
//     + b_ih + b_hh, is hoisted out of the loop by the caller (g0, one GEMM per sequence);
//   * the 7 per-step inputs (current (dx,dy,dt) + 4 dynamic features, RG.py:486-506) are a
//     K = 8 MFMA contraction;
//   * layer-0 W_hh stays resident in registers when H <= 128; layer-1 W_ih / W_hh and the
//     head stream from L2 in fragment-major order;
//   * per-row scalars (position, current input, teacher-forcing position) live in the
//     registers of threads 0..31.
// Teacher forcing is a per-step host decision for the whole batch (RG.py:588): tf_mask[t].
#include "lstm_tile.h"

namespace {

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus
__device__ __forceinline__ float lrelu_f(float x) { return x > 0.f ? x : 0.2f * x; }

// TRAIN: every saved-activation buffer is present (the host entry point checks all-or-none): no
// per-row pointer tests.  Otherwise a no-grad rollout that stores only out_seq.
template <int NW, bool RES0, bool TRAIN>
__global__ __launch_bounds__(NW * 64, 1)
void gen_fwd_kernel(gmt_gen_fwd_args a)
{
    constexpr int H = NW * 32, KS = H / 2, HH = H / 2, NH = (HH + 31) / 32, NT = NW * 64;
    extern __shared__ float smem[];
    float* h0T = smem;                                 // [2][H][LDA]
    // H = 256 does not fit a double-buffered h1 next to the dropout buffer in 160 KB: it keeps ONE h1
    // buffer and pays one more barrier per step (between layer 1's reads and its writes)
    constexpr bool H1S = (NW > 4);
    constexpr int H1B = H1S ? 1 : 2;
    float* h1T = h0T + 2 * H * GMT_LDA;                // [H1B][H][LDA]
    float* yT = h1T + H1B * H * GMT_LDA;               // [NH*32][LDA]
    float* uT = yT + NH * 32 * GMT_LDA;                // [8][LDA]
    float* h0dT = uT + 8 * GMT_LDA;                    // [H][LDA] layer-0 output after dropout (dropout_p > 0)
    float* w2S = h0dT + H * GMT_LDA;                   // [3][HH] head weights
    float* oP = w2S + 3 * HH;                          // [NT/32][3][32] partial head outputs
    // at H = 128 the last KL k-steps of W_hh_l0 live in LDS, not registers (the serial tail spills otherwise)
    constexpr int KL = (RES0 && NW == 4) ? 16 : 0;
    constexpr int KR = KS - KL;
    float4* wl = reinterpret_cast<float4*>(oP + (NT / 32) * 96) + (size_t)(threadIdx.x >> 6) * (KL / 4) * 4 * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * GMT_TM;
    const int hi = lane >> 5, lo = lane & 31;
    const int col = wave * 32 + lo;
    const int T = a.T, B = a.B;

    const float4* wu = reinterpret_cast<const float4*>(a.w_u_pk) + (size_t)wave * 1 * 4 * 64;
    const float4* whh0 = reinterpret_cast<const float4*>(a.w_hh0_pk) + (size_t)wave * (H / 8) * 4 * 64;
    const float4* wih1 = reinterpret_cast<const float4*>(a.w_ih1_pk) + (size_t)wave * (H / 8) * 4 * 64;
    const float4* whh1 = reinterpret_cast<const float4*>(a.w_hh1_pk) + (size_t)wave * (H / 8) * 4 * 64;
    const float4* w1 = reinterpret_cast<const float4*>(a.w1_pk) + (size_t)wave * (H / 8) * 64;

    float wr0[4][RES0 ? KR : 1];
    if constexpr (RES0) {
        gmt_load_resident<4, KR>(wr0, whh0, lane);
        for (int i = lane; i < (KL / 4) * 4 * 64; i += 64) wl[i] = whh0[(KR / 4) * 4 * 64 + i];   // visible after the first barrier
    }

    for (int i = tid; i < (2 + H1B) * H * GMT_LDA + NH * 32 * GMT_LDA + 8 * GMT_LDA; i += NT) smem[i] = 0.f;
    for (int i = tid; i < 3 * HH; i += NT) w2S[i] = a.w2[i];
    const bool drop = a.dropout_p > 0.f;
    const uint32_t thresh24 = (uint32_t)(a.dropout_p * 16777216.0f);
    const float keep_scale = 1.0f / (1.0f - a.dropout_p);

    float bz1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bz1[g] = a.bias1[g * H + col];
    const float b1v = (col < HH) ? a.b1[col] : 0.f;

    float c0[16], c1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }

    // per-row scalar state (threads 0..31 own row tid)
    float px = 0.f, py = 0.f, ex = 0.f, ey = 0.f, tpx = 0.f, tpy = 0.f, in0 = 0.f, in1 = 0.f, in2 = 0.f;
    if (tid < GMT_TM) {
        const size_t row = (size_t)row0 + tid;
        px = a.starts[row * 2]; py = a.starts[row * 2 + 1];
        ex = a.ends[row * 2]; ey = a.ends[row * 2 + 1];
        tpx = px; tpy = py;
        in0 = a.init_in[0]; in1 = a.init_in[1]; in2 = a.init_in[2];
    }
    // sampling with early stop (Generator.generate, RG.py:671-686): a row is done at the first step whose new
    // position is within stop_distance of its end point; the tile leaves the loop once all its rows are done
    __shared__ int s_stop;
    const bool stopping = !TRAIN && a.stop_distance > 0.f;
    int first_hit = 0, t_exit = a.T;
    if (tid == 0) s_stop = 0;
    const size_t lane_row = (size_t)row0 + 4 * hi;
    int cur = 0;
    f32x16 acc0[4];
    auto g0_load = [&]() {
        const float* gp = gmt_opaque_v(a.g0 + lane_row * (4 * H) + 4 * col);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 v = gmt_ld4(gp + (size_t)((r & 3) + 8 * (r >> 2)) * (4 * H));
            acc0[0][r] = v.x; acc0[1][r] = v.y; acc0[2][r] = v.z; acc0[3][r] = v.w;
        }
    };
    g0_load();
    __syncthreads();

    GMT_PH_DECL;
    for (int t = 0; t < T; ++t) {
        const size_t rs = (size_t)t * B + lane_row;
        GMT_PH(7);
        // ---- S1: dynamic features of the current position (RG.py:486-506) + current input
        if (tid < GMT_TM) {
            const float rx = ex - px, ry = ey - py;
            const float dist = sqrtf(rx * rx + ry * ry + 1e-8f);
            const float ang = atan2f(ry, rx);
            uT[0 * GMT_LDA + tid] = in0; uT[1 * GMT_LDA + tid] = in1; uT[2 * GMT_LDA + tid] = in2;
            uT[3 * GMT_LDA + tid] = rx; uT[4 * GMT_LDA + tid] = ry; uT[5 * GMT_LDA + tid] = dist;
            uT[6 * GMT_LDA + tid] = ang; uT[7 * GMT_LDA + tid] = 0.f;
            if (TRAIN) {
                float4* up = reinterpret_cast<float4*>(a.u_all + ((size_t)t * B + row0 + tid) * 8);
                up[0] = make_float4(in0, in1, in2, rx);
                up[1] = make_float4(ry, dist, ang, 0.f);
            }
        }
        GMT_PH(0);
        __syncthreads();
        GMT_PH(1);
        if constexpr (!TRAIN) {
            if (stopping && s_stop) { t_exit = t; break; }          // uniform: written before the barrier above
        }
        // ---- S2: layer 0   (acc0 = the time-invariant g0 tile, requested during the previous step's serial tail)
        {
            f32x16 (&acc)[4] = acc0;
            gmt_mma_stream<4>(acc, uT, wu, 8, lane);
            if constexpr (RES0) gmt_mma_resident<4, KS, KR>(acc, h0T + (size_t)cur * H * GMT_LDA, wr0, lane, wl);
            else gmt_mma_stream<4>(acc, h0T + (size_t)cur * H * GMT_LDA, whh0, H, lane);
            float* hn = h0T + (size_t)(cur ^ 1) * H * GMT_LDA + col * GMT_LDA + 4 * hi;
            float* sg = TRAIN ? a.gates0 + rs * (4 * H) + 4 * col : nullptr;
            float* sc = TRAIN ? a.c0_all + rs * H + col : nullptr;
            float* sh = TRAIN ? a.h0_all + rs * H + col : nullptr;
            float* shd = TRAIN ? a.h0d_all + rs * H + col : nullptr;
            float* hd = h0dT + col * GMT_LDA + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (r & 3) + 8 * (r >> 2);
                const float ig = gmt_sigmoid(acc[0][r]), fg = gmt_sigmoid(acc[1][r]);
                const float gg = gmt_tanh(acc[2][r]), og = gmt_sigmoid(acc[3][r]);
                const float cn = fg * c0[r] + ig * gg;
                const float hv = og * gmt_tanh(cn);
                c0[r] = cn;
                hn[mr] = hv;
                if (drop) {
                    const uint32_t idx = (uint32_t)((rs + mr) * H + col);
                    const float hdv = gmt_keep(idx, a.seed, thresh24) ? hv * keep_scale : 0.f;
                    hd[mr] = hdv;
                    if (TRAIN) shd[(size_t)mr * H] = hdv;
                }
                if (TRAIN) {
                    gmt_st4(sg + (size_t)mr * (4 * H), ig, fg, gg, og);
                    sc[(size_t)mr * H] = cn;
                    sh[(size_t)mr * H] = hv;
                }
            }
        }
        GMT_PH(2);
        __syncthreads();
        // ---- S3: layer 1
        {
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = bz1[g];
            gmt_mma_stream<4>(acc, drop ? h0dT : h0T + (size_t)(cur ^ 1) * H * GMT_LDA, wih1, H, lane);
            gmt_mma_stream<4>(acc, h1T + (size_t)(H1S ? 0 : cur) * H * GMT_LDA, whh1, H, lane);
            if constexpr (H1S) __syncthreads();
            float* hn = h1T + (size_t)(H1S ? 0 : (cur ^ 1)) * H * GMT_LDA + col * GMT_LDA + 4 * hi;
            float* sg = TRAIN ? a.gates1 + rs * (4 * H) + 4 * col : nullptr;
            float* sc = TRAIN ? a.c1_all + rs * H + col : nullptr;
            float* sh = TRAIN ? a.h1_all + rs * H + col : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (r & 3) + 8 * (r >> 2);
                const float ig = gmt_sigmoid(acc[0][r]), fg = gmt_sigmoid(acc[1][r]);
                const float gg = gmt_tanh(acc[2][r]), og = gmt_sigmoid(acc[3][r]);
                const float cn = fg * c1[r] + ig * gg;
                const float hv = og * gmt_tanh(cn);
                c1[r] = cn;
                hn[mr] = hv;
                if (TRAIN) {
                    gmt_st4(sg + (size_t)mr * (4 * H), ig, fg, gg, og);
                    sc[(size_t)mr * H] = cn;
                    sh[(size_t)mr * H] = hv;
                }
            }
        }
        GMT_PH(3);
        __syncthreads();
        // ---- S4: head layer 1, Linear(H -> H/2) + LeakyReLU (RG.py:460-462)
        if (wave < NH) {
            f32x16 acc[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = b1v;
            gmt_mma_stream<1>(acc, h1T + (size_t)(H1S ? 0 : (cur ^ 1)) * H * GMT_LDA, w1, H, lane);
            float* yp = yT + col * GMT_LDA + 4 * hi;
            float* sy = (TRAIN && col < HH) ? a.y_all + rs * HH + col : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (r & 3) + 8 * (r >> 2);
                const float y = lrelu_f(acc[0][r]);
                yp[mr] = y;
                if (sy) sy[(size_t)mr * HH] = y;
            }
        }
        GMT_PH(4);
        __syncthreads();
        GMT_PH(5);
        // ---- S5: Linear(H/2 -> 3): every thread sums a slice of k for one row, threads 0..31 finish;
        //          softplus on dt, position / next-input update (RG.py:574-595)
        g0_load();                                     // next step's layer-0 accumulator init, in flight under the serial tail
        {
            constexpr int NP = NT / 32, KP = (HH + NP - 1) / NP;
            const int part = tid >> 5, row = tid & 31;
            float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const int k = part * KP + j;
                if (k < HH) {
                    const float y = yT[k * GMT_LDA + row];
                    p0 = fmaf(y, w2S[k], p0); p1 = fmaf(y, w2S[HH + k], p1); p2 = fmaf(y, w2S[2 * HH + k], p2);
                }
            }
            oP[(part * 3 + 0) * 32 + row] = p0; oP[(part * 3 + 1) * 32 + row] = p1; oP[(part * 3 + 2) * 32 + row] = p2;
        }
        __syncthreads();
        if (tid < GMT_TM) {
            constexpr int NP = NT / 32;
            const float* b2 = gmt_opaque_s(a.b2);
            float o0 = b2[0], o1 = b2[1], o2 = b2[2];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                o0 += oP[(p * 3 + 0) * 32 + tid]; o1 += oP[(p * 3 + 1) * 32 + tid]; o2 += oP[(p * 3 + 2) * 32 + tid];
            }
            const float dt = softplus_f(o2) + 0.001f;
            const size_t ro = ((size_t)t * B + row0 + tid) * 3;
            a.out_seq[ro] = o0; a.out_seq[ro + 1] = o1; a.out_seq[ro + 2] = dt;
            if (TRAIN) a.oraw_all[(size_t)t * B + row0 + tid] = o2;
            px += o0; py += o1;                                          // RG.py:585
            bool tf = false;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
            if (a.target) {
                t0 = a.target[ro]; t1 = a.target[ro + 1]; t2 = a.target[ro + 2];
                tpx += t0; tpy += t1;                                    // running form of RG.py:591-593
                tf = a.tf_mask && a.tf_mask[t];
            }
            if (tf) { in0 = t0; in1 = t1; in2 = t2; px = tpx; py = tpy; }
            else { in0 = o0; in1 = o1; in2 = dt; }
            if constexpr (!TRAIN) {
                if (stopping) {
                    const float qx = px - ex, qy = py - ey;
                    const bool hit = sqrtf(qx * qx + qy * qy) < a.stop_distance || (int)(row0 + tid) >= a.rows_valid;
                    if (hit && first_hit == 0) first_hit = t + 1;            // RG.py:675-680
                    if (__all(first_hit != 0) && tid == 0) s_stop = 1;       // the 32 row-owning lanes vote
                }
            }
        }
        GMT_PH(6);
        cur ^= 1;
    }
    GMT_PH_DUMP;
    if constexpr (!TRAIN) {
        if (stopping) {
            // steps the tile never ran: zero (the reference keeps generating them until every row of the BATCH
            // is done; they lie beyond each of these rows' lengths)
            for (int tt = t_exit; tt < T; ++tt)
                if (tid < 3 * GMT_TM) a.out_seq[((size_t)tt * B + row0) * 3 + tid] = 0.f;
            if (tid < GMT_TM && a.lengths_out) a.lengths_out[row0 + tid] = first_hit ? first_hit : T;
        }
    }
}

// ---------------------------------------------------------------------------- backward
__device__ __forceinline__ void cell_bwd(float dht, float& dc, float ig, float fg, float gg, float og,
                                         float cn, float cp, float& dai, float& daf, float& dag, float& dao)
{
    const float tc = gmt_tanh(cn);
    const float dog = dht * tc;
    const float dct = dc + dht * og * (1.f - tc * tc);
    dai = dct * gg * ig * (1.f - ig);
    daf = dct * cp * fg * (1.f - fg);
    dag = dct * ig * (1.f - gg * gg);
    dao = dog * og * (1.f - og);
    dc = dct * fg;
}

// Every operand of step t-1 is requested while step t's contractions run (one accumulator row of
// loads ahead of a group of MFMAs): layer 0's saved gates under B4, layer 1's and the per-row
// scalars under B6.  c_t is carried from the c_{t-1} fetch of the step before.
// SUMG: d g0 = sum over t of d(pre-activation of layer 0) (g0 enters every step's gates) is accumulated in LDS - element
// (row, unit) belongs to one lane for the whole launch, so a plain float4 read-modify-write per accumulator row and step
// (~1 k of a step's ~80 k cycles) - and written once at the end: the separate reduction pass over gates0 [T, B, 4H]
// (1 GB at B = 4096, T = 128: 0.45 ms as an ATen kernel) is gone.  Needs 32 x 4H floats of LDS: H <= 128.
template <int NW, bool RES0, bool SUMG>
__global__ __launch_bounds__(NW * 64, 1)
void gen_bwd_kernel(gmt_gen_bwd_args a)
{
    constexpr int H = NW * 32, HH = H / 2, NT = NW * 64, NP = NT / 32;
    constexpr int KS4 = 2 * H;                         // k-steps over K = 4H
    constexpr int NY = (GMT_TM * HH) / NT;             // head activations per thread (B2)
    extern __shared__ float smem[];
    float* daT = smem;                                 // [4H][LDA]
    float* dyT = daT + 4 * H * GMT_LDA;                // [HH][LDA]; later aliased by the du partials [NP][8][32]
    float* doS = dyT + HH * GMT_LDA;                   // [3][32]
    constexpr bool WUL = (NW <= 4);                    // W_u [4H][8] (zero-padded) cached in LDS when it fits: read by every thread in B6
    float* wuS = doS + 96;
    float* sgS = wuS + (WUL ? 4 * H * 8 : 0);          // [32][4H]  (SUMG)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * GMT_TM;
    if constexpr (SUMG) {
        for (int i = tid; i < GMT_TM * 4 * H; i += NT) sgS[i] = 0.f;
    }
    const int hi = lane >> 5, lo = lane & 31;
    const int col = wave * 32 + lo;
    const int T = a.T, B = a.B;

    const float4* w1T = reinterpret_cast<const float4*>(a.w1T_pk) + (size_t)wave * (HH / 8) * 64;
    const float4* wl1T = reinterpret_cast<const float4*>(a.w_l1T_pk) + (size_t)wave * (4 * H / 8) * 2 * 64;
    const float4* whh0T = reinterpret_cast<const float4*>(a.w_hh0T_pk) + (size_t)wave * (4 * H / 8) * 64;
    float wr0[1][RES0 ? KS4 : 1];
    if constexpr (RES0) gmt_load_resident<1, KS4>(wr0, whh0T, lane);
    if constexpr (WUL) {
        for (int i = tid; i < 4 * H * 8; i += NT) wuS[i] = a.w_uP[i];
    }

    float dh0[16], dc0[16], dh1[16], dc1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dh0[r] = 0.f; dc0[r] = 0.f; dh1[r] = 0.f; dc1[r] = 0.f; }
    float din0 = 0.f, din1 = 0.f, din2 = 0.f, dpx = 0.f, dpy = 0.f;    // carried (threads 0..31)
    const bool drop = a.dropout_p > 0.f;
    const uint32_t thresh24 = (uint32_t)(a.dropout_p * 16777216.0f);
    const float keep_scale = 1.0f / (1.0f - a.dropout_p);
    const size_t lane_row = (size_t)row0 + 4 * hi;

    // prefetch registers
    float g1[4][16], cp1[16], cn1[16];                 // layer 1: gates of step t, c_{t-1}, c_t
    float g0[4][16], cp0[16], cn0[16];                 // layer 0
    float py[NY];                                      // y_all of step t (B2 map: row tid/HH + (NT/HH) j, unit tid%HH)
    float pdo0 = 0.f, pdo1 = 0.f, pdo2 = 0.f, po2 = 0.f, prx = 0.f, pry = 0.f, pdist = 1.f;   // threads 0..31
    static_assert(NT % HH == 0, "B2 thread map");
    auto fetch_row1 = [&](int t, int r) {
        const int mr = (r & 3) + 8 * (r >> 2);
        const float4 q = gmt_ld4(a.gates1 + (((size_t)t * B + lane_row) + mr) * (4 * H) + 4 * col);
        g1[0][r] = q.x; g1[1][r] = q.y; g1[2][r] = q.z; g1[3][r] = q.w;
        const int tp = t > 0 ? t - 1 : 0;              // (t = 0: c_{-1} = 0 is selected at use)
        cp1[r] = a.c1_all[(((size_t)tp * B + lane_row) + mr) * H + col];
    };
    auto fetch_row0 = [&](int t, int r) {
        const int mr = (r & 3) + 8 * (r >> 2);
        const float4 q = gmt_ld4(a.gates0 + (((size_t)t * B + lane_row) + mr) * (4 * H) + 4 * col);
        g0[0][r] = q.x; g0[1][r] = q.y; g0[2][r] = q.z; g0[3][r] = q.w;
        const int tp = t > 0 ? t - 1 : 0;
        cp0[r] = a.c0_all[(((size_t)tp * B + lane_row) + mr) * H + col];
    };
    auto fetch_small = [&](int t) {
        const size_t yb = ((size_t)t * B + row0 + tid / HH) * HH + tid % HH;
#pragma unroll
        for (int j = 0; j < NY; ++j) py[j] = a.y_all[yb + (size_t)j * (NT / HH) * HH];
        if (tid < GMT_TM) {
            const size_t rr = (size_t)t * B + row0 + tid;
            pdo0 = a.d_out[rr * 3]; pdo1 = a.d_out[rr * 3 + 1]; pdo2 = a.d_out[rr * 3 + 2];
            po2 = a.oraw_all[rr];
            prx = a.u_all[rr * 8 + 3]; pry = a.u_all[rr * 8 + 4]; pdist = a.u_all[rr * 8 + 5];
        }
    };
    {
        const int t = T - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            fetch_row1(t, r);
            const int mr = (r & 3) + 8 * (r >> 2);
            cn1[r] = a.c1_all[(((size_t)t * B + lane_row) + mr) * H + col];
            cn0[r] = a.c0_all[(((size_t)t * B + lane_row) + mr) * H + col];
        }
        fetch_small(t);
    }
    __syncthreads();

    GMT_PH_DECL;
    for (int t = T - 1; t >= 0; --t) {
        const size_t rs = (size_t)t * B + lane_row;
        const int tn = t > 0 ? t - 1 : 0;
        GMT_PH(7);              // the step fetched next (clamped: no branch around loads)
        const bool tf = a.tf_mask && a.tf_mask[t];     // input of step t+1 came from the target
        // ---- B1: gradient of out_t = external + feedback through next input / position
        if (tid < GMT_TM) {
            const size_t ro = ((size_t)t * B + row0 + tid) * 3;
            float q0 = pdo0, q1 = pdo1, q2 = pdo2;
            if (!tf) { q0 += din0 + dpx; q1 += din1 + dpy; q2 += din2; }
            else { dpx = 0.f; dpy = 0.f; }             // pos_{t+1} was reset from the targets: path cut
            const float sg = po2 > 20.f ? 1.f : 1.f / (1.f + expf(-po2));     // d softplus
            q2 *= sg;
            doS[0 * 32 + tid] = q0; doS[1 * 32 + tid] = q1; doS[2 * 32 + tid] = q2;
            a.d_o[ro] = q0; a.d_o[ro + 1] = q1; a.d_o[ro + 2] = q2;
        }
        __syncthreads();
        GMT_PH(0);
        // ---- B2: head backward, d(ypre) = (d_o W2) * lrelu'
        {
            const float* w2 = gmt_opaque_s(a.w2);
            const int k = tid % HH;
            const float w20 = w2[k], w21 = w2[HH + k], w22 = w2[2 * HH + k];
            const size_t yb = ((size_t)t * B + row0 + tid / HH) * HH + k;
#pragma unroll
            for (int j = 0; j < NY; ++j) {
                const int row = tid / HH + j * (NT / HH);
                float v = doS[row] * w20 + doS[32 + row] * w21 + doS[64 + row] * w22;
                v *= (py[j] > 0.f) ? 1.f : 0.2f;
                dyT[k * GMT_LDA + row] = v;
                a.d_ypre[yb + (size_t)j * (NT / HH) * HH] = v;
            }
        }
        __syncthreads();
        GMT_PH(1);
        // ---- B3: dh1 += d(ypre) W1 ; layer-1 cell backward -> da1
        {
            f32x16 acc[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = dh1[r];
            gmt_mma_stream<1>(acc, dyT, w1T, HH, lane);
            float* gp = a.gates1 + rs * (4 * H) + 4 * col;
            float* dl = daT + col * GMT_LDA + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (r & 3) + 8 * (r >> 2);
                float* q = gp + (size_t)mr * (4 * H);
                const float cp = t > 0 ? cp1[r] : 0.f;
                float dai, daf, dag, dao;
                cell_bwd(acc[0][r], dc1[r], g1[0][r], g1[1][r], g1[2][r], g1[3][r], cn1[r], cp, dai, daf, dag, dao);
                cn1[r] = cp1[r];
                gmt_st4(q, dai, daf, dag, dao);
                dl[0 * H * GMT_LDA + mr] = dai; dl[1 * H * GMT_LDA + mr] = daf;
                dl[2 * H * GMT_LDA + mr] = dag; dl[3 * H * GMT_LDA + mr] = dao;
            }
        }
        __syncthreads();
        GMT_PH(2);
        // ---- B4: dh1_prev = da1 W_hh1 ; dh0_total = dh0 + da1 W_ih1   (layer 0's operands of this step arrive)
        f32x16 acc2[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc2[0][r] = 0.f; acc2[1][r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) fetch_row0(t, r);
        GMT_PH(3);
        gmt_mma_stream<2>(acc2, daT, wl1T, 4 * H, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dh1[r] = acc2[0][r];
            float g = acc2[1][r];                      // gradient of layer 1's (dropped) input
            if (drop) {
                const uint32_t idx = (uint32_t)((rs + (r & 3) + 8 * (r >> 2)) * H + col);
                g = gmt_keep(idx, a.seed, thresh24) ? g * keep_scale : 0.f;
            }
            acc2[1][r] = dh0[r] + g;
        }
        __syncthreads();
        GMT_PH(4);
        // ---- B5: layer-0 cell backward -> da0
        {
            float* gp = a.gates0 + rs * (4 * H) + 4 * col;
            float* dl = daT + col * GMT_LDA + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (r & 3) + 8 * (r >> 2);
                float* q = gp + (size_t)mr * (4 * H);
                const float cp = t > 0 ? cp0[r] : 0.f;
                float dai, daf, dag, dao;
                cell_bwd(acc2[1][r], dc0[r], g0[0][r], g0[1][r], g0[2][r], g0[3][r], cn0[r], cp, dai, daf, dag, dao);
                cn0[r] = cp0[r];
                gmt_st4(q, dai, daf, dag, dao);
                if constexpr (SUMG) {
                    float4* sp = reinterpret_cast<float4*>(sgS + (size_t)(mr + 4 * hi) * (4 * H) + 4 * col);
                    float4 sv = *sp;
                    sv.x += dai; sv.y += daf; sv.z += dag; sv.w += dao;
                    *sp = sv;
                }
                dl[0 * H * GMT_LDA + mr] = dai; dl[1 * H * GMT_LDA + mr] = daf;
                dl[2 * H * GMT_LDA + mr] = dag; dl[3 * H * GMT_LDA + mr] = dao;
            }
        }
        __syncthreads();
        GMT_PH(5);
        // ---- B6: dh0_prev = da0 W_hh0 ; partial d(u) = da0 W_u over this thread's 64-wide k slice
        //          (layer 1's operands and the per-row scalars of step t-1 arrive)
        float urx = prx, ury = pry, udist = pdist;      // this step's values, before the fetch overwrites them
        {
            f32x16 acc[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
            if constexpr (RES0) {
                gmt_mma_resident<1, KS4>(acc, daT, wr0, lane, nullptr, [&](int grp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((r * (KS4 / 8)) / 16 == grp) fetch_row1(tn, r);
                    if (grp == KS4 / 8 - 1) fetch_small(tn);
                });
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) fetch_row1(tn, r);
                fetch_small(tn);
                gmt_mma_stream<1>(acc, daT, whh0T, 4 * H, lane);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dh0[r] = acc[0][r];
            const int part = tid >> 5, row = tid & 31;
            float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int k0 = part * 64;
            const float* wup = WUL ? wuS : gmt_opaque_s(a.w_uP);
#pragma unroll 8
            for (int k = k0; k < k0 + 64; ++k) {
                const float d = daT[k * GMT_LDA + row];
                const float4 wa = reinterpret_cast<const float4*>(wup + (size_t)k * 8)[0];
                const float4 wb = reinterpret_cast<const float4*>(wup + (size_t)k * 8)[1];
                s[0] = fmaf(d, wa.x, s[0]); s[1] = fmaf(d, wa.y, s[1]); s[2] = fmaf(d, wa.z, s[2]);
                s[3] = fmaf(d, wa.w, s[3]); s[4] = fmaf(d, wb.x, s[4]); s[5] = fmaf(d, wb.y, s[5]);
                s[6] = fmaf(d, wb.z, s[6]);
            }
            float* pp = dyT + (size_t)part * 8 * 32 + row;   // aliases dyT (dead since B3)
#pragma unroll
            for (int j = 0; j < 7; ++j) pp[j * 32] = s[j];
        }
        __syncthreads();
        GMT_PH(6);
        // ---- B7: reduce d(u); dynamic-feature backward (RG.py:502-506) -> d(pos_t); carry
        if (tid < GMT_TM) {
            float du[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 7; ++j) du[j] += dyT[(size_t)p * 8 * 32 + j * 32 + tid];
            const float rx = urx, ry = ury, dist = udist;
            const float r2 = rx * rx + ry * ry;
            const float inv = r2 > 0.f ? 1.f / r2 : 0.f;
            const float drx = du[3] + du[5] * rx / dist - du[6] * ry * inv;
            const float dry = du[4] + du[5] * ry / dist + du[6] * rx * inv;
            // rem = end - pos  ->  d(pos_t) = -d(rem) + what flows on to pos_{t+1}
            dpx = dpx - drx; dpy = dpy - dry;
            din0 = du[0]; din1 = du[1]; din2 = du[2];
            if (t == 0) {
                float* o = a.d_in0 + ((size_t)row0 + tid) * 3;
                o[0] = din0; o[1] = din1; o[2] = din2;
            }
        }
    }
    if constexpr (SUMG) {
        // (each lane reads back only what it wrote itself: no barrier needed)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float4 sv = *reinterpret_cast<const float4*>(sgS + (size_t)row * (4 * H) + 4 * col);
            *reinterpret_cast<float4*>(a.d_g0 + ((size_t)row0 + row) * (4 * H) + 4 * col) = sv;
        }
    }
    GMT_PH_DUMP;
}

template <int NW>
int launch_gen_fwd(const gmt_gen_fwd_args& a, hipStream_t stream)
{
    constexpr int H = NW * 32, HH = H / 2, NH = (HH + 31) / 32;
    constexpr bool RES0 = (NW <= 4);
    const size_t lds = ((size_t)((NW > 4 ? 3 : 4) * H + NH * 32 + 8 + H) * GMT_LDA + 3 * HH + NW * 2 * 96
                        + ((RES0 && NW == 4) ? NW * 16 * 4 * 64 : 0)) * sizeof(float);
    if (lds > 160 * 1024 - 256) return GMT_ERR_UNSUPPORTED;
    auto k = a.gates0 ? gen_fwd_kernel<NW, RES0, true> : gen_fwd_kernel<NW, RES0, false>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GMT_ERR_HIP;
    hipLaunchKernelGGL(k, dim3(a.B / GMT_TM), dim3(NW * 64), lds, stream, a);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

template <int NW>
int launch_gen_bwd(const gmt_gen_bwd_args& a, hipStream_t stream)
{
    constexpr int H = NW * 32, HH = H / 2;
    constexpr bool RES0 = (NW <= 2);             // at H >= 128 the prefetch registers need the room: W_hh_l0^T streams from L2
    constexpr bool CAN_SUM = (NW <= 4);
    if (a.d_g0 && !CAN_SUM) return GMT_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)(4 * H + HH) * GMT_LDA + 96 + (NW <= 4 ? 4 * H * 8 : 0) + (a.d_g0 ? GMT_TM * 4 * H : 0)) * sizeof(float);
    if (lds > 160 * 1024 - 256) return GMT_ERR_UNSUPPORTED;
    auto k = (a.d_g0 && CAN_SUM) ? gen_bwd_kernel<NW, RES0, CAN_SUM> : gen_bwd_kernel<NW, RES0, false>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GMT_ERR_HIP;
    hipLaunchKernelGGL(k, dim3(a.B / GMT_TM), dim3(NW * 64), lds, stream, a);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

}  // namespace

extern "C" int gmt_gen_rollout_fwd(const gmt_gen_fwd_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && a->g0 && a->w_u_pk && a->w_hh0_pk && a->w_ih1_pk && a->w_hh1_pk && a->bias1 && a->w1_pk &&
                  a->b1 && a->w2 && a->b2 && a->starts && a->ends && a->init_in && a->out_seq);
    GMT_CHECK_ARG(a->T >= 1 && a->B >= GMT_TM && (a->B % GMT_TM) == 0);
    GMT_CHECK_ARG(!(a->stop_distance > 0.f) || (!a->gates0 && !a->target && a->rows_valid >= 1 && a->rows_valid <= a->B));
    {   // the saved-activation buffers come all together (training) or not at all (no-grad rollout)
        const int n_saved = !!a->gates0 + !!a->c0_all + !!a->h0_all + !!a->gates1 + !!a->c1_all + !!a->h1_all + !!a->u_all +
                            !!a->y_all + !!a->oraw_all;
        GMT_CHECK_ARG(n_saved == 0 || n_saved == 9);
        GMT_CHECK_ARG(!(n_saved == 9 && a->dropout_p > 0.f) || a->h0d_all);
    }
    switch (a->H) {
        case 32: return launch_gen_fwd<1>(*a, stream);
        case 64: return launch_gen_fwd<2>(*a, stream);
        case 128: return launch_gen_fwd<4>(*a, stream);
        case 256: return launch_gen_fwd<8>(*a, stream);
        default: return GMT_ERR_UNSUPPORTED;
    }
}

extern "C" int gmt_gen_rollout_bwd(const gmt_gen_bwd_args* a, hipStream_t stream)
{
    GMT_CHECK_ARG(a && a->gates0 && a->c0_all && a->gates1 && a->c1_all && a->y_all && a->oraw_all && a->u_all &&
                  a->d_out && a->w1T_pk && a->w_l1T_pk && a->w_hh0T_pk && a->w_uP && a->w2 && a->d_ypre && a->d_o &&
                  a->d_in0);
    GMT_CHECK_ARG(a->T >= 1 && a->B >= GMT_TM && (a->B % GMT_TM) == 0);
    switch (a->H) {
        case 32: return launch_gen_bwd<1>(*a, stream);
        case 64: return launch_gen_bwd<2>(*a, stream);
        case 128: return launch_gen_bwd<4>(*a, stream);
        case 256: return launch_gen_bwd<8>(*a, stream);
        default: return GMT_ERR_UNSUPPORTED;
    }
}
