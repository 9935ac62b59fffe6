// Weight-gradient contraction of the recurrent layers:  dW[n][i] = sum_r dA[r][n] * X[r][i]
// over R = L*B row-steps (K ~ 5e5, N = 4H = 512, I = 128 / 256) - the "plain GEMM over time"
// that autograd runs for RecurrentGAN.py:1021 / 1109.  Its shape (tiny output, huge K) is
// the worst case of a tiled library GEMM, so it is a split-K fp32-MFMA kernel here:
//   * a workgroup owns a 256 (n) x 128 (i) output tile for one slab of rows and keeps it in
//     128 accumulator registers per lane (4 waves x [64 x 128]);
//   * both operands are row-major with the contraction index r as the ROW, which is exactly
//     the k-major order the MFMA fragments want: staging is a straight float4 copy
//     global -> registers -> LDS (no transpose), issued one chunk ahead (T14 split);
//   * partial tiles go to a [KSPLIT][N][I] slab buffer and a second pass sums them in a fixed
//     order (deterministic; float atomics would make gradients run-to-run different);
//   * the column sums of dA (= bias gradient) are accumulated by the staging threads for free.
#include "lstm_tile.h"
#include <cstdlib>
#include <mutex>

// Scratch of gmt_wgrad_noalloc (include/gmt_hip.h): the split-K slab buffer that gmt_wgrad takes from its
// caller as `workspace`, as one file-scope device array per GPU (65 MiB, held until the process ends).  In
// host code the NAME of this array is only a stand-in: its device address comes from hipGetSymbolAddress,
// once per device, and that pointer is what the kernels and every other HIP call are given.
static __device__ __align__(256) float gmt_wgrad_scratch[GMT_WGRAD_NOALLOC_MAX_FLOATS];

namespace {

constexpr int WG_RC = 16;                               // rows per chunk

// NB: 32-wide n blocks per wave (tile n = 128 NB), QB: 32-wide i blocks (tile i = 32 QB).
//   <2, 4>  256 x 128  the recurrent weights (N = 4H, I = H / 2H)
//   <1, 4>  128 x 128  N = 64 / 128 (feature encoder, output heads); columns n >= N read as zero
//   <*, 1>  x 32       skinny inputs (I <= 32, any row pitch: scalar guarded loads), HBM-bound
// Workgroups of one row slab sit on the same XCD (blockIdx % 8), so the dA / X rows they share are
// fetched into that XCD's L2 once.
template <int NB, int QB>
__global__ __launch_bounds__(256, 2)
void wgrad_kernel(const float* __restrict__ dA, long long lda, const float* __restrict__ X, long long ldx,
                  float* __restrict__ slab, float* __restrict__ db_slab, int N, int I, long long R, int ksplit,
                  float x_drop_p, uint32_t x_drop_seed)
{
    // X is read through an inverted-dropout mask (the consumer layer applied the same one to its input
    // in LDS, csrc/lstm_seq.hip): element (r, i) is kept iff gmt_keep16(r*ldx + i, seed).  Wide tiles only.
    const uint32_t x_thresh16 = (uint32_t)(x_drop_p * 65536.0f);
    const float x_keep_scale = 1.0f / (1.0f - x_drop_p);
    constexpr int TN = 128 * NB, TI = 32 * QB;
    constexpr int LDA_ = TN + 4, LDX_ = TI + 4;         // LDS pitches (floats): +4 keeps float4 alignment
    constexpr int AC4 = TN / 4, ARG = 256 / AC4, AJ = WG_RC / ARG;   // A staging: float4 columns, row groups, float4 per thread
    constexpr bool XV = (QB == 4);                      // X staged as float4 (aligned, I % 128 == 0)
    constexpr int XC4 = TI / 4, XRG = 256 / XC4, XJ = XV ? WG_RC / XRG : 1;
    __shared__ float sA[2][WG_RC][LDA_];
    __shared__ float sX[2][WG_RC][LDX_];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    const int n_tiles = (N + TN - 1) / TN, i_tiles = (I + TI - 1) / TI;
    // blockIdx = xcd + 8 * (tile + tiles * ks_group), ks = 8 * ks_group + xcd
    int bid = blockIdx.x;
    const int xcd = bid & 7; bid >>= 3;
    const int it = bid % i_tiles; bid /= i_tiles;
    const int nt = bid % n_tiles; bid /= n_tiles;
    const int ks = bid * 8 + xcd;
    if (ks >= ksplit) return;
    const long long n_chunks = R / WG_RC;
    const long long cpk = (n_chunks + ksplit - 1) / ksplit;
    const long long c0 = (long long)ks * cpk;
    long long c1 = c0 + cpk; if (c1 > n_chunks) c1 = n_chunks;

    const int a_c4 = tid % AC4, a_r0 = tid / AC4;       // rows a_r0 + ARG * j
    const bool a_ok = nt * TN + 4 * a_c4 < N;           // N % 4 == 0: whole float4 in or out
    // X: float4 map (x_c4, rows x_r0 + XRG * j) or scalar map (2 elements: rows tid/32 and tid/32 + 8, column tid%32)
    const int x_c4 = tid % XC4, x_r0 = tid / XC4;
    const int xs_c = tid & 31, xs_r = tid >> 5;
    const bool xs_ok = it * TI + xs_c < I;
    // addressing: one uniform base per chunk (scalar registers) + per-thread 32-bit element offsets
    // that never change (a 16-row chunk spans < 2^31 elements)
    const float* tA = dA + (long long)nt * TN;
    const float* tX = X + (long long)it * TI;
    unsigned offA[AJ], offX[XV ? XJ : 2];
#pragma unroll
    for (int j = 0; j < AJ; ++j) offA[j] = (unsigned)((a_r0 + ARG * j) * lda + 4 * a_c4);
    if constexpr (XV) {
#pragma unroll
        for (int j = 0; j < XJ; ++j) offX[j] = (unsigned)((x_r0 + XRG * j) * ldx + 4 * x_c4);
    } else {
        offX[0] = (unsigned)(xs_r * ldx + xs_c);
        offX[1] = (unsigned)((xs_r + 8) * ldx + xs_c);
    }

    struct Stage { float4 ra[AJ]; float4 rx[XV ? XJ : 1]; };
    float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto g_load = [&](Stage& st, long long c) {
        const float* bA = tA + c * WG_RC * lda;
        const float* bX = tX + c * WG_RC * ldx;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            st.ra[j] = a_ok ? *reinterpret_cast<const float4*>(bA + offA[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (XV) {
#pragma unroll
            for (int j = 0; j < XJ; ++j) st.rx[j] = *reinterpret_cast<const float4*>(bX + offX[j]);
        } else {
            st.rx[0].x = xs_ok ? bX[offX[0]] : 0.f;
            st.rx[0].y = xs_ok ? bX[offX[1]] : 0.f;
        }
    };
    auto s_store = [&](Stage& st, int buf, long long c) {
        if constexpr (XV) {
            if (x_thresh16) {                           // uniform
                const uint32_t i0 = (uint32_t)(c * WG_RC * ldx) + (uint32_t)(it * TI);
#pragma unroll
                for (int j = 0; j < XJ; ++j) gmt_drop4(st.rx[j], i0 + offX[j], x_drop_seed, x_thresh16, x_keep_scale);
            }
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            *reinterpret_cast<float4*>(&sA[buf][a_r0 + ARG * j][4 * a_c4]) = st.ra[j];
            colsum.x += st.ra[j].x; colsum.y += st.ra[j].y; colsum.z += st.ra[j].z; colsum.w += st.ra[j].w;
        }
        if constexpr (XV) {
#pragma unroll
            for (int j = 0; j < XJ; ++j) *reinterpret_cast<float4*>(&sX[buf][x_r0 + XRG * j][4 * x_c4]) = st.rx[j];
        } else {
            sX[buf][xs_r][xs_c] = st.rx[0].x;
            sX[buf][xs_r + 8][xs_c] = st.rx[0].y;
        }
    };

    f32x16 acc[NB][QB];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto compute = [&](int buf) {
        const float* pa = &sA[buf][hi][wave * (32 * NB) + lo];
        const float* px = &sX[buf][hi][lo];
#pragma unroll
        for (int s = 0; s < WG_RC / 2; ++s) {
            float av[NB], bv[QB];
#pragma unroll
            for (int a = 0; a < NB; ++a) av[a] = pa[2 * s * LDA_ + 32 * a];
#pragma unroll
            for (int q = 0; q < QB; ++q) bv[q] = px[2 * s * LDX_ + 32 * q];
#pragma unroll
            for (int q = 0; q < QB; ++q)
#pragma unroll
                for (int a = 0; a < NB; ++a)
                    acc[a][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[q], acc[a][q], 0, 0, 0);
        }
    };

    // Two register stages in turn: chunk c+2 is requested while chunk c is contracted and chunk c+1
    // (requested one trip earlier) moves registers -> LDS, so a row chunk has two trips (~3 us) to
    // arrive from HBM.  Indices past the slab are clamped (a harmless re-read), never branched on.
    Stage s0, s1;
    const long long cl = c1 - 1;
    if (c0 < c1) {
        g_load(s0, c0);
        s_store(s0, 0, c0);
        g_load(s0, c0 + 1 < c1 ? c0 + 1 : cl);
    }
    __syncthreads();
    long long c = c0;
    for (; c + 1 < c1; c += 2) {                        // LDS[0] = chunk c, s0 = chunk c+1
        g_load(s1, c + 2 < c1 ? c + 2 : cl);
        compute(0);
        s_store(s0, 1, c + 1);
        __syncthreads();
        g_load(s0, c + 3 < c1 ? c + 3 : cl);
        compute(1);
        if (c + 2 < c1) s_store(s1, 0, c + 2);               // (a clamped re-read must not reach the column sums)
        __syncthreads();
    }
    if (c < c1) compute(0);                             // odd tail
    // partial tile -> slab[ks][n][i]   (rows n >= N / columns i >= I of an edge tile are dropped)
    {
        const int n0 = nt * TN + wave * (32 * NB), i0 = it * TI + lo;
        float* out = slab + ((long long)ks * N + n0) * I + i0;
        const bool whole = (nt + 1) * TN <= N && (it + 1) * TI <= I;
        if (whole) {
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int q = 0; q < QB; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        out[(long long)(a * 32 + gmt_acc_row(r, lane)) * I + 32 * q] = acc[a][q][r];
        } else {
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int q = 0; q < QB; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dn = a * 32 + gmt_acc_row(r, lane);
                        if (n0 + dn < N && i0 + 32 * q < I) out[(long long)dn * I + 32 * q] = acc[a][q][r];
                    }
        }
    }
    // bias gradient: column sums of this slab of dA (only the it == 0 workgroups write)
    if (db_slab && it == 0) {
        __syncthreads();
        float* red = &sA[0][0][0];                      // reuse: [ARG][TN]
        red[a_r0 * TN + 4 * a_c4 + 0] = colsum.x; red[a_r0 * TN + 4 * a_c4 + 1] = colsum.y;
        red[a_r0 * TN + 4 * a_c4 + 2] = colsum.z; red[a_r0 * TN + 4 * a_c4 + 3] = colsum.w;
        __syncthreads();
        if (tid < TN && nt * TN + tid < N) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < ARG; ++g) v += red[g * TN + tid];
            db_slab[(long long)ks * N + (long long)nt * TN + tid] = v;
        }
    }
}

// out[j] (+)= sum_k slab[k][j]; float4 lanes.  A block is 16 float4 columns x 16 slab groups: group y sums slabs
// [y*ksplit/16, (y+1)*ksplit/16) into four interleaved partial sums (k % 4: four independent loads in flight per thread),
// adds those in order, then the sixteen group sums are added in order - one fixed association of the additions whatever the
// launch geometry (deterministic), with enough loads in flight to pull the slabs (67 MB at N = 512, I = 128, 256 slabs) at
// memory speed: 28 -> 15 us per call, 47 calls per iteration (round 3: the 64-column x 4-group form kept ~1 load per thread).
// perm_h > 0: the rows n of the result are in the gate buffers' UNIT-MAJOR order (n = 4 unit + gate, H = perm_h units)
// and leave in torch's gate-major order (row gate H + unit), so that no separate permutation pass follows; I % 4 == 0
// (whole float4 per row) or I == 1 (the bias gradient: one float4 = the four gates of one unit).
constexpr int SR_COLS = 16, SR_GROUPS = 16;
__global__ __launch_bounds__(256)
void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long long n, int ksplit,
                        int accumulate, int I, int perm_h)
{
    __shared__ float4 part[SR_GROUPS][SR_COLS];
    const int x = threadIdx.x % SR_COLS, y = threadIdx.x / SR_COLS;
    const long long n4 = n >> 2;
    const int k0 = (int)((long long)ksplit * y / SR_GROUPS), k1 = (int)((long long)ksplit * (y + 1) / SR_GROUPS);
    for (long long base = (long long)blockIdx.x * SR_COLS; base < n4; base += (long long)gridDim.x * SR_COLS) {
        const long long i = base + x;
        float4 s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4) {
            const float4* p = reinterpret_cast<const float4*>(slab) + i;
            const long long n4s = n >> 2;
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = p[(long long)(k + q) * n4s];
#pragma unroll
                for (int q = 0; q < 4; ++q) { s[q].x += v[q].x; s[q].y += v[q].y; s[q].z += v[q].z; s[q].w += v[q].w; }
            }
            for (int q = 0; k < k1; ++k, ++q) {
                const float4 v = p[(long long)k * n4s];
                s[q].x += v.x; s[q].y += v.y; s[q].z += v.z; s[q].w += v.w;
            }
        }
        float4 t = s[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) { t.x += s[q].x; t.y += s[q].y; t.z += s[q].z; t.w += s[q].w; }
        part[y][x] = t;
        __syncthreads();
        if (y == 0 && i < n4) {
            t = part[0][x];
#pragma unroll
            for (int q = 1; q < SR_GROUPS; ++q) { t.x += part[q][x].x; t.y += part[q][x].y; t.z += part[q][x].z; t.w += part[q][x].w; }
            if (perm_h > 0 && I == 1) {                         // bias: elements 4 i .. 4 i + 3 are gates 0..3 of unit i
                float* o = out + i;
                if (accumulate) { t.x += o[0]; t.y += o[perm_h]; t.z += o[2 * (long long)perm_h]; t.w += o[3 * (long long)perm_h]; }
                o[0] = t.x; o[perm_h] = t.y; o[2 * (long long)perm_h] = t.z; o[3 * (long long)perm_h] = t.w;
            } else {
                long long d = i;
                if (perm_h > 0) {
                    const long long row = (4 * i) / I, c4 = ((4 * i) % I) >> 2;
                    d = (((row & 3) * perm_h + (row >> 2)) * (long long)I >> 2) + c4;
                }
                if (accumulate) { const float4 a = reinterpret_cast<const float4*>(out)[d]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
                reinterpret_cast<float4*>(out)[d] = t;
            }
        }
        __syncthreads();
    }
}

// the same one float at a time (N * I not a multiple of 4, e.g. I = 9)
__global__ void slab_reduce_scalar_kernel(const float* __restrict__ slab, float* __restrict__ out, long long n, int ksplit,
                                          int accumulate)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = accumulate ? out[i] : 0.f;
        for (int k = 0; k < ksplit; ++k) s += slab[(long long)k * n + i];
        out[i] = s;
    }
}

}  // namespace

// tile shape for (N, I): 0 = <2,4>, 1 = <1,4>, 2 = <2,1>, 3 = <1,1>; -1 unsupported
static int wgrad_shape(int N, int I, long long ldx)
{
    if (N < 4 || (N % 4) != 0 || I < 1) return -1;
    // wide tiles read whole 128-column blocks of X: I % 128 == 0, or 32 < I with rows padded (readable, finite) up to the
    // next multiple of 128 - columns >= I of the edge tile are dropped at the store
    const bool wide = (ldx % 4) == 0 && ((I % 128) == 0 || (I > 32 && ldx >= ((I + 127) / 128) * 128));
    if (!wide && I > 32) return -1;
    const bool big = (N % 256) == 0;                    // (128 x 128 tiles for N = 512 measured no faster)
    return wide ? (big ? 0 : 1) : (big ? 2 : 3);
}

extern "C" int gmt_wgrad_workspace_floats(int N, int I, long long R, int* ksplit_out, long long* floats_out)
{
    const int shape = wgrad_shape(N, I, ((I + 127) / 128) * 128);   // (gmt_wgrad checks the real pitch)
    GMT_CHECK_ARG(shape >= 0 && R >= WG_RC && (R % WG_RC) == 0);
    const int TN = (shape == 0 || shape == 2) ? 256 : 128, TI = (shape <= 1) ? 128 : 32;
    const int tiles = ((N + TN - 1) / TN) * ((I + TI - 1) / TI);
    static const int per_cu = [] { const char* e = getenv("GMT_WGRAD_WG_PER_CU"); int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    int ks = (per_cu * 256 + tiles - 1) / tiles;        // ~2 workgroups per CU (tuning knob: GMT_WGRAD_WG_PER_CU)
    const long long n_chunks = R / WG_RC;
    if (ks > n_chunks) ks = (int)n_chunks;
    if (ks < 1) ks = 1;
    if (ksplit_out) *ksplit_out = ks;
    if (floats_out) *floats_out = (long long)ks * N * I + (long long)ks * N;
    return GMT_OK;
}

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
float* g_scratch_ptr[kMaxDevices];       // per device: each GPU holds its own copy of the array at its own address

// device address of gmt_wgrad_scratch on the CURRENT device (looked up once per device)
int wgrad_scratch_ptr(float** out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return GMT_ERR_HIP;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(gmt_wgrad_scratch)) != hipSuccess || !p) return GMT_ERR_HIP;
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    *out = g_scratch_ptr[dev];
    return GMT_OK;
}
}  // namespace

extern "C" int gmt_slab_sum(const float* slab, float* out, long long n, int k, int accumulate, hipStream_t stream)
{
    GMT_CHECK_ARG(slab && out && n >= 4 && (n % 4) == 0 && k >= 1 && ((((uintptr_t)slab) | ((uintptr_t)out)) & 15) == 0);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(gmt_grid_1d(n / 4, SR_COLS)), dim3(256), 0, stream, slab, out, n, k, accumulate, 4, 0);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_wgrad_noalloc_init(void)
{
    float* p = nullptr;
    return wgrad_scratch_ptr(&p);
}

extern "C" int gmt_wgrad_noalloc(const float* dA, long long lda, const float* X, long long ldx, long long R, int N, int I,
                                 float* dW, float* db, int accumulate, float x_dropout_p, unsigned x_dropout_seed,
                                 hipStream_t stream)
{
    int ks = 0; long long need = 0;
    const int rc = gmt_wgrad_workspace_floats(N, I, R, &ks, &need);
    if (rc != GMT_OK) return rc;
    if (need > (long long)GMT_WGRAD_NOALLOC_MAX_FLOATS) return GMT_ERR_WORKSPACE;   // beyond the static scratch: use gmt_wgrad
    float* ws = nullptr;
    const int rp = wgrad_scratch_ptr(&ws);
    if (rp != GMT_OK) return rp;
    // every slab element the reduction reads is written by this call's wgrad_kernel launch first, so the
    // previous call's contents never show and nothing has to be cleared
    return gmt_wgrad(dA, lda, X, ldx, R, N, I, dW, db, accumulate, ws, (long long)GMT_WGRAD_NOALLOC_MAX_FLOATS,
                     x_dropout_p, x_dropout_seed, stream);
}

extern "C" int gmt_wgrad(const float* dA, long long lda, const float* X, long long ldx, long long R, int N, int I,
                         float* dW, float* db, int accumulate, float* workspace, long long workspace_floats,
                         float x_dropout_p, unsigned x_dropout_seed, hipStream_t stream)
{
    return gmt_wgrad_gates(dA, lda, X, ldx, R, N, I, dW, db, accumulate, workspace, workspace_floats, x_dropout_p,
                           x_dropout_seed, 0, stream);
}

extern "C" int gmt_wgrad_gates(const float* dA, long long lda, const float* X, long long ldx, long long R, int N, int I,
                               float* dW, float* db, int accumulate, float* workspace, long long workspace_floats,
                               float x_dropout_p, unsigned x_dropout_seed, int unit_major_h, hipStream_t stream)
{
    GMT_CHECK_ARG(dA && X && dW && workspace);
    GMT_CHECK_ARG(unit_major_h == 0 || (N == 4 * unit_major_h && (I % 4) == 0 && (((uintptr_t)dW) & 15) == 0));
    GMT_CHECK_ARG(x_dropout_p >= 0.f && x_dropout_p < 1.f);
    const int shape = wgrad_shape(N, I, ldx);
    GMT_CHECK_ARG(shape >= 0);
    if (x_dropout_p > 0.f) GMT_CHECK_ARG(shape <= 1 && R * ldx < (1ll << 32));   // wide tiles; 32-bit mask counter
    GMT_CHECK_ARG((lda % 4) == 0 && (((uintptr_t)dA | (uintptr_t)workspace) & 15) == 0);
    if (shape <= 1) GMT_CHECK_ARG(((uintptr_t)X & 15) == 0);
    int ks = 0; long long need = 0;
    const int rc = gmt_wgrad_workspace_floats(N, I, R, &ks, &need);
    if (rc != GMT_OK) return rc;
    if (workspace_floats < need) return GMT_ERR_WORKSPACE;
    float* slab = workspace;
    float* db_slab = db ? workspace + (long long)ks * N * I : nullptr;
    const int TN = (shape == 0 || shape == 2) ? 256 : 128, TI = (shape <= 1) ? 128 : 32;
    const int tiles = ((N + TN - 1) / TN) * ((I + TI - 1) / TI);
    const int grid = 8 * tiles * ((ks + 7) / 8);
    switch (shape) {
    case 0: hipLaunchKernelGGL((wgrad_kernel<2, 4>), dim3(grid), dim3(256), 0, stream, dA, lda, X, ldx, slab, db_slab, N, I, R, ks, x_dropout_p, x_dropout_seed); break;
    case 1: hipLaunchKernelGGL((wgrad_kernel<1, 4>), dim3(grid), dim3(256), 0, stream, dA, lda, X, ldx, slab, db_slab, N, I, R, ks, x_dropout_p, x_dropout_seed); break;
    case 2: hipLaunchKernelGGL((wgrad_kernel<2, 1>), dim3(grid), dim3(256), 0, stream, dA, lda, X, ldx, slab, db_slab, N, I, R, ks, x_dropout_p, x_dropout_seed); break;
    default: hipLaunchKernelGGL((wgrad_kernel<1, 1>), dim3(grid), dim3(256), 0, stream, dA, lda, X, ldx, slab, db_slab, N, I, R, ks, x_dropout_p, x_dropout_seed); break;
    }
    GMT_LAUNCH_CHECK();
    const long long n = (long long)N * I;
    if ((n % 4) == 0 && (((uintptr_t)dW) & 15) == 0)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(gmt_grid_1d(n / 4, SR_COLS)), dim3(256), 0, stream, slab, dW, n, ks, accumulate,
                           I, unit_major_h);
    else
        hipLaunchKernelGGL(slab_reduce_scalar_kernel, dim3(gmt_grid_1d(n, 256)), dim3(256), 0, stream, slab, dW, n, ks, accumulate);
    GMT_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(gmt_grid_1d(N / 4, SR_COLS)), dim3(256), 0, stream, db_slab, db,
                           (long long)N, ks, accumulate, 1, unit_major_h);
        GMT_LAUNCH_CHECK();
    }
    return GMT_OK;
}
