// Fused front end of the critic: kinematic features -> SN-Linear(9 -> H) -> LeakyReLU -> SN-Linear(H -> H)
// -> LeakyReLU for every point of every trajectory, in one pass (compute_kinematics RG.py:315-389 called at
// RG.py:757, Discriminator.feature_encoder RG.py:712-717 applied at RG.py:761).  As ATen ops this is ~25
// launches and five passes over [L*B, H] tensors per critic call; here the 9 features and the first
// activation never leave the CU unless the backward pass needs them (feat / h1 outputs).
//
// A workgroup (H/32 waves) walks tiles of 128 consecutive row-steps (time-major rows r = t*B + b):
//   phase 1  threads 0..127: the 9 features of one point each (gmt_kin::kin_fwd_vals) -> LDS, transposed;
//   phase 2  h1 = lrelu(feat W1^T + b1): K = 16 (9 used) MFMA contraction, W1 fragments in registers;
//   phase 3  e  = lrelu(h1 W2^T + b2):   K = H, W2 (H x H fp32) RESIDENT in registers for the whole kernel
//            (H = 128: 64 VGPRs a lane), four 32-row blocks per weight fragment read.
// Wave w owns hidden units [32w, 32w + 32) (lstm_tile.h decomposition).  MFMA-bound: 2 (16 + H) H FLOP per
// row-step; two workgroups per CU so that one's feature phase hides under the other's contractions.
#include "lstm_tile.h"
#include "kinematics_point.h"

namespace {

constexpr int ENC_RT = 128;                            // rows per tile
constexpr int ENC_LDP = ENC_RT + 4;                    // LDS pitch (floats) of a transposed operand row

template <int NW>
__global__ __launch_bounds__(NW * 64)
void encoder_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ dts,
                        const float* __restrict__ w1_pk, const float* __restrict__ b1,
                        const float* __restrict__ w2_pk, const float* __restrict__ b2,
                        float* __restrict__ feat_out, float* __restrict__ h1_out, float* __restrict__ e_out,
                        int L, int B, float eps, float slope, long long R, long long n_tiles)
{
    constexpr int H = NW * 32, NT = NW * 64, KS = H / 2, RB = ENC_RT / 32;
    extern __shared__ float smem[];
    float* fT = smem;                                  // [16][LDP]  features (rows 9..15 stay zero)
    float* hT = smem + 16 * ENC_LDP;                   // [H][LDP]   first activation
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    const int col = wave * 32 + lo;

    float w1r[1][8], w2r[1][KS];
    gmt_load_resident<1, 8>(w1r, reinterpret_cast<const float4*>(w1_pk) + (size_t)wave * 2 * 64, lane);
    gmt_load_resident<1, KS>(w2r, reinterpret_cast<const float4*>(w2_pk) + (size_t)wave * (H / 8) * 64, lane);
    const float b1v = b1[col], b2v = b2[col];
    for (int i = tid; i < 16 * ENC_LDP; i += NT) fT[i] = 0.f;
    __syncthreads();

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row0 = tile * ENC_RT;
        // ---- phase 1: features
        for (int i = tid; i < ENC_RT; i += NT) {
            const long long idx = row0 + i;
            if (idx < R) {
                float o[9];
                gmt_kin::kin_fwd_vals(idx, pos, dts, o, L, B, eps);
#pragma unroll
                for (int c = 0; c < 9; ++c) fT[c * ENC_LDP + i] = o[c];
                if (feat_out) {
#pragma unroll
                    for (int c = 0; c < 9; ++c) feat_out[idx * 9 + c] = o[c];
                }
            }
        }
        __syncthreads();
        // ---- phase 2: h1 = lrelu(feat W1^T + b1)
        {
            f32x16 acc[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][r] = b1v;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const float* ap = fT + (2 * ks + hi) * ENC_LDP + lo;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[rb * 32], w1r[0][ks], acc[rb], 0, 0, 0);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (row0 + rb * 32 >= R) break;        // uniform: R is a multiple of 32
                float* hp = hT + col * ENC_LDP + rb * 32 + 4 * hi;
                float* gp = h1_out ? h1_out + (size_t)(row0 + rb * 32 + 4 * hi) * H + col : nullptr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mr = (r & 3) + 8 * (r >> 2);
                    const float v = acc[rb][r] > 0.f ? acc[rb][r] : slope * acc[rb][r];
                    hp[mr] = v;
                    if (gp) gp[(size_t)mr * H] = v;
                }
            }
        }
        __syncthreads();
        // ---- phase 3: e = lrelu(h1 W2^T + b2)
        {
            f32x16 acc[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][r] = b2v;
            const float* ap = hT + hi * ENC_LDP + lo;
            float a[2][RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[0][rb] = ap[rb * 32];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a[(ks + 1) & 1][rb] = ap[2 * (ks + 1) * ENC_LDP + rb * 32];
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks & 1][rb], w2r[0][ks], acc[rb], 0, 0, 0);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (row0 + rb * 32 >= R) break;
                float* gp = e_out + (size_t)(row0 + rb * 32 + 4 * hi) * H + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mr = (r & 3) + 8 * (r >> 2);
                    const float v = acc[rb][r];
                    gp[(size_t)mr * H] = v > 0.f ? v : slope * v;
                }
            }
        }
        // (no barrier here: the next tile's phase 1 writes fT only, which nobody reads after the barrier above;
        //  its phase 2 writes hT after the next barrier, which every wave reaches after its phase 3)
    }
}

template <int NW>
int launch_enc_fwd(const float* pos, const float* dts, const float* w1_pk, const float* b1, const float* w2_pk,
                   const float* b2, float* feat, float* h1, float* e, int L, int B, float eps, float slope,
                   hipStream_t stream)
{
    constexpr int H = NW * 32;
    const size_t lds = (size_t)(16 + H) * ENC_LDP * sizeof(float);
    if (lds > 160 * 1024 - 256) return GMT_ERR_UNSUPPORTED;
    auto k = encoder_fwd_kernel<NW>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GMT_ERR_HIP;
    const long long R = (long long)L * B;
    const long long n_tiles = (R + ENC_RT - 1) / ENC_RT;
    const int per_cu = (2 * lds <= 150 * 1024) ? 2 : 1;
    long long grid = 256LL * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NW * 64), lds, stream, pos, dts, w1_pk, b1, w2_pk, b2, feat, h1, e,
                       L, B, eps, slope, R, n_tiles);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

}  // namespace

extern "C" int gmt_encoder_fwd(const float* pos, const float* dts, const float* w1_pk, const float* b1,
                               const float* w2_pk, const float* b2, float* feat, float* h1, float* e,
                               int L, int B, int H, float eps, float slope, hipStream_t stream)
{
    GMT_CHECK_ARG(pos && dts && w1_pk && b1 && w2_pk && b2 && e && L >= 1 && B >= GMT_TM && (B % GMT_TM) == 0);
    switch (H) {
        case 32: return launch_enc_fwd<1>(pos, dts, w1_pk, b1, w2_pk, b2, feat, h1, e, L, B, eps, slope, stream);
        case 64: return launch_enc_fwd<2>(pos, dts, w1_pk, b1, w2_pk, b2, feat, h1, e, L, B, eps, slope, stream);
        case 128: return launch_enc_fwd<4>(pos, dts, w1_pk, b1, w2_pk, b2, feat, h1, e, L, B, eps, slope, stream);
        case 256: return launch_enc_fwd<8>(pos, dts, w1_pk, b1, w2_pk, b2, feat, h1, e, L, B, eps, slope, stream);
        default: return GMT_ERR_UNSUPPORTED;
    }
}
