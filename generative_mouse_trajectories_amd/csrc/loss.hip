// Per-trajectory bookkeeping around the two networks, one thread per trajectory, fp32:
//   * gmt_traj_from_seq / _bwd: trajectory_to_absolute (RG.py:392-417: pos[0] = start, pos[t+1] = start +
//     cumsum(deltas)[t]) fused with the dt padding F.pad(dt, (1, 0), 0.001) of RG.py:934 / 949 / 1061 and with the
//     transposition into the kernels' time-major layout; its backward is the reverse cumulative sum;
//   * gmt_gen_loss_fwd / _bwd: the generator's auxiliary losses (RG.py:1072-1095): endpoint squared error at each
//     row's own length, the masked direction-cosine terms and the mask count, plus the critic-score sum of
//     RG.py:1069 - all reductions of the step in one kernel, wavefront-shuffle sums per block; the block partials
//     are added in a fixed order (deterministic) by a one-wave second launch (the kernel boundary is the
//     cross-XCD hand-off), results in one 4-float buffer.
// The per-row cumulative sum runs sequentially in double and is rounded to fp32 per step: exactly what
// torch.cumsum does for float tensors on the CPU (the reference's device), so positions are bit-identical to it.
#include "gmt_common.h"

namespace {

constexpr int LB = 256;

__global__ __launch_bounds__(LB)
void traj_from_seq_kernel(const float* __restrict__ seq, long long sb, long long st, const float* __restrict__ starts,
                          float* __restrict__ pos, float* __restrict__ dts, int T, int B, int n_valid, float first_dt)
{
    const int b = blockIdx.x * LB + threadIdx.x;
    if (b >= B) return;
    float2* p2 = reinterpret_cast<float2*>(pos);
    if (b >= n_valid) {                                 // batch padding
        for (int t = 0; t <= T; ++t) { p2[(size_t)t * B + b] = make_float2(0.f, 0.f); dts[(size_t)t * B + b] = 0.f; }
        return;
    }
    const float sx = starts[2 * b], sy = starts[2 * b + 1];
    double cx = 0.0, cy = 0.0;
    p2[b] = make_float2(sx, sy);
    dts[b] = first_dt;
    const float* s = seq + (size_t)b * sb;
    // one thread walks a whole trajectory: the loads do not depend on the running sums, so eight steps' worth are
    // requested at once (the sums stay sequential: same bits) - the loop was one memory latency per step, 80-120 us
    constexpr int U = 8;
    int t = 0;
    for (; t + U <= T; t += U) {
        float vx[U], vy[U], vt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { vx[u] = s[(size_t)(t + u) * st]; vy[u] = s[(size_t)(t + u) * st + 1]; vt[u] = s[(size_t)(t + u) * st + 2]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cx += (double)vx[u]; cy += (double)vy[u];
            p2[(size_t)(t + u + 1) * B + b] = make_float2(sx + (float)cx, sy + (float)cy);
            dts[(size_t)(t + u + 1) * B + b] = vt[u];
        }
    }
    for (; t < T; ++t) {
        const float dx = s[(size_t)t * st], dy = s[(size_t)t * st + 1], dt = s[(size_t)t * st + 2];
        cx += (double)dx; cy += (double)dy;
        p2[(size_t)(t + 1) * B + b] = make_float2(sx + (float)cx, sy + (float)cy);
        dts[(size_t)(t + 1) * B + b] = dt;
    }
}

// gseq[b][t] = (sum_{s > t} gpos[s][b], gdts[t+1][b]); rows >= n_valid get zeros
__global__ __launch_bounds__(LB)
void traj_from_seq_bwd_kernel(const float* __restrict__ gpos, const float* __restrict__ gdts, float* __restrict__ gseq,
                              long long sb, long long st, int T, int B, int n_valid, int n_rows_out)
{
    const int b = blockIdx.x * LB + threadIdx.x;
    if (b >= B || b >= n_rows_out) return;
    const float2* g2 = reinterpret_cast<const float2*>(gpos);
    float* o = gseq + (size_t)b * sb;
    float ax = 0.f, ay = 0.f;
    const bool live = b < n_valid;
    constexpr int U = 8;
    int t = T - 1;
    for (; t - U + 1 >= 0; t -= U) {                     // eight steps' loads in flight, sums in the same order
        float2 g[U]; float gd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { g[u] = g2[(size_t)(t - u + 1) * B + b]; gd[u] = gdts[(size_t)(t - u + 1) * B + b]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ax += g[u].x; ay += g[u].y;
            o[(size_t)(t - u) * st] = live ? ax : 0.f;
            o[(size_t)(t - u) * st + 1] = live ? ay : 0.f;
            o[(size_t)(t - u) * st + 2] = live ? gd[u] : 0.f;
        }
    }
    for (; t >= 0; --t) {
        const float2 g = g2[(size_t)(t + 1) * B + b];
        ax += g.x; ay += g.y;
        o[(size_t)t * st] = live ? ax : 0.f;
        o[(size_t)t * st + 1] = live ? ay : 0.f;
        o[(size_t)t * st + 2] = live ? gdts[(size_t)(t + 1) * B + b] : 0.f;
    }
}

// Generated (dx, dy, dt) sequences -> the telemetry kernels' input planes: x, y in pixels and the raw time between
// movements, f64, [T+1][n] (point 0 = the start point, dt 0) - the move from the generator's normalised fp32 output
// to what MT.rs:550-654 consumes (BASELINE config 5: sampling + metric recompute).  Cumulative sum in f64.
__global__ __launch_bounds__(LB)
void rollout_to_xyt_kernel(const float* __restrict__ seq, long long sb, long long st, const float* __restrict__ starts,
                           double* __restrict__ x, double* __restrict__ y, double* __restrict__ dt, int T, long long n,
                           double scale_x, double scale_y)
{
    const long long b = (long long)blockIdx.x * LB + threadIdx.x;
    if (b >= n) return;
    double cx = (double)starts[2 * b], cy = (double)starts[2 * b + 1];
    x[b] = cx * scale_x; y[b] = cy * scale_y; dt[b] = 0.0;
    const float* s = seq + (size_t)b * sb;
    for (int t = 0; t < T; ++t) {
        cx += (double)s[(size_t)t * st]; cy += (double)s[(size_t)t * st + 1];
        const size_t o = (size_t)(t + 1) * n + b;
        x[o] = cx * scale_x; y[o] = cy * scale_y; dt[o] = (double)s[(size_t)t * st + 2];
    }
}

struct DirTerm { float tx, ty; };

__device__ __forceinline__ DirTerm target_dir(float sx, float sy, float ex, float ey)
{
    const float dx = ex - sx, dy = ey - sy;
    const float n = sqrtf(dx * dx + dy * dy) + 1e-8f;                // RG.py:1078: eps outside the square root
    return {dx / n, dy / n};
}

// scal[0] = sum_b |final_b - end_b|^2, scal[1] = sum (1 - cos) mask, scal[2] = sum mask, scal[3] = sum_b score_b
__global__ __launch_bounds__(LB)
void gen_loss_fwd_kernel(const float* __restrict__ seq, long long sb, long long st, const float* __restrict__ starts,
                         const float* __restrict__ ends, const int* __restrict__ lengths,
                         const float* __restrict__ score, float* __restrict__ final_pos,
                         float* __restrict__ partial, int T, int n_valid)
{
    __shared__ float red[LB / 64][4];
    const int b = blockIdx.x * LB + threadIdx.x;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (b < n_valid) {
        const float sx = starts[2 * b], sy = starts[2 * b + 1], ex = ends[2 * b], ey = ends[2 * b + 1];
        const DirTerm td = target_dir(sx, sy, ex, ey);
        int len = lengths[b];
        len = len < 0 ? 0 : (len > T ? T : len);
        const float* s = seq + (size_t)b * sb;
        double cx = 0.0, cy = 0.0;
        float dsum = 0.f;
        constexpr int U = 8;
        int t = 0;
        for (; t + U <= len; t += U) {                   // (loads of eight steps in flight; sums in step order)
            float vx[U], vy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { vx[u] = s[(size_t)(t + u) * st]; vy[u] = s[(size_t)(t + u) * st + 1]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float dx = vx[u], dy = vy[u];
                cx += (double)dx; cy += (double)dy;
                const float n = sqrtf(dx * dx + dy * dy + 1e-8f);
                dsum += 1.0f - ((dx / n) * td.tx + (dy / n) * td.ty);
            }
        }
        for (; t < len; ++t) {
            const float dx = s[(size_t)t * st], dy = s[(size_t)t * st + 1];
            cx += (double)dx; cy += (double)dy;
            const float n = sqrtf(dx * dx + dy * dy + 1e-8f);           // RG.py:1081: eps inside
            dsum += 1.0f - ((dx / n) * td.tx + (dy / n) * td.ty);       // RG.py:1082-1095
        }
        const float fx = sx + (float)cx, fy = sy + (float)cy;          // fake_traj[b, len] (RG.py:1072)
        final_pos[2 * b] = fx; final_pos[2 * b + 1] = fy;
        v[0] = (fx - ex) * (fx - ex) + (fy - ey) * (fy - ey);
        v[1] = dsum;
        v[2] = (float)len;
        v[3] = score ? score[b] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = gmt_wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < LB / 64; ++w) s += red[w][threadIdx.x];
        partial[(size_t)blockIdx.x * 4 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(64)
void gen_loss_finish_kernel(const float* __restrict__ partial, float* __restrict__ scal, int nb)
{
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int k = 0; k < nb; ++k) s += partial[(size_t)k * 4 + threadIdx.x];
        scal[threadIdx.x] = s;
    }
}

// gseq[b][t] = [t < len_b] * (gscal[0] * 2 (final_b - end_b) + gscal[1] * d(1 - cos)/d(delta_t)), dt column 0
__global__ __launch_bounds__(LB)
void gen_loss_bwd_kernel(const float* __restrict__ seq, long long sb, long long st, const float* __restrict__ starts,
                         const float* __restrict__ ends, const int* __restrict__ lengths,
                         const float* __restrict__ final_pos, const float* __restrict__ gscal,
                         float* __restrict__ gseq, long long gsb, long long gst, int T, int B, int n_valid,
                         const float* __restrict__ gtot, float* __restrict__ d_score)
{
    // gtot != null (gmt_gen_step_loss_bwd): gscal holds the three coefficients of gen_step_finish_kernel, to be scaled
    // by the adjoint of the total loss; d_score[b] = gtot * gscal[2] on the valid rows, 0 on the batch padding
    const int b = blockIdx.x * LB + threadIdx.x;
    if (b >= B) return;
    const float gt = gtot ? gtot[0] : 1.0f;
    if (d_score) d_score[b] = b < n_valid ? gt * gscal[2] : 0.f;
    float* o = gseq + (size_t)b * gsb;
    int len = 0;
    float ge_x = 0.f, ge_y = 0.f, gd = 0.f;
    DirTerm td = {0.f, 0.f};
    if (b < n_valid) {
        len = lengths[b];
        len = len < 0 ? 0 : (len > T ? T : len);
        const float g0 = gt * gscal[0];
        gd = gt * gscal[1];
        ge_x = 2.f * g0 * (final_pos[2 * b] - ends[2 * b]);
        ge_y = 2.f * g0 * (final_pos[2 * b + 1] - ends[2 * b + 1]);
        td = target_dir(starts[2 * b], starts[2 * b + 1], ends[2 * b], ends[2 * b + 1]);
    }
    const float* s = seq + (size_t)b * sb;
    auto one = [&](int t, float dx, float dy) {
        float ox = 0.f, oy = 0.f;
        if (t < len) {
            const float n = sqrtf(dx * dx + dy * dy + 1e-8f);
            const float c = (dx / n) * td.tx + (dy / n) * td.ty;        // cosine
            // d(1 - cos)/d(dx) = -(tx / n - cos dx / n^2)
            ox = ge_x - gd * (td.tx / n - c * dx / (n * n));
            oy = ge_y - gd * (td.ty / n - c * dy / (n * n));
        }
        o[(size_t)t * gst] = ox; o[(size_t)t * gst + 1] = oy; o[(size_t)t * gst + 2] = 0.f;
    };
    constexpr int U = 8;
    int t = 0;
    for (; t + U <= T; t += U) {                         // eight steps' loads in flight (steps are independent here)
        float vx[U], vy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { vx[u] = s[(size_t)(t + u) * st]; vy[u] = s[(size_t)(t + u) * st + 1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) one(t + u, vx[u], vy[u]);
    }
    for (; t < T; ++t) one(t, s[(size_t)t * st], s[(size_t)t * st + 1]);
}

// The scalar arithmetic of train_generator_step after the reductions (RG.py:1069, 1072-1097), one thread:
//   out[0] = -mean(score) (g_loss), out[1] = scal0 / (2B) (endpoint, F.mse_loss over (B, 2)), out[2] = num / den (direction;
//   nd != null: the numerator / denominator summed over the ranks of a data-parallel run), out[3] = mean(score),
//   total[0] = g_loss + w_e endpoint + w_d direction,
//   coef[0..2] = d total / d scal0, d total / d (local) num, d total / d score_b  - what gen_loss_bwd_kernel multiplies
//   by the adjoint of total (world / den: the ranks' gradients are averaged afterwards, see Trainer).
__global__ __launch_bounds__(64)
void gen_step_finish_kernel(const float* __restrict__ scal, const float* __restrict__ nd, float world, int B,
                            float w_e, float w_d, float* __restrict__ total, float* __restrict__ out,
                            float* __restrict__ coef)
{
    if (threadIdx.x == 0) {
        const float fB = (float)B;
        const float mean_s = scal[3] / fB;
        const float g_loss = -mean_s;
        const float endpoint = scal[0] / (2.0f * fB);
        const float num = nd ? nd[0] : scal[1], den = nd ? nd[1] : scal[2];
        const float direction = num / den;
        out[0] = g_loss; out[1] = endpoint; out[2] = direction; out[3] = mean_s;
        total[0] = g_loss + w_e * endpoint + w_d * direction;
        coef[0] = w_e / (2.0f * fB);
        coef[1] = w_d * (world / den);
        coef[2] = -1.0f / fB;
    }
}

// Critic loss of train_discriminator_step (RG.py:963): d_loss = mean(fake) - mean(real) over the B valid rows of the two
// padded score vectors; out = (mean real, mean fake).  One block; fixed summation order.
__global__ __launch_bounds__(LB)
void critic_loss_fwd_kernel(const float* __restrict__ real, const float* __restrict__ fake, int B,
                            float* __restrict__ loss, float* __restrict__ out)
{
    __shared__ float red[LB / 64][2];
    float sr = 0.f, sf = 0.f;
    for (int b = threadIdx.x; b < B; b += LB) { sr += real[b]; sf += fake[b]; }
    sr = gmt_wave_sum(sr); sf = gmt_wave_sum(sf);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sr; red[threadIdx.x >> 6][1] = sf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tr = 0.f, tf = 0.f;
#pragma unroll
        for (int w = 0; w < LB / 64; ++w) { tr += red[w][0]; tf += red[w][1]; }
        const float mr = tr / (float)B, mf = tf / (float)B;
        out[0] = mr; out[1] = mf;
        loss[0] = mf - mr;
    }
}

__global__ __launch_bounds__(LB)
void critic_loss_bwd_kernel(const float* __restrict__ g, int B, int Bp, float* __restrict__ d_real, float* __restrict__ d_fake)
{
    const int b = blockIdx.x * LB + threadIdx.x;
    if (b >= Bp) return;
    const float v = b < B ? g[0] / (float)B : 0.f;
    d_real[b] = -v;
    d_fake[b] = v;
}

}  // namespace

extern "C" int gmt_critic_loss_fwd(const float* real, const float* fake, int B, float* loss, float* out, hipStream_t stream)
{
    GMT_CHECK_ARG(real && fake && loss && out && B >= 1);
    hipLaunchKernelGGL(critic_loss_fwd_kernel, dim3(1), dim3(LB), 0, stream, real, fake, B, loss, out);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_critic_loss_bwd(const float* g, int B, int Bp, float* d_real, float* d_fake, hipStream_t stream)
{
    GMT_CHECK_ARG(g && d_real && d_fake && B >= 1 && Bp >= B);
    hipLaunchKernelGGL(critic_loss_bwd_kernel, dim3((Bp + LB - 1) / LB), dim3(LB), 0, stream, g, B, Bp, d_real, d_fake);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_gen_step_loss_finish(const float* scal, const float* nd, float world, int n_valid, float w_endpoint,
                                        float w_direction, float* total, float* out, float* coef, hipStream_t stream)
{
    GMT_CHECK_ARG(scal && total && out && coef && n_valid >= 1 && world >= 1.0f);
    hipLaunchKernelGGL(gen_step_finish_kernel, dim3(1), dim3(64), 0, stream, scal, nd, world, n_valid, w_endpoint,
                       w_direction, total, out, coef);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_gen_step_loss_bwd(const float* seq, long long stride_b, long long stride_t, const float* starts,
                                     const float* ends, const int* lengths, const float* final_pos, const float* coef,
                                     const float* g_total, float* gseq, long long gstride_b, long long gstride_t,
                                     float* d_score, int T, int B, int n_valid, hipStream_t stream)
{
    GMT_CHECK_ARG(seq && starts && ends && lengths && final_pos && coef && g_total && gseq && d_score && T >= 1 && B >= 1 &&
                  n_valid >= 0 && n_valid <= B);
    hipLaunchKernelGGL(gen_loss_bwd_kernel, dim3((B + LB - 1) / LB), dim3(LB), 0, stream, seq, stride_b, stride_t, starts,
                       ends, lengths, final_pos, coef, gseq, gstride_b, gstride_t, T, B, n_valid, g_total, d_score);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_traj_from_seq(const float* seq, long long stride_b, long long stride_t, const float* starts,
                                 float* pos, float* dts, int T, int B, int n_valid, float first_dt, hipStream_t stream)
{
    GMT_CHECK_ARG(seq && starts && pos && dts && T >= 1 && B >= 1 && n_valid >= 0 && n_valid <= B);
    hipLaunchKernelGGL(traj_from_seq_kernel, dim3((B + LB - 1) / LB), dim3(LB), 0, stream, seq, stride_b, stride_t, starts,
                       pos, dts, T, B, n_valid, first_dt);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_traj_from_seq_bwd(const float* gpos, const float* gdts, float* gseq, long long stride_b,
                                     long long stride_t, int T, int B, int n_valid, int n_rows_out, hipStream_t stream)
{
    GMT_CHECK_ARG(gpos && gdts && gseq && T >= 1 && B >= 1 && n_valid >= 0 && n_valid <= B && n_rows_out >= n_valid &&
                  n_rows_out <= B);
    hipLaunchKernelGGL(traj_from_seq_bwd_kernel, dim3((B + LB - 1) / LB), dim3(LB), 0, stream, gpos, gdts, gseq, stride_b,
                       stride_t, T, B, n_valid, n_rows_out);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_rollout_to_xyt_f64(const float* seq, long long stride_b, long long stride_t, const float* starts,
                                      double* x, double* y, double* dt, int T, long long n, double scale_x,
                                      double scale_y, hipStream_t stream)
{
    GMT_CHECK_ARG(seq && starts && x && y && dt && T >= 1 && n >= 1);
    hipLaunchKernelGGL(rollout_to_xyt_kernel, dim3((unsigned)((n + LB - 1) / LB)), dim3(LB), 0, stream, seq, stride_b,
                       stride_t, starts, x, y, dt, T, n, scale_x, scale_y);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_gen_loss_workspace_floats(int n_valid, long long* floats_out)
{
    GMT_CHECK_ARG(floats_out && n_valid >= 1);
    *floats_out = 4LL * ((n_valid + LB - 1) / LB);          // block partials
    return GMT_OK;
}

extern "C" int gmt_gen_loss_fwd(const float* seq, long long stride_b, long long stride_t, const float* starts,
                                const float* ends, const int* lengths, const float* score, float* final_pos,
                                float* scal, float* workspace, int T, int n_valid, hipStream_t stream)
{
    GMT_CHECK_ARG(seq && starts && ends && lengths && final_pos && scal && workspace && T >= 1 && n_valid >= 1);
    const int nb = (n_valid + LB - 1) / LB;
    hipLaunchKernelGGL(gen_loss_fwd_kernel, dim3(nb), dim3(LB), 0, stream, seq, stride_b, stride_t, starts, ends, lengths,
                       score, final_pos, workspace, T, n_valid);
    GMT_LAUNCH_CHECK();
    hipLaunchKernelGGL(gen_loss_finish_kernel, dim3(1), dim3(64), 0, stream, workspace, scal, nb);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_gen_loss_bwd(const float* seq, long long stride_b, long long stride_t, const float* starts,
                                const float* ends, const int* lengths, const float* final_pos, const float* gscal,
                                float* gseq, long long gstride_b, long long gstride_t, int T, int B, int n_valid,
                                hipStream_t stream)
{
    GMT_CHECK_ARG(seq && starts && ends && lengths && final_pos && gscal && gseq && T >= 1 && B >= 1 && n_valid >= 0 &&
                  n_valid <= B);
    hipLaunchKernelGGL(gen_loss_bwd_kernel, dim3((B + LB - 1) / LB), dim3(LB), 0, stream, seq, stride_b, stride_t, starts,
                       ends, lengths, final_pos, gscal, gseq, gstride_b, gstride_t, T, B, n_valid, nullptr, nullptr);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
