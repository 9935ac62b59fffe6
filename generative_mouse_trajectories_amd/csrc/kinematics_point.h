// Per-point bodies of the kinematics kernels (RecurrentGAN.py:315-389, "mode A").
// Shared by kinematics.hip (device) and by the host-side simulation build that the CPU
// tests use to check the arithmetic without a GPU (tests/hostsim).
#pragma once
#include <math.h>
#ifndef GMT_HD
#  ifdef __HIPCC__
#    define GMT_HD __host__ __device__ __forceinline__
#  else
#    define GMT_HD static inline
#  endif
#endif

#ifdef __HIPCC__
#  define GMT_HDM __host__ __device__ __forceinline__
#else
#  define GMT_HDM inline
#endif

// a / b: IEEE division on the host (reference arithmetic of the CPU checks); on the device one v_rcp_f32 and a
// multiply (<= 1.5 ulp; the parity budget of these features is 2e-6 relative), which takes the kernels from
// VALU-bound to HBM-bound
#if defined(__HIP_DEVICE_COMPILE__)
#  define GMT_DIV(a, b) __fdividef((a), (b))
#else
#  define GMT_DIV(a, b) ((a) / (b))
#endif

namespace gmt_kin {

struct F2 { float x, y; };

struct Pt {            // quantities of point t that need points t and t-1 only
    float dx, dy, td, disp, vel, vx, vy;
};

GMT_HD Pt make_pt(float x, float y, float xp, float yp, float d, bool has_prev, float eps) {
    Pt p;
    p.dx = has_prev ? x - xp : 0.0f;                               // RG.py:335-338
    p.dy = has_prev ? y - yp : 0.0f;
    p.td = d + eps;
    p.disp = sqrtf(p.dx * p.dx + p.dy * p.dy + eps);               // RG.py:341
    p.vel = GMT_DIV(p.disp, p.td);                                         // RG.py:342
    p.vx = GMT_DIV(p.dx, p.td);                                            // RG.py:345-346
    p.vy = GMT_DIV(p.dy, p.td);
    return p;
}

GMT_HD void kin_fwd_vals(long long idx, const float* pos, const float* dts, float (&o)[9], int L, int B, float eps)
{
    const int t = (int)(idx / B);
    // window t-3 .. t (clamped loads; the has_prev flags zero what does not exist)
    float x[4], y[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int tt = t - 3 + k;
        long long j = idx - (long long)(3 - k) * B;
        if (tt < 0) j = idx - (long long)t * B;
        const F2 p = reinterpret_cast<const F2*>(pos)[j];
        x[k] = p.x; y[k] = p.y; d[k] = dts[j];
    }
    Pt p0 = make_pt(x[3], y[3], x[2], y[2], d[3], t >= 1, eps);       // t
    Pt p1 = make_pt(x[2], y[2], x[1], y[1], d[2], t >= 2, eps);       // t-1
    Pt p2 = make_pt(x[1], y[1], x[0], y[0], d[1], t >= 3, eps);       // t-2
    float acc = 0.f, accp = 0.f, jerk = 0.f, ax = 0.f, ay = 0.f;
    if (t >= 1) {
        float m = (d[3] + d[2]) * 0.5f + eps;                         // RG.py:351
        acc = GMT_DIV(p0.vel - p1.vel, m);                               // RG.py:352
        ax = GMT_DIV(p0.vx - p1.vx, m);                                   // RG.py:363-364
        ay = GMT_DIV(p0.vy - p1.vy, m);
        if (t >= 2) {
            float mp = (d[2] + d[1]) * 0.5f + eps;
            accp = GMT_DIV(p1.vel - p2.vel, mp);
            jerk = GMT_DIV(acc - accp, m + eps);                       // RG.py:356-357
        }
    }
    float u = p0.vx * p0.vx + p0.vy * p0.vy + eps;
    float curv = GMT_DIV(fabsf(p0.vx * ay - p0.vy * ax), u * sqrtf(u));     // RG.py:366-368
    o[0] = x[3]; o[1] = y[3]; o[2] = p0.dx; o[3] = p0.dy; o[4] = d[3];
    o[5] = fminf(fmaxf(p0.vel, 0.f), 100.f);                          // RG.py:371-374
    o[6] = fminf(fmaxf(acc, -1000.f), 1000.f);
    o[7] = fminf(fmaxf(jerk, -10000.f), 10000.f);
    o[8] = fminf(fmaxf(curv, 0.f), 100.f);
}

GMT_HD void kin_fwd_point(long long idx, const float* pos, const float* dts, float* feat, int L, int B, float eps)
{
    float o[9];
    kin_fwd_vals(idx, pos, dts, o, L, B, eps);
    for (int c = 0; c < 9; ++c) feat[idx * 9 + c] = o[c];
}

// Backward for point (s, b): gathers d(loss)/d(pos_s), d(loss)/d(dts_s) from the feature
// gradients of points s .. s+3, recomputing the forward window s-3 .. s+3.
// `grad(k, c)` returns d(loss)/d(feat[point s+k][c]) for k = 0..3 with s+k <= L-1 (the kernel serves it from
// LDS, the host simulation from memory); results: (gx, gy) and gd.
template <class Grad>
GMT_HD void kin_bwd_vals(long long idx, const float* pos, const float* dts, Grad grad,
                         float& gx_out, float& gy_out, float& gd_out, int L, int B, float eps)
{
    const int s = (int)(idx / B);
    // points s-3 .. s+3 -> local index k = tt - (s-3)
    float x[7], y[7], d[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        int tt = s - 3 + k;
        tt = tt < 0 ? 0 : (tt > L - 1 ? L - 1 : tt);
        long long j = idx + (long long)(tt - s) * B;
        const F2 p = reinterpret_cast<const F2*>(pos)[j];
        x[k] = p.x; y[k] = p.y; d[k] = dts[j];
    }
    // Pt for tt = s-2 .. s+3 -> P[k], tt = s-2+k
    Pt P[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int tt = s - 2 + k;
        P[k] = make_pt(x[k + 1], y[k + 1], x[k], y[k], d[k + 1], tt >= 1, eps);
    }
    // m, acc, ax, ay for tt = s-1 .. s+3 -> index k, tt = s-1+k  (valid when 1 <= tt <= L-1)
    float m[5], acc[5], axv[5], ayv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int tt = s - 1 + k;
        bool ok = (tt >= 1) && (tt <= L - 1);
        m[k] = (d[k + 2] + d[k + 1]) * 0.5f + eps;
        acc[k] = ok ? GMT_DIV(P[k + 1].vel - P[k].vel, m[k]) : 0.f;
        axv[k] = ok ? GMT_DIV(P[k + 1].vx - P[k].vx, m[k]) : 0.f;
        ayv[k] = ok ? GMT_DIV(P[k + 1].vy - P[k].vy, m[k]) : 0.f;
    }
    // feature gradients of tt = s .. s+3 (zero outside the sequence)
    float g[4][9];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int tt = s + k;
        bool ok = tt <= L - 1;
#pragma unroll
        for (int c = 0; c < 9; ++c) g[k][c] = ok ? grad(k, c) : 0.f;
    }
    // jerk adjoint Jb[k], tt = s+k (k = 0..3); jerk value jv[k]
    float Jb[4], jv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int tt = s + k;
        bool ok = (tt >= 2) && (tt <= L - 1);
        float j = ok ? GMT_DIV(acc[k + 1] - acc[k], m[k + 1] + eps) : 0.f;
        jv[k] = j;
        Jb[k] = (ok && j >= -10000.f && j <= 10000.f) ? g[k][7] : 0.f;
    }
    // acc adjoint Ab[k], tt = s+k (k = 0..2)
    float Ab[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int tt = s + k;
        bool ok = (tt >= 1) && (tt <= L - 1);
        float a = acc[k + 1];
        float v = (a >= -1000.f && a <= 1000.f) ? g[k][6] : 0.f;
        v += GMT_DIV(Jb[k], m[k + 1] + eps);
        v -= GMT_DIV(Jb[k + 1], m[k + 2] + eps);     // Jb is zero when tt+1 is outside
        Ab[k] = ok ? v : 0.f;
    }
    // curvature adjoints at tt = s+k (k = 0..2): qb, ub, axb, ayb
    float qb[3], ub[3], axb[3], ayb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int tt = s + k;
        bool ok = tt <= L - 1;
        const Pt& p = P[k + 2];
        float ax = axv[k + 1], ay = ayv[k + 1];
        float q = p.vx * ay - p.vy * ax;
        float u = p.vx * p.vx + p.vy * p.vy + eps;
        float su = sqrtf(u);
        float s3 = u * su;
        float cross = fabsf(q);
        float curv = GMT_DIV(cross, s3);
        float Cb = (ok && curv >= 0.f && curv <= 100.f) ? g[k][8] : 0.f;
        float sgn = q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f);
        qb[k] = GMT_DIV(Cb * sgn, s3);
        ub[k] = GMT_DIV(-Cb * curv * 1.5f, u);         // d(cross/u^1.5)/du = -1.5 * curv / u
        bool okA = ok && tt >= 1;
        axb[k] = okA ? -qb[k] * p.vy : 0.f;
        ayb[k] = okA ? qb[k] * p.vx : 0.f;
    }
    // per tt = s+k (k = 0..1): Vb, VXb, VYb, mb, then dxb, dyb, tdb
    float dxb[2], dyb[2], mb[2], tdb0 = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int tt = s + k;
        bool ok = tt <= L - 1;
        const Pt& p = P[k + 2];
        float ax = axv[k + 1], ay = ayv[k + 1];
        float Vb = (p.vel >= 0.f && p.vel <= 100.f) ? g[k][5] : 0.f;
        Vb += GMT_DIV(Ab[k], m[k + 1]);               // Ab is zero for tt < 1
        Vb -= GMT_DIV(Ab[k + 1], m[k + 2]);            // zero when tt+1 outside
        float VXb = qb[k] * ay + 2.f * ub[k] * p.vx + GMT_DIV(axb[k], m[k + 1]) - GMT_DIV(axb[k + 1], m[k + 2]);
        float VYb = -qb[k] * ax + 2.f * ub[k] * p.vy + GMT_DIV(ayb[k], m[k + 1]) - GMT_DIV(ayb[k + 1], m[k + 2]);
        float dispb = GMT_DIV(Vb, p.td);
        float tdb = GMT_DIV(-Vb * p.vel, p.td) - GMT_DIV(VXb * p.vx, p.td) - GMT_DIV(VYb * p.vy, p.td);
        dxb[k] = ok ? (g[k][2] + GMT_DIV(VXb, p.td) + GMT_DIV(dispb * p.dx, p.disp)) : 0.f;
        dyb[k] = ok ? (g[k][3] + GMT_DIV(VYb, p.td) + GMT_DIV(dispb * p.dy, p.disp)) : 0.f;
        bool okM = ok && tt >= 1;
        mb[k] = okM ? (GMT_DIV(-Ab[k] * acc[k + 1], m[k + 1]) - GMT_DIV(Jb[k] * jv[k], m[k + 1] + eps)
                       - GMT_DIV(axb[k] * ax, m[k + 1]) - GMT_DIV(ayb[k] * ay, m[k + 1])) : 0.f;
        if (k == 0) tdb0 = tdb;
    }
    float gx = g[0][0] + (s >= 1 ? dxb[0] : 0.f) - dxb[1];
    float gy = g[0][1] + (s >= 1 ? dyb[0] : 0.f) - dyb[1];
    float gd = g[0][4] + tdb0 + 0.5f * (mb[0] + mb[1]);
    gx_out = gx; gy_out = gy; gd_out = gd;
}

// ------------------------------------------------------------------------------------------------------------
// Forward-mode derivative of the features w.r.t. the positions: fhat[t] = J_t(pos, dts) . phat  (dts fixed).
// This is the kinematics piece of the gradient penalty's SECOND backward (RG.py:829-835: autograd.grad with
// create_graph=True differentiates the first backward, gpos = J^T gfeat, once more; the adjoint of gfeat in
// that map is J phat with phat = the adjoint of gpos).  The interpolated positions are a leaf that no
// parameter feeds, so the adjoint of pos itself (the Hessian term) is never needed.  Same formulas as
// kin_fwd_vals in dual numbers; a clamp passes a tangent exactly where kin_bwd_vals passes a gradient
// (value inside the closed interval).
// ------------------------------------------------------------------------------------------------------------
struct Du { float v, d; };
GMT_HD Du du_add(Du a, Du b) { return Du{a.v + b.v, a.d + b.d}; }
GMT_HD Du du_sub(Du a, Du b) { return Du{a.v - b.v, a.d - b.d}; }
GMT_HD Du du_mul(Du a, Du b) { return Du{a.v * b.v, a.d * b.v + a.v * b.d}; }
GMT_HD Du du_addf(Du a, float c) { return Du{a.v + c, a.d}; }
GMT_HD Du du_divf(Du a, float c) { return Du{GMT_DIV(a.v, c), GMT_DIV(a.d, c)}; }
GMT_HD Du du_div(Du a, Du b) { const float q = GMT_DIV(a.v, b.v); return Du{q, GMT_DIV(a.d - q * b.d, b.v)}; }
GMT_HD Du du_sqrt(Du a) { const float r = sqrtf(a.v); return Du{r, GMT_DIV(0.5f * a.d, r)}; }
GMT_HD float du_clamp_tan(Du a, float lo, float hi) { return (a.v >= lo && a.v <= hi) ? a.d : 0.0f; }

struct PtD { Du dx, dy, vel, vx, vy; };

GMT_HD PtD make_ptd(Du x, Du y, Du xp, Du yp, float d, bool has_prev, float eps) {
    PtD p;
    const Du z{0.0f, 0.0f};
    p.dx = has_prev ? du_sub(x, xp) : z;
    p.dy = has_prev ? du_sub(y, yp) : z;
    const float td = d + eps;
    const Du disp = du_sqrt(du_addf(du_add(du_mul(p.dx, p.dx), du_mul(p.dy, p.dy)), eps));
    p.vel = du_divf(disp, td);
    p.vx = du_divf(p.dx, td);
    p.vy = du_divf(p.dy, td);
    return p;
}

GMT_HD void kin_jvp_vals(long long idx, const float* pos, const float* dts, const float* phat, float (&o)[9],
                         int L, int B, float eps)
{
    const int t = (int)(idx / B);
    Du x[4], y[4];
    float d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int tt = t - 3 + k;
        long long j = idx - (long long)(3 - k) * B;
        if (tt < 0) j = idx - (long long)t * B;
        const F2 p = reinterpret_cast<const F2*>(pos)[j];
        const F2 h = reinterpret_cast<const F2*>(phat)[j];
        x[k] = Du{p.x, h.x}; y[k] = Du{p.y, h.y}; d[k] = dts[j];
    }
    const PtD p0 = make_ptd(x[3], y[3], x[2], y[2], d[3], t >= 1, eps);
    const PtD p1 = make_ptd(x[2], y[2], x[1], y[1], d[2], t >= 2, eps);
    const PtD p2 = make_ptd(x[1], y[1], x[0], y[0], d[1], t >= 3, eps);
    const Du z{0.0f, 0.0f};
    Du acc = z, jerk = z, ax = z, ay = z;
    if (t >= 1) {
        const float m = (d[3] + d[2]) * 0.5f + eps;
        acc = du_divf(du_sub(p0.vel, p1.vel), m);
        ax = du_divf(du_sub(p0.vx, p1.vx), m);
        ay = du_divf(du_sub(p0.vy, p1.vy), m);
        if (t >= 2) {
            const float mp = (d[2] + d[1]) * 0.5f + eps;
            const Du accp = du_divf(du_sub(p1.vel, p2.vel), mp);
            jerk = du_divf(du_sub(acc, accp), m + eps);
        }
    }
    const Du u = du_addf(du_add(du_mul(p0.vx, p0.vx), du_mul(p0.vy, p0.vy)), eps);
    const Du q = du_sub(du_mul(p0.vx, ay), du_mul(p0.vy, ax));
    const float sgn = q.v > 0.f ? 1.f : (q.v < 0.f ? -1.f : 0.f);
    const Du cross{fabsf(q.v), sgn * q.d};
    const Du curv = du_div(cross, du_mul(u, du_sqrt(u)));
    o[0] = x[3].d; o[1] = y[3].d; o[2] = p0.dx.d; o[3] = p0.dy.d; o[4] = 0.0f;
    o[5] = du_clamp_tan(p0.vel, 0.f, 100.f);
    o[6] = du_clamp_tan(acc, -1000.f, 1000.f);
    o[7] = du_clamp_tan(jerk, -10000.f, 10000.f);
    o[8] = du_clamp_tan(curv, 0.f, 100.f);
}

GMT_HD void kin_jvp_point(long long idx, const float* pos, const float* dts, const float* phat, float* fhat,
                          int L, int B, float eps)
{
    float o[9];
    kin_jvp_vals(idx, pos, dts, phat, o, L, B, eps);
    for (int c = 0; c < 9; ++c) fhat[idx * 9 + c] = o[c];
}

struct GradFromMemory {
    const float* gfeat; long long idx; int B;
    GMT_HDM float operator()(int k, int c) const { return gfeat[(idx + (long long)k * B) * 9 + c]; }
};

GMT_HD void kin_bwd_point(long long idx, const float* pos, const float* dts, const float* gfeat,
                          float* gpos, float* gdts, int L, int B, float eps)
{
    float gx, gy, gd;
    GradFromMemory gm{gfeat, idx, B};
    kin_bwd_vals(idx, pos, dts, gm, gx, gy, gd, L, B, eps);
    gpos[idx * 2] = gx; gpos[idx * 2 + 1] = gy;
    gdts[idx] = gd;
}

}  // namespace gmt_kin
