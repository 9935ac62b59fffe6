// Per-trajectory bodies of the telemetry ("mode B") kernels: f64 restatement of
// crates/mouse-telemetry/src/mouse_telemetry.rs:550-654 (kinematics) and :664-756
// (derived running statistics).  Layout: time-major, padded, SoA -
// x/y/tm [n_max][n_traj], one thread per trajectory, so that at every step the 64
// lanes of a wave touch 64 consecutive doubles (512 B) of each plane.
// Shared by telemetry.hip (device) and the host simulation build of the CPU tests.
#pragma once
#include <math.h>
#ifndef GMT_HD
#  ifdef __HIPCC__
#    define GMT_HD __host__ __device__ __forceinline__
#  else
#    define GMT_HD static inline
#  endif
#endif

namespace gmt_tel {

// Running state of one trajectory and the update for one event (MT.rs:550-654).  kin_traj below walks a
// trajectory with it (host simulation, simple device path); the device kernel of telemetry.hip walks the same
// step function over register-staged chunks of events.
struct KinState {
    double path, v_prev, a_prev, vx_prev, vy_prev, eff_prev, hover, total, xp, yp, tp, t0;
};
struct KinOut { double v, a, j, pl, dir, curv, hv, tot; };

GMT_HD void kin_init(KinState& s)
{
    s.path = s.v_prev = s.a_prev = s.vx_prev = s.vy_prev = s.eff_prev = 0.0;
    s.hover = s.total = s.xp = s.yp = s.tp = s.t0 = 0.0;
}

// event i of a trajectory (live: i < its length; padded rows produce zeros and leave the state alone)
GMT_HD KinOut kin_step(KinState& s, int i, bool live, double xi, double yi, double ti, int time_is_dt,
                       double min_dt, double alpha)
{
    KinOut o;
    o.v = o.a = o.j = o.pl = o.dir = o.curv = o.hv = o.tot = 0.0;
    if (live) {
        if (i == 0) s.t0 = ti;
        if (i > 0) {
            const double raw_dt = time_is_dt ? ti : ti - s.tp;               // MT.rs:580-585
            if (time_is_dt) s.total += raw_dt; else s.total = ti - s.t0;     // MT.rs:570-574
            const double eff = raw_dt > min_dt ? raw_dt : min_dt;            // MT.rs:587
            const double dx = xi - s.xp, dy = yi - s.yp;                     // MT.rs:591-594
            const double dist = sqrt(dx * dx + dy * dy);                     // MT.rs:597
            s.path += dist;                                                  // MT.rs:600
            const double v_raw = eff > 0.0 ? dist / eff : 0.0;               // MT.rs:603-607
            o.v = (i == 1) ? v_raw : alpha * v_raw + (1.0 - alpha) * s.v_prev;   // MT.rs:609-613
            const double a_raw = (i > 1 && eff > 0.0) ? (o.v - s.v_prev) / eff : 0.0;   // MT.rs:616-620
            o.a = (i <= 1) ? a_raw : alpha * a_raw + (1.0 - alpha) * s.a_prev;   // MT.rs:622-626
            o.j = (i > 1 && eff > 0.0) ? (o.a - s.a_prev) / eff : 0.0;       // MT.rs:629-633
            o.dir = atan2(dy, dx);                                           // MT.rs:636
            s.hover = (dist < 1.0) ? s.hover + raw_dt : 0.0;                 // MT.rs:638-644
            const double vx = dx / eff, vy = dy / eff;   // curvature: RecurrentGAN.py:359-368 form
            if (i > 1) {
                const double dt_mid = 0.5 * (eff + s.eff_prev);
                const double ax = (vx - s.vx_prev) / dt_mid, ay = (vy - s.vy_prev) / dt_mid;
                const double u = vx * vx + vy * vy + 1e-8;
                o.curv = fabs(vx * ay - vy * ax) / (u * sqrt(u));
            }
            s.vx_prev = vx; s.vy_prev = vy; s.eff_prev = eff; s.v_prev = o.v; s.a_prev = o.a;
        }
        s.xp = xi; s.yp = yi; s.tp = time_is_dt ? 0.0 : ti;
        o.pl = s.path; o.hv = s.hover; o.tot = s.total;
    }
    return o;
}

// out6 planes: velocity, acceleration, jerk, path_length, direction, curvature
// hover / total (optional, may be null): hover_time, total_duration planes.
GMT_HD void kin_traj(long long r, long long n_traj, int n_max, int len,
                     const double* x, const double* y, const double* tm, int time_is_dt,
                     double min_dt, double alpha, double* out6, double* hover_out, double* total_out)
{
    const long long plane = (long long)n_max * n_traj;
    KinState s;
    kin_init(s);
    for (int i = 0; i < n_max; ++i) {
        const long long o = (long long)i * n_traj + r;
        const bool live = i < len;
        const KinOut k = kin_step(s, i, live, live ? x[o] : 0.0, live ? y[o] : 0.0, live ? tm[o] : 0.0, time_is_dt,
                                  min_dt, alpha);
        out6[0 * plane + o] = k.v;
        out6[1 * plane + o] = k.a;
        out6[2 * plane + o] = k.j;
        out6[3 * plane + o] = k.pl;
        out6[4 * plane + o] = k.dir;
        out6[5 * plane + o] = k.curv;
        if (hover_out) hover_out[o] = k.hv;
        if (total_out) total_out[o] = k.tot;
    }
}

// derived: 11 planes [11][n_max][n_traj] in the CSV column order of MT.rs:820-830.
GMT_HD void derived_traj(long long r, long long n_traj, int n_max, int len,
                         const double* x, const double* y, const double* raw_dt, const double* total_duration,
                         const double* kin6, double* out11)
{
    const long long plane = (long long)n_max * n_traj;
    const double* velocity = kin6;
    const double* acceleration = kin6 + plane;
    const double* jerk = kin6 + 2 * plane;
    const double* path_length = kin6 + 3 * plane;
    double cum_v = 0.0, max_v = 0.0, cum_a = 0.0, max_a = 0.0, cum_j = 0.0, cum_active = 0.0;
    double cum_m = 0.0, max_m = 0.0, x0 = 0.0, y0 = 0.0;
    int jerk_count = 0;
    for (int i = 0; i < n_max; ++i) {
        const long long o = (long long)i * n_traj + r;
        double d[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) d[k] = 0.0;
        if (i < len) {
            const double xi = x[o], yi = y[o], td = total_duration[o];
            if (i == 0) { x0 = xi; y0 = yi; }
            if (i >= 1) {                                                        // MT.rs:689-709
                const double v = velocity[o];
                cum_v += v; if (v > max_v) max_v = v;
                d[0] = cum_v / (double)i; d[1] = max_v;
                const double m = fabs(v);
                cum_m += m; if (m > max_m) max_m = m;
                d[4] = cum_m / (double)i; d[5] = max_m;
                cum_active += raw_dt[o];
                d[8] = td - cum_active;
            }
            if (i >= 2) {                                                        // MT.rs:711-719
                const double a = acceleration[o];
                cum_a += a; if (a > max_a) max_a = a;
                d[2] = cum_a / (double)(i - 1); d[3] = max_a;
            }
            if (i >= 3) {                                                        // MT.rs:721-730
                cum_j += fabs(jerk[o]); jerk_count += 1;
                const double avg = cum_j / (double)jerk_count;
                d[6] = avg > 0.0 ? 1.0 / avg : 0.0;
            }
            const double ix = xi - x0, iy = yi - y0;                             // MT.rs:732-745
            const double ideal = sqrt(ix * ix + iy * iy);
            d[7] = ideal > 0.0 ? path_length[o] / ideal : 0.0;
            d[9] = log2(ideal / 100.0 + 1.0);
            d[10] = td;
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) out11[k * plane + o] = d[k];
    }
}

}  // namespace gmt_tel
