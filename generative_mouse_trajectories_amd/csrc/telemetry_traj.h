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

// out6 planes: velocity, acceleration, jerk, path_length, direction, curvature
// hover / total (optional, may be null): hover_time, total_duration planes.
GMT_HD void kin_traj(long long r, long long n_traj, int n_max, int len,
                     const double* x, const double* y, const double* tm, int time_is_dt,
                     double min_dt, double alpha, double* out6, double* hover_out, double* total_out)
{
    const long long plane = (long long)n_max * n_traj;
    double path = 0.0, v_prev = 0.0, a_prev = 0.0, vx_prev = 0.0, vy_prev = 0.0, eff_prev = 0.0;
    double hover = 0.0, total = 0.0;
    double xp = 0.0, yp = 0.0, tp = 0.0, t0 = 0.0;
    for (int i = 0; i < n_max; ++i) {
        const long long o = (long long)i * n_traj + r;
        double v = 0.0, a = 0.0, j = 0.0, dir = 0.0, curv = 0.0, pl = 0.0, hv = 0.0, tot = 0.0;
        if (i < len) {
            const double xi = x[o], yi = y[o], ti = tm[o];
            if (i == 0) t0 = ti;
            if (i > 0) {
                const double raw_dt = time_is_dt ? ti : ti - tp;                 // MT.rs:580-585
                if (time_is_dt) total += raw_dt; else total = ti - t0;           // MT.rs:570-574
                const double eff = raw_dt > min_dt ? raw_dt : min_dt;            // MT.rs:587
                const double dx = xi - xp, dy = yi - yp;                         // MT.rs:591-594
                const double dist = sqrt(dx * dx + dy * dy);                     // MT.rs:597
                path += dist;                                                    // MT.rs:600
                const double v_raw = eff > 0.0 ? dist / eff : 0.0;               // MT.rs:603-607
                v = (i == 1) ? v_raw : alpha * v_raw + (1.0 - alpha) * v_prev;   // MT.rs:609-613
                const double a_raw = (i > 1 && eff > 0.0) ? (v - v_prev) / eff : 0.0;   // MT.rs:616-620
                a = (i <= 1) ? a_raw : alpha * a_raw + (1.0 - alpha) * a_prev;   // MT.rs:622-626
                j = (i > 1 && eff > 0.0) ? (a - a_prev) / eff : 0.0;             // MT.rs:629-633
                dir = atan2(dy, dx);                                             // MT.rs:636
                hover = (dist < 1.0) ? hover + raw_dt : 0.0;                     // MT.rs:638-644
                const double vx = dx / eff, vy = dy / eff;   // curvature: RecurrentGAN.py:359-368 form
                if (i > 1) {
                    const double dt_mid = 0.5 * (eff + eff_prev);
                    const double ax = (vx - vx_prev) / dt_mid, ay = (vy - vy_prev) / dt_mid;
                    const double u = vx * vx + vy * vy + 1e-8;
                    curv = fabs(vx * ay - vy * ax) / (u * sqrt(u));
                }
                vx_prev = vx; vy_prev = vy; eff_prev = eff; v_prev = v; a_prev = a;
            }
            xp = xi; yp = yi; tp = time_is_dt ? 0.0 : ti;
            pl = path; hv = hover; tot = total;
        }
        out6[0 * plane + o] = v;
        out6[1 * plane + o] = a;
        out6[2 * plane + o] = j;
        out6[3 * plane + o] = pl;
        out6[4 * plane + o] = dir;
        out6[5 * plane + o] = curv;
        if (hover_out) hover_out[o] = hv;
        if (total_out) total_out[o] = tot;
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
