/* CPU oracle for the mouse-telemetry kinematics (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the arithmetic of the reference's Rust crate
 *   crates/mouse-telemetry/src/mouse_telemetry.rs  ("MT.rs")
 *     calculate_collection_attributes   MT.rs:550-654
 *     calculate_derived_attributes      MT.rs:664-756
 *     constants MIN_DERIVATIVE_DT / SMOOTHING_ALPHA   MT.rs:31,33
 * The Rust toolchain is absent from the image, so the crate itself cannot be
 * built; this restatement is PINNED against the reference's own data files
 * (data/mouse_data_*.csv hold the inputs and the outputs of exactly these two
 * functions; tests/golden/telemetry.npz is their parsed form) - all rows are
 * reproduced bit-exactly in f64 (tests/test_telemetry_oracle.py).
 * `direction` is computed by MT.rs:636 but never written to the CSV, and
 * `curvature` does not exist in MT.rs at all (it follows the formula of
 * RecurrentGAN.py:359-368 on the clamped dt): those two outputs are
 * "parity unpinned".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stddef.h>

#define TEL_CURV_EPS 1e-8

/* One trajectory of n events.  If time_is_dt != 0, tm[i] is the raw time
 * between movements (tm[0] ignored), otherwise tm[i] is an absolute timestamp.
 * Outputs (each n doubles, may be NULL): velocity, acceleration, jerk,
 * path_length, direction, curvature, hover_time, total_duration. */
void tel_kinematics(int n, const double *x, const double *y, const double *tm, int time_is_dt,
                    double min_dt, double alpha,
                    double *velocity, double *acceleration, double *jerk, double *path_length,
                    double *direction, double *curvature, double *hover_time, double *total_duration)
{
    double path = 0.0, v_prev = 0.0, a_prev = 0.0, hover = 0.0, total = 0.0;
    double vx_prev = 0.0, vy_prev = 0.0, eff_prev = 0.0;
    for (int i = 0; i < n; ++i) {
        double v = 0.0, a = 0.0, j = 0.0, dir = 0.0, curv = 0.0;
        if (i > 0) {
            double raw_dt = time_is_dt ? tm[i] : tm[i] - tm[i - 1];          /* MT.rs:580-585 */
            if (time_is_dt) total += raw_dt; else total = tm[i] - tm[0];     /* MT.rs:570-574 */
            double eff = raw_dt > min_dt ? raw_dt : min_dt;                  /* MT.rs:587 */
            double dx = x[i] - x[i - 1], dy = y[i] - y[i - 1];               /* MT.rs:591-594 */
            double dist = sqrt(dx * dx + dy * dy);                           /* MT.rs:597 */
            path += dist;                                                    /* MT.rs:600 */
            double v_raw = eff > 0.0 ? dist / eff : 0.0;                     /* MT.rs:603-607 */
            v = (i == 1) ? v_raw : alpha * v_raw + (1.0 - alpha) * v_prev;   /* MT.rs:609-613 */
            double a_raw = (i > 1 && eff > 0.0) ? (v - v_prev) / eff : 0.0;  /* MT.rs:616-620 */
            a = (i <= 1) ? a_raw : alpha * a_raw + (1.0 - alpha) * a_prev;   /* MT.rs:622-626 */
            j = (i > 1 && eff > 0.0) ? (a - a_prev) / eff : 0.0;             /* MT.rs:629-633 */
            dir = atan2(dy, dx);                                             /* MT.rs:636 */
            hover = (dist < 1.0) ? hover + raw_dt : 0.0;                     /* MT.rs:638-644 */
            /* curvature: RecurrentGAN.py:359-368 on the clamped dt (no Rust counterpart) */
            double vx = dx / eff, vy = dy / eff;
            if (i > 1) {
                double dt_mid = 0.5 * (eff + eff_prev);
                double ax = (vx - vx_prev) / dt_mid, ay = (vy - vy_prev) / dt_mid;
                curv = fabs(vx * ay - vy * ax) / pow(vx * vx + vy * vy + TEL_CURV_EPS, 1.5);
            }
            vx_prev = vx; vy_prev = vy; eff_prev = eff;
            v_prev = v; a_prev = a;
        }
        if (velocity) velocity[i] = v;
        if (acceleration) acceleration[i] = a;
        if (jerk) jerk[i] = j;
        if (path_length) path_length[i] = path;
        if (direction) direction[i] = dir;
        if (curvature) curvature[i] = curv;
        if (hover_time) hover_time[i] = hover;
        if (total_duration) total_duration[i] = total;
    }
}

/* Derived running statistics, MT.rs:664-756.  out is n x 11 row-major:
 * average_velocity, peak_velocity, average_acceleration, peak_acceleration,
 * average_momentum, peak_momentum, smoothness, deviation_from_ideal_path,
 * idle_time, fitts_index_of_difficulty, fitts_movement_time. */
void tel_derived(int n, const double *x, const double *y, const double *raw_dt, const double *total_duration,
                 const double *velocity, const double *acceleration, const double *jerk,
                 const double *path_length, double *out)
{
    double cum_v = 0.0, max_v = 0.0, cum_a = 0.0, max_a = 0.0, cum_j = 0.0, cum_active = 0.0;
    double cum_m = 0.0, max_m = 0.0;
    int jerk_count = 0;
    for (int i = 0; i < n; ++i) {
        double *d = out + (size_t)i * 11;
        for (int k = 0; k < 11; ++k) d[k] = 0.0;
        if (i >= 1) {                                                        /* MT.rs:689-709 */
            double v = velocity[i];
            cum_v += v; if (v > max_v) max_v = v;
            d[0] = cum_v / (double)i; d[1] = max_v;
            double m = fabs(v);
            cum_m += m; if (m > max_m) max_m = m;
            d[4] = cum_m / (double)i; d[5] = max_m;
            cum_active += raw_dt[i];
            d[8] = total_duration[i] - cum_active;
        }
        if (i >= 2) {                                                        /* MT.rs:711-719 */
            double a = acceleration[i];
            cum_a += a; if (a > max_a) max_a = a;
            d[2] = cum_a / (double)(i - 1); d[3] = max_a;
        }
        if (i >= 3) {                                                        /* MT.rs:721-730 */
            cum_j += fabs(jerk[i]); jerk_count += 1;
            double avg = cum_j / (double)jerk_count;
            d[6] = avg > 0.0 ? 1.0 / avg : 0.0;
        }
        double ix = x[i] - x[0], iy = y[i] - y[0];                           /* MT.rs:732-745 */
        double ideal = sqrt(ix * ix + iy * iy);
        d[7] = ideal > 0.0 ? path_length[i] / ideal : 0.0;
        d[9] = log2(ideal / 100.0 + 1.0);
        d[10] = total_duration[i];
    }
}

/* Batched driver over a time-major padded layout [n_max][n_traj] (the layout
 * the HIP kernel uses); out6 is [6][n_max][n_traj] = velocity, acceleration,
 * jerk, path_length, direction, curvature.  Used as the CPU baseline. */
void tel_kinematics_batch(int n_traj, int n_max, const int *lengths,
                          const double *x, const double *y, const double *tm, int time_is_dt,
                          double min_dt, double alpha, double *out6)
{
    size_t plane = (size_t)n_max * (size_t)n_traj;
    for (int r = 0; r < n_traj; ++r) {
        int n = lengths ? lengths[r] : n_max;
        double path = 0.0, v_prev = 0.0, a_prev = 0.0, vx_prev = 0.0, vy_prev = 0.0, eff_prev = 0.0;
        for (int i = 0; i < n_max; ++i) {
            size_t o = (size_t)i * n_traj + r;
            double v = 0.0, a = 0.0, j = 0.0, dir = 0.0, curv = 0.0;
            if (i > 0 && i < n) {
                size_t p = o - n_traj;
                double raw_dt = time_is_dt ? tm[o] : tm[o] - tm[p];
                double eff = raw_dt > min_dt ? raw_dt : min_dt;
                double dx = x[o] - x[p], dy = y[o] - y[p];
                double dist = sqrt(dx * dx + dy * dy);
                path += dist;
                double v_raw = eff > 0.0 ? dist / eff : 0.0;
                v = (i == 1) ? v_raw : alpha * v_raw + (1.0 - alpha) * v_prev;
                double a_raw = (i > 1 && eff > 0.0) ? (v - v_prev) / eff : 0.0;
                a = (i <= 1) ? a_raw : alpha * a_raw + (1.0 - alpha) * a_prev;
                j = (i > 1 && eff > 0.0) ? (a - a_prev) / eff : 0.0;
                dir = atan2(dy, dx);
                double vx = dx / eff, vy = dy / eff;
                if (i > 1) {
                    double dt_mid = 0.5 * (eff + eff_prev);
                    double ax = (vx - vx_prev) / dt_mid, ay = (vy - vy_prev) / dt_mid;
                    curv = fabs(vx * ay - vy * ax) / pow(vx * vx + vy * vy + TEL_CURV_EPS, 1.5);
                }
                vx_prev = vx; vy_prev = vy; eff_prev = eff; v_prev = v; a_prev = a;
            }
            if (i >= n) path = 0.0;
            out6[0 * plane + o] = v;
            out6[1 * plane + o] = a;
            out6[2 * plane + o] = j;
            out6[3 * plane + o] = (i < n) ? path : 0.0;
            out6[4 * plane + o] = dir;
            out6[5 * plane + o] = curv;
        }
    }
}
