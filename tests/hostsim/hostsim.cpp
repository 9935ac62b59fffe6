// Host-side simulation of the element-wise kernel bodies (CPU tests only).
// The per-point / per-trajectory / per-element functions in csrc/*_point.h, *_traj.h and
// *_elem.h are the exact bodies the gfx950 kernels run; compiling them with g++ lets the
// "not gpu" test-suite check their arithmetic against the oracle without a GPU.
// This file is test infrastructure: it is never linked into libgmt_hip.so.
#include "../../generative_mouse_trajectories_amd/csrc/kinematics_point.h"
#include "../../generative_mouse_trajectories_amd/csrc/telemetry_traj.h"
#include "../../generative_mouse_trajectories_amd/csrc/adam_elem.h"

extern "C" {
void sim_kin_fwd(const float* pos, const float* dts, float* feat, int L, int B, float eps) {
    for (long long i = 0; i < (long long)L * B; ++i) gmt_kin::kin_fwd_point(i, pos, dts, feat, L, B, eps);
}
void sim_kin_bwd(const float* pos, const float* dts, const float* gfeat, float* gpos, float* gdts,
                 int L, int B, float eps) {
    for (long long i = 0; i < (long long)L * B; ++i) gmt_kin::kin_bwd_point(i, pos, dts, gfeat, gpos, gdts, L, B, eps);
}
void sim_tel_kin(const double* x, const double* y, const double* tm, const int* lengths, double* out6,
                 double* hover, double* total, long long n_traj, int n_max, int time_is_dt, double min_dt, double alpha) {
    for (long long r = 0; r < n_traj; ++r)
        gmt_tel::kin_traj(r, n_traj, n_max, lengths ? lengths[r] : n_max, x, y, tm, time_is_dt, min_dt, alpha, out6, hover, total);
}
void sim_tel_derived(const double* x, const double* y, const double* raw_dt, const double* total, const double* kin6,
                     const int* lengths, double* out11, long long n_traj, int n_max) {
    for (long long r = 0; r < n_traj; ++r)
        gmt_tel::derived_traj(r, n_traj, n_max, lengths ? lengths[r] : n_max, x, y, raw_dt, total, kin6, out11);
}
void sim_adam(float* p, const float* g, float* m, float* v, long long n, float b1, float b2, float eps,
              float step_size, float bc2_sqrt) {
    for (long long i = 0; i < n; ++i) gmt_adam::adam_elem(p[i], g[i], m[i], v[i], b1, b2, eps, step_size, bc2_sqrt);
}
}
