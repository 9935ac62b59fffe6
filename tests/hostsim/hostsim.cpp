// Host-side simulation of the element-wise kernel bodies (CPU tests only).
// The per-point / per-trajectory / per-element functions in csrc/*_point.h, *_traj.h and
// *_elem.h are the exact bodies the gfx950 kernels run; compiling them with g++ lets the
// "not gpu" test-suite check their arithmetic against the oracle without a GPU.
// This file is test infrastructure: it is never linked into libgmt_hip.so.
#include "../../generative_mouse_trajectories_amd/csrc/kinematics_point.h"
#include "../../generative_mouse_trajectories_amd/csrc/telemetry_traj.h"
#include "../../generative_mouse_trajectories_amd/csrc/adam_elem.h"
#include "../../generative_mouse_trajectories_amd/csrc/bf16_maps.h"

extern "C" {
void sim_kin_fwd(const float* pos, const float* dts, float* feat, int L, int B, float eps) {
    for (long long i = 0; i < (long long)L * B; ++i) gmt_kin::kin_fwd_point(i, pos, dts, feat, L, B, eps);
}
void sim_kin_jvp(const float* pos, const float* dts, const float* phat, float* fhat, int L, int B, float eps) {
    for (long long i = 0; i < (long long)L * B; ++i) gmt_kin::kin_jvp_point(i, pos, dts, phat, fhat, L, B, eps);
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
// index maps of the bf16 kernels (csrc/bf16_maps.h)
int sim_stage_src_chunk(int row, int pos) { return gmt_bf16_maps::stage_src_chunk(row, pos); }
int sim_stage_addr(int row, int chunk) { return gmt_bf16_maps::stage_addr(row, chunk); }
int sim_stage32_src_chunk(int row, int pos) { return gmt_bf16_maps::stage32_src_chunk(row, pos); }
int sim_stage32_addr(int row, int chunk) { return gmt_bf16_maps::stage32_addr(row, chunk); }
int sim_image_src_chunk(int rowb, int row, int lane) { return gmt_bf16_maps::image_src_chunk(rowb, row, lane); }
int sim_image_addr(int rowb, int row, int off) { return gmt_bf16_maps::image_addr(rowb, row, off); }
int sim_wg_src_chunk(int row, int pos) { return gmt_bf16_maps::wg_src_chunk(row, pos); }
int sim_wg_addr(int row, int col) { return gmt_bf16_maps::wg_addr(row, col); }
void sim_wg_tr_lane(int lane, int col0, int s, int half, int* row, int* col) { gmt_bf16_maps::wg_tr_lane(lane, col0, s, half, *row, *col); }
int sim_slab_col(int H, int d, int u, int g) { return gmt_bf16_maps::slab_col(H, d, u, g); }
void sim_tile_of_block(long long bid, long long nb, int n_p, long long n_q, int* tp, long long* tq) {
    gmt_bf16_maps::tile_of_block(bid, nb, n_p, n_q, *tp, *tq);
}
void sim_patch_of_block(int bid, int nb, int n_mt, int n_pt, int* d, int* pt, int* mt) {
    gmt_bf16_maps::patch_of_block(bid, nb, n_mt, n_pt, *d, *pt, *mt);
}
}
