// Telemetry ("mode B") kinematic-metric recompute, f64, from raw (x, y, t) batches:
// mouse_telemetry.rs:550-756 semantics.  One thread per trajectory walks its events in
// order (first-order recurrences: EMA velocity / acceleration, running path length);
// the planes are time-major so every step of a wave is 64 consecutive doubles per plane.
// HBM-bound: 24 B in + 48 B out per point.
#include "gmt_common.h"
#include "telemetry_traj.h"

namespace {

// One thread per trajectory; the events are walked in chunks of TEL_C: the x / y / t of the NEXT chunk (3 x TEL_C
// loads of 8 B per lane = 12 KB per wave) are requested before the current chunk's recurrences run, so a wave
// always has a chunk of reads in flight under its arithmetic and stores (the dependent chain of one event is
// ~60 f64 operations; without the staging each step waited for its own three loads).
#ifndef GMT_TEL_C
#define GMT_TEL_C 4
#endif
constexpr int TEL_C = GMT_TEL_C;

__global__ __launch_bounds__(64)
void tel_kin_kernel(const double* __restrict__ x, const double* __restrict__ y,
                    const double* __restrict__ tm, const int* __restrict__ lengths,
                    double* __restrict__ out6, double* __restrict__ hover, double* __restrict__ total,
                    long long n_traj, int n_max, int time_is_dt, double min_dt, double alpha)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_traj) return;
    int len = lengths ? lengths[r] : n_max;
    len = len < 0 ? 0 : (len > n_max ? n_max : len);
    const long long plane = (long long)n_max * n_traj;
    gmt_tel::KinState s;
    gmt_tel::kin_init(s);
    double cx[TEL_C], cy[TEL_C], ct[TEL_C], nx[TEL_C], ny[TEL_C], nt[TEL_C];
    auto fetch = [&](double (&fx)[TEL_C], double (&fy)[TEL_C], double (&ft)[TEL_C], int i0) {
#pragma unroll
        for (int k = 0; k < TEL_C; ++k) {
            int i = i0 + k;
            i = i < n_max ? i : n_max - 1;              // clamped (a harmless re-read), never branched on
            const long long o = (long long)i * n_traj + r;
            fx[k] = x[o]; fy[k] = y[o]; ft[k] = tm[o];
        }
    };
    fetch(cx, cy, ct, 0);
    for (int i0 = 0; i0 < n_max; i0 += TEL_C) {
        fetch(nx, ny, nt, i0 + TEL_C);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < TEL_C; ++k) {
            const int i = i0 + k;
            if (i < n_max) {
                const gmt_tel::KinOut q = gmt_tel::kin_step(s, i, i < len, cx[k], cy[k], ct[k], time_is_dt, min_dt, alpha);
                const long long o = (long long)i * n_traj + r;
                out6[0 * plane + o] = q.v;
                out6[1 * plane + o] = q.a;
                out6[2 * plane + o] = q.j;
                out6[3 * plane + o] = q.pl;
                out6[4 * plane + o] = q.dir;
                out6[5 * plane + o] = q.curv;
                if (hover) hover[o] = q.hv;
                if (total) total[o] = q.tot;
            }
        }
#pragma unroll
        for (int k = 0; k < TEL_C; ++k) { cx[k] = nx[k]; cy[k] = ny[k]; ct[k] = nt[k]; }
    }
}

__global__ void tel_derived_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                   const double* __restrict__ raw_dt, const double* __restrict__ total,
                                   const double* __restrict__ kin6, const int* __restrict__ lengths,
                                   double* __restrict__ out11, long long n_traj, int n_max)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_traj) return;
    int len = lengths ? lengths[r] : n_max;
    len = len < 0 ? 0 : (len > n_max ? n_max : len);
    gmt_tel::derived_traj(r, n_traj, n_max, len, x, y, raw_dt, total, kin6, out11);
}

}  // namespace

extern "C" int gmt_telemetry_kinematics_f64(const double* x, const double* y, const double* tm, const int* lengths,
                                            double* out6, double* hover_time, double* total_duration,
                                            long long n_traj, int n_max, int time_is_dt,
                                            double min_dt, double alpha, hipStream_t stream)
{
    GMT_CHECK_ARG(x && y && tm && out6 && n_traj >= 1 && n_max >= 1 && n_traj < (1LL << 40));
    const int block = 64;   // one wave per block: 1 M trajectories -> 16 K waves over 256 CUs
    const long long grid = (n_traj + block - 1) / block;
    GMT_CHECK_ARG(grid <= 0x7fffffffLL);
    hipLaunchKernelGGL(tel_kin_kernel, dim3((unsigned)grid), dim3(block), 0, stream, x, y, tm, lengths, out6,
                       hover_time, total_duration, n_traj, n_max, time_is_dt, min_dt, alpha);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}

extern "C" int gmt_telemetry_derived_f64(const double* x, const double* y, const double* raw_dt,
                                         const double* total_duration, const double* kin6, const int* lengths,
                                         double* out11, long long n_traj, int n_max, hipStream_t stream)
{
    GMT_CHECK_ARG(x && y && raw_dt && total_duration && kin6 && out11 && n_traj >= 1 && n_max >= 1);
    const int block = 64;
    const long long grid = (n_traj + block - 1) / block;
    GMT_CHECK_ARG(grid <= 0x7fffffffLL);
    hipLaunchKernelGGL(tel_derived_kernel, dim3((unsigned)grid), dim3(block), 0, stream, x, y, raw_dt, total_duration,
                       kin6, lengths, out11, n_traj, n_max);
    GMT_LAUNCH_CHECK();
    return GMT_OK;
}
