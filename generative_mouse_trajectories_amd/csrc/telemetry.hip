// Telemetry ("mode B") kinematic-metric recompute, f64, from raw (x, y, t) batches:
// mouse_telemetry.rs:550-756 semantics.  One thread per trajectory walks its events in
// order (first-order recurrences: EMA velocity / acceleration, running path length);
// the planes are time-major so every step of a wave is 64 consecutive doubles per plane.
// HBM-bound: 24 B in + 48 B out per point.
#include "gmt_common.h"
#include "telemetry_traj.h"

namespace {

__global__ void tel_kin_kernel(const double* __restrict__ x, const double* __restrict__ y,
                               const double* __restrict__ tm, const int* __restrict__ lengths,
                               double* __restrict__ out6, double* __restrict__ hover, double* __restrict__ total,
                               long long n_traj, int n_max, int time_is_dt, double min_dt, double alpha)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_traj) return;
    int len = lengths ? lengths[r] : n_max;
    len = len < 0 ? 0 : (len > n_max ? n_max : len);
    gmt_tel::kin_traj(r, n_traj, n_max, len, x, y, tm, time_is_dt, min_dt, alpha, out6, hover, total);
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
