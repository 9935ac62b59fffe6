/* libgmt_hip.so - C ABI of the MI355X (gfx950) kernels behind the WGAN-GP trajectory
 * train step of jrcalgo/generative-mouse-trajectories.
 *
 * The reference has no FFI for this path: its hot path is Python
 * (python/learning/RecurrentGAN.py, "RG.py") calling stock PyTorch ops, and its telemetry
 * arithmetic is a Rust function (crates/mouse-telemetry/src/mouse_telemetry.rs, "MT.rs").
 * Each entry point below names the reference code it replaces.  The host side
 * (the generative_mouse_trajectories_amd package, same class / function surface as RG.py) binds
 * this library with ctypes; INTEGRATION.md shows the binding a maintainer of the reference
 * would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current device; the caller owns all memory,
 *     workspace included - nothing here allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t); calls are asynchronous;
 *   - return value: 0 (GMT_OK) or a negative GMT_ERR_* code; the host turns it into an
 *     exception;
 *   - no global mutable state; re-entrant (the autograd engine calls from its own thread);
 *   - tensors are fp32 and TIME-MAJOR unless said otherwise: [L][B][...] so that one
 *     timestep of all trajectories is contiguous.
 */
#ifndef GMT_HIP_H
#define GMT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define GMT_OK 0
#define GMT_ERR_BAD_ARG (-1)      /* null pointer, bad shape, misaligned buffer          */
#define GMT_ERR_UNSUPPORTED (-2)  /* hidden size / tile configuration not instantiated   */
#define GMT_ERR_HIP (-3)          /* a HIP launch failed                                 */
#define GMT_ERR_WORKSPACE (-4)    /* caller's workspace is too small                     */

/* ABI version of this header; gmt_version() of the loaded library must match. */
#define GMT_ABI_VERSION 1
int gmt_version(void);

/* ---- kinematic features, "mode A" ---------------------------------------------------
 * Replaces compute_kinematics, RG.py:315-389 (called from Discriminator.forward,
 * RG.py:757).  pos [L][B][2], dts [L][B] -> feat [L][B][9] =
 * (x, y, dx, dy, dt, velocity, acceleration, jerk, curvature), clamps of RG.py:371-374. */
int gmt_kinematics_fwd(const float* pos, const float* dts, float* feat,
                       int L, int B, float eps, hipStream_t stream);
/* Its backward (autograd does this in the reference): gfeat [L][B][9] ->
 * gpos [L][B][2], gdts [L][B]. */
int gmt_kinematics_bwd(const float* pos, const float* dts, const float* gfeat,
                       float* gpos, float* gdts, int L, int B, float eps, hipStream_t stream);

/* ---- telemetry kinematics, "mode B", f64 ----------------------------------------------
 * Replaces MouseCollector::calculate_collection_attributes, MT.rs:550-654
 * (constants MT.rs:31,33).  x, y, tm: [n_max][n_traj] planes (time-major, padded);
 * tm holds absolute timestamps, or the raw time between movements when time_is_dt != 0.
 * lengths [n_traj] (may be NULL = n_max).  out6: [6][n_max][n_traj] = velocity,
 * acceleration, jerk, path_length, direction, curvature (curvature follows
 * RG.py:359-368; MT.rs has none).  hover_time / total_duration: optional planes. */
int gmt_telemetry_kinematics_f64(const double* x, const double* y, const double* tm, const int* lengths,
                                 double* out6, double* hover_time, double* total_duration,
                                 long long n_traj, int n_max, int time_is_dt,
                                 double min_dt, double alpha, hipStream_t stream);
/* Replaces calculate_derived_attributes, MT.rs:664-756: out11 [11][n_max][n_traj] in the
 * CSV column order of MT.rs:820-830. */
int gmt_telemetry_derived_f64(const double* x, const double* y, const double* raw_dt,
                              const double* total_duration, const double* kin6, const int* lengths,
                              double* out11, long long n_traj, int n_max, hipStream_t stream);

/* ---- fused Adam -----------------------------------------------------------------------
 * Replaces torch.optim.Adam.step for g_optimizer / d_optimizer, RG.py:864-873, 1022, 1110.
 * One launch over one flat, 16-byte aligned buffer per model.  `step` is the 1-based step
 * count, grad_scale multiplies the gradient first (1/world_size under data parallelism). */
int gmt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                  hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GMT_HIP_H */
