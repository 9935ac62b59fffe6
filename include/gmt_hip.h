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
 *   - re-entrant (the autograd engine calls from its own thread); no global mutable state with two exceptions, both
 *     per device and documented at their entry points: the bf16 row-group side stream + events (gmt_bf16_set_row_groups)
 *     and the static scratch array of gmt_wgrad_noalloc;
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
#define GMT_ABI_VERSION 9
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

/* Forward-mode derivative of the same features w.r.t. the positions (dts fixed): phat [L][B][2] -> fhat [L][B][9] =
 * J(pos, dts) phat.  The kinematics piece of the second backward of compute_gradient_penalty (RG.py:829-835,
 * create_graph=True): the adjoint of gfeat in gpos = J^T gfeat.  Clamped features pass no tangent, as in _bwd. */
int gmt_kinematics_jvp(const float* pos, const float* dts, const float* phat, float* fhat,
                       int L, int B, float eps, hipStream_t stream);
/* out = g * LeakyReLU'(act) element-wise (n % 4 == 0, 16-byte aligned): the activation masks of the feature encoder
 * (RG.py:714, 716) in the penalty's first and second backward, where they cannot ride in a GEMM epilogue. */
int gmt_lrelu_mask_mul(const float* g, const float* act, float* out, float slope, long long n, hipStream_t stream);

/* ---- fused critic front end ----------------------------------------------------------------
 * Replaces compute_kinematics (RG.py:315-389, called at RG.py:757) + Discriminator.feature_encoder
 * (RG.py:712-717, applied at RG.py:761) in one launch: pos [L][B][2], dts [L][B] ->
 * e [L][B][H] = lrelu(W2 lrelu(W1 feat + b1) + b2), slope 0.2 in the reference.  w1_pk / w2_pk: the
 * (spectrally normalised) weights [H][9] / [H][H] packed by gmt_pack_weight (K 16 with k_valid 9 / K H;
 * n_waves H/32, 1 gate, stride_k 1, stride_n 9 / H).  feat [L][B][9] and h1 [L][B][H] (first activation)
 * are optional outputs for the backward pass.  B % 32 == 0, H in {32, 64, 128, 256}. */
int gmt_encoder_fwd(const float* pos, const float* dts, const float* w1_pk, const float* b1,
                    const float* w2_pk, const float* b2, float* feat, float* h1, float* e,
                    int L, int B, int H, float eps, float slope, hipStream_t stream);

/* ---- input-gradient contraction -----------------------------------------------------------------
 * dX[r][n] = (sum_d sum_k dA[d][r][k] * W_d[k][n]) * (act ? (act[r][n] > 0 ? 1 : slope) : 1): the dx GEMMs autograd
 * runs for RG.py:1021 / 1109 - the input gradient of an LSTM layer (dA = d(pre-activation) of both directions,
 * K = 4H, W = W_ih with its 4H axis in the buffers' unit-major order) and of the encoder's Linears with their
 * LeakyReLU backward fused (act = the layer's activated output).  dA [ndir][R][K], dX [R][N] contiguous;
 * N in {32, 64, 128, 256, 512}, K in {N, 2N, 4N} (<= 1024), R % 32 == 0.
 * w_pk [ndir] packed by gmt_pack_weight with stride_k N, stride_n 1 and
 *   - gmt_dx_gemm_resident(K, N) != 0 (the hot shapes: weights stay in registers): n_waves N/32, 1 gate;
 *   - otherwise (weights streamed): n_waves min(4, N/32), n_gates N / (32 n_waves), gate_stride 32 n_waves. */
int gmt_dx_gemm_resident(int K, int N);
int gmt_dx_gemm(const float* dA, const float* w_pk, float* dX, const float* act, float slope,
                long long R, int K, int N, int ndir, hipStream_t stream);

/* ---- trajectory bookkeeping and generator losses ----------------------------------------------------
 * gmt_traj_from_seq: trajectory_to_absolute (RG.py:392-417) + F.pad(dt, (1, 0), first_dt) (RG.py:934, 949, 1061)
 * + the transposition to time-major: seq element (b, t, c) at seq[b*stride_b + t*stride_t + c], c = dx, dy, dt;
 * starts [B][2] -> pos [T+1][B][2], dts [T+1][B].  Rows >= n_valid (batch padding) are zero.  _bwd: the reverse
 * cumulative sum, (gpos, gdts) -> gseq in the given strides; gseq has n_rows_out rows (n_valid <= n_rows_out <= B),
 * those >= n_valid are zero. */
int gmt_traj_from_seq(const float* seq, long long stride_b, long long stride_t, const float* starts,
                      float* pos, float* dts, int T, int B, int n_valid, float first_dt, hipStream_t stream);
int gmt_traj_from_seq_bwd(const float* gpos, const float* gdts, float* gseq, long long stride_b,
                          long long stride_t, int T, int B, int n_valid, int n_rows_out, hipStream_t stream);
/* gmt_rollout_to_xyt_f64: generated (dx, dy, dt) sequences (Generator.generate, RG.py:606-690; same strides as
 * above) -> the telemetry kernels' f64 input planes x, y (scaled to pixels: RG.py:168-189 divides by the screen
 * size) and raw time between movements, each [T+1][n], point 0 = the start point with dt 0 - the seam between
 * the sampling kernel and gmt_telemetry_kinematics_f64 (time_is_dt = 1). */
int gmt_rollout_to_xyt_f64(const float* seq, long long stride_b, long long stride_t, const float* starts,
                           double* x, double* y, double* dt, int T, long long n, double scale_x, double scale_y,
                           hipStream_t stream);
/* gmt_gen_loss_fwd: every reduction of train_generator_step (RG.py:1069-1095) in one pass:
 * scal[0] = sum_b |final_b - end_b|^2 (final_b = fake_traj[b, len_b], RG.py:1072), scal[1] = sum (1 - cos) mask,
 * scal[2] = sum mask (RG.py:1077-1095), scal[3] = sum_b score_b (score may be NULL).  final_pos [n_valid][2] is
 * saved for the backward; workspace: gmt_gen_loss_workspace_floats.  _bwd: gseq = gscal[0] d(scal[0]) +
 * gscal[1] d(scal[1]) w.r.t. the sequence (gscal: 2 device floats), rows >= n_valid zero. */
int gmt_gen_loss_workspace_floats(int n_valid, long long* floats_out);
int gmt_gen_loss_fwd(const float* seq, long long stride_b, long long stride_t, const float* starts,
                     const float* ends, const int* lengths, const float* score, float* final_pos,
                     float* scal, float* workspace, int T, int n_valid, hipStream_t stream);
int gmt_gen_loss_bwd(const float* seq, long long stride_b, long long stride_t, const float* starts,
                     const float* ends, const int* lengths, const float* final_pos, const float* gscal,
                     float* gseq, long long gstride_b, long long gstride_t, int T, int B, int n_valid,
                     hipStream_t stream);

/* The scalar arithmetic of the two step functions, so that no element-wise torch kernels sit between the critic call and
 * the backward pass (~45 launches per iteration otherwise).  They have the reference's arithmetic order.
 * gmt_critic_loss_fwd / _bwd: d_loss = mean(fake) - mean(real) of RG.py:963 over the B valid rows of two padded score
 *   vectors; out = (mean real, mean fake) (the d_real_score / d_fake_score metrics of RG.py:1024-1027); backward writes
 *   -+ g / B on the valid rows and zeros on the padding.
 * gmt_gen_step_loss_finish (after gmt_gen_loss_fwd with the score vector given: scal[4]): total = g_loss + w_endpoint endpoint + w_direction direction (RG.py:1097), out = (g_loss,
 *   endpoint, direction, mean score), coef = the three factors the backward kernel scales by the adjoint of total;
 *   nd (or NULL) = the direction numerator / denominator summed over the ranks of a data-parallel run, world = their
 *   number (RG.py:1095 divides by the mask sum of the whole batch).
 * gmt_gen_step_loss_bwd: gradient of total wrt the rollout (gseq [T][B][3]) and wrt the padded score vector, one launch. */
int gmt_critic_loss_fwd(const float* real, const float* fake, int B, float* loss, float* out, hipStream_t stream);
int gmt_critic_loss_bwd(const float* g, int B, int Bp, float* d_real, float* d_fake, hipStream_t stream);
int gmt_gen_step_loss_finish(const float* scal, const float* nd, float world, int n_valid, float w_endpoint,
                             float w_direction, float* total, float* out, float* coef, hipStream_t stream);
int gmt_gen_step_loss_bwd(const float* seq, long long stride_b, long long stride_t, const float* starts,
                          const float* ends, const int* lengths, const float* final_pos, const float* coef,
                          const float* g_total, float* gseq, long long gstride_b, long long gstride_t,
                          float* d_score, int T, int B, int n_valid, hipStream_t stream);

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

/* ---- spectral normalisation of the critic's Linear weights ---------------------------------
 * Replaces the torch.nn.utils.spectral_norm hook the reference wraps its five critic Linears in
 * (RG.py:713, 715, 731, 734, 736; the hook's compute_weight runs at every forward, RG.py:761 / 781).
 * One workgroup per layer, all layers of a critic call in one launch (n_layers <= GMT_SN_MAX_LAYERS).
 * power_iterate != 0 (training mode): v <- normalize(W^T u), u <- normalize(W v) written back to
 * the u / v buffers (eps as F.normalize, 1e-12 in the reference); always: sigma = u . (W v),
 * w_out = w_orig / sigma.  sigma [1] and uv_saved [out+in] (the u, v the sigma was formed with)
 * are what gmt_spectral_norm_bwd needs; they may be NULL for an inference call.
 * Backward: d_w_orig = d_w_sn / sigma - (sum d_w_sn (.) w_orig) / sigma^2 * u v^T. */
#define GMT_SN_MAX_LAYERS 8
typedef struct gmt_sn_fwd_layer {
    const float* w_orig;   /* [out][in] */
    float* u;              /* [out]  in / out */
    float* v;              /* [in]   in / out */
    float* w_out;          /* [out][in] */
    float* sigma;          /* [1] or NULL */
    float* uv_saved;       /* [out + in] or NULL */
    int out_features, in_features;
} gmt_sn_fwd_layer;
typedef struct gmt_sn_bwd_layer {
    const float* d_w_sn;   /* [out][in] */
    const float* w_orig;
    const float* uv_saved;
    const float* sigma;
    float* d_w_orig;       /* [out][in] */
    int out_features, in_features;
} gmt_sn_bwd_layer;
int gmt_spectral_norm_fwd(const gmt_sn_fwd_layer* layers, int n_layers, float eps, int power_iterate,
                          hipStream_t stream);
int gmt_spectral_norm_bwd(const gmt_sn_bwd_layer* layers, int n_layers, hipStream_t stream);

/* ---- fused Adam -----------------------------------------------------------------------
 * Replaces torch.optim.Adam.step for g_optimizer / d_optimizer, RG.py:864-873, 1022, 1110.
 * One launch over one flat, 16-byte aligned buffer per model.  `step` is the 1-based step
 * count, grad_scale multiplies the gradient first (1/world_size under data parallelism). */
int gmt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                  hipStream_t stream);

/* ---- persistent LSTM layer (discriminator BiLSTM) ----------------------------------------
 * gmt_pack_weight: re-orders a weight block into the per-lane MFMA fragment order the
 * recurrent kernels stream (see csrc/lstm_tile.h).  The logical operand is B[k][n] =
 * src[k*stride_k + n*stride_n] with n = g*gate_stride + 32*w + j (g < n_gates, w < n_waves,
 * j < 32); rows k >= k_valid and columns n >= n_valid are zero padding (K % 8 == 0).
 * dst holds n_waves*n_gates*32*K floats.  Run once per optimizer step per weight. */
int gmt_pack_weight(const float* src, float* dst, int K, int k_valid, int n_valid, int n_waves, int n_gates,
                    int gate_stride, long long stride_k, long long stride_n, hipStream_t stream);

/* gmt_lstm_layer_fwd: one layer, ndir directions, all timesteps of a packed (per-row
 * length-masked) LSTM.  Replaces the torch.nn.LSTM call at RG.py:772 (construction
 * RG.py:720-727; gate order i,f,g,o).  H in {32, 64, 128, 256}.
 *   x        [L][B][I]   layer input (fused input contraction), or NULL with
 *   pregates [ndir][L][B][4H] = x W_ih^T + b_ih + b_hh precomputed by the caller, in the
 *            unit-major column order of the gate buffers (below)
 *   w_ih_pk  [ndir] packed (K = I,  n_gates 4, gate_stride H, stride_k 1, stride_n I)
 *   w_hh_pk  [ndir] packed (K = H,  same form)
 *   bias     [ndir][4H]  b_ih + b_hh (only read with x)
 *   lengths  [B] int32 or NULL (= L); clamped to [1, L] as RG.py:766 does
 * Gate buffers (gates, d(pre-activation), dgates_extra, g_gates; gates0/1 of the rollout) are
 * row-major [..][B][4H] with UNIT-MAJOR columns: (unit u, gate g in i,f,g,o) at 4*u + g, so
 * that a lane moves its four gate values as one float4.  A GEMM that meets such a buffer
 * uses the weight with its 4H axis permuted the same way (row 4*u+g <- torch row g*H+u).
 * outputs (each may be NULL): gates [ndir][L][B][4H] activated i,f,g,o (may alias
 * pregates), c_all [ndir][L][B][H], h_all [L][B][ndir*H] (zero at padded steps),
 * h_final [B][ndir*H] (forward: state at len-1; reverse: state after t = 0, RG.py:776-778).
 * x_dropout_p > 0 (needs x): nn.LSTM's inter-layer dropout (RG.py:725) fused into the CONSUMING
 * layer - element e of x (flat index into [L][B][I]) is read as x[e]/(1-p) if
 * hash(e, x_dropout_seed) keeps it, else 0 (counter-based mask, csrc/lstm_tile.h gmt_keep16; no mask
 * tensor exists).  The same (p, seed) then goes to the producing layer's gmt_lstm_layer_bwd and to
 * gmt_wgrad for this layer's dW_ih. */
int gmt_lstm_layer_fwd(const float* x, const float* pregates, const float* w_ih_pk, const float* w_hh_pk,
                       const float* bias, const int* lengths, float* gates, float* c_all, float* h_all,
                       float* h_final, int L, int B, int I, int H, int ndir,
                       float x_dropout_p, unsigned x_dropout_seed, hipStream_t stream);

/* gmt_lstm_xproj: the time-parallel half of the same layer, x W_ih^T + (b_ih + b_hh), for all R = L*B row-steps and
 * both directions in one launch (RG.py:772: what nn.LSTM computes per step): x [R][I], w_ih_pk as above,
 * bias [ndir][4H] or NULL -> pregates [ndir][R][4H] (unit-major columns), to be passed to gmt_lstm_layer_fwd as
 * `pregates` (x = NULL).  The gradient-penalty sub-batch (a quarter of the batch: a quarter of the CUs in the
 * persistent kernels) runs this part on the whole machine.  I in {H, 2H}, R % 32 == 0. */
int gmt_lstm_xproj(const float* x, const float* w_ih_pk, const float* bias, float* pregates,
                   long long R, int I, int H, int ndir, hipStream_t stream);

/* gmt_lstm_layer_bwd: reverse sweep of the same layer (what autograd does for RG.py:1021,
 * 1109).  In: gates / c_all saved by the forward, dh_all [L][B][ndir*H] and / or dh_final
 * [B][ndir*H] (either may be NULL), w_hh_pkT [ndir] packed (K = 4H, n_gates 1,
 * stride_k H, stride_n 1).  Out: `gates` is overwritten with d(pre-activation) (zero at
 * padded steps); the caller contracts it with x / h_prev / W_ih for dW and dx.
 * dgates_extra / dc_extra: optional additional adjoints of the saved activations (used by
 * the gradient-penalty double backward).  da_out: when non-NULL, d(pre-activation) goes
 * there and `gates` is left intact; dct_out / dht_out [ndir][L][B][H]: optional, the total
 * cell / hidden adjoint of every step (inputs of gmt_lstm_layer_bwd2).  H in {32,64,128,256}.
 * dy_dropout_p > 0: dh_all is the adjoint of the DROPPED output (the next layer ran with
 * x_dropout_p): element e of dh_all is read through the same mask.  Ordinary call only (no
 * extra adjoints / outputs). */
int gmt_lstm_layer_bwd(float* gates, const float* c_all, const float* w_hh_pkT, const int* lengths,
                       const float* dh_all, const float* dh_final, const float* dgates_extra,
                       const float* dc_extra, float* da_out, float* dct_out, float* dht_out,
                       int L, int B, int H, int ndir, float dy_dropout_p, unsigned dy_dropout_seed,
                       hipStream_t stream);

/* gmt_lstm_layer_bwd_fx (H = 128): the plain backward sweep of a two-layer stack with the top layer's INPUT gradient
 * contracted inside its time loop (what autograd computes for the input of layer 1 of torch.nn.LSTM, RG.py:772, in
 * RG.py:1021 / 1109).  Called twice:
 *   producer (top layer): dx_pairs given, dh_pairs NULL, dh_final = the adjoint of the final states, I = 2H;
 *     w_ih_pkT = per direction gmt_pack_weight of W_ih [4H][I] as it is (K = 4H gate-major, 4 waves, I / 128 gates,
 *     gate_stride 128, stride_k I, stride_n 1); dx_pairs [L*B][I][2]: slot d = direction d's partial dx (steps beyond a
 *     tile's longest row are zeroed);
 *   consumer (layer below): dh_pairs = that buffer (its two slots are added at use, through the inter-layer dropout mask
 *     when dy_dropout_p > 0), dx_pairs NULL.
 * Otherwise as gmt_lstm_layer_bwd: gates are overwritten in place by d(pre-activation). */
int gmt_lstm_layer_bwd_fx(float* gates, const float* c_all, const float* w_hh_pkT, const int* lengths,
                          const float* dh_pairs, const float* dh_final, const float* w_ih_pkT, float* dx_pairs,
                          int L, int B, int H, int I, int ndir, float dy_dropout_p, unsigned dy_dropout_seed,
                          hipStream_t stream);

/* gmt_lstm_layer_bwd2: adjoint of gmt_lstm_layer_bwd itself - the LSTM part of
 * torch.autograd.grad(..., create_graph=True) in compute_gradient_penalty, RG.py:829-835.
 * In: xhat [L][B][I] = adjoint of dX, the forward-packed weights, the saved gates / c_all
 * and the dct / dht buffers of the first backward sweep.  Out: hhat_all [L][B][ndir*H]
 * (adjoint of dh_all; may be NULL), hhat_final [B][ndir*H] (adjoint of dh_final; may be
 * NULL), g_gates [ndir][L][B][4H] and g_c [ndir][L][B][H] (adjoints of the saved
 * activations, to be passed to gmt_lstm_layer_bwd as dgates_extra / dc_extra).
 * Hoisted form: w_ih_pk == NULL and xhat = [ndir][L][B][4H] = xhat W_ih^T computed for all timesteps by
 * gmt_lstm_xproj (bias NULL); I is ignored. */
int gmt_lstm_layer_bwd2(const float* xhat, const float* w_ih_pk, const float* w_hh_pk, const int* lengths,
                        const float* gates, const float* c_all, const float* dct_all, const float* dht_all,
                        float* hhat_all, float* hhat_final, float* g_gates, float* g_c,
                        int L, int B, int I, int H, int ndir, hipStream_t stream);

/* 16-row-tile forms of the three sweeps above for the gradient-penalty pass (RG.py:790-842; H = 128, B % 16 == 0):
 * a quarter of the batch fills a quarter of the CUs with 32-row tiles, these fill twice as many (csrc/lstm_gp16.hip).
 * Hoisted inputs only: pregates / pre = the gmt_lstm_xproj result ([ndir][L][B][4H]; pregates may alias gates);
 * w_hh [ndir][4H][H] = the layer's weight_hh parameters as they are (row-major, gate order i,f,g,o - no packing).
 * _bwd is always out of place (da_out); dgates_extra / dc_extra and dct_out / dht_out are each given as a pair or not
 * at all.  Same buffers, masks and results as gmt_lstm_layer_fwd / _bwd / _bwd2 (which remain the general forms). */
int gmt_lstm16_supported(int H, int B);
int gmt_lstm16_layer_fwd(const float* pregates, const float* w_hh, const int* lengths, float* gates, float* c_all,
                         float* h_all, float* h_final, int L, int B, int H, int ndir, hipStream_t stream);
int gmt_lstm16_layer_bwd(const float* gates, const float* c_all, const float* w_hh, const int* lengths,
                         const float* dh_all, const float* dh_final, const float* dgates_extra, const float* dc_extra,
                         float* da_out, float* dct_out, float* dht_out, int L, int B, int H, int ndir,
                         hipStream_t stream);
int gmt_lstm16_layer_bwd2(const float* pre, const float* w_hh, const int* lengths, const float* gates,
                          const float* c_all, const float* dct_all, const float* dht_all, float* hhat_all,
                          float* hhat_final, float* g_gates, float* g_c, int L, int B, int H, int ndir,
                          hipStream_t stream);

/* ---- weight-gradient contraction over time -------------------------------------------------
 * dW[n][i] (+)= sum_r dA[r][n] * X[r][i],  db[n] (+)= sum_r dA[r][n]  over R = L*B row-steps: the
 * dW_ih / dW_hh / db GEMMs autograd runs for RG.py:1021, 1109.  Split-K fp32-MFMA kernel with a
 * deterministic slab reduction.  N % 4 == 0 (256-wide tiles when N % 256 == 0, else 128-wide with the
 * columns beyond N read as zero); I % 128 == 0, or I <= 32 (skinny X, any row pitch), or - round 3 - any
 * I > 32 whose rows are readable and ZERO up to the next multiple of 128 columns (ldx >= that multiple:
 * the saved activations of the dense-stack kernels); R % 16 == 0; lda / ldx = row pitches in floats (lda
 * and, for the wide tiles, ldx multiples of 4); db may be NULL; accumulate != 0 adds to dW / db.
 * gmt_wgrad_workspace_floats tells the caller how large `workspace` must be.
 * x_dropout_p > 0 (I % 128 == 0 only): X is read through the dropout mask of gmt_lstm_layer_fwd,
 * element (r, i) with counter r*ldx + i. */
int gmt_wgrad_workspace_floats(int N, int I, long long R, int* ksplit_out, long long* floats_out);
int gmt_wgrad(const float* dA, long long lda, const float* X, long long ldx, long long R, int N, int I,
              float* dW, float* db, int accumulate, float* workspace, long long workspace_floats,
              float x_dropout_p, unsigned x_dropout_seed, hipStream_t stream);

/* gmt_wgrad_gates: gmt_wgrad for a dA that is a gate buffer (unit-major columns 4 u + g, N = 4 unit_major_h): dW / db leave
 * in torch's gate-major row order (g H + u) straight from the slab reduction - no permutation pass afterwards.
 * unit_major_h == 0 is gmt_wgrad.  I % 4 == 0. */
int gmt_wgrad_gates(const float* dA, long long lda, const float* X, long long ldx, long long R, int N, int I,
                    float* dW, float* db, int accumulate, float* workspace, long long workspace_floats,
                    float x_dropout_p, unsigned x_dropout_seed, int unit_major_h, hipStream_t stream);

/* gmt_wgrad_noalloc: gmt_wgrad for callers that cannot allocate - code under HIP graph capture, tight
 * serving / training loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference (which
 * never allocates device memory itself).  Same kernels and same results as gmt_wgrad bit for bit; the split-K
 * slab buffer is a static device array inside the library (GMT_WGRAD_NOALLOC_MAX_FLOATS floats = 65 MiB on
 * each GPU the library runs on, held until the process ends) instead of a caller workspace.  Calls no
 * hipMalloc / hipFree or any allocating routine.  Shapes whose gmt_wgrad_workspace_floats exceeds the maximum
 * return GMT_ERR_WORKSPACE.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU
 * (two streams, two threads) race - serialise them.  Nothing needs clearing between calls (every slab element
 * read is written first by the same call).  gmt_wgrad_noalloc_init looks the array's device address up for
 * the current device (hipGetSymbolAddress, once per device; otherwise done by the first call on that device):
 * call it once per device before capturing a graph. */
#define GMT_WGRAD_NOALLOC_MAX_FLOATS 17039360LL
int gmt_wgrad_noalloc_init(void);
int gmt_wgrad_noalloc(const float* dA, long long lda, const float* X, long long ldx, long long R, int N, int I,
                      float* dW, float* db, int accumulate, float x_dropout_p, unsigned x_dropout_seed,
                      hipStream_t stream);

/* gmt_slab_sum: out[j] (+)= sum_k slab[k][j], j < n (n % 4 == 0, 16-byte aligned), in one fixed order - the second pass of
 * every split reduction here (gmt_wgrad's slabs, the per-tile LayerNorm parameter gradients of gmt_cond_encoder_bwd). */
int gmt_slab_sum(const float* slab, float* out, long long n, int k, int accumulate, hipStream_t stream);

/* ---- the two small dense stacks: generator condition encoder, critic output head -------------------------------
 * One launch per direction each; a workgroup owns 32 rows (trajectories) and all columns, every activation of the
 * stack stays in LDS between layers, weights are read row-major as they are (no packing).  H = 32 NW, NW in {1,2,4,8,16}.
 *
 * gmt_cond_encoder_fwd = Generator._compute_condition RG.py:469-484 (start, end, ||end - start|| = sqrt(. + 1e-8),
 * atan2(dy, dx), z -> [B, 6 + latent]) + Generator.condition_encoder RG.py:440-447 (Linear, LayerNorm, LeakyReLU(0.2),
 * Linear, LayerNorm, LeakyReLU(0.2)) + the time-invariant part of the layer-0 gate pre-activations that the rollout
 * hoists out of its loop, g0 = cond . W_ih_l0[:, 3:3+H]^T + b_ih_l0 + b_hh_l0 (the `condition` columns of the LSTM
 * input assembled at RG.py:566-569, applied by nn.LSTM at RG.py:571).  Rows >= B (batch padding up to Bp, Bp % 32 == 0)
 * leave as zeros in every output. */
typedef struct gmt_cond_fwd_args {
    const float* start; const float* end; const float* z;      /* [B][2], [B][2], [B][latent] */
    long long ld_start, ld_end, ld_z;                           /* row pitches (floats) */
    const float* w1; const float* b1; const float* ln1_g; const float* ln1_b;   /* [H][6+latent], [H], [H], [H] */
    const float* w2; const float* b2; const float* ln2_g; const float* ln2_b;   /* [H][H], [H], [H], [H] */
    const float* w_g; long long ld_wg;     /* W_ih_l0 + 3 (first condition column), row pitch 3 + H + 4; NULL: no g0 */
    const float* bias_a; const float* bias_b;                   /* b_ih_l0, b_hh_l0 [4H] */
    float* cond;            /* [Bp][H]  out */
    float* g0;              /* [Bp][4H] out (w_g given) */
    /* saved for gmt_cond_encoder_bwd: all given (training) or all NULL */
    float* c_in;            /* [Bp][c_pitch]: the 6 + latent inputs of the first Linear, zero-padded to the pitch */
    float* xhat1; float* rstd1; float* a1;                      /* [Bp][H], [Bp], [Bp][H] */
    float* xhat2; float* rstd2;                                 /* [Bp][H], [Bp] */
    int B, Bp, H, latent, c_pitch;
    int g0_unit_major;      /* 1: g0 columns 4 u + g (the fp32 rollout's gate buffers), 0: torch's g H + u */
    float slope, ln_eps, dist_eps;
} gmt_cond_fwd_args;
int gmt_cond_encoder_fwd(const gmt_cond_fwd_args* args, hipStream_t stream);

/* Backward of the same (autograd over RG.py:469-484 / 440-447 in the reference): the data path only - the adjoints of
 * the two Linear pre-activations (dy2, dy1) and per-tile partial LayerNorm parameter gradients; the caller contracts
 * d_g0 / dy2 / dy1 with cond / a1 / c_in on gmt_wgrad and sums ln_part with gmt_slab_sum.  d_g0 rows >= B are
 * zeroed IN PLACE (so that the caller's weight-gradient contraction may run over all Bp rows). */
typedef struct gmt_cond_bwd_args {
    float* d_g0;            /* [Bp][4H] gradient of g0, or NULL */
    const float* d_cond;    /* [Bp][H] gradient of cond, or NULL */
    const float* w_g; long long ld_wg;
    const float* cond; const float* xhat2; const float* rstd2; const float* ln2_g;
    const float* w2; const float* a1; const float* xhat1; const float* rstd1; const float* ln1_g;
    float* dy2; float* dy1; /* [Bp][H] out */
    float* ln_part;         /* [Bp/32][4][H] out: per tile d ln2_g, d ln2_b, d ln1_g, d ln1_b */
    int B, Bp, H, g0_unit_major;
    float slope;
} gmt_cond_bwd_args;
int gmt_cond_encoder_bwd(const gmt_cond_bwd_args* args, hipStream_t stream);

/* gmt_critic_head_fwd = Discriminator.output_layers RG.py:730-737 applied at RG.py:781: Linear(2H -> H), LeakyReLU(0.2),
 * Dropout(p), Linear(H -> H/2), LeakyReLU(0.2), Linear(H/2 -> 1) on the final hidden states; the weights are the
 * spectrally normalised ones (gmt_spectral_norm_fwd).  Dropout is a counter-based mask (gmt_keep of element
 * row * H + column, lstm_tile.h): backward regenerates it, nothing is stored; p = 0 in eval mode. */
typedef struct gmt_head_fwd_args {
    const float* x;         /* [Bp][2H] */
    const float* w1; const float* b1; const float* w2; const float* b2; const float* w3; const float* b3;
    float* score;           /* [Bp] out */
    float* a1;              /* [Bp][H] out: first activation after dropout (saved), or NULL */
    float* a2;              /* [Bp][a2_pitch] out: second activation, zero-padded to the pitch (saved), or NULL */
    int Bp, H, a2_pitch;
    float slope, dropout_p; unsigned int seed;
    /* mask mode (both given, b1..b3 unused): m1 [Bp][H], m2 [Bp][a2_pitch] = the saved a1 / a2 of an earlier call; the
     * LeakyReLUs are replaced by a multiplication with LeakyReLU'(m) (and the same dropout factor) and there are no
     * biases: score = ((x W1^T * f1) W2^T * f2) w3^T, a1 / a2 receive the two intermediate products.  This is the adjoint
     * of gmt_critic_head_bwd as a function of d_score - the head's piece of the penalty's second backward (RG.py:829-835). */
    const float* m1; const float* m2;
} gmt_head_fwd_args;
int gmt_critic_head_fwd(const gmt_head_fwd_args* args, hipStream_t stream);

/* Backward (autograd over RG.py:781): dx and the adjoints of the three Linear pre-activations, which the caller
 * contracts with x / a1 / a2 on gmt_wgrad (ds4 = d_score padded to 4 columns for that kernel's N % 4 rule). */
typedef struct gmt_head_bwd_args {
    const float* d_score;   /* [Bp] */
    const float* a1; const float* a2;
    const float* w1; const float* w2; const float* w3;
    float* dx;              /* [Bp][2H] out */
    float* dy1; float* dy2; float* ds4;   /* [Bp][H], [Bp][H/2], [Bp][4] out */
    int Bp, H, a2_pitch;
    float slope, dropout_p; unsigned int seed;
} gmt_head_bwd_args;
int gmt_critic_head_bwd(const gmt_head_bwd_args* args, hipStream_t stream);

/* ---- persistent generator rollout -----------------------------------------------------------
 * Replaces the autoregressive loop of Generator.forward, RG.py:508-604 (dynamic features
 * RG.py:486-506, nn.LSTM step RG.py:571, output_layer RG.py:460-464/574, softplus RG.py:577-580,
 * position update RG.py:585, teacher forcing RG.py:588-595) - T steps in one launch.
 * All buffers time-major; B % 32 == 0; H in {32, 64, 128, 256}; HH = H/2.  Packed operands
 * are produced by gmt_pack_weight (form given per field). */
typedef struct gmt_gen_fwd_args {
    const float* g0;        /* [B][4H]  condition . W_ih_l0[:, 3:3+H]^T + b_ih_l0 + b_hh_l0 (time-invariant), unit-major columns */
    const float* w_u_pk;    /* W_u = W_ih_l0[:, {0,1,2, 3+H..3+H+3}] ([4H][7]); K 8, k_valid 7, 4 gates, stride_n 7 */
    const float* w_hh0_pk;  /* K H, 4 gates */
    const float* w_ih1_pk;  /* K H, 4 gates */
    const float* w_hh1_pk;  /* K H, 4 gates */
    const float* bias1;     /* [4H] b_ih_l1 + b_hh_l1 */
    const float* w1_pk;     /* output_layer.0.weight [HH][H]: K H, 1 gate, n_waves max(1,H/64), n_valid HH, stride_n H */
    const float* b1;        /* [HH] */
    const float* w2;        /* output_layer.2.weight [3][HH], plain */
    const float* b2;        /* [3] */
    const float* starts;    /* [B][2] */
    const float* ends;      /* [B][2] */
    const float* init_in;   /* [3]  initial_input parameter */
    const float* target;    /* [T][B][3] or NULL */
    const unsigned char* tf_mask; /* [T] (device) or NULL: 1 = step t+1 is fed the target of step t */
    float* out_seq;         /* [T][B][3] (dx, dy, dt) */
    /* saved for the backward sweep: all nine given (training) or all NULL (no-grad rollout) */
    float* gates0; float* c0_all; float* h0_all;   /* [T][B][4H], [T][B][H], [T][B][H] */
    float* gates1; float* c1_all; float* h1_all;
    float* u_all;           /* [T][B][8]  (in0,in1,in2, rem_x, rem_y, dist, angle, 0) */
    float* y_all;           /* [T][B][HH] head activation after LeakyReLU */
    float* oraw_all;        /* [T][B]     pre-softplus dt logit */
    float* h0d_all;         /* [T][B][H]  layer-0 output after inter-layer dropout (only written when dropout_p > 0) */
    int T, B, H;
    float dropout_p;        /* inter-layer dropout of nn.LSTM(dropout=0.1), RG.py:456; 0 = off (eval / parity) */
    unsigned int seed;      /* mask = f(seed, t, row, unit): regenerated identically by the backward launch */
    /* sampling with early stop (Generator.generate, RG.py:671-686; no-grad rollout without target only):
     * stop_distance > 0: row r is done at the first step whose new position lies within stop_distance of
     * ends[r]; lengths_out[r] = that step count (T if never; rows >= rows_valid, the batch padding, count
     * as done); a 32-row tile stops once all its rows are done and its remaining out_seq steps are zero. */
    float stop_distance;
    int rows_valid;
    int* lengths_out;       /* [B] or NULL */
} gmt_gen_fwd_args;
int gmt_gen_rollout_fwd(const gmt_gen_fwd_args* args, hipStream_t stream);

/* BPTT of the same loop (autograd over RG.py:559-595 in the reference).  gates0 / gates1 are
 * overwritten with d(pre-activation) of layer 0 / 1; the caller contracts them (and d_ypre,
 * d_o) with the saved activations for the weight gradients. */
typedef struct gmt_gen_bwd_args {
    float* gates0; const float* c0_all;
    float* gates1; const float* c1_all;
    const float* y_all; const float* oraw_all; const float* u_all;
    const float* d_out;      /* [T][B][3] gradient of out_seq */
    const float* w1T_pk;     /* B[k][n] = W1[k][n]: K HH, 1 gate, n_waves H/32, stride_k H, stride_n 1 */
    const float* w_l1T_pk;   /* cat([W_hh_l1, W_ih_l1], dim=1) [4H][2H]: K 4H, 2 gates, gate_stride H, stride_k 2H */
    const float* w_hh0T_pk;  /* B[k][n] = W_hh_l0[k][n]: K 4H, 1 gate, stride_k H */
    const float* w_uP;       /* [4H][8] plain: W_u zero-padded to 8 columns */
    const float* w2;         /* [3][HH] */
    const unsigned char* tf_mask;
    float* d_ypre;           /* [T][B][HH] out */
    float* d_o;              /* [T][B][3] out: gradient of the head's raw output */
    float* d_in0;            /* [B][3] out: gradient of the initial input, per row */
    float* d_g0;             /* [B][4H] out or NULL (H <= 128): gradient of g0 = sum over t of d(pre-activation of layer 0),
                              * unit-major columns like g0 - accumulated on chip, no reduction pass over gates0 afterwards */
    int T, B, H;
    float dropout_p; unsigned int seed;   /* as in the forward launch */
} gmt_gen_bwd_args;
int gmt_gen_rollout_bwd(const gmt_gen_bwd_args* args, hipStream_t stream);

/* ---- bf16-operand / fp32-accumulate path (BASELINE.json configs[2]: H = 512, T = 256, "bf16 compute / fp32 accum") ----
 * At H = 512 the recurrent weights (4H x H = 2 MB in bf16) no longer fit a CU, so this family is organised differently
 * from the persistent fp32 kernels above: everything that does not sit on the recurrence is hoisted into dense GEMMs over
 * all L*B row-steps, and the recurrence itself is one launch per timestep (the kernel boundary is the grid-wide hand-off
 * of h_t).  Activations and weights are bf16 in memory, accumulators / cell state / gradients of parameters are fp32.
 * All matrices are row-major with the contraction index contiguous; M % 256 == 0, N % 256 == 0, K % 64 == 0. */

/* C[m][n] = epilogue(sum_k A[m][k] W[n][k] (+ bias[n])): nn.LSTM's x W_ih^T + b for all timesteps (RG.py:720-727), the
 * feature encoder's Linears (RG.py:712-717, epilogue 1 = LeakyReLU) and autograd's dX = dA W (W passed transposed;
 * epilogue 2 multiplies by LeakyReLU'(act[m][n])).  A, W, C, act: bf16; bias: fp32 or NULL. */
int gmt_bf16_gemm_nt(const void* A, long long lda, const void* W, long long ldw, void* C, long long ldc,
                     const float* bias, const void* act, long long ldact, int epilogue, float slope,
                     long long M, int N, int K, hipStream_t stream);

/* One bidirectional (ndir = 2) or unidirectional LSTM layer over all L timesteps, one launch per timestep (csrc/lstm_bf16.hip;
 * nn.LSTM on the packed sequence, RG.py:720-727, 764-772).  Layouts, time-major, B % 256 == 0, H % 128 == 0:
 *   gates [L][B][ndir*4H] bf16 - in: x W_ih^T + b_ih + b_hh from gmt_bf16_gemm_nt, out: the activated i,f,g,o; the column of
 *         (direction d, unit u, gate g) is d*4H + (u/32)*128 + g*32 + u%32 ("slab" order; the weights' 4H axis is permuted the
 *         same way by the caller);
 *   c_all [L][B][ndir*H] fp32 and h_all [L][B][ndir*H] bf16 (out), column d*H + u;  w_hh [ndir][4H slab order][H] bf16.
 * ragged != 0: lengths[b] <= L valid steps per row (direction 0 freezes the state of a finished row, so h_all[L-1] holds its
 * final state; direction 1 starts at t = lengths[b]-1 from zero state, final state = h_all[0]).
 * drop_p > 0: h_drop [L][B][ndir*H] bf16 (out) = h_all after nn.LSTM's inter-layer dropout (RG.py:725) - the next layer's
 * input; the mask is the counter-based function of (drop_seed, element index) the fp32 kernels use (gmt_keep16). */
int gmt_bf16_lstm_layer_fwd(void* gates, float* c_all, void* h_all, void* h_drop, const void* w_hh, const int* lengths,
                            int ragged, int L, int B, int H, int ndir, float drop_p, unsigned int drop_seed,
                            hipStream_t stream);
/* Backward sweep of the same layer (autograd's BPTT): gates is overwritten in place with d(pre-activation) - the dA that the
 * weight-gradient and input-gradient GEMMs then contract.  dh_out [L][B][ndir*H] bf16 or NULL: gradient of h_all from the layer
 * above; dh_fin [ndir][B][H] fp32: gradient of the two final hidden states; dc_carry (and, for ragged batches, dh_carry)
 * [ndir][B][H] fp32: scratch carried from launch to launch (no initialisation needed);  w_hhT [ndir][H][4H slab order] bf16.
 * drop_p > 0: dh_out is the gradient of the forward launch's h_drop; the same mask is applied to it on the way in. */
int gmt_bf16_lstm_layer_bwd(void* gates, const float* c_all, const void* dh_out, const float* dh_fin, float* dc_carry,
                            float* dh_carry, const void* w_hhT, const int* lengths, int ragged, int L, int B, int H,
                            int ndir, float drop_p, unsigned int drop_seed, hipStream_t stream);

/* Row groups (no counterpart in the reference: a scheduling knob of this library).  The rows of a batch are independent
 * recurrences, so gmt_bf16_lstm_layer_fwd / _bwd and gmt_bf16_gen_rollout_fwd / _bwd can run a call as two independent chains of
 * step launches over B / 2 rows each - chain 0 on the caller's stream, chain 1 on a stream the library owns (one per device,
 * created on first use, default priority, non-blocking), forked from and joined to the caller's stream by events, so the call
 * stays in order for the caller.  The workgroups of one launch move in lock-step (contraction, then the HBM-heavy epilogue);
 * two chains overlap one's epilogue and launch boundary with the other's MFMAs.  Results do not depend on n (bit-identical).
 * n = 1 or 2 (default 2, or the environment's GMT_BF16_ROW_GROUPS); a call uses one chain when B / 2 is not a multiple of four
 * row tiles.  The library deliberately owns ONE extra stream per device: a process has GPU_MAX_HW_QUEUES (4) hardware queues,
 * and once default-priority streams fill them (three library streams + the default stream) a later two-stream sequence of
 * the fp32 path (Trainer.overlap_g_forward's high-priority torch stream) never completes on this stack; with
 * GPU_MAX_HW_QUEUES=8 it does.
 * With n = 2 these calls must not be made under single-stream HIP graph capture. */
int gmt_bf16_set_row_groups(int n);

/* Gradient penalty at this width (compute_gradient_penalty, RG.py:790-842).  The two entry points below sweep full-length
 * rows; the host runs ragged batches on them behind a pre-activation mask (ops_bf16._mask_g_: the hoisted part of gate g of
 * direction 1 is zeroed at t >= len, which keeps that direction's state exactly zero there).
 * gmt_bf16_lstm_layer_bwd_gp: the backward sweep out of place (gates is kept, dA [L][B][ndir*4H] bf16 is written), which can also
 * keep the per-step totals dct_all / dht_all [L][B][ndir*H] fp32 the second-order sweep needs (first backward under
 * create_graph=True), and take the adjoints of the saved activations as extra inputs (gg_in [L][B][ndir*4H], gc_in
 * [L][B][ndir*H], fp32; the pass after the second-order sweep).
 * gmt_bf16_lstm_layer_bwd2: the second-order sweep.  pre_hat [L][B][ndir*4H] bf16 = xhat W_ih^T (xhat = adjoint of the sweep's
 * dX); outputs hhat_all [L][B][ndir*H] bf16 (adjoint of dh_out = the xhat of the layer above), hhat_fin [ndir][B][H] fp32 (adjoint
 * of dh_fin), g_gates / g_c (adjoints of the saved activated gates / cell states, fp32).  chat_carry [ndir][B][H] fp32: scratch. */
int gmt_bf16_lstm_layer_bwd_gp(const void* gates, void* dA, const float* c_all, const void* dh_out, const float* dh_fin,
                               const float* gg_in, const float* gc_in, float* dct_all, float* dht_all, float* dc_carry,
                               const void* w_hhT, int L, int B, int H, int ndir, hipStream_t stream);
int gmt_bf16_lstm_layer_bwd2(const void* pre_hat, const void* gates, const float* c_all, const float* dct_all,
                             const float* dht_all, void* hhat_all, float* hhat_fin, float* g_gates, float* g_c,
                             float* chat_carry, const void* w_hh, int L, int B, int H, int ndir, hipStream_t stream);

/* Weight gradients of the bf16 path: slabs[s][n][i] = sum over the s-th of n_slabs row ranges of dA[r][n] * X[r][i]
 * (autograd's dW = dA^T X for the LSTM and encoder weights; the caller sums the n_slabs fp32 partial results - deterministic).
 * dA [M][lda], X [M][ldx] bf16; M % 64 == 0, N % 256 == 0, I % 256 == 0. */
int gmt_bf16_wgrad_tn(const void* dA, long long lda, const void* X, long long ldx, float* slabs,
                      long long M, int N, int I, int n_slabs, hipStream_t stream);

/* Generator rollout at H = 512 (bf16 operands / fp32 accumulation), no-grad form: Generator.forward under torch.no_grad (the fake
 * batch of a D step, RG.py:940-945) and Generator.generate (RG.py:606-690); three launches per timestep (csrc/gen_bf16.hip).
 * Weights (bf16, 4H axis in the slab order of gmt_bf16_lstm_layer_fwd): w0 [4H][H+64] = [W_hh_l0 | W_u | W_u | 0] with W_u the
 * 7 input columns (cur 3, dyn 4) of weight_ih_l0; w1 [4H][2H] = [W_ih_l1 | W_hh_l1]; wh1 [H/2][H] = output_layer.0.weight.
 * cond_pre [B][4H] bf16 = condition W_cond^T + b_ih_l0 + b_hh_l0 (slab order; constant over time, RG.py:564-568).
 * Workspace, caller-initialised: x0 [2][B][H+64] bf16 = zeros except x0[0][:, H:H+14] = the first step's 7 inputs as bf16 hi then
 * lo parts (initial_input, end - start, its norm with eps 1e-8 inside the root, atan2); x1 [2][B][2H] bf16, c0 / c1 [2][B][H] fp32
 * = zeros; pos [2][B][2] fp32 = starts twice (position, teacher-forced position).  B % 256 == 0, H == 512. */
typedef struct gmt_bf16_gen_args {
    const void* cond_pre; const void* w0; const void* w1; const float* bias1;
    const void* wh1; const float* bh1; const float* wh2; const float* bh2;
    const float* ends;            /* [B][2] */
    const float* target;          /* [T][B][3] or NULL */
    const unsigned char* tf_mask; /* [T] (device) or NULL: 1 = step t+1 is fed the target of step t (RG.py:588-595) */
    float* out_seq;               /* [T][B][3] (dx, dy, dt) */
    void* x0; void* x1; float* c0; float* c1; float* pos;
    /* training rollout (all four given, else all NULL): every step keeps its state for the BPTT sweep - x0 is then
     * [T+1][B][GMT_BF16_GEN_X0_TRAIN_WIDTH] (slot t = the operand of step t; the extra width makes it a weight-gradient tile
     * multiple), x1 [T+1][B][2H], c0 / c1 [T+1][B][H] (slot t = state BEFORE step t; slots 0 caller-zeroed), and */
    void* gates0; void* gates1;   /* [T][B][4H] bf16: activated i,f,g,o of layer 0 / 1 (slab order) */
    void* y_all;                  /* [T][B][H/2] bf16: head activation after LeakyReLU */
    float* oraw_all;              /* [T][B]: pre-softplus dt logit */
    int T, B, H;
    float dropout_p;              /* inter-layer dropout of nn.LSTM(dropout=0.1), RG.py:451-457; 0 = off */
    unsigned int seed;            /* mask = f(seed, (t*B + row)*H + unit): regenerated by the backward launch */
} gmt_bf16_gen_args;
#define GMT_BF16_GEN_X0_TRAIN_WIDTH 768
int gmt_bf16_gen_rollout_fwd(const gmt_bf16_gen_args* args, hipStream_t stream);

/* BPTT of the training rollout above (autograd over RG.py:559-595): three launches per timestep in reverse order.  gates0 /
 * gates1 are overwritten with d(pre-activation) of layer 0 / 1; the caller contracts them (and dypre_all, d_o) with the saved
 * operands x0 / x1 / y_all for the weight gradients (gmt_bf16_wgrad_tn).  Transposed weights, bf16, K-contiguous, 4H axis in
 * slab order: w1T [H][H/2] = output_layer.0.weight^T, whh1T / wih1T / whh0T [H][4H], w_u [4H][8] (7 input columns + a zero).
 * tf_mask_host: the T teacher-forcing flags as a HOST array (they steer pointer choices).  Scratch, no initialisation needed:
 * dc0 / dc1 [B][H] fp32, p_carry [B][2] fp32. */
typedef struct gmt_bf16_gen_bwd_args {
    void* gates0; void* gates1;
    const float* c0; const float* c1;     /* [T+1][B][H] from the forward launch */
    const void* x0;                       /* [T+1][B][GMT_BF16_GEN_X0_TRAIN_WIDTH] from the forward launch */
    const void* y_all; const float* oraw_all;
    const float* d_out;                   /* [T][B][3] gradient of out_seq */
    const void* w1T; const void* whh1T; const void* wih1T; const void* whh0T; const void* w_u; const float* wh2;
    void* dypre_all;                      /* [T][B][H/2] bf16 out: gradient of the head's pre-activation */
    float* d_o;                           /* [T][B][3] out: gradient of the head's raw output */
    float* d_in0;                         /* [B][3] out: gradient of the initial input, per row */
    float* dc0; float* dc1; float* p_carry;
    const unsigned char* tf_mask_host;
    int T, B, H;
    float dropout_p; unsigned int seed;   /* as in the forward launch */
} gmt_bf16_gen_bwd_args;
int gmt_bf16_gen_rollout_bwd(const gmt_bf16_gen_bwd_args* args, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GMT_HIP_H */
