"""Twice-differentiable discriminator forward out of plain torch ops ON THE GPU, used only by
the WGAN-GP gradient penalty (RecurrentGAN.py:790-842), which needs the gradient of a
gradient through the BiLSTM, the spectral-norm linears and the kinematic features.

This is the interim path for the penalty sub-batch (25 % of the batch every 4th step): the
hand-written kernels of ops.py implement first-order backward only.  It is not a CPU
fallback - it refuses CPU tensors like the rest of the package - and it shares the
Discriminator's parameters and spectral-norm buffers (one power iteration per call, like
any other forward)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib as L_


def kinematics_torch(pos: torch.Tensor, dts: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """Time-major RecurrentGAN.py:315-389: pos [L,B,2], dts [L,B] -> [L,B,9]."""
    x, y = pos[..., 0], pos[..., 1]
    z1 = torch.zeros_like(x[:1])
    dx = torch.cat([z1, x[1:] - x[:-1]], 0)
    dy = torch.cat([z1, y[1:] - y[:-1]], 0)
    td = dts + eps
    vel = torch.sqrt(dx * dx + dy * dy + eps) / td
    vx, vy = dx / td, dy / td
    dt_mid = (dts[1:] + dts[:-1]) / 2 + eps
    acc = torch.cat([z1, (vel[1:] - vel[:-1]) / dt_mid], 0)
    if pos.shape[0] >= 3:
        jerk = torch.cat([z1, z1, (acc[2:] - acc[1:-1]) / (dt_mid[1:] + eps)], 0)
    else:
        jerk = torch.zeros_like(acc)
    ax = torch.cat([z1, (vx[1:] - vx[:-1]) / dt_mid], 0)
    ay = torch.cat([z1, (vy[1:] - vy[:-1]) / dt_mid], 0)
    curv = (vx * ay - vy * ax).abs() / (vx * vx + vy * vy + eps) ** 1.5
    return torch.stack([x, y, dx, dy, dts, vel.clamp(0, 100), acc.clamp(-1000, 1000),
                        jerk.clamp(-10000, 10000), curv.clamp(0, 100)], dim=-1)


GP_KERNEL_HEAD = True           # False: the critic head of the penalty pass as torch ops (A/B switch)
GP_KERNEL_FRONT_END = True      # False: the torch-composite front end (kept for widths the kernels do not cover, and as an A/B switch)


def discriminator_forward_gp(D, trajectories: torch.Tensor, dts: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """Twice-differentiable discriminator forward for the gradient penalty: the BiLSTM runs in
    the HIP kernels (ops.LSTMLayerGPFn: forward, first backward and the second-order sweep
    are each one persistent launch per layer), and so do the front end (ops.EncoderGPFn) and the
    output head (ops.CriticHeadGPFn) when their widths are in the kernels' tables - torch ops otherwise, which autograd
    differentiates twice; the spectral-norm weights come from gmt_spectral_norm_fwd and are only differentiated once (they
    do not depend on the input)."""
    from . import ops
    from .models import _pad_batch
    L_.require_cuda(trajectories, dts)
    cfg = D.config
    B = trajectories.shape[0]
    Bp = ops.pad_rows(B)
    pos = _pad_batch(trajectories.transpose(0, 1), Bp, 1)
    d = _pad_batch(dts.transpose(0, 1), Bp, 1)
    lens = _pad_batch(lengths.to(device=trajectories.device, dtype=torch.int32).clamp(min=1), Bp, 0, 1).contiguous()
    sn_w = D.spectral_weights()
    fe = D.feature_encoder
    H = cfg.discriminator_hidden_dim
    if (GP_KERNEL_FRONT_END and not getattr(D, 'bf16', False) and ops.encoder_gp_supported(H)
            and fe[3].negative_slope == fe[1].negative_slope):
        # kinematics + both Linears + both LeakyReLUs in one launch, twice differentiable through hand-written
        # first / second backward passes (ops.EncoderGPFn): no torch composite on the fp32 penalty path
        w1, w2 = sn_w.pop(0), sn_w.pop(0)
        x = ops.encoder_gp(pos.contiguous(), d.contiguous(), w1, fe[0].bias, w2, fe[2].bias, float(fe[1].negative_slope), 1e-8)
    else:
        x = D.run_sequential(D.feature_encoder, kinematics_torch(pos, d), sn_w)
    nl = cfg.discriminator_num_layers
    if getattr(D, 'bf16', False):
        # H = 512 (BASELINE configs[2]): the whole BiLSTM stack is ONE twice-differentiable op on the bf16 step kernels
        from . import ops_bf16
        Bq = (B + ops_bf16.ROW_TILE - 1) // ops_bf16.ROW_TILE * ops_bf16.ROW_TILE
        if Bq != Bp:
            x = _pad_batch(x[:, :B], Bq, 1)
            lens = _pad_batch(lens[:B], Bq, 0, 1)
        lw = [w for l in range(nl) for w in D.lstm.layer_weights(l)]
        drop_p = cfg.lstm_dropout if (D.training and nl > 1) else 0.0
        h_final = ops_bf16.lstm_stack_gp(x.contiguous(), nl, lw, lens, drop_p)
        return _gp_head(D, h_final, sn_w, B)
    h_final = None
    for l in range(nl):
        h_all, h_final, _, _ = ops.LSTMLayerGPFn.apply(x.contiguous(), lens, *D.lstm.layer_weights(l))
        x = h_all
        if l + 1 < nl and D.training and cfg.lstm_dropout > 0:
            x = F.dropout(x, cfg.lstm_dropout, True)
    return _gp_head(D, h_final, sn_w, B)


def _gp_head(D, h_final: torch.Tensor, sn_w, B: int) -> torch.Tensor:
    """output_layers on the (padded) final states of the penalty pass -> scores of the B valid rows."""
    from . import ops
    ol = D.output_layers
    H = D.config.discriminator_hidden_dim
    if (GP_KERNEL_HEAD and ops.critic_head_supported(H) and len(ol) == 6 and len(sn_w) == 3 and h_final.dtype == torch.float32
            and h_final.shape[0] % ops.TM == 0 and ol[1].negative_slope == ol[4].negative_slope):
        # the 3-Linear head in kernels too, twice differentiable (ops.CriticHeadGPFn / CriticHeadBwdGPFn); its Dropout is
        # the counter-based mask of the ordinary critic call, drawn per call
        p = float(ol[2].p) if D.training else 0.0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0
        w1, w2, w3 = sn_w
        return ops.critic_head_gp(h_final, w1, ol[0].bias, w2, ol[3].bias, w3, ol[5].bias, float(ol[1].negative_slope), p, seed)[:B]
    return D.run_sequential(D.output_layers, h_final[:B], sn_w)
