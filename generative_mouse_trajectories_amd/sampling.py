"""Generator-only sampling at scale + GPU recompute of the telemetry metrics (BASELINE.json configs[4]):
`Generator.generate` (RecurrentGAN.py:606-690) without the early-stop shortcut - every trajectory runs all
`max_generation_steps` in one launch of the no-grad rollout kernel - followed by the f64 kinematic metrics of
mouse-telemetry (mouse_telemetry.rs:550-654) on the generated (x, y, t), all on the GPU:

    rollout kernel -> gmt_rollout_to_xyt_f64 (f64 pixel planes) -> gmt_telemetry_kinematics_f64
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib as L_
from . import ops
from .telemetry import telemetry_kinematics


@torch.no_grad()
def sample_rollout(generator, start: torch.Tensor, end: torch.Tensor, z: Optional[torch.Tensor] = None,
                   steps: Optional[int] = None) -> torch.Tensor:
    """start / end (N,2), z (N,latent) -> out (T,Np,3) time-major (dx, dy, dt), Np = N padded to 32 rows.
    Eval mode (no dropout), no target, no early stop."""
    N = start.shape[0]
    if z is None:
        z = torch.randn(N, generator.config.latent_dim, device=start.device)
    T = int(steps or generator.config.max_generation_steps)
    was_training = generator.training
    generator.eval()
    try:
        return generator._rollout(start, end, z, None, None, T, padded=True)
    finally:
        if was_training:
            generator.train()


def rollout_to_xyt(out_tm: torch.Tensor, start: torch.Tensor, screen_w: float, screen_h: float):
    """out_tm (T,Np,3), start (N,2) -> x, y, dt f64 planes (T+1, N): pixel positions incl. the start point and the
    raw time between movements (dt[0] = 0)."""
    L_.require_cuda(out_tm, start)
    T, Np, _ = out_tm.shape
    N = start.shape[0]
    assert out_tm.is_contiguous() and N <= Np
    x, y, dt = (torch.empty(T + 1, N, device=out_tm.device, dtype=torch.float64) for _ in range(3))
    L_.call('gmt_rollout_to_xyt_f64', L_.ptr(out_tm), 3, 3 * Np, L_.ptr(start.contiguous().float()), L_.ptr(x), L_.ptr(y),
            L_.ptr(dt), T, N, float(screen_w), float(screen_h), L_.stream())
    return x, y, dt


@torch.no_grad()
def sample_and_measure(generator, start: torch.Tensor, end: torch.Tensor, z: Optional[torch.Tensor] = None,
                       steps: Optional[int] = None, kin_out: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """One config-5 launch group: N trajectories x `steps` generated, then their six kinematic metric planes
    [6, steps + 1, N] (velocity, acceleration, jerk, path_length, direction, curvature)."""
    cfg = generator.config
    out = sample_rollout(generator, start, end, z, steps)
    x, y, dt = rollout_to_xyt(out, start, cfg.screen_width, cfg.screen_height)
    kin = telemetry_kinematics(x, y, dt, None, time_is_dt=True, out=kin_out)
    return {'out': out, 'x': x, 'y': y, 'dt': dt, 'kin6': kin}
