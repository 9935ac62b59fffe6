"""GPU recompute of the mouse-telemetry kinematic metrics from raw (x, y, t) batches
(crates/mouse-telemetry/src/mouse_telemetry.rs:550-756 semantics, f64).  Inputs are
time-major padded planes [n_max, n_traj] on the GPU."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L_

MIN_DERIVATIVE_DT = 0.004   # MT.rs:31
SMOOTHING_ALPHA = 0.3       # MT.rs:33
KIN_NAMES = ('velocity', 'acceleration', 'jerk', 'path_length', 'direction', 'curvature')
DERIVED_NAMES = ('average_velocity', 'peak_velocity', 'average_acceleration', 'peak_acceleration',
                 'average_momentum', 'peak_momentum', 'smoothness', 'deviation_from_ideal_path',
                 'idle_time', 'fitts_index_of_difficulty', 'fitts_movement_time')


def telemetry_kinematics(x: torch.Tensor, y: torch.Tensor, tm: torch.Tensor, lengths: Optional[torch.Tensor] = None,
                         time_is_dt: bool = False, min_dt: float = MIN_DERIVATIVE_DT, alpha: float = SMOOTHING_ALPHA,
                         return_aux: bool = False, out: Optional[torch.Tensor] = None):
    """Returns kin6 [6, n_max, n_traj] (KIN_NAMES order); with return_aux a dict that also
    holds hover_time and total_duration planes."""
    L_.require_cuda(x, y, tm)
    assert x.dtype == torch.float64 and x.dim() == 2 and x.shape == y.shape == tm.shape
    x, y, tm = x.contiguous(), y.contiguous(), tm.contiguous()
    n_max, n_traj = x.shape
    kin6 = out if out is not None else torch.empty(6, n_max, n_traj, device=x.device, dtype=torch.float64)
    hov = tot = None
    if return_aux:
        hov = torch.empty_like(x)
        tot = torch.empty_like(x)
    if lengths is not None:
        lengths = lengths.to(torch.int32).contiguous()
    L_.call('gmt_telemetry_kinematics_f64', L_.ptr(x), L_.ptr(y), L_.ptr(tm), L_.ptr(lengths), L_.ptr(kin6),
            L_.ptr(hov), L_.ptr(tot), n_traj, n_max, int(time_is_dt), min_dt, alpha, L_.stream(),
            nbytes=72.0 * n_traj * n_max)
    if return_aux:
        return {'kin6': kin6, 'hover_time': hov, 'total_duration': tot}
    return kin6


def telemetry_derived(x, y, raw_dt, total_duration, kin6, lengths=None):
    """Returns [11, n_max, n_traj] in DERIVED_NAMES (CSV column) order, MT.rs:664-756."""
    L_.require_cuda(x, y, raw_dt, total_duration, kin6)
    n_max, n_traj = x.shape
    out = torch.empty(11, n_max, n_traj, device=x.device, dtype=torch.float64)
    if lengths is not None:
        lengths = lengths.to(torch.int32).contiguous()
    L_.call('gmt_telemetry_derived_f64', L_.ptr(x.contiguous()), L_.ptr(y.contiguous()), L_.ptr(raw_dt.contiguous()),
            L_.ptr(total_duration.contiguous()), L_.ptr(kin6.contiguous()), L_.ptr(lengths), L_.ptr(out),
            n_traj, n_max, L_.stream())
    return out
