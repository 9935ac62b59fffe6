"""ctypes binding of oracle/telemetry_oracle.c (TEST INFRASTRUCTURE ONLY).

Restates crates/mouse-telemetry/src/mouse_telemetry.rs:550-756; see the header
of telemetry_oracle.c for the parity status (pinned by the reference CSVs)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libtelemetry_oracle.so')
MIN_DT = 0.004      # MT.rs:31
ALPHA = 0.3         # MT.rs:33
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'telemetry_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'libtelemetry_oracle.so'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def kinematics(x, y, tm, time_is_dt=True, min_dt=MIN_DT, alpha=ALPHA):
    """One trajectory.  Returns dict of float64 arrays."""
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    tm = np.ascontiguousarray(tm, np.float64)
    n = x.shape[0]
    names = ('velocity', 'acceleration', 'jerk', 'path_length', 'direction', 'curvature',
             'hover_time', 'total_duration')
    out = {k: np.zeros(n, np.float64) for k in names}
    lib().tel_kinematics(ctypes.c_int(n), _p(x), _p(y), _p(tm), ctypes.c_int(int(time_is_dt)),
                         ctypes.c_double(min_dt), ctypes.c_double(alpha), *[_p(out[k]) for k in names])
    return out


DERIVED = ('average_velocity', 'peak_velocity', 'average_acceleration', 'peak_acceleration',
           'average_momentum', 'peak_momentum', 'smoothness', 'deviation_from_ideal_path',
           'idle_time', 'fitts_index_of_difficulty', 'fitts_movement_time')


def derived(x, y, raw_dt, total_duration, velocity, acceleration, jerk, path_length):
    arrs = [np.ascontiguousarray(a, np.float64) for a in
            (x, y, raw_dt, total_duration, velocity, acceleration, jerk, path_length)]
    n = arrs[0].shape[0]
    out = np.zeros((n, 11), np.float64)
    lib().tel_derived(ctypes.c_int(n), *[_p(a) for a in arrs], _p(out))
    return out


def kinematics_batch(x, y, tm, lengths=None, time_is_dt=False, min_dt=MIN_DT, alpha=ALPHA):
    """Time-major padded batch [n_max, n_traj] -> [6, n_max, n_traj]."""
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    tm = np.ascontiguousarray(tm, np.float64)
    n_max, n_traj = x.shape
    out = np.zeros((6, n_max, n_traj), np.float64)
    lp = None
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, np.int32)
        lp = lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    lib().tel_kinematics_batch(ctypes.c_int(n_traj), ctypes.c_int(n_max), lp, _p(x), _p(y), _p(tm),
                               ctypes.c_int(int(time_is_dt)), ctypes.c_double(min_dt),
                               ctypes.c_double(alpha), _p(out))
    return out
