"""The mouse-telemetry on-disk format (the 23-column CSV written by
crates/mouse-telemetry/src/mouse_telemetry.rs:763-870, "MT.rs") around the GPU metric
kernels: read a recording, recompute its metric columns on the MI355X, emit generated
trajectories in the same format.

A *collection* is what MT.rs writes per batch (MT.rs:504-520: the events up to and including a
left click): its first row has total_duration == 0 (MT.rs:570-574), which is also how
RecurrentGAN.py:151 splits a file into trajectories.  Here a collection is a dict of
equally long columns: float64 numpy arrays for the 21 numeric columns, lists of str for
click_events / scroll_events.

Only the I/O and the text formatting run on the host; every metric value comes from
gmt_telemetry_kinematics_f64 / gmt_telemetry_derived_f64 (no CPU arithmetic path).
"""
from __future__ import annotations

import csv
import os
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L_
from . import telemetry as TEL

# MT.rs:807-831, in file order
COLUMNS = ('timestamp', 'total_duration', 'time_between_movements', 'hover_time', 'position_x', 'position_y',
           'velocity', 'acceleration', 'jerk', 'path_length', 'click_events', 'scroll_events',
           'average_velocity', 'peak_velocity', 'average_acceleration', 'peak_acceleration',
           'average_momentum', 'peak_momentum', 'smoothness', 'deviation_from_ideal_path', 'idle_time',
           'fitts_index_of_difficulty', 'fitts_movement_time')
TEXT_COLUMNS = ('click_events', 'scroll_events')
NUMERIC_COLUMNS = tuple(c for c in COLUMNS if c not in TEXT_COLUMNS)
KINEMATIC_COLUMNS = ('velocity', 'acceleration', 'jerk', 'path_length')        # recomputed, MT.rs:550-654
DERIVED_COLUMNS = TEL.DERIVED_NAMES                                           # recomputed, MT.rs:664-756

Collection = Dict[str, object]


def format_f64(v: float) -> str:
    """Rust's `f64::to_string()` (MT.rs:843-864): the shortest digit string that round-trips,
    always positional (no exponent), no trailing '.0'."""
    v = float(v)
    if v != v:
        return 'NaN'
    if v in (float('inf'), float('-inf')):
        return 'inf' if v > 0 else '-inf'
    return np.format_float_positional(v, unique=True, trim='-')


def format_timestamp(secs: float) -> str:
    """MT.rs:768-775: seconds since the epoch with four decimals."""
    return f'{float(secs):.4f}'


# ------------------------------------------------------------------------------------------
# reading
# ------------------------------------------------------------------------------------------
def read_collections(path: str) -> List[Collection]:
    """Parse one CSV into its collections (split at rows with total_duration == 0)."""
    out: List[Collection] = []
    cur: Optional[Dict[str, list]] = None
    with open(path, 'r', newline='') as fh:
        rd = csv.reader(fh)
        header = next(rd, None)
        if header is None:
            return out
        if tuple(header) != COLUMNS:
            raise ValueError(f'{path}: not a mouse-telemetry recording (header {header[:3]}...)')
        for row in rd:
            if len(row) != len(COLUMNS):
                raise ValueError(f'{path}: row with {len(row)} fields')
            if cur is None or float(row[1]) == 0.0:
                if cur is not None:
                    out.append(_finish(cur))
                cur = {c: [] for c in COLUMNS}
            for c, tok in zip(COLUMNS, row):
                cur[c].append(tok if c in TEXT_COLUMNS else float(tok))
    if cur is not None:
        out.append(_finish(cur))
    return out


def _finish(cur: Dict[str, list]) -> Collection:
    return {c: (list(v) if c in TEXT_COLUMNS else np.asarray(v, np.float64)) for c, v in cur.items()}


# ------------------------------------------------------------------------------------------
# writing
# ------------------------------------------------------------------------------------------
def write_collections(path: str, collections: Iterable[Collection], append: bool = True) -> int:
    """Append collections to `path` the way MT.rs:763-870 does (header only into an empty file,
    '\\n' record terminator).  Returns the number of rows written."""
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    mode = 'a' if append else 'w'
    rows = 0
    with open(path, mode, newline='') as fh:
        wr = csv.writer(fh, lineterminator='\n')
        if fh.tell() == 0:
            wr.writerow(COLUMNS)
        for col in collections:
            n = len(col['position_x'])
            cells = []
            for c in COLUMNS:
                if c in TEXT_COLUMNS:
                    cells.append([str(s) for s in col[c]])
                elif c == 'timestamp':
                    cells.append([format_timestamp(v) for v in col[c]])
                else:
                    cells.append([format_f64(v) for v in col[c]])
                if len(cells[-1]) != n:
                    raise ValueError(f'column {c}: {len(cells[-1])} rows, expected {n}')
            wr.writerows(zip(*cells))
            rows += n
    return rows


# ------------------------------------------------------------------------------------------
# GPU recompute
# ------------------------------------------------------------------------------------------
def _planes(cols: Sequence[np.ndarray], n_max: int, device) -> torch.Tensor:
    """ragged per-collection columns -> one time-major padded plane [n_max, n_traj] on the GPU."""
    host = np.zeros((n_max, len(cols)), np.float64)
    for r, a in enumerate(cols):
        host[:len(a), r] = a
    return torch.from_numpy(host).to(device)


def compute_metrics(xs: Sequence[np.ndarray], ys: Sequence[np.ndarray], dts: Sequence[np.ndarray],
                    total_durations: Optional[Sequence[np.ndarray]] = None,
                    device: str = 'cuda') -> List[Dict[str, np.ndarray]]:
    """Metric columns of a ragged batch of trajectories from raw (x, y, time_between_movements):
    one launch each of the kinematics and the derived-statistics kernel over the padded batch.
    total_durations: the recorded t_i - t_0 column when there is one (MT.rs derives idle_time and
    fitts_movement_time from timestamps); otherwise the kernel's own running sum of dt is used."""
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise L_.GmtError('telemetry metrics are computed on the GPU; there is no CPU path')
    n = len(xs)
    lens = np.array([len(a) for a in xs], np.int32)
    if n == 0:
        return []
    if any(len(y) != l or len(t) != l for y, t, l in zip(ys, dts, lens)):
        raise ValueError('x, y and time_between_movements of a trajectory must be equally long')
    n_max = int(lens.max())
    tx, ty, tdt = _planes(xs, n_max, dev), _planes(ys, n_max, dev), _planes(dts, n_max, dev)
    tl = torch.from_numpy(lens).to(dev)
    out = TEL.telemetry_kinematics(tx, ty, tdt, tl, time_is_dt=True, return_aux=True)
    ttot = out['total_duration'] if total_durations is None else _planes(total_durations, n_max, dev)
    der = TEL.telemetry_derived(tx, ty, tdt, ttot, out['kin6'], tl)
    kin6 = out['kin6'].cpu().numpy()
    hov, tot, der = out['hover_time'].cpu().numpy(), ttot.cpu().numpy(), der.cpu().numpy()
    res = []
    for r, l in enumerate(lens):
        m = {k: kin6[j, :l, r].copy() for j, k in enumerate(TEL.KIN_NAMES)}
        m['hover_time'] = hov[:l, r].copy()
        m['total_duration'] = tot[:l, r].copy()
        for j, k in enumerate(DERIVED_COLUMNS):
            m[k] = der[j, :l, r].copy()
        res.append(m)
    return res


def recompute_collections(collections: Sequence[Collection], device: str = 'cuda') -> List[Collection]:
    """The same collections with hover_time, velocity, acceleration, jerk, path_length and the 11
    derived columns recomputed on the GPU from position_x, position_y, time_between_movements
    (and the recorded total_duration).  The timestamp column is not a time input: it is rounded
    to 1e-4 s on disk (MT.rs:774)."""
    ms = compute_metrics([c['position_x'] for c in collections], [c['position_y'] for c in collections],
                         [c['time_between_movements'] for c in collections],
                         [c['total_duration'] for c in collections], device)
    out = []
    for c, m in zip(collections, ms):
        d = dict(c)
        for k in KINEMATIC_COLUMNS + ('hover_time',) + tuple(DERIVED_COLUMNS):
            d[k] = m[k]
        out.append(d)
    return out


def collections_from_trajectories(xs: Sequence[np.ndarray], ys: Sequence[np.ndarray], dts: Sequence[np.ndarray],
                                  t0: float = 0.0, device: str = 'cuda') -> List[Collection]:
    """Generated trajectories (absolute pixel positions and per-step dt, dt[0] ignored) as
    collections: metrics from the GPU kernels, consecutive timestamps starting at t0, the last
    event of each trajectory marked as the left click that ends a batch (MT.rs:504-520)."""
    xs = [np.asarray(a, np.float64) for a in xs]
    ys = [np.asarray(a, np.float64) for a in ys]
    dts = [np.concatenate([[0.0], np.asarray(a, np.float64)[1:]]) for a in dts]
    ms = compute_metrics(xs, ys, dts, None, device)
    out, t = [], float(t0)
    for x, y, dt, m in zip(xs, ys, dts, ms):
        n = len(x)
        col: Collection = {'timestamp': t + np.cumsum(dt), 'total_duration': m['total_duration'],
                           'time_between_movements': dt, 'hover_time': m['hover_time'],
                           'position_x': x, 'position_y': y,
                           'click_events': ['None'] * (n - 1) + ['Left'], 'scroll_events': ['None'] * n}
        for k in KINEMATIC_COLUMNS + tuple(DERIVED_COLUMNS):
            col[k] = m[k]
        out.append(col)
        t = float(col['timestamp'][-1]) if n else t
    return out
