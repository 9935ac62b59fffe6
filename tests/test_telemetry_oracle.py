"""Pins oracle/telemetry_oracle.c to the reference's own CSV data (the inputs and
outputs of mouse_telemetry.rs:550-756).  CPU only."""
import numpy as np

import helpers as H
from oracle import telemetry as TO


def segments(g):
    d = g['data']
    cols = list(g['columns'])
    fid = g['file_id']
    td = d[:, cols.index('total_duration')]
    starts = [i for i in range(len(d)) if td[i] == 0 and (i == 0 or True)]
    # a segment starts at every row with total_duration == 0 (RecurrentGAN.py:151) or at a file change
    cut = sorted(set(starts) | {i for i in range(1, len(d)) if fid[i] != fid[i - 1]} | {0})
    cut.append(len(d))
    return d, cols, [(cut[i], cut[i + 1]) for i in range(len(cut) - 1) if cut[i + 1] > cut[i]]


def test_kinematics_bit_exact_all_rows():
    g = H.load('telemetry.npz')
    d, cols, segs = segments(g)
    c = {k: d[:, cols.index(k)] for k in cols}
    assert len(segs) == 307 and len(d) == 13115
    worst = 0.0
    for a, b in segs:
        o = TO.kinematics(c['position_x'][a:b], c['position_y'][a:b], c['time_between_movements'][a:b], True)
        for k in ('velocity', 'acceleration', 'jerk', 'path_length', 'hover_time'):
            assert np.array_equal(o[k], c[k][a:b]), (k, a, b)
        # total_duration re-summed from dt: not bit exact (the CSV holds t_i - t_0)
        err = np.abs(o['total_duration'] - c['total_duration'][a:b]).max()
        worst = max(worst, err / max(c['total_duration'][a:b].max(), 1e-30))
        dv = TO.derived(c['position_x'][a:b], c['position_y'][a:b], c['time_between_movements'][a:b],
                        c['total_duration'][a:b], o['velocity'], o['acceleration'], o['jerk'], o['path_length'])
        for j, k in enumerate(TO.DERIVED):
            ref = c[k][a:b]
            if k == 'idle_time':
                # the reference subtracts timestamps; only the raw dt survives in the CSV
                assert np.abs(dv[:, j] - ref).max() <= 1e-9, k
            else:
                assert np.abs(dv[:, j] - ref).max() <= 4e-16 * max(np.abs(ref).max(), 1.0), (k, a, b)
    assert worst < 1e-14


def test_timestamp_mode_and_batch_layout():
    g = H.load('telemetry.npz')
    d, cols, segs = segments(g)
    c = {k: d[:, cols.index(k)] for k in cols}
    segs = segs[:40]
    n_max = max(b - a for a, b in segs)
    x = np.zeros((n_max, len(segs))); y = np.zeros_like(x); t = np.zeros_like(x)
    lens = np.array([b - a for a, b in segs], np.int32)
    for r, (a, b) in enumerate(segs):
        x[:b - a, r] = c['position_x'][a:b]; y[:b - a, r] = c['position_y'][a:b]
        tt = np.cumsum(c['time_between_movements'][a:b]); tt[0] = 0.0
        t[:b - a, r] = tt
    out = TO.kinematics_batch(x, y, t, lens, time_is_dt=False)
    for r, (a, b) in enumerate(segs):
        for j, k in enumerate(('velocity', 'acceleration', 'jerk', 'path_length')):
            ref = c[k][a:b]
            # BASELINE tolerance: 1e-6 relative to the Rust output
            assert np.abs(out[j, :b - a, r] - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-30), (k, r)
        assert np.all(out[:, b - a:, r] == 0.0)
        o = TO.kinematics(x[:b - a, r], y[:b - a, r], t[:b - a, r], False)
        assert np.array_equal(o['direction'], out[4, :b - a, r])
        assert np.array_equal(o['curvature'], out[5, :b - a, r])
