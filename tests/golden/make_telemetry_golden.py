#!/usr/bin/env python3
"""Parse the reference's own telemetry CSVs (inputs AND outputs of
mouse_telemetry.rs:550-756) into one compact float64 fixture.  Build container only.

    python tests/golden/make_telemetry_golden.py
"""
import csv
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
COLS = ['total_duration', 'time_between_movements', 'hover_time', 'position_x', 'position_y',
        'velocity', 'acceleration', 'jerk', 'path_length',
        'average_velocity', 'peak_velocity', 'average_acceleration', 'peak_acceleration',
        'average_momentum', 'peak_momentum', 'smoothness', 'deviation_from_ideal_path',
        'idle_time', 'fitts_index_of_difficulty', 'fitts_movement_time']


def main():
    rows, file_id, clicks = [], [], []
    files = sorted(glob.glob('/root/reference/data/*.csv'))
    for fi, f in enumerate(files):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append([float(r[c]) for c in COLS])
                file_id.append(fi)
                clicks.append(r['click_events'] == 'Left')
    a = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'telemetry.npz'), columns=np.array(COLS), data=a,
                        file_id=np.array(file_id, np.int8), left_click=np.array(clicks),
                        files=np.array([os.path.basename(f) for f in files]))
    print(a.shape)
    # a verbatim excerpt (data rows only: the reference's recorded CSV text) for the reader / writer tests:
    # the first three collections of the oldest and of the newest recording
    lines = []
    for f in (files[0], files[-1]):
        with open(f) as fh:
            header = fh.readline()
            if not lines:
                lines.append(header)
            n_coll = 0
            for ln in fh:
                if ln.split(',')[1] == '0':
                    n_coll += 1
                    if n_coll > 3:
                        break
                lines.append(ln)
    with open(os.path.join(HERE, 'telemetry_sample.csv'), 'w', newline='') as fh:
        fh.writelines(lines)
    print(len(lines) - 1, 'sample rows')


if __name__ == '__main__':
    main()
