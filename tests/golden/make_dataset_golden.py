#!/usr/bin/env python3
"""Golden vectors for the data path: the REAL reference's MouseTrajectoryDataset and
collate_trajectories (RecurrentGAN.py:98-308) run on tests/golden/telemetry_sample.csv - an excerpt of the
reference's own data file - with and without the x4 mirror augmentation.  Build container only:

    MPLBACKEND=Agg python tests/golden/make_dataset_golden.py      ->  tests/golden/dataset.npz

Only inputs (the CSV excerpt, already a fixture) and outputs are stored; nothing of the reference's source."""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference  # noqa: E402


def main():
    ref = load_reference()
    out = {}
    csv_path = os.path.join(HERE, 'telemetry_sample.csv')
    for tag, augment, normalize in (('aug', True, True), ('plain', False, True), ('raw', False, False)):
        cfg = ref.Config()
        with contextlib.redirect_stdout(io.StringIO()):
            ds = ref.MouseTrajectoryDataset(csv_path, cfg, normalize=normalize, augment=augment)
        n = len(ds)
        out[f'{tag}_n'] = np.asarray(n)
        out[f'{tag}_lengths'] = np.asarray([ds[i]['length'] for i in range(n)], dtype=np.int64)
        # every trajectory, through the reference's collate (sorted by length, zero padded), in two batches
        for b, idx in enumerate((list(range(0, n, 2)), list(range(1, n, 2)))):
            col = ref.collate_trajectories([ds[i] for i in idx])
            for k, v in col.items():
                out[f'{tag}_b{b}_{k}'] = v.numpy()
    np.savez_compressed(os.path.join(HERE, 'dataset.npz'), **out)
    print({k: v.shape for k, v in out.items() if k.endswith('sequences')}, [int(out[f'{t}_n']) for t in ('aug', 'plain', 'raw')])


if __name__ == '__main__':
    main()
