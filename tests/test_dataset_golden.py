"""Data path against the reference itself: MouseTrajectoryDataset + collate_trajectories
of this package on tests/golden/telemetry_sample.csv must reproduce, bit for bit, what the reference's own classes
(RecurrentGAN.py:98-308) produced on the same file (tests/golden/make_dataset_golden.py imported them in the build
container): trajectory count, lengths, and the collated starts / ends / sequences / lengths with and without the x4
mirror augmentation and without normalisation."""
import contextlib
import io
import os

import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize('tag,augment,normalize', [('aug', True, True), ('plain', False, True), ('raw', False, False)])
def test_dataset_and_collate_match_the_reference_bit_for_bit(tag, augment, normalize):
    from generative_mouse_trajectories_amd.config import Config
    from generative_mouse_trajectories_amd.data import MouseTrajectoryDataset, collate_trajectories
    g = H.load('dataset.npz')
    with contextlib.redirect_stdout(io.StringIO()) as log:
        ds = MouseTrajectoryDataset(os.path.join(H.GOLD, 'telemetry_sample.csv'), Config(), normalize=normalize, augment=augment)
    n = int(g[f'{tag}_n'])
    assert len(ds) == n and n > 0
    assert 'base trajectories' in log.getvalue()
    assert np.array_equal(np.asarray([ds[i]['length'] for i in range(n)]), g[f'{tag}_lengths'])
    for b, idx in enumerate((list(range(0, n, 2)), list(range(1, n, 2)))):
        col = collate_trajectories([ds[i] for i in idx])
        assert sorted(col) == ['ends', 'lengths', 'sequences', 'starts']
        for k, v in col.items():
            ref = g[f'{tag}_b{b}_{k}']
            assert v.numpy().dtype == ref.dtype and v.numpy().shape == ref.shape, (k, v.dtype, ref.dtype, v.shape, ref.shape)
            assert np.array_equal(v.numpy(), ref), f'{tag} batch {b} {k}: max diff {np.abs(v.numpy() - ref).max()}'
