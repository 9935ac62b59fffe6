#!/usr/bin/env python3
"""BASELINE.json configs[0] as a golden LOSS CURVE: the real reference (RecurrentGAN.py) trained on CPU on its own recording
data/mouse_data_20250813_235208.csv with default hyper-parameters (G 256 / D 128, latent 64, B = 32), 8 iterations of
train_discriminator_step + train_generator_step (RG.py:1130-1142) over consecutive batches of the reference's own
MouseTrajectoryDataset / collate_trajectories, dropout zeroed (the determinism recipe of SURVEY.md section 8c), every RNG
draw recorded.  Run in the build container only:

    MPLBACKEND=Agg python tests/golden/make_config1_golden.py

Writes tests/golden/config1_curve.npz: the batches (inputs), the recorded noise and the per-iteration metric dicts (outputs).
Nothing of the reference's source is stored."""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

CSV = '/root/reference/data/mouse_data_20250813_235208.csv'
ITERS, B = 8, 32
EPOCHS = [0, 0, 10, 10, 30, 30, 60, 60]          # teacher-forcing ratio 1.0 .. 0.4 along the curve (RG.py:903-911)


def main():
    ref = MG.load_reference()
    cfg = ref.Config()
    assert (cfg.generator_hidden_dim, cfg.discriminator_hidden_dim, cfg.latent_dim, cfg.batch_size) == (256, 128, 64, 32)
    ds = ref.MouseTrajectoryDataset(CSV, cfg)
    n = len(ds)
    print('dataset samples:', n)
    G, D, Pg, Pd, Bd = MG.build_models(ref, cfg, 301, 302)
    torch.manual_seed(9876)
    tr = ref.Trainer(G, D, cfg, torch.device('cpu'), '/tmp/runs_golden_c1')
    G.train(); D.train()
    out = {'epochs': np.array(EPOCHS), 'n_samples': np.array(n)}
    for it in range(ITERS):
        idx = [(it * B + j) % n for j in range(B)]
        batch = ref.collate_trajectories([ds[i] for i in idx])
        for k, v in batch.items():
            out[f'it{it}_batch/{k}'] = v.numpy().copy()
        Tn = batch['sequences'].shape[1]
        tr.epoch = EPOCHS[it]
        for kind in ('D', 'G'):
            st = torch.get_rng_state()
            d_step_before = tr.d_step
            m = tr.train_discriminator_step(batch) if kind == 'D' else tr.train_generator_step(batch)
            after = torch.get_rng_state()
            torch.set_rng_state(st)                 # replay the draws to record them (SURVEY.md section 8c order)
            out[f'it{it}_{kind}_z'] = torch.randn(B, cfg.latent_dim).numpy()
            coins = np.array([torch.rand(1).item() for _ in range(Tn)])
            if kind == 'D':
                if d_step_before % cfg.gp_every_n == 0:
                    b = max(1, int(math.ceil(B * cfg.gp_batch_frac)))
                    if b < B:
                        out[f'it{it}_D_gp_idx'] = torch.randperm(B).numpy()
                    out[f'it{it}_D_alpha'] = torch.rand(b, 1, 1).reshape(b).numpy()
            else:
                out[f'it{it}_G_tf_mask'] = coins < tr.get_teacher_forcing_ratio()
            assert torch.equal(torch.get_rng_state(), after), 'RNG replay out of sync'
            keys = sorted(m.keys())
            out[f'it{it}_{kind}_metric_names'] = np.array(keys)
            out[f'it{it}_{kind}_metrics'] = np.array([m[k] for k in keys], dtype=np.float64)
            print(it, kind, {k: round(m[k], 6) for k in keys}, flush=True)
    np.savez_compressed(os.path.join(HERE, 'config1_curve.npz'), **out)
    print('config1_curve.npz', len(out))


if __name__ == '__main__':
    main()
