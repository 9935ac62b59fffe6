"""Shared test helpers: golden loading, oracle parameter construction, comparison."""
from __future__ import annotations

import os

import numpy as np
import torch

import paramgen as PG
from oracle import gan_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

TINY = dict(latent_dim=8, generator_hidden_dim=32, discriminator_hidden_dim=32)
MID = dict(latent_dim=64, generator_hidden_dim=128, discriminator_hidden_dim=128)
DEFAULT = dict()
TINY_LENGTHS = [9, 9, 7, 5, 4, 1]
MID_LENGTHS = [12, 12, 12, 11, 9, 8, 6, 5]
DEFAULT_LENGTHS = [8, 8, 6, 5]


def load(name):
    return np.load(os.path.join(GOLD, name))


def t(a, dtype=torch.float32):
    a = np.asarray(a)
    if a.dtype.kind in 'iub':
        return torch.from_numpy(a.copy())
    return torch.from_numpy(a.copy()).to(dtype)


def oracle_models(cfg_over, seed_g, seed_d, dtype=torch.float32):
    cfg = O.OracleConfig(**cfg_over)
    Pg = {k: t(v, dtype) for k, v in PG.make_generator_params(cfg, seed_g).items()}
    pd, bd = PG.make_discriminator_params(cfg, seed_d)
    Pd = {k: t(v, dtype) for k, v in pd.items()}
    Bd = {k: t(v, dtype) for k, v in bd.items()}
    return cfg, Pg, Pd, Bd


def batch_t(B, T, seed, lengths, dtype=torch.float32):
    b = PG.make_batch(B, T, seed, lengths)
    return {k: t(v, dtype) for k, v in b.items()}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    return float(np.abs(a - b).max() / den) if b.size else 0.0


def assert_close(a, b, rtol, name='', atol_frac=None):
    """max-norm relative error: |a-b|_inf <= rtol * |b|_inf."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    assert np.asarray(a).shape == np.asarray(b).shape, f'{name}: shape {np.asarray(a).shape} vs {np.asarray(b).shape}'
    e = relerr(a, b)
    assert e <= rtol, f'{name}: rel err {e:.3e} > {rtol:.1e}'
    return e


def curve_check(it, kind, got, ref_names, ref_vals, bar, exact=None):
    """Config-1 loss curve (tests/golden/config1_curve.npz): a step's metrics against the reference's.  Scores, penalty and
    auxiliary losses relative to their own size; d_loss is the difference of two scores that are ~100 x larger than it (an
    untrained critic scores real and fake alike), so it is held to the same ABSOLUTE error as the scores it is made of.
    exact (optional): the same step's metrics from the fp64 oracle.  Where the reference's own fp32 value is further than
    `bar` from it - the penalty of a later iteration is such a quantity: clamps in the kinematic features make the critic's
    input gradient discontinuous, 790.9 in the reference, 950.7 in fp64 - the value must be no further from the fp64 one than
    1.25 x the reference is (the rule of test_gpu_parity._replay).  Returns the worst error / bar ratio among the metrics
    held to `bar`."""
    ref = dict(zip(list(ref_names), [float(v) for v in ref_vals]))
    worst = 0.0
    for k, v in ref.items():
        scale = abs(v)
        if k == 'd_loss':
            scale = max(abs(ref['d_real_score']), abs(ref['d_fake_score']))
        if k == 'g_loss':
            scale = abs(ref['g_fake_score'])
        scale = max(scale, 1e-3)
        e = abs(float(got[k]) - v) / scale
        if e > bar and exact is not None and abs(v - float(exact[k])) / scale > bar:
            e_ref, e_got = abs(v - float(exact[k])) / scale, abs(float(got[k]) - float(exact[k])) / scale
            assert e_got <= 1.25 * e_ref + bar, (f'iteration {it} {kind} step, {k}: {got[k]} is {e_got:.2e} from the fp64 oracle '
                                                 f'({exact[k]}), the reference ({v}) {e_ref:.2e}')
            continue
        worst = max(worst, e / bar)
        assert e <= bar, f'iteration {it} {kind} step, {k}: {got[k]} vs {v} (error {e:.2e} of scale, bar {bar:.1e})'
    return worst


def curve_oracle64(g, n_iter):
    """The fp64 oracle's metrics for the first n_iter iterations of the config-1 curve: [(D metrics, G metrics)]."""
    import torch
    from oracle import gan_oracle as O
    dt = torch.float64
    cfg, Pg, Pd, Bd = oracle_models(DEFAULT, 301, 302, dt)
    opt_g, opt_d, out = {}, {}, []
    cv = lambda v: v.to(dt) if v.dtype.is_floating_point else v
    for it in range(n_iter):
        batch = {k: cv(t(g[f'it{it}_batch/{k}'])) for k in ('starts', 'ends', 'sequences', 'lengths')}
        noise = {'z': cv(t(g[f'it{it}_D_z']))}
        if f'it{it}_D_alpha' in g:
            noise['alpha'] = cv(t(g[f'it{it}_D_alpha']))
            noise['gp_idx'] = t(g[f'it{it}_D_gp_idx'])
        md, _, _ = O.discriminator_step(Pg, Pd, Bd, opt_d, batch, noise, cfg, it)
        ratio = O.teacher_forcing_ratio(int(g['epochs'][it]), cfg)
        mg, _, _ = O.generator_step(Pg, Pd, Bd, opt_g, batch, {'z': cv(t(g[f'it{it}_G_z']))}, cfg, list(g[f'it{it}_G_tf_mask']), ratio)
        out.append((md, mg))
    return out
