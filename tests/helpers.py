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
