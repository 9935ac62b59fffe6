"""Deterministic parameter / batch generators shared by the golden-vector
script (which feeds them to the real reference) and by the tests (which feed
them to the oracle and to the HIP path).  numpy PCG64 only, so the values do
not depend on the torch version."""
from __future__ import annotations

import math
import os
import sys
from typing import Dict

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
from oracle import gan_oracle as O  # noqa: E402


def _fill(rng, name: str, shape, hidden: int) -> np.ndarray:
    k = 1.0 / math.sqrt(hidden)
    if name == 'initial_input':
        return rng.uniform(-0.05, 0.05, size=shape)
    if name.startswith('condition_encoder.1') or name.startswith('condition_encoder.4'):
        if name.endswith('weight'):
            return 1.0 + rng.uniform(-0.1, 0.1, size=shape)
        return rng.uniform(-0.1, 0.1, size=shape)
    if name.endswith('weight_u') or name.endswith('weight_v'):
        v = rng.normal(size=shape)
        return v / max(np.linalg.norm(v), 1e-12)
    return rng.uniform(-k, k, size=shape)


def make_generator_params(cfg, seed: int) -> Dict[str, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(seed))
    H = cfg.generator_hidden_dim
    return {n: _fill(rng, n, s, H).astype(np.float32) for n, s in O.generator_shapes(cfg).items()}


def make_discriminator_params(cfg, seed: int):
    rng = np.random.Generator(np.random.PCG64(seed))
    H = cfg.discriminator_hidden_dim
    ps, bs = O.discriminator_shapes(cfg)
    P = {n: _fill(rng, n, s, H).astype(np.float32) for n, s in ps.items()}
    Bf = {n: _fill(rng, n, s, H).astype(np.float32) for n, s in bs.items()}
    return P, Bf


def make_batch(B: int, T: int, seed: int, lengths=None) -> Dict[str, np.ndarray]:
    """Synthetic batch in the collate_trajectories contract (RG.py:303-308):
    sorted by length descending, zero padded (SURVEY.md section 8d generator)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    starts = rng.uniform(0.0, 1.0, size=(B, 2))
    deltas = rng.normal(0.0, 0.01, size=(B, T, 2))
    dts = rng.uniform(0.004, 0.024, size=(B, T))
    if lengths is None:
        lengths = np.full((B,), T, dtype=np.int64)
    lengths = np.sort(np.asarray(lengths, dtype=np.int64))[::-1].copy()
    mask = (np.arange(T)[None, :] < lengths[:, None])
    deltas = deltas * mask[:, :, None]
    dts = dts * mask
    ends = starts + deltas.sum(axis=1)
    seq = np.concatenate([deltas, dts[:, :, None]], axis=-1)
    return {'starts': starts.astype(np.float32), 'ends': ends.astype(np.float32),
            'sequences': seq.astype(np.float32), 'lengths': lengths}


def digest(a: np.ndarray, n: int = 16) -> np.ndarray:
    """[l2 norm, sum, n strided samples] in float64."""
    f = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, f.size - 1, num=min(n, f.size)).astype(np.int64)
    out = np.zeros(2 + n, dtype=np.float64)
    out[0] = np.sqrt((f * f).sum())
    out[1] = f.sum()
    out[2:2 + idx.size] = f[idx]
    return out
