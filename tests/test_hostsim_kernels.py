"""CPU checks of the element-wise kernel bodies (same source the gfx950 kernels compile),
built for the host with g++ (tests/hostsim/hostsim.cpp), against the golden vectors and
the oracle.  No GPU."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch

import helpers as H
from oracle import gan_oracle as O
from oracle import telemetry as TO

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'hostsim', 'libhostsim.so')


@pytest.fixture(scope='module')
def sim():
    src = os.path.join(HERE, 'hostsim', 'hostsim.cpp')
    subprocess.check_call(['g++', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-std=c++17',
                           '-Dunroll=', '-Wno-unknown-pragmas', '-o', SO, src])
    return ctypes.CDLL(SO)


def fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_kinematics_bodies(sim):
    g = H.load('kinematics.npz')
    for name in ('kat', 'a', 'b', 'c', 'd', 'e', 'pad'):
        pos = g[f'kin_{name}_pos']; dts = g[f'kin_{name}_dts']
        B, L, _ = pos.shape
        pos_t = np.ascontiguousarray(pos.transpose(1, 0, 2)); dts_t = np.ascontiguousarray(dts.T)
        feat = np.zeros((L, B, 9), np.float32)
        sim.sim_kin_fwd(fp(pos_t), fp(dts_t), fp(feat), L, B, ctypes.c_float(1e-8))
        H.assert_close(feat.transpose(1, 0, 2), g[f'kin_{name}_feat'], 2e-6, f'fwd {name}')
        w = np.ascontiguousarray(g[f'kin_{name}_w'].transpose(1, 0, 2))
        gpos = np.zeros((L, B, 2), np.float32); gdts = np.zeros((L, B), np.float32)
        sim.sim_kin_bwd(fp(pos_t), fp(dts_t), fp(w), fp(gpos), fp(gdts), L, B, ctypes.c_float(1e-8))
        H.assert_close(gpos.transpose(1, 0, 2), g[f'kin_{name}_gpos'], 2e-5, f'gpos {name}')
        H.assert_close(gdts.T, g[f'kin_{name}_gdts'], 2e-5, f'gdts {name}')


def test_kinematics_jvp_against_fp64_oracle(sim):
    """kin_jvp_vals (the kinematics piece of the gradient penalty's second backward): fhat = J(pos, dts) phat against
    torch.autograd.functional.jvp of the oracle's compute_kinematics restatement in fp64, and the adjoint identity
    <gfeat, J phat> = <J^T gfeat, phat> against the backward body (the pair the penalty differentiates through)."""
    rng = np.random.default_rng(5)
    B, L = 9, 37
    pos = (rng.uniform(0, 1, (B, 1, 2)) + np.cumsum(rng.normal(0, 0.01, (B, L, 2)), 1)).astype(np.float32)
    dts = rng.uniform(0.002, 0.03, (B, L)).astype(np.float32)
    dts[:, 5] *= 20.0                                   # a few slow steps: some features inside, some outside their clamps
    phat = rng.normal(size=(B, L, 2)).astype(np.float32)
    p64, d64 = torch.tensor(pos, dtype=torch.float64), torch.tensor(dts, dtype=torch.float64)
    _, ref = torch.autograd.functional.jvp(lambda q: O.kinematics(q, d64), (p64,), (torch.tensor(phat, dtype=torch.float64),))
    pos_t = np.ascontiguousarray(pos.transpose(1, 0, 2)); dts_t = np.ascontiguousarray(dts.T)
    ph_t = np.ascontiguousarray(phat.transpose(1, 0, 2))
    fhat = np.zeros((L, B, 9), np.float32)
    sim.sim_kin_jvp(fp(pos_t), fp(dts_t), fp(ph_t), fp(fhat), L, B, ctypes.c_float(1e-8))
    got = fhat.transpose(1, 0, 2)
    feat = O.kinematics(p64, d64).numpy()
    inside = np.ones_like(got, dtype=bool)              # compare where the fp64 feature is clear of a clamp boundary
    for c, (lo, hi) in {5: (0, 100), 6: (-1000, 1000), 7: (-10000, 10000), 8: (0, 100)}.items():
        inside[:, :, c] = (np.abs(feat[:, :, c] - lo) > 1e-3 * (hi - lo)) & (np.abs(feat[:, :, c] - hi) > 1e-3 * (hi - lo))
    assert inside.mean() > 0.9 and (feat[:, :, 7] == 10000).any()
    for c in range(9):
        r, gt = ref.numpy()[:, :, c][inside[:, :, c]], got[:, :, c][inside[:, :, c]]
        assert np.abs(gt - r).max() <= 2e-4 * max(np.abs(r).max(), 1e-30), (c, np.abs(gt - r).max(), np.abs(r).max())
    w = rng.normal(size=(B, L, 9)).astype(np.float32)
    gpos = np.zeros((L, B, 2), np.float32); gdts = np.zeros((L, B), np.float32)
    sim.sim_kin_bwd(fp(pos_t), fp(dts_t), fp(np.ascontiguousarray(w.transpose(1, 0, 2))), fp(gpos), fp(gdts), L, B, ctypes.c_float(1e-8))
    lhs = float((w.astype(np.float64) * got).sum()); rhs = float((gpos.transpose(1, 0, 2).astype(np.float64) * phat).sum())
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs)), (lhs, rhs)


def test_kinematics_bwd_against_fp64_oracle(sim):
    rng = np.random.default_rng(3)
    B, L = 7, 40
    pos = (rng.uniform(0, 1, (B, 1, 2)) + np.cumsum(rng.normal(0, 0.01, (B, L, 2)), 1)).astype(np.float32)
    dts = rng.uniform(0.002, 0.03, (B, L)).astype(np.float32)
    w = rng.normal(size=(B, L, 9)).astype(np.float32)
    p = torch.tensor(pos, dtype=torch.float64, requires_grad=True)
    d = torch.tensor(dts, dtype=torch.float64, requires_grad=True)
    (O.kinematics(p, d) * torch.tensor(w, dtype=torch.float64)).sum().backward()
    pos_t = np.ascontiguousarray(pos.transpose(1, 0, 2)); dts_t = np.ascontiguousarray(dts.T)
    gpos = np.zeros((L, B, 2), np.float32); gdts = np.zeros((L, B), np.float32)
    sim.sim_kin_bwd(fp(pos_t), fp(dts_t), fp(np.ascontiguousarray(w.transpose(1, 0, 2))), fp(gpos), fp(gdts),
                    L, B, ctypes.c_float(1e-8))
    H.assert_close(gpos.transpose(1, 0, 2), p.grad.numpy(), 1e-4, 'gpos vs fp64')
    H.assert_close(gdts.T, d.grad.numpy(), 1e-4, 'gdts vs fp64')


def test_telemetry_bodies_bit_exact(sim):
    import test_telemetry_oracle as TT
    g = H.load('telemetry.npz')
    d, cols, segs = TT.segments(g)
    c = {k: d[:, cols.index(k)] for k in cols}
    n_max = max(b - a for a, b in segs)
    n = len(segs)
    x = np.zeros((n_max, n)); y = np.zeros_like(x); dt = np.zeros_like(x); tdur = np.zeros_like(x)
    lens = np.array([b - a for a, b in segs], np.int32)
    for r, (a, b) in enumerate(segs):
        x[:b - a, r] = c['position_x'][a:b]; y[:b - a, r] = c['position_y'][a:b]
        dt[:b - a, r] = c['time_between_movements'][a:b]; tdur[:b - a, r] = c['total_duration'][a:b]
    out6 = np.zeros((6, n_max, n)); hov = np.zeros((n_max, n)); tot = np.zeros((n_max, n))
    sim.sim_tel_kin(fp(x), fp(y), fp(dt), fp(lens), fp(out6), fp(hov), fp(tot), ctypes.c_longlong(n), n_max, 1,
                    ctypes.c_double(TO.MIN_DT), ctypes.c_double(TO.ALPHA))
    out11 = np.zeros((11, n_max, n))
    sim.sim_tel_derived(fp(x), fp(y), fp(dt), fp(tdur), fp(out6), fp(lens), fp(out11), ctypes.c_longlong(n), n_max)
    for r, (a, b) in enumerate(segs):
        for j, k in enumerate(('velocity', 'acceleration', 'jerk', 'path_length')):
            assert np.array_equal(out6[j, :b - a, r], c[k][a:b]), (k, r)
        assert np.array_equal(hov[:b - a, r], c['hover_time'][a:b])
        for j, k in enumerate(TO.DERIVED):
            ref = c[k][a:b]
            tol = 1e-9 if k == 'idle_time' else 4e-16 * max(np.abs(ref).max(), 1.0)
            assert np.abs(out11[j, :b - a, r] - ref).max() <= tol, (k, r)
        assert np.all(out6[:, b - a:, r] == 0)


def test_adam_body_matches_torch(sim):
    rng = np.random.default_rng(0)
    n = 1003
    p0 = rng.normal(size=n).astype(np.float32)
    P = {'w': torch.tensor(p0.copy())}
    st = {}
    p = p0.copy(); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    tp = torch.nn.Parameter(torch.tensor(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-4, betas=(0.5, 0.9))
    for step in range(1, 6):
        gr = (rng.normal(size=n) * 10.0 ** rng.uniform(-6, 0, size=n)).astype(np.float32)
        O.adam_update(P, {'w': torch.tensor(gr)}, st, 1e-4, (0.5, 0.9))
        tp.grad = torch.tensor(gr); opt.step()
        bc1 = 1 - 0.5 ** step; bc2 = 1 - 0.9 ** step
        sim.sim_adam(fp(p), fp(gr), fp(m), fp(v), ctypes.c_longlong(n), ctypes.c_float(0.5), ctypes.c_float(0.9),
                     ctypes.c_float(1e-8), ctypes.c_float(1e-4 / bc1), ctypes.c_float(math.sqrt(bc2)))
        np.testing.assert_allclose(P['w'].numpy(), tp.detach().numpy(), rtol=0, atol=1e-9)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1.3e-7, atol=2e-9)   # <= 1 ulp
        np.testing.assert_allclose(m, st['w']['exp_avg'].numpy(), rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(v, st['w']['exp_avg_sq'].numpy(), rtol=1e-6, atol=1e-20)
