"""GPU parity of the individual HIP kernels (through the C ABI) against the CPU oracle and
the golden vectors of the reference."""
import math

import numpy as np
import pytest
import torch

import helpers as H
from oracle import gan_oracle as O
from oracle import telemetry as TO

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    from generative_mouse_trajectories_amd import ops as ops_
    from generative_mouse_trajectories_amd import _lib
    _lib.lib()
    return ops_


def test_kinematics_golden(ops):
    g = H.load('kinematics.npz')
    for name in ('kat', 'a', 'b', 'c', 'd', 'e', 'pad'):
        pos = H.t(g[f'kin_{name}_pos']).to(DEV).transpose(0, 1).contiguous().requires_grad_(True)
        dts = H.t(g[f'kin_{name}_dts']).to(DEV).t().contiguous().requires_grad_(True)
        feat = ops.kinematics_tm(pos, dts)
        H.assert_close(feat.transpose(0, 1), g[f'kin_{name}_feat'], 2e-6, f'feat {name}')
        w = H.t(g[f'kin_{name}_w']).to(DEV).transpose(0, 1)
        (feat * w).sum().backward()
        H.assert_close(pos.grad.transpose(0, 1), g[f'kin_{name}_gpos'], 2e-5, f'gpos {name}')
        H.assert_close(dts.grad.t(), g[f'kin_{name}_gdts'], 2e-5, f'gdts {name}')


def test_kinematics_large_vs_oracle(ops):
    torch.manual_seed(0)
    B, L = 300, 129
    pos = (torch.rand(B, 1, 2) + torch.cumsum(torch.randn(B, L, 2) * 0.01, 1))
    dts = torch.rand(B, L) * 0.02 + 0.004
    w = torch.randn(B, L, 9)
    p64 = pos.double().requires_grad_(True); d64 = dts.double().requires_grad_(True)
    f64 = O.kinematics(p64, d64)
    (f64 * w.double()).sum().backward()
    pg = pos.to(DEV).transpose(0, 1).contiguous().requires_grad_(True)
    dg = dts.to(DEV).t().contiguous().requires_grad_(True)
    f = ops.kinematics_tm(pg, dg)
    (f * w.to(DEV).transpose(0, 1)).sum().backward()
    H.assert_close(f.transpose(0, 1), f64.detach(), 1e-5, 'feat')
    H.assert_close(pg.grad.transpose(0, 1), p64.grad, 1e-4, 'gpos')
    H.assert_close(dg.grad.t(), d64.grad, 1e-4, 'gdts')


def test_adam_vs_torch(ops):
    torch.manual_seed(1)
    n = 100003
    p0 = torch.randn(n)
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([tp], lr=1e-4, betas=(0.5, 0.9))
    npad = (n + 3) // 4 * 4
    p = torch.zeros(npad, device=DEV); p[:n] = p0.to(DEV)
    m = torch.zeros(npad, device=DEV); v = torch.zeros(npad, device=DEV)
    for step in range(1, 5):
        g = torch.randn(n) * 10.0 ** torch.empty(n).uniform_(-6, 0)
        tp.grad = g.clone(); opt.step()
        gg = torch.zeros(npad, device=DEV); gg[:n] = g.to(DEV)
        ops.adam_step_(p, gg, m, v, 1e-4, (0.5, 0.9), 1e-8, step)
        np.testing.assert_allclose(p[:n].cpu().numpy(), tp.detach().numpy(), rtol=2e-7, atol=2e-9)
        np.testing.assert_allclose(m[:n].cpu().numpy(), opt.state[tp]['exp_avg'].numpy(), rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(v[:n].cpu().numpy(), opt.state[tp]['exp_avg_sq'].numpy(), rtol=1e-6, atol=1e-20)


def _lstm_oracle(x, lengths, W, nd, dt=torch.float64):
    """x (L,B,I) time-major -> oracle outputs time-major, via the batch-first oracle code."""
    xb = x.transpose(0, 1).to(dt)
    outs, hf = [], []
    for d in range(nd):
        o, h = O._masked_lstm_dir(xb, lengths, W[4 * d].to(dt), W[4 * d + 1].to(dt), W[4 * d + 2].to(dt),
                                  W[4 * d + 3].to(dt), reverse=(d == 1))
        outs.append(o); hf.append(h)
    return torch.cat(outs, -1).transpose(0, 1), torch.cat(hf, -1)


@pytest.mark.parametrize('Hd,I_mult,B,L,fuse', [(32, 1, 32, 7, False), (32, 1, 32, 7, True), (32, 2, 64, 9, True),
                                               (64, 1, 32, 5, True), (128, 1, 64, 12, False), (128, 1, 64, 12, True),
                                               (128, 2, 32, 10, True), (256, 1, 32, 6, True)])
def test_lstm_layer_fwd_bwd(ops, Hd, I_mult, B, L, fuse):
    torch.manual_seed(Hd + B + L)
    I = Hd * I_mult
    nd = 2
    k = 1.0 / math.sqrt(Hd)
    W = []
    for d in range(nd):
        W += [torch.empty(4 * Hd, I).uniform_(-k, k), torch.empty(4 * Hd, Hd).uniform_(-k, k),
              torch.empty(4 * Hd).uniform_(-k, k), torch.empty(4 * Hd).uniform_(-k, k)]
    x = torch.randn(L, B, I) * 0.5
    lengths = torch.randint(1, L + 1, (B,))
    lengths[0] = L
    lengths[B // 2] = 1
    # oracle in fp64 with autograd
    x64 = x.double().requires_grad_(True)
    W64 = [w.double().requires_grad_(True) for w in W]
    ho, hf = _lstm_oracle(x64, lengths, W64, nd)
    wy = torch.randn(L, B, nd * Hd).double(); wf = torch.randn(B, nd * Hd).double()
    mask = (torch.arange(L)[:, None] < lengths[None, :]).double()[:, :, None]
    ((ho * wy * mask).sum() + (hf * wf).sum()).backward()
    # HIP
    xg = x.to(DEV).requires_grad_(True)
    Wg = [w.to(DEV).requires_grad_(True) for w in W]
    h_all, h_final = ops.lstm_layer(xg, lengths.to(DEV).int(), Wg, fuse_x=fuse)
    H.assert_close(h_all, ho.detach(), 2e-5, 'h_all')
    H.assert_close(h_final, hf.detach(), 2e-5, 'h_final')
    ((h_all * (wy * mask).float().to(DEV)).sum() + (h_final * wf.float().to(DEV)).sum()).backward()
    H.assert_close(xg.grad, x64.grad, 1e-4, 'dx')
    for i, nm in enumerate(['w_ih', 'w_hh', 'b_ih', 'b_hh'] * nd):
        H.assert_close(Wg[i].grad, W64[i].grad, 1e-4, f'd{nm}[{i // 4}]')


def test_telemetry_kernel_bit_exact():
    import test_telemetry_oracle as TT
    from generative_mouse_trajectories_amd import telemetry as TEL
    g = H.load('telemetry.npz')
    d, cols, segs = TT.segments(g)
    c = {k: d[:, cols.index(k)] for k in cols}
    n_max = max(b - a for a, b in segs); n = len(segs)
    x = np.zeros((n_max, n)); y = np.zeros_like(x); dt = np.zeros_like(x); tdur = np.zeros_like(x)
    lens = np.array([b - a for a, b in segs], np.int32)
    for r, (a, b) in enumerate(segs):
        x[:b - a, r] = c['position_x'][a:b]; y[:b - a, r] = c['position_y'][a:b]
        dt[:b - a, r] = c['time_between_movements'][a:b]; tdur[:b - a, r] = c['total_duration'][a:b]
    tx, ty, tdt = (torch.tensor(a_, device=DEV) for a_ in (x, y, dt))
    out = TEL.telemetry_kinematics(tx, ty, tdt, torch.tensor(lens, device=DEV), time_is_dt=True, return_aux=True)
    kin6 = out['kin6'].cpu().numpy(); hov = out['hover_time'].cpu().numpy()
    der = TEL.telemetry_derived(tx, ty, tdt, torch.tensor(tdur, device=DEV), out['kin6'],
                                torch.tensor(lens, device=DEV)).cpu().numpy()
    worst = 0.0
    for r, (a, b) in enumerate(segs):
        for j, k in enumerate(('velocity', 'acceleration', 'jerk', 'path_length')):
            ref = c[k][a:b]
            # BASELINE tolerance is 1e-6 relative; the f64 kernel is expected to be bit exact
            err = np.abs(kin6[j, :b - a, r] - ref).max() / max(np.abs(ref).max(), 1e-300)
            worst = max(worst, err)
            assert err <= 1e-12, (k, r, err)
        assert np.abs(hov[:b - a, r] - c['hover_time'][a:b]).max() <= 1e-15
        o = TO.kinematics(x[:b - a, r], y[:b - a, r], dt[:b - a, r], True)
        assert np.abs(kin6[4, :b - a, r] - o['direction']).max() <= 1e-12
        ref = o['curvature']
        assert np.abs(kin6[5, :b - a, r] - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-300)
        for j, k in enumerate(TO.DERIVED):
            ref = c[k][a:b]
            tol = 1e-9 if k == 'idle_time' else 1e-12 * max(np.abs(ref).max(), 1.0)
            assert np.abs(der[j, :b - a, r] - ref).max() <= tol, (k, r)
    print('telemetry worst rel err', worst)


@pytest.mark.parametrize('R,N,I,ldx,shift', [(32 * 40, 512, 128, 256, 0), (32 * 129, 512, 256, 256, 32),
                                            (32 * 700, 512, 128, 128, 0), (64, 256, 128, 128, 0),
                                            # the other tilings: N = 128 / 64 (zero-padded tile), skinny X (I <= 32,
                                            # odd row pitch), odd chunk counts, a slab per chunk
                                            (16 * 37, 128, 128, 128, 0), (32 * 100, 64, 128, 128, 0),
                                            (32 * 300, 128, 9, 9, 0), (32 * 129, 512, 8, 8, 32),
                                            (16 * 3, 384, 128, 128, 0), (16, 512, 256, 256, 0), (32 * 50, 36, 7, 11, 0)])
def test_wgrad_kernel(ops, R, N, I, ldx, shift):
    torch.manual_seed(R + I)
    dA = torch.randn(R + shift, N, device=DEV)
    Xf = torch.randn(R + shift, ldx, device=DEV)
    X = Xf[:, ldx - I:] if ldx > I else Xf
    a, x = dA[shift:], X[:R]
    dW, db = ops.wgrad(a, x, want_db=True)
    ref = a.double().t() @ x.double()
    H.assert_close(dW, ref, 2e-6, 'dW')
    H.assert_close(db, a.double().sum(0), 2e-6, 'db')
    dW2, _ = ops.wgrad(a, x)
    assert torch.equal(dW, dW2), 'the slab reduction must be deterministic'


def test_linear_fn_matches_autograd(ops):
    torch.manual_seed(5)
    x = torch.randn(12, 64, 9, device=DEV, requires_grad=True)
    w = torch.randn(128, 9, device=DEV, requires_grad=True)
    b = torch.randn(128, device=DEV, requires_grad=True)
    g = torch.randn(12, 64, 128, device=DEV)
    y = ops.linear(x, w, b)
    dx, dw, db = torch.autograd.grad(y, (x, w, b), g)
    yr = torch.nn.functional.linear(x.double(), w.double(), b.double())
    rx, rw, rb = torch.autograd.grad(yr, (x, w, b), g.double())
    H.assert_close(y, yr.detach(), 1e-6, 'y')
    H.assert_close(dx, rx, 1e-6, 'dx'); H.assert_close(dw, rw, 2e-6, 'dw'); H.assert_close(db, rb, 2e-6, 'db')
    # double backward (gradient-penalty use): d/dw of |dL/dx|^2
    y = ops.linear(x, w, b)
    gx, = torch.autograd.grad(y, x, g, create_graph=True)
    (gx ** 2).sum().backward()
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    gx64, = torch.autograd.grad(torch.nn.functional.linear(x64, w64, b.double()), x64, g.double(), create_graph=True)
    (gx64 ** 2).sum().backward()
    H.assert_close(w.grad, w64.grad, 1e-5, 'd2w')
