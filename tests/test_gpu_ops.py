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


# ---- round 2: fused critic front end, dx contraction, trajectory bookkeeping, generator losses -----------------
@pytest.mark.parametrize('Hd,B,L', [(128, 96, 37), (32, 32, 9), (64, 64, 5), (256, 32, 12)])
def test_fused_encoder_vs_oracle(ops, Hd, B, L):
    """gmt_encoder_fwd (kinematics + Linear 9->H + LeakyReLU + Linear H->H + LeakyReLU, RG.py:315-389, 712-717)
    and its backward (gmt_dx_gemm with the fused LeakyReLU', gmt_wgrad, gmt_kinematics_bwd) against the same
    composition in fp64 torch ops."""
    g = torch.Generator().manual_seed(Hd + L)
    pos = torch.rand(B, 1, 2, generator=g) + torch.cumsum(torch.randn(B, L, 2, generator=g) * 0.01, 1)
    dts = torch.rand(B, L, generator=g) * 0.02 + 0.004
    w1 = torch.randn(Hd, 9, generator=g) * 0.05; b1 = torch.randn(Hd, generator=g) * 0.1
    w2 = torch.randn(Hd, Hd, generator=g) / math.sqrt(Hd); b2 = torch.randn(Hd, generator=g) * 0.1
    w = torch.randn(B, L, Hd, generator=g)
    leaves = [t.double().requires_grad_(True) for t in (pos, dts, w1, b1, w2, b2)]
    f = O.kinematics(leaves[0], leaves[1])
    h1 = torch.nn.functional.leaky_relu(torch.nn.functional.linear(f, leaves[2], leaves[3]), 0.2)
    e64 = torch.nn.functional.leaky_relu(torch.nn.functional.linear(h1, leaves[4], leaves[5]), 0.2)
    (e64 * w.double()).sum().backward()
    dl = [pos.transpose(0, 1).contiguous(), dts.t().contiguous(), w1, b1, w2, b2]
    dl = [t.to(DEV).requires_grad_(True) for t in dl]
    for pre in (False, True):
        for t in dl:
            t.grad = None
        e = ops.encoder(*dl, 0.2, 1e-8, pre)
        H.assert_close(e.transpose(0, 1), e64.detach(), 2e-5, f'encoder output H={Hd}')
        gin = w.to(DEV).transpose(0, 1)
        if pre:                                   # the consumer hands over d(pre-activation): apply LeakyReLU' here
            gin = gin * torch.where(e.detach() > 0, 1.0, 0.2)
        e.backward(gin.contiguous())
        H.assert_close(dl[0].grad.transpose(0, 1), leaves[0].grad, 1e-4, 'encoder gpos')
        H.assert_close(dl[1].grad.t(), leaves[1].grad, 1e-4, 'encoder gdts')
        for k in range(2, 6):
            H.assert_close(dl[k].grad, leaves[k].grad, 1e-4, f'encoder weight grad {k}')


@pytest.mark.parametrize('K,N,nd,R', [(512, 256, 2, 32 * 700), (512, 128, 2, 32 * 513), (128, 128, 1, 32 * 40),
                                       (128, 32, 2, 64), (256, 64, 1, 32 * 33), (1024, 256, 2, 32 * 20), (512, 512, 1, 32 * 9),
                                       (512, 128, 2, 32 * 3), (512, 256, 1, 32), (256, 128, 2, 32 * 300), (256, 256, 2, 32 * 129),
                                       (128, 128, 2, 32 * 1000)])
def test_dx_gemm_vs_fp64(ops, K, N, nd, R):
    """gmt_dx_gemm: dX = sum_d dA_d W_d, optionally times LeakyReLU'(act) - more tiles than workgroups, both
    tile-walk parities, one and two directions."""
    g = torch.Generator().manual_seed(K + N + R)
    dA = torch.randn(nd, R, K, generator=g)
    W = torch.randn(nd, K, N, generator=g) / math.sqrt(K)
    act = torch.randn(R, N, generator=g)
    ref = sum(dA[d].double() @ W[d].double() for d in range(nd))
    w_pk = torch.stack([ops.pack_dx_weight(W[d].to(DEV).contiguous()) for d in range(nd)])
    out = ops.dx_gemm(dA.to(DEV), w_pk, K, N)
    H.assert_close(out, ref, 2e-6, 'dx_gemm')
    out = ops.dx_gemm(dA.to(DEV), w_pk, K, N, act.to(DEV), 0.2)
    H.assert_close(out, ref * torch.where(act > 0, 1.0, 0.2).double(), 2e-6, 'dx_gemm with LeakyReLU\'')


def test_traj_from_seq_and_generator_losses_vs_oracle(ops):
    """gmt_traj_from_seq (RG.py:392-417 + the dt padding) and gmt_gen_loss_fwd / _bwd (RG.py:1072-1095) against the
    oracle's generator_losses and autograd, ragged lengths, batch padding rows, both sequence layouts."""
    B, T = 70, 19
    Bp = ops.pad_rows(B)
    g = torch.Generator().manual_seed(5)
    lengths = torch.randint(1, T + 1, (B,), generator=g)
    lengths[0] = T
    starts = torch.rand(B, 2, generator=g)
    ends = starts + 0.2 * torch.randn(B, 2, generator=g)
    seq = torch.cat([torch.randn(B, T, 2, generator=g) * 0.01, torch.rand(B, T, 1, generator=g) * 0.02 + 0.004], -1)
    seq[3, 2, :2] = 0.0                                   # a zero step: the eps inside the square root matters
    # batch-first data layout
    pos, dts = ops.traj_from_seq(seq.to(DEV), starts.to(DEV), False, B, Bp)
    ref_pos = O.to_absolute(starts, seq[:, :, :2])
    assert torch.equal(pos[:, :B].transpose(0, 1).cpu(), ref_pos)             # the CPU reference's association, bit for bit
    assert torch.equal(dts[:, :B].t().cpu(), torch.nn.functional.pad(seq[:, :, 2], (1, 0), value=0.001))
    assert float(pos[:, B:].abs().max()) == 0 and float(dts[:, B:].abs().max()) == 0
    # time-major rollout layout, with gradients
    seq_tm = torch.zeros(T, Bp, 3)
    seq_tm[:, :B] = seq.transpose(0, 1)
    seq_tm[:, B:] = 7.0                                   # garbage in the padding rows must not leak
    sg = seq_tm.to(DEV).requires_grad_(True)
    pos2, dts2 = ops.traj_from_seq(sg, starts.to(DEV), True, B, Bp)
    assert torch.equal(pos2[:, :B].cpu(), pos[:, :B].cpu()) and torch.equal(dts2[:, :B].cpu(), dts[:, :B].cpu())
    wp, wd = torch.randn(T + 1, Bp, 2, generator=g), torch.randn(T + 1, Bp, generator=g)
    scal = ops.gen_aux_losses(sg, starts.to(DEV), ends.to(DEV), lengths.to(torch.int32).to(DEV))
    cw = torch.tensor([0.7, -1.3, 0.0, 0.0])
    ((pos2 * wp.to(DEV)).sum() + (dts2 * wd.to(DEV)).sum() + (scal * cw.to(DEV)).sum()).backward()
    s64 = seq.double().requires_grad_(True)
    p64 = O.to_absolute(starts.double(), s64[:, :, :2])
    d64 = torch.nn.functional.pad(s64[:, :, 2], (1, 0), value=0.001)
    _, endpoint, direction = O.generator_losses(s64, torch.zeros(B, 1, dtype=torch.float64), starts.double(), ends.double(), lengths)
    mask_sum = float(lengths.sum())
    H.assert_close(scal[:3].cpu(), torch.stack([endpoint * 2 * B, direction * mask_sum, torch.tensor(mask_sum, dtype=torch.float64)]).detach(),
                   2e-6, 'loss sums')
    tot = (p64 * wp[:, :B].transpose(0, 1).double()).sum() + (d64 * wd[:, :B].t().double()).sum() \
        + 0.7 * endpoint * 2 * B - 1.3 * direction * mask_sum
    tot.backward()
    H.assert_close(sg.grad[:, :B].transpose(0, 1), s64.grad, 2e-5, 'd(sequence)')
    assert float(sg.grad[:, B:].abs().max()) == 0
