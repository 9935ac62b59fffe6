"""Pins the CPU oracle (oracle/gan_oracle.py) to the real reference through the
golden vectors that tests/golden/make_golden.py produced from RecurrentGAN.py.
CPU only."""
import math

import numpy as np
import pytest
import torch

import helpers as H
import paramgen as PG
from oracle import gan_oracle as O

RT = 2e-5      # fp32 CPU restatement vs fp32 CPU reference (different op order)


def test_kinematics_and_absolute():
    g = H.load('kinematics.npz')
    for name in ('kat', 'a', 'b', 'c', 'd', 'e', 'pad'):
        pos = H.t(g[f'kin_{name}_pos']).requires_grad_(True)
        dts = H.t(g[f'kin_{name}_dts']).requires_grad_(True)
        f = O.kinematics(pos, dts)
        H.assert_close(f, g[f'kin_{name}_feat'], 1e-6, f'feat {name}')
        (f * H.t(g[f'kin_{name}_w'])).sum().backward()
        H.assert_close(pos.grad, g[f'kin_{name}_gpos'], RT, f'gpos {name}')
        H.assert_close(dts.grad, g[f'kin_{name}_gdts'], RT, f'gdts {name}')
    # SURVEY section 8c known answer, row t=2
    f = O.kinematics(H.t(g['kin_kat_pos']), H.t(g['kin_kat_dts']))[0, 2].numpy()
    np.testing.assert_allclose(f, [0.2, 0.1, 0.1, 0.1, 0.02, 7.0711, -195.26, -1e4, 9.4281], rtol=2e-4)
    out = O.to_absolute(H.t(g['abs_start']), H.t(g['abs_deltas']))
    H.assert_close(out, g['abs_out'], 1e-7, 'to_absolute')


def test_modules_tiny():
    g = H.load('tiny_modules.npz')
    cfg, Pg, Pd, Bd = H.oracle_models(H.TINY, 11, 12)
    b = H.batch_t(6, 9, 13, H.TINY_LENGTHS)
    z = H.t(g['tiny_z'])
    T = 9
    for name, mask in (('tf0', [False] * T), ('tf1', [True] * T), ('mix', list(g['tiny_gfwd_mix_mask']))):
        seq = O.generator_forward(Pg, b['starts'], b['ends'], z, b['sequences'], mask)
        H.assert_close(seq, g[f'tiny_gfwd_{name}'], RT, f'G fwd {name}')
    Pg_r = O._req(Pg)
    seq = O.generator_forward(Pg_r, b['starts'], b['ends'], z, b['sequences'], list(g['tiny_gfwd_mix_mask']))
    gr = torch.autograd.grad((seq * H.t(g['tiny_gfwd_mix_w'])).sum(), list(Pg_r.values()))
    for k, v in zip(Pg_r.keys(), gr):
        H.assert_close(v, g[f'tiny_gbwd_mix_grad/{k}'], 1e-4, f'G bwd {k}')

    traj = H.t(g['tiny_d_traj']).requires_grad_(True)
    dts = H.t(g['tiny_d_dts']).requires_grad_(True)
    Pd_r = O._req(Pd)
    score, inter = O.discriminator_forward(Pd_r, Bd, traj, dts, b['lengths'] + 1, return_intermediates=True)
    H.assert_close(score, g['tiny_d_score'], RT, 'D score')
    H.assert_close(inter['h_combined'], g['tiny_d_hcomb'], RT, 'D hcomb')
    H.assert_close(inter['encoded'], g['tiny_d_enc'], RT, 'D enc')
    names = list(Pd_r.keys())
    gr = torch.autograd.grad((score * H.t(g['tiny_d_ws'])).sum(), [Pd_r[k] for k in names] + [traj, dts])
    for k, v in zip(names, gr):
        H.assert_close(v, g[f'tiny_d_grad/{k}'], 1e-4, f'D grad {k}')
    H.assert_close(gr[-2], g['tiny_d_gtraj'], 1e-4, 'D gtraj')
    H.assert_close(gr[-1], g['tiny_d_gdts'], 1e-4, 'D gdts')
    for k in Bd:
        H.assert_close(Bd[k], g[f'tiny_d_buf_after1/{k}'], 1e-6, f'buf {k}')
    # gradient penalty with double backward
    Pd_r = O._req(Pd)
    gp = O.gradient_penalty(Pd_r, Bd, H.t(g['tiny_d_traj']), H.t(g['tiny_gp_fake']), H.t(g['tiny_d_dts']),
                            H.t(g['tiny_gp_fdts']), b['lengths'] + 1, H.t(g['tiny_gp_alpha']))
    assert abs(float(gp) - float(g['tiny_gp_value'])) <= 1e-5 * abs(float(g['tiny_gp_value']))
    gr = torch.autograd.grad(gp, [Pd_r[k] for k in names], allow_unused=True)
    for k, v in zip(names, gr):
        v = torch.zeros_like(Pd_r[k]) if v is None else v
        H.assert_close(v, g[f'tiny_gp_grad/{k}'], 2e-4, f'GP grad {k}')
    for k in Bd:
        H.assert_close(Bd[k], g[f'tiny_d_buf_after2/{k}'], 1e-6, f'buf2 {k}')
    # generate()
    cfg.max_generation_steps = 12
    gs, gl = O.generator_generate(Pg, b['starts'], b['ends'], z, cfg)
    H.assert_close(gs, g['tiny_gen_out'], RT, 'generate out')
    assert np.array_equal(gl.numpy(), g['tiny_gen_len'])


def _replay(tag, cfg_over, B, T, lengths, full, gtol):
    g = H.load(f'{tag}_steps.npz')
    cfg, Pg, Pd, Bd = H.oracle_models(cfg_over, 21, 22)
    batch = H.batch_t(B, T, 23, lengths)
    plan = [s.split(':') for s in g[f'{tag}_plan']]
    opt_g, opt_d = {}, {}
    d_step = 0

    def check(name, val, rtol):
        ref = g[name]
        if not full:
            val = PG.digest(val.detach().numpy() if isinstance(val, torch.Tensor) else val)
            # the norm and the samples; the plain sum cancels heavily, so scale it by the norm
            e = np.abs(val - ref)
            scale = max(ref[0], 1e-30)
            assert e[0] <= rtol * scale, f'{name} norm {e[0] / scale:.2e}'
            assert e[2:].max() <= rtol * max(np.abs(ref[2:]).max(), 1e-30) * 4, f'{name} samples'
        else:
            H.assert_close(val, ref, rtol, name)

    for i, (kind, epoch) in enumerate(plan):
        epoch = int(epoch)
        z = H.t(g[f'{tag}_s{i}_z'])
        names = list(g[f'{tag}_s{i}_metric_names'])
        ref_m = dict(zip(names, g[f'{tag}_s{i}_metrics']))
        if kind == 'D':
            noise = {'z': z}
            if f'{tag}_s{i}_alpha' in g:
                noise['alpha'] = H.t(g[f'{tag}_s{i}_alpha'])
                if f'{tag}_s{i}_gp_idx' in g:
                    noise['gp_idx'] = H.t(g[f'{tag}_s{i}_gp_idx'])
            m, grads, _ = O.discriminator_step(Pg, Pd, Bd, opt_d, batch, noise, cfg, d_step)
            d_step += 1
            P, opt = Pd, opt_d
        else:
            mask = list(g[f'{tag}_s{i}_tf_mask'])
            ratio = O.teacher_forcing_ratio(epoch, cfg)
            assert abs(ratio - float(g[f'{tag}_s{i}_tf_ratio'])) < 1e-12
            m, grads, _ = O.generator_step(Pg, Pd, Bd, opt_g, batch, {'z': z}, cfg, mask, ratio)
            P, opt = Pg, opt_g
        for k, v in ref_m.items():
            assert abs(m[k] - v) <= 1e-4 * max(abs(v), 1e-3), f'step {i} metric {k}: {m[k]} vs {v}'
        for k in P:
            gk = grads[k] if grads[k] is not None else torch.zeros_like(P[k])
            check(f'{tag}_s{i}_grad/{k}', gk, gtol)
            check(f'{tag}_s{i}_m/{k}', opt[k]['exp_avg'], gtol)
            check(f'{tag}_s{i}_v/{k}', opt[k]['exp_avg_sq'], 2 * gtol)
            # parameters: the Adam step is lr*sign-like, compare the update through the params
            check(f'{tag}_s{i}_param/{k}', P[k], 2e-5)
        for k in Bd:
            check(f'{tag}_s{i}_buf/{k}', Bd[k], 1e-5)


def test_train_steps_tiny():
    _replay('tiny', H.TINY, 6, 9, H.TINY_LENGTHS, True, 2e-4)


def test_train_steps_mid_digest():
    _replay('mid', H.MID, 8, 12, H.MID_LENGTHS, False, 3e-4)


def test_train_steps_default_digest():
    _replay('default', H.DEFAULT, 4, 8, H.DEFAULT_LENGTHS, False, 3e-4)


def test_fp64_oracle_agrees_with_fp32():
    cfg, Pg, Pd, Bd = H.oracle_models(H.TINY, 11, 12, torch.float64)
    b = H.batch_t(6, 9, 13, H.TINY_LENGTHS, torch.float64)
    g = H.load('tiny_modules.npz')
    seq = O.generator_forward(Pg, b['starts'], b['ends'], H.t(g['tiny_z'], torch.float64), b['sequences'], [False] * 9)
    H.assert_close(seq.float(), g['tiny_gfwd_tf0'], 2e-5, 'fp64 G fwd')


def test_config1_loss_curve_first_iterations():
    """BASELINE configs[0]: the reference trained on its own recording (data/mouse_data_20250813_235208.csv) with default
    hyper-parameters, B = 32 - tests/golden/config1_curve.npz holds its batches, noise and per-iteration metrics
    (tests/golden/make_config1_golden.py).  The oracle's step functions replay the first 3 of the 8 iterations here (CPU
    time); the GPU suite replays all 8 through the Trainer."""
    g = H.load('config1_curve.npz')
    cfg, Pg, Pd, Bd = H.oracle_models(H.DEFAULT, 301, 302)
    opt_g, opt_d = {}, {}
    for it in range(3):
        batch = {k: H.t(g[f'it{it}_batch/{k}']) for k in ('starts', 'ends', 'sequences', 'lengths')}
        noise = {'z': H.t(g[f'it{it}_D_z'])}
        if f'it{it}_D_alpha' in g:
            noise['alpha'] = H.t(g[f'it{it}_D_alpha'])
            noise['gp_idx'] = H.t(g[f'it{it}_D_gp_idx'])
        # 1e-4 on the same weights (iteration 0); once both sides have taken their own Adam steps the bar is 5e-4 (the
        # oracle - CPU torch like the reference, another association of the same sums - is 1.4e-4 off at iteration 1)
        bar = 1e-4 if it == 0 else 5e-4
        m, _, _ = O.discriminator_step(Pg, Pd, Bd, opt_d, batch, noise, cfg, it)
        H.curve_check(it, 'D', m, g[f'it{it}_D_metric_names'], g[f'it{it}_D_metrics'], bar)
        ratio = O.teacher_forcing_ratio(int(g['epochs'][it]), cfg)
        m, _, _ = O.generator_step(Pg, Pd, Bd, opt_g, batch, {'z': H.t(g[f'it{it}_G_z'])}, cfg, list(g[f'it{it}_G_tf_mask']), ratio)
        H.curve_check(it, 'G', m, g[f'it{it}_G_metric_names'], g[f'it{it}_G_metrics'], bar)
