#!/usr/bin/env python3
"""A/B of the penalty's front end (kernelised ops.EncoderGPFn vs the torch composite) on the config-1 curve data: replays
iterations 0-3 with the composite, then takes iteration 4's D step (the ill-conditioned penalty) with either path from the
same state and compares every D gradient with the fp64 oracle's.  GPU + CPU oracle; diagnosis only."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import helpers as H
import test_gpu_parity as TP
from oracle import gan_oracle as O
from generative_mouse_trajectories_amd import composite

g = H.load('config1_curve.npz')
dev = torch.device('cuda:0')


def noise_of(it):
    n = {'z': H.t(g[f'it{it}_D_z'])}
    if f'it{it}_D_alpha' in g:
        n['alpha'] = H.t(g[f'it{it}_D_alpha']); n['gp_idx'] = H.t(g[f'it{it}_D_gp_idx'])
    return n


def batch_of(it):
    return {k: H.t(g[f'it{it}_batch/{k}']) for k in ('starts', 'ends', 'sequences', 'lengths')}


def run(front_end_at_4, upto=4):
    gmt, cfg, G, D = TP.build(H.DEFAULT, 301, 302)
    tr = gmt.Trainer(G, D, cfg, dev, '/tmp/runs_diag')
    for it in range(upto + 1):
        composite.GP_KERNEL_FRONT_END = front_end_at_4 if it == upto else False
        tr.epoch = int(g['epochs'][it])
        m = dict(tr.train_discriminator_step(batch_of(it), noise_of(it)))
        if it == upto:
            return m, {k: p.grad.detach().cpu().double().clone() for k, p in D.named_parameters()}
        tr.train_generator_step(batch_of(it), {'z': H.t(g[f'it{it}_G_z']), 'tf_mask': list(g[f'it{it}_G_tf_mask'])})


for upto in (0, 4):
    m_k, g_k = run(True, upto)
    m_c, g_c = run(False, upto)
    dt = torch.float64
    cfg, Pg, Pd, Bd = H.oracle_models(H.DEFAULT, 301, 302, dt)
    opt_g, opt_d = {}, {}
    cv = lambda v: v.to(dt) if v.dtype.is_floating_point else v
    for it in range(upto + 1):
        b = {k: cv(v) for k, v in batch_of(it).items()}
        n = {k: cv(v) for k, v in noise_of(it).items()}
        md, gd, _ = O.discriminator_step(Pg, Pd, Bd, opt_d, b, n, cfg, it)
        if it < upto:
            O.generator_step(Pg, Pd, Bd, opt_g, b, {'z': cv(H.t(g[f'it{it}_G_z']))}, cfg, list(g[f'it{it}_G_tf_mask']),
                             O.teacher_forcing_ratio(int(g['epochs'][it]), cfg))
    print(f'--- iteration {upto}: penalty kernel path {m_k["d_gp"]:.3f}, composite {m_c["d_gp"]:.3f}, fp64 oracle {float(md["d_gp"]):.3f}')
    for k in g_k:
        ref = gd[k].double() if gd[k] is not None else torch.zeros_like(g_k[k])
        den = max(float(ref.abs().max()), 1e-30)
        print(f'  {k:42s} |g| {den:9.3e}  kernel-vs-fp64 {float((g_k[k] - ref).abs().max()) / den:8.2e}  composite-vs-fp64 '
              f'{float((g_c[k] - ref).abs().max()) / den:8.2e}  kernel-vs-composite {float((g_k[k] - g_c[k]).abs().max()) / den:8.2e}')
