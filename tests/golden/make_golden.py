#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (RecurrentGAN.py) on CPU.

Run in the build container only (the reference never travels to the GPU box):

    MPLBACKEND=Agg python tests/golden/make_golden.py

Writes tests/golden/*.npz.  The reference is imported by path with a no-op
stand-in for the (not installed) `torch.utils.tensorboard` module; nothing of
the reference's source is stored, only inputs and outputs.
"""
from __future__ import annotations

import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import paramgen as PG  # noqa: E402

REF = '/root/reference/python/learning/RecurrentGAN.py'


def load_reference():
    stub = types.ModuleType('torch.utils.tensorboard')

    class SummaryWriter:  # no-op
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_figure(self, *a, **k): pass
        def close(self): pass
    stub.SummaryWriter = SummaryWriter
    sys.modules['torch.utils.tensorboard'] = stub
    spec = importlib.util.spec_from_file_location('RecurrentGAN', REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_cfg(ref, **over):
    cfg = ref.Config()
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def build_models(ref, cfg, seed_g, seed_d):
    Pg = PG.make_generator_params(cfg, seed_g)
    Pd, Bd = PG.make_discriminator_params(cfg, seed_d)
    G = ref.Generator(cfg)
    D = ref.Discriminator(cfg)
    G.load_state_dict({k: T(v) for k, v in Pg.items()}, strict=True)
    sd = {k: T(v) for k, v in Pd.items()}
    sd.update({k: T(v) for k, v in Bd.items()})
    D.load_state_dict(sd, strict=True)
    # parity mode: zero the hard-coded dropouts
    G.lstm.dropout = 0.0
    D.lstm.dropout = 0.0
    D.output_layers[2].p = 0.0
    G.train()
    D.train()
    return G, D, Pg, Pd, Bd


def tbatch(b):
    return {k: T(v) for k, v in b.items()}


# ---------------------------------------------------------------- kinematics
def gold_kinematics(ref, out):
    rng = np.random.Generator(np.random.PCG64(7))
    cases = {}
    # known-answer example of SURVEY.md section 8c
    pos = np.array([[[0.0, 0.0], [0.1, 0.0], [0.2, 0.1], [0.25, 0.3], [0.2, 0.5]]], dtype=np.float32)
    dts = np.array([[0.001, 0.01, 0.02, 0.015, 0.03]], dtype=np.float32)
    cases['kat'] = (pos, dts)
    for name, (B, L) in {'a': (3, 7), 'b': (5, 33), 'c': (2, 1), 'd': (4, 2), 'e': (2, 130)}.items():
        start = rng.uniform(0, 1, size=(B, 1, 2))
        p = start + np.cumsum(rng.normal(0, 0.01, size=(B, L, 2)), axis=1)
        d = rng.uniform(0.004, 0.024, size=(B, L))
        d[:, 0] = 0.001
        cases[name] = (p.astype(np.float32), d.astype(np.float32))
    # padded style: dt == 0 rows and repeated positions (collate zero padding)
    p, d = cases['b']
    p2, d2 = p.copy(), d.copy()
    p2[:, 20:] = p2[:, 19:20]
    d2[:, 20:] = 0.0
    cases['pad'] = (p2, d2)
    for name, (p, d) in cases.items():
        tp = T(p).clone().requires_grad_(True)
        td = T(d).clone().requires_grad_(True)
        f = ref.compute_kinematics(tp, td)
        w = T(rng.normal(size=f.shape).astype(np.float32))
        (f * w).sum().backward()
        out[f'kin_{name}_pos'] = p
        out[f'kin_{name}_dts'] = d
        out[f'kin_{name}_feat'] = f.detach().numpy()
        out[f'kin_{name}_w'] = w.numpy()
        out[f'kin_{name}_gpos'] = tp.grad.numpy()
        out[f'kin_{name}_gdts'] = td.grad.numpy()
    s = T(rng.uniform(0, 1, size=(4, 2)).astype(np.float32))
    dl = T(rng.normal(0, 0.01, size=(4, 11, 2)).astype(np.float32))
    out['abs_start'], out['abs_deltas'] = s.numpy(), dl.numpy()
    out['abs_out'] = ref.trajectory_to_absolute(s, dl).numpy()


# ---------------------------------------------------------------- modules
def gold_modules(ref, out, cfg_over, B, Tn, lengths, tag):
    cfg = make_cfg(ref, **cfg_over)
    G, D, Pg, Pd, Bd = build_models(ref, cfg, 11, 12)
    batch = PG.make_batch(B, Tn, 13, lengths)
    tb = tbatch(batch)
    rng = np.random.Generator(np.random.PCG64(14))
    z = T(rng.normal(size=(B, cfg.latent_dim)).astype(np.float32))
    out[f'{tag}_z'] = z.numpy()
    # teacher forcing: ratio 0 -> never, ratio 1.0 -> rand()<1 always
    for name, ratio in (('tf0', 0.0), ('tf1', 1.0)):
        seq, _ = G(tb['starts'], tb['ends'], z, target_sequences=tb['sequences'],
                   target_lengths=tb['lengths'], teacher_forcing_ratio=ratio)
        out[f'{tag}_gfwd_{name}'] = seq.detach().numpy()
    # mixed mask with recorded coins
    torch.manual_seed(99)
    st = torch.get_rng_state()
    seq, _ = G(tb['starts'], tb['ends'], z, target_sequences=tb['sequences'],
               target_lengths=tb['lengths'], teacher_forcing_ratio=0.5)
    torch.set_rng_state(st)
    coins = np.array([torch.rand(1).item() for _ in range(Tn)])
    out[f'{tag}_gfwd_mix'] = seq.detach().numpy()
    out[f'{tag}_gfwd_mix_mask'] = (coins < 0.5)
    # gradient of a random functional through the mixed rollout
    w = T(rng.normal(size=seq.shape).astype(np.float32))
    G.zero_grad()
    (seq * w).sum().backward()
    out[f'{tag}_gfwd_mix_w'] = w.numpy()
    for k, p in G.named_parameters():
        out[f'{tag}_gbwd_mix_grad/{k}'] = p.grad.numpy().copy()

    # discriminator forward (one call = one power iteration)
    traj = ref.trajectory_to_absolute(tb['starts'], tb['sequences'][:, :, :2])
    dts = torch.nn.functional.pad(tb['sequences'][:, :, 2], (1, 0), value=0.001)
    grabbed = {}
    hk = D.output_layers.register_forward_hook(lambda m, i, o: grabbed.__setitem__('hcomb', i[0].detach()))
    hk2 = D.feature_encoder.register_forward_hook(lambda m, i, o: grabbed.__setitem__('enc', o.detach()))
    traj_r = traj.clone().requires_grad_(True)
    dts_r = dts.clone().requires_grad_(True)
    score = D(traj_r, dts_r, tb['lengths'] + 1)
    hk.remove(); hk2.remove()
    D.zero_grad()
    ws = T(rng.normal(size=score.shape).astype(np.float32))
    (score * ws).sum().backward()
    out[f'{tag}_d_traj'], out[f'{tag}_d_dts'] = traj.numpy(), dts.numpy()
    out[f'{tag}_d_score'] = score.detach().numpy()
    out[f'{tag}_d_hcomb'] = grabbed['hcomb'].numpy()
    out[f'{tag}_d_enc'] = grabbed['enc'].numpy()
    out[f'{tag}_d_ws'] = ws.numpy()
    out[f'{tag}_d_gtraj'] = traj_r.grad.numpy()
    out[f'{tag}_d_gdts'] = dts_r.grad.numpy()
    for k, p in D.named_parameters():
        out[f'{tag}_d_grad/{k}'] = p.grad.numpy().copy()
    for k, b in D.named_buffers():
        out[f'{tag}_d_buf_after1/{k}'] = b.numpy().copy()
    # gradient penalty (second power iteration), explicit alpha via seed replay
    fake = traj + T(rng.normal(0, 0.01, size=tuple(traj.shape)).astype(np.float32))
    fdts = dts * T(rng.uniform(0.5, 1.5, size=tuple(dts.shape)).astype(np.float32))
    torch.manual_seed(5)
    st = torch.get_rng_state()
    D.zero_grad()
    gp = ref.compute_gradient_penalty(D, traj, fake, dts, fdts, tb['lengths'] + 1, torch.device('cpu'))
    gp.backward()
    torch.set_rng_state(st)
    alpha = torch.rand(B, 1, 1)
    out[f'{tag}_gp_fake'], out[f'{tag}_gp_fdts'] = fake.numpy(), fdts.numpy()
    out[f'{tag}_gp_alpha'] = alpha.reshape(B).numpy()
    out[f'{tag}_gp_value'] = np.array(gp.item())
    for k, p in D.named_parameters():
        out[f'{tag}_gp_grad/{k}'] = (p.grad.numpy().copy() if p.grad is not None
                                     else np.zeros(tuple(p.shape), np.float32))
    for k, b in D.named_buffers():
        out[f'{tag}_d_buf_after2/{k}'] = b.numpy().copy()
    # generate() in eval mode
    G.eval()
    with torch.no_grad():
        cfg.max_generation_steps = 12
        gs, gl = G.generate(tb['starts'], tb['ends'], z)
    out[f'{tag}_gen_out'], out[f'{tag}_gen_len'] = gs.numpy(), gl.numpy()
    G.train()


# ---------------------------------------------------------------- train steps
def gold_steps(ref, out, cfg_over, B, Tn, lengths, plan, tag, full):
    cfg = make_cfg(ref, **cfg_over)
    G, D, Pg, Pd, Bd = build_models(ref, cfg, 21, 22)
    batch = PG.make_batch(B, Tn, 23, lengths)
    tb = tbatch(batch)
    torch.manual_seed(4321)
    tr = ref.Trainer(G, D, cfg, torch.device('cpu'), '/tmp/runs_golden')
    G.train(); D.train()
    out[f'{tag}_plan'] = np.array([f'{k}:{e}' for k, e in plan])

    def store(name, arr):
        arr = np.array(arr, copy=True)   # snapshot: the tensors are updated in place later
        out[name] = arr if full else PG.digest(arr)

    for i, (kind, epoch) in enumerate(plan):
        tr.epoch = epoch
        st = torch.get_rng_state()
        d_step_before = tr.d_step
        if kind == 'D':
            m = tr.train_discriminator_step(tb)
        else:
            m = tr.train_generator_step(tb)
        after = torch.get_rng_state()
        # replay the reference's draws (SURVEY.md section 8c order) to record them
        torch.set_rng_state(st)
        z = torch.randn(B, cfg.latent_dim)
        coins = np.array([torch.rand(1).item() for _ in range(Tn)])
        out[f'{tag}_s{i}_z'] = z.numpy()
        if kind == 'D':
            if d_step_before % cfg.gp_every_n == 0:
                b = max(1, int(math.ceil(B * cfg.gp_batch_frac)))
                if b < B:
                    perm = torch.randperm(B)
                    out[f'{tag}_s{i}_gp_idx'] = perm.numpy()
                alpha = torch.rand(b, 1, 1)
                out[f'{tag}_s{i}_alpha'] = alpha.reshape(b).numpy()
        else:
            ratio = tr.get_teacher_forcing_ratio()
            out[f'{tag}_s{i}_tf_mask'] = coins < ratio
            out[f'{tag}_s{i}_tf_ratio'] = np.array(ratio)
        assert torch.equal(torch.get_rng_state(), after), 'RNG replay out of sync'
        keys = sorted(m.keys())
        out[f'{tag}_s{i}_metric_names'] = np.array(keys)
        out[f'{tag}_s{i}_metrics'] = np.array([m[k] for k in keys], dtype=np.float64)
        model, opt = (D, tr.d_optimizer) if kind == 'D' else (G, tr.g_optimizer)
        for k, p in model.named_parameters():
            store(f'{tag}_s{i}_grad/{k}', p.grad.numpy())
            store(f'{tag}_s{i}_param/{k}', p.detach().numpy())
            stt = opt.state[p]
            store(f'{tag}_s{i}_m/{k}', stt['exp_avg'].numpy())
            store(f'{tag}_s{i}_v/{k}', stt['exp_avg_sq'].numpy())
        for k, b_ in D.named_buffers():
            store(f'{tag}_s{i}_buf/{k}', b_.numpy())
    return


def main():
    ref = load_reference()
    out = {}
    gold_kinematics(ref, out)
    np.savez_compressed(os.path.join(HERE, 'kinematics.npz'), **out)
    print('kinematics.npz', len(out))

    tiny = dict(latent_dim=8, generator_hidden_dim=32, discriminator_hidden_dim=32)
    out = {}
    gold_modules(ref, out, tiny, 6, 9, [9, 9, 7, 5, 4, 1], 'tiny')
    np.savez_compressed(os.path.join(HERE, 'tiny_modules.npz'), **out)
    print('tiny_modules.npz', len(out))

    plan = [('D', 50), ('G', 50), ('D', 50), ('G', 200), ('D', 50), ('D', 50), ('D', 50), ('G', 0)]
    out = {}
    gold_steps(ref, out, tiny, 6, 9, [9, 9, 7, 5, 4, 1], plan, 'tiny', full=True)
    np.savez_compressed(os.path.join(HERE, 'tiny_steps.npz'), **out)
    print('tiny_steps.npz', len(out))

    mid = dict(latent_dim=64, generator_hidden_dim=128, discriminator_hidden_dim=128)
    out = {}
    gold_steps(ref, out, mid, 8, 12, [12, 12, 12, 11, 9, 8, 6, 5], plan[:4], 'mid', full=False)
    np.savez_compressed(os.path.join(HERE, 'mid_steps.npz'), **out)
    print('mid_steps.npz', len(out))

    out = {}
    gold_steps(ref, out, {}, 4, 8, [8, 8, 6, 5], plan[:4], 'default', full=False)
    np.savez_compressed(os.path.join(HERE, 'default_steps.npz'), **out)
    print('default_steps.npz', len(out))


if __name__ == '__main__':
    main()
