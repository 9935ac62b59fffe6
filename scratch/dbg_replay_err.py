"""Prints, per step of the tiny / mid golden replays, the worst gradient error against the reference goldens."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import helpers as H
import paramgen as PG
from test_gpu_parity import build, DEV

def run(tag, over, B, T, lengths, full):
    g = H.load(f'{tag}_steps.npz')
    gmt, cfg, G, D = build(over, 21, 22)
    tr = gmt.Trainer(G, D, cfg, torch.device(DEV), '/tmp/runs_dbg')
    batch = H.batch_t(B, T, 23, lengths)
    plan = [s.split(':') for s in g[f'{tag}_plan']]
    for i, (kind, epoch) in enumerate(plan):
        tr.epoch = int(epoch)
        z = H.t(g[f'{tag}_s{i}_z'])
        if kind == 'D':
            noise = {'z': z}
            if f'{tag}_s{i}_alpha' in g:
                noise['alpha'] = H.t(g[f'{tag}_s{i}_alpha'])
                if f'{tag}_s{i}_gp_idx' in g:
                    noise['gp_idx'] = H.t(g[f'{tag}_s{i}_gp_idx'])
            m = tr.train_discriminator_step(batch, noise); model = D
        else:
            m = tr.train_generator_step(batch, {'z': z, 'tf_mask': list(g[f'{tag}_s{i}_tf_mask'])}); model = G
        ref_m = dict(zip(list(g[f'{tag}_s{i}_metric_names']), g[f'{tag}_s{i}_metrics']))
        em = max(abs(m[k] - v) / max(abs(v), 1e-3) for k, v in ref_m.items())
        worst = ('', 0.0)
        for k, p in model.named_parameters():
            ref = g[f'{tag}_s{i}_grad/{k}']
            val = p.grad.detach().cpu().numpy()
            if not full:
                val = PG.digest(val); e = abs(val[0] - ref[0]) / max(ref[0], 1e-30)
            else:
                e = H.relerr(val, ref)
            if e > worst[1]:
                worst = (k, e)
        print(f'{tag} step {i} {kind}: metric err {em:.2e}  worst grad {worst[0]} {worst[1]:.2e}', flush=True)

run('tiny', H.TINY, 6, 9, H.TINY_LENGTHS, True)
run('mid', H.MID, 8, 12, H.MID_LENGTHS, False)
run('default', H.DEFAULT, 4, 8, H.DEFAULT_LENGTHS, False)
