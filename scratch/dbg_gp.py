import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch, numpy as np
import helpers as H
import test_gpu_parity as TP
from generative_mouse_trajectories_amd import composite as C
g = H.load('tiny_modules.npz')
gmt, cfg, G, D = TP.build(H.TINY, 11, 12)
DEV = 'cuda:0'
b = TP.dev_batch(H.batch_t(6, 9, 13, H.TINY_LENGTHS))
traj = H.t(g['tiny_d_traj']).to(DEV); dts = H.t(g['tiny_d_dts']).to(DEV)
lens = b['lengths'] + 1
bufs0 = {k: v.clone() for k, v in D.named_buffers()}
res = {}
for name, fn in (('torch', C.discriminator_forward_torch), ('gp', C.discriminator_forward_gp)):
    for k, v in D.named_buffers():
        v.copy_(bufs0[k])
    x = traj.clone().requires_grad_(True)
    s = fn(D, x, dts, lens)
    gr = torch.autograd.grad(s, x, torch.ones_like(s), create_graph=True)[0]
    gp = ((gr.reshape(6, -1).norm(2, dim=1) - 1) ** 2).mean()
    D.zero_grad()
    gp.backward()
    res[name] = (s.detach().cpu(), gr.detach().cpu(), gp.item(), {k: (p.grad.clone().cpu() if p.grad is not None else None) for k, p in D.named_parameters()})
print('score diff', (res['torch'][0] - res['gp'][0]).abs().max().item(), res['torch'][0].abs().max().item())
print('grad diff', (res['torch'][1] - res['gp'][1]).abs().max().item(), res['torch'][1].abs().max().item())
print('gp', res['torch'][2], res['gp'][2])
for k in res['torch'][3]:
    a, c = res['torch'][3][k], res['gp'][3][k]
    if a is None or c is None:
        print(k, 'None', a is None, c is None); continue
    print(k, 'relerr %.3e' % ((a - c).abs().max() / a.abs().max().clamp(min=1e-30)).item())
