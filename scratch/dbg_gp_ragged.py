"""Ragged gradient penalty on the bf16 kernels: a batch whose rows all have the SAME length 10 inside T = 16 (mask path) against the
same data truncated to T = 10 (full-length path): identical arithmetic at the valid steps, so the gradients must agree to ~1e-3."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tests', 'golden')]
import numpy as np, torch
import helpers as H, paramgen as PG
from oracle import gan_oracle as O
import generative_mouse_trajectories_amd as gmt
DEV = torch.device('cuda:0')
over = dict(latent_dim=8, generator_hidden_dim=32, discriminator_hidden_dim=256)
cfg = gmt.Config()
for k, v in over.items(): setattr(cfg, k, v)
cfg.lstm_dropout = cfg.head_dropout = 0.0
cfg.compute_dtype = 'bf16'
pd, bd = PG.make_discriminator_params(O.OracleConfig(**over), 131)
def critic():
    D = gmt.Discriminator(cfg)
    sd = {k: torch.from_numpy(v) for k, v in pd.items()}; sd.update({k: torch.from_numpy(v) for k, v in bd.items()})
    D.load_state_dict(sd, strict=True)
    return D.to(DEV).train()
Bs, T, Tv = 64, 16, 10
br = H.batch_t(Bs, T, 132, [Tv] * Bs, torch.float32); bf = H.batch_t(Bs, T, 133, [Tv] * Bs, torch.float32)
alpha = torch.rand(Bs, 1, 1, generator=torch.Generator().manual_seed(134))
def run(Tc):
    rt = gmt.trajectory_to_absolute(br['starts'], br['sequences'][:, :Tc, :2]); ft = gmt.trajectory_to_absolute(bf['starts'], bf['sequences'][:, :Tc, :2])
    rd = torch.nn.functional.pad(br['sequences'][:, :Tc, 2], (1, 0), value=0.001) * 50; fd = torch.nn.functional.pad(bf['sequences'][:, :Tc, 2], (1, 0), value=0.001) * 50
    lens = torch.full((Bs,), Tv + 1)
    D = critic()
    gp = gmt.compute_gradient_penalty(D, rt.to(DEV), ft.to(DEV), rd.to(DEV), fd.to(DEV), lens.to(DEV), DEV, alpha=alpha.to(DEV))
    gp.backward()
    return gp.item(), {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None}
v_full, g_full = run(Tv)
v_rag, g_rag = run(T)
print('penalty', v_full, v_rag)
for k in g_full:
    a, c = g_full[k].double().flatten(), g_rag[k].double().flatten()
    print(f'{k:36s} rel L2 {float((a - c).norm() / a.norm()):.2e}')
