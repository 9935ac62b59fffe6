import sys, os, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R_, R_ + '/tests', R_ + '/tests/golden'): sys.path.insert(0, p)
import generative_mouse_trajectories_amd as gmt
import helpers as H, paramgen as PG
from oracle import gan_oracle as O
DEV = torch.device('cuda:0')
over = dict(latent_dim=16, generator_hidden_dim=512, discriminator_hidden_dim=32)
cfg = gmt.Config()
for k, v in over.items(): setattr(cfg, k, v)
cfg.compute_dtype = 'bf16'; cfg.lstm_dropout = 0.0
Bs, seed = 200, 121
pg = PG.make_generator_params(O.OracleConfig(**over), seed)
pg['output_layer.0.bias'] = pg['output_layer.0.bias'] + 3.0      # every head unit far on the linear side of its LeakyReLU
G = gmt.Generator(cfg); G.load_state_dict({k: torch.from_numpy(v) for k, v in pg.items()}, strict=True); G = G.to(DEV).train()
for T in (1, 4, 12):
    b = H.batch_t(Bs, T, seed + 1, [T] * Bs, torch.float32)
    z = torch.randn(Bs, 16, generator=torch.Generator().manual_seed(seed + 2))
    R = torch.randn(Bs, T, 3, generator=torch.Generator().manual_seed(seed + 3))
    tf = [False] * T if T < 12 else [True, False, False, True, False, True, True, False, False, False, True, False]
    G.zero_grad()
    out, _ = G(b['starts'].to(DEV), b['ends'].to(DEV), z.to(DEV), b['sequences'].to(DEV), tf_mask=tf)
    (out * R.to(DEV)).sum().backward()
    res = {}
    for dt in (torch.float32, torch.float64):
        P = {k: H.t(v, dt).requires_grad_(True) for k, v in pg.items()}
        ref = O.generator_forward(P, b['starts'].to(dt), b['ends'].to(dt), z.to(dt), b['sequences'].to(dt), tf)
        keys = list(P.keys())
        res[dt] = dict(zip(keys, torch.autograd.grad((ref * R.to(dt)).sum(), [P[k] for k in keys], allow_unused=True)))
    line = []
    for k, prm in G.named_parameters():
        g64 = res[torch.float64][k]
        if g64 is None: continue
        line.append(f'{k.replace("condition_encoder", "ce").replace("output_layer", "ol").replace("weight", "w")} {H.relerr(prm.grad.cpu().reshape(g64.shape), g64):.1e}/{H.relerr(res[torch.float32][k], g64):.0e}')
    print(f'T={T}: ' + ', '.join(line))
