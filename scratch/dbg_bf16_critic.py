import sys, os, copy, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import generative_mouse_trajectories_amd as gmt
import helpers as H, paramgen as PG
from oracle import gan_oracle as O
DEV = torch.device('cuda:0')
Hd, Bs, T, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 48, 64, 71
over = dict(latent_dim=8, generator_hidden_dim=32, discriminator_hidden_dim=Hd)
def mk(dtype):
    cfg = gmt.Config()
    for k, v in over.items(): setattr(cfg, k, v)
    cfg.lstm_dropout = cfg.head_dropout = 0.0
    cfg.compute_dtype = dtype
    D = gmt.Discriminator(cfg)
    pd, bd = PG.make_discriminator_params(O.OracleConfig(**over), seed)
    sd = {k: torch.from_numpy(v) for k, v in pd.items()}; sd.update({k: torch.from_numpy(v) for k, v in bd.items()})
    D.load_state_dict(sd, strict=True)
    return D.to(DEV).train()
rng = np.random.default_rng(seed + 1)
lengths = rng.integers(T // 2, T + 1, Bs); lengths[:2] = T
b = H.batch_t(Bs, T, seed + 2, lengths.tolist(), torch.float32)
traj = gmt.trajectory_to_absolute(b['starts'], b['sequences'][:, :, :2])
dts = torch.nn.functional.pad(b['sequences'][:, :, 2], (1, 0), value=0.001)
lp1 = b['lengths'] + 1
cap = {}
res = {}
for dt in ('fp32', 'bf16'):
    D = mk(dt)
    D.output_layers[0].register_forward_pre_hook(lambda m, a, dt=dt: cap.__setitem__(dt, a[0].detach().clone()))
    res[dt] = D(traj.to(DEV), dts.to(DEV), lp1.to(DEV)).detach()
hf32, hfb = cap['fp32'][:Bs], cap['bf16'][:Bs]
print('h_final: max|fp32| %.4f  max abs diff %.3e  rel %.3e' % (hf32.abs().max(), (hf32 - hfb).abs().max(), (hf32 - hfb).abs().max() / hf32.abs().max()))
print('score: fp32', res['fp32'][:4].flatten().tolist(), 'bf16', res['bf16'][:4].flatten().tolist())
print('score rel', ((res['fp32'] - res['bf16']).abs().max() / res['fp32'].abs().max()).item())

# ---- torch emulation of the same precision policy (bf16 where the kernels store bf16, fp32 elsewhere)
from generative_mouse_trajectories_amd import ops
BF = torch.bfloat16
rb = lambda x: x.to(BF).float()
D = mk('bf16')
with torch.no_grad():
    Bp = 256
    pos = torch.zeros(T + 1, Bp, 2, device=DEV); pos[:, :Bs] = traj.to(DEV).transpose(0, 1)
    dd = torch.zeros(T + 1, Bp, device=DEV); dd[:, :Bs] = dts.to(DEV).transpose(0, 1)
    ln = torch.ones(Bp, device=DEV, dtype=torch.int32); ln[:Bs] = lp1.to(DEV)
    sn = D.spectral_weights()
    w1, w2 = sn[0], sn[1]
    fe = D.feature_encoder
    feat = ops.kinematics_tm(pos, dd)
    lre = lambda v: torch.nn.functional.leaky_relu(v, 0.2)
    h1 = rb(lre(feat @ w1.t() + fe[0].bias))
    print('max |feat| %.3g  max |h1| %.3g' % (feat.abs().max(), h1.abs().max()))
    x = rb(lre(h1 @ rb(w2).t() + fe[2].bias))
    print('max |e| %.3g' % x.abs().max())
    L = T + 1
    for l in range(2):
        ws = D.lstm.layer_weights(l)
        outs = []
        for d in range(2):
            wih, whh, bih, bhh = ws[4 * d: 4 * d + 4]
            pre = rb(x @ rb(wih).t() + (bih + bhh))
            h = torch.zeros(Bp, Hd, device=DEV); c = torch.zeros(Bp, Hd, device=DEV)
            o_t = [None] * L
            for t in (range(L) if d == 0 else range(L - 1, -1, -1)):
                i, f, g, o = (pre[t] + h @ rb(whh).t()).split(Hd, 1)
                cn = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                hn = rb(torch.sigmoid(o) * torch.tanh(cn))
                m = (t < ln).float().unsqueeze(1)
                c = m * cn + (1 - m) * c
                h = m * hn + (1 - m) * h
                o_t[t] = h if d == 0 else h * m
            outs.append(torch.stack(o_t))
        x = torch.cat(outs, 2)
        fin = torch.cat([outs[0][L - 1], outs[1][0]], 1)
    print('emulated h_final vs kernel bf16: max abs diff %.3e ; emulated vs fp32 path %.3e' % ((fin[:Bs] - hfb).abs().max(), (fin[:Bs] - hf32).abs().max()))
