"""Trainer.overlap_g_forward with the bf16 kernels: same G step as an ordinary trainer fed the same noise?"""
import sys, os, copy, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import generative_mouse_trajectories_amd as gmt
faulthandler.dump_traceback_later(80, exit=True)
cfg = gmt.Config()
cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 16, 512, 256
cfg.lstm_dropout, cfg.head_dropout = 0.0, 0.0
cfg.compute_dtype = 'bf16'
torch.manual_seed(3)
G, D = gmt.Generator(cfg), gmt.Discriminator(cfg)
G2, D2 = copy.deepcopy(G), copy.deepcopy(D)
dev = torch.device('cuda:0')
tr1 = gmt.Trainer(G, D, cfg, dev, log_dir='/tmp/gmt_runs_t1')
tr2 = gmt.Trainer(G2, D2, cfg, dev, log_dir='/tmp/gmt_runs_t2')
tr1.overlap_g_forward = True
B, T = 96, 12
g = torch.Generator().manual_seed(11)
deltas = torch.randn(B, T, 2, generator=g) * 0.01
starts = torch.rand(B, 2, generator=g)
batch = {'starts': starts.to(dev), 'ends': (starts + deltas.sum(1)).to(dev),
         'sequences': torch.cat([deltas, torch.rand(B, T, 1, generator=g) * 0.02 + 0.004], -1).to(dev),
         'lengths': torch.randint(3, T + 1, (B,), generator=g).sort(descending=True).values.to(dev)}
for it in range(2):
    nd = {'z': torch.randn(B, cfg.latent_dim, generator=g), 'gp_idx': torch.randperm(B, generator=g), 'alpha': torch.rand(-(-B // 4), generator=g)}
    r1 = tr1.train_discriminator_step(batch, dict(nd))
    spec = tr1._spec
    zs, mask = spec['z'].clone(), list(spec['tf_mask'])
    r2 = tr2.train_discriminator_step(batch, dict(nd))
    print('D', {k: (round(r1[k], 6), round(r2[k], 6)) for k in r1})
    g1 = tr1.train_generator_step(batch)
    g2 = tr2.train_generator_step(batch, {'z': zs, 'tf_mask': mask})
    print('G', {k: (round(g1[k], 6), round(g2[k], 6)) for k in g1})
print('max |dG|', float((tr1.g_flat.flat - tr2.g_flat.flat).abs().max()))
