import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import generative_mouse_trajectories_amd as gmt
DEV = torch.device('cuda:0')
cfg = gmt.Config()
cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 16, 64, 64
cfg.min_trajectory_length, cfg.batch_size, cfg.gp_every_n = 3, 8, 2
cfg.lstm_dropout = cfg.head_dropout = 0.0
ds = gmt.MouseTrajectoryDataset('tests/golden/telemetry_sample.csv', cfg)
torch.manual_seed(0)
tr = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, DEV, log_dir='/tmp/r1')
batch = gmt.collate_trajectories([ds[i] for i in range(8)])
for _ in range(3):
    tr.train_discriminator_step(batch); tr.train_generator_step(batch)
tr.save_checkpoint('/tmp/now.pt')
torch.manual_seed(1)
tr2 = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, DEV, log_dir='/tmp/r2')
tr2.load_checkpoint('/tmp/now.pt')
for name, a, b in (('G', tr.generator, tr2.generator), ('D', tr.discriminator, tr2.discriminator)):
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if not torch.equal(sa[k], sb[k]):
            print(name, k, (sa[k] - sb[k]).abs().max().item())
print('flat equal', torch.equal(tr.d_flat.flat, tr2.d_flat.flat), torch.equal(tr.g_flat.flat, tr2.g_flat.flat))
print('m equal', torch.equal(tr.d_optimizer.exp_avg, tr2.d_optimizer.exp_avg), torch.equal(tr.d_optimizer.exp_avg_sq, tr2.d_optimizer.exp_avg_sq))
B, T = batch['sequences'].shape[:2]
g = torch.Generator().manual_seed(9)
tr2.d_step = tr.d_step
noise_d = {'z': torch.randn(B, cfg.latent_dim, generator=g), 'gp_idx': torch.randperm(B, generator=g), 'alpha': torch.rand(2, 1, 1, generator=g)}
for t_ in (tr, tr2):
    t_.generator.train(); t_.discriminator.train()
print(tr.discriminator.training, tr2.discriminator.training, tr.d_step, tr2.d_step)
print(tr.train_discriminator_step(batch, noise_d)); print(tr2.train_discriminator_step(batch, noise_d))
