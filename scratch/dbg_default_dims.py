"""Reference default dims (G 256 / D 128, latent 64) at B=4096, T=128: time per iteration."""
import sys, time, torch
sys.path.insert(0, '.')
import generative_mouse_trajectories_amd as gmt
from bench import synthetic_batch
dev = torch.device('cuda:0')
cfg = gmt.Config()
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 128
torch.manual_seed(0)
tr = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, dev, log_dir='/tmp/gmt_runs_d')
tr.epoch = cfg.teacher_forcing_decay_epochs
tr.overlap_g_forward = True
batch = synthetic_batch(B, T, 1234, dev)
tf = [False] * T
def step():
    a = tr.train_discriminator_step(batch); b = tr.train_generator_step(batch, {'tf_mask': tf}); return a, b
for _ in range(4): m = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): m = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print(f'G{cfg.generator_hidden_dim}/D{cfg.discriminator_hidden_dim} B={B} T={T}: {dt * 1e3:.1f} ms/iter, {B / dt:.0f} traj/s', m)
print('peak GB', torch.cuda.max_memory_allocated() / 1e9)
