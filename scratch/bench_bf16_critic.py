"""Critic forward + backward at the config-3 shape (8192 rows = real + fake, T = 256, H = 512, bf16): per-kernel table."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generative_mouse_trajectories_amd as gmt
from generative_mouse_trajectories_amd import _lib as L_
DEV = torch.device('cuda:0')
B, T, Hd = int(os.environ.get('B', 8192)), int(os.environ.get('T', 256)), 512
cfg = gmt.Config(); cfg.discriminator_hidden_dim = Hd; cfg.generator_hidden_dim = 128; cfg.compute_dtype = 'bf16'
cfg.lstm_dropout = float(os.environ.get('DROP', 0.1))
torch.manual_seed(0)
D = gmt.Discriminator(cfg).to(DEV).train()
g = torch.Generator().manual_seed(1)
pos = (torch.rand(1, B, 2, generator=g) + torch.cumsum(torch.randn(T + 1, B, 2, generator=g) * 0.01, 0)).to(DEV)
dts = (torch.rand(T + 1, B, generator=g) * 0.02 + 0.004).to(DEV)
lens = torch.full((B,), T + 1, dtype=torch.int32, device=DEV)
def step():
    D.zero_grad(set_to_none=True)
    s = D.forward_tm(pos, dts, lens)
    s.mean().backward()
step(); torch.cuda.synchronize()
print('peak GB', torch.cuda.max_memory_allocated() / 1e9)
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); t1 = time.perf_counter()
d = 2 * (9 * Hd + Hd * Hd) + 32 * Hd * Hd + 48 * Hd * Hd
F_D = (T + 1) * d
print(f'fwd+bwd {1e3 * (t1 - t0):.1f} ms  = {3 * F_D * B / (t1 - t0) / 1e12:.0f} TFLOP/s algorithmic (3 F_D per row)')
L_.profile_start()
step(); torch.cuda.synchronize()
prof = L_.profile_stop()
tot = 0
for k, (n, ms, fl, nb) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print(f'{k:24s} n={n:3d} {ms:9.2f} ms' + (f'  {fl / ms / 1e9:7.0f} TFLOP/s' if fl else ''))
print('labelled total', tot)
