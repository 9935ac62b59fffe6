import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generative_mouse_trajectories_amd as gmt
from generative_mouse_trajectories_amd import _lib as L_
DEV = torch.device('cuda:0')
B, T, H = int(os.environ.get('B', 4096)), 256, 512
cfg = gmt.Config(); cfg.generator_hidden_dim = H; cfg.discriminator_hidden_dim = 128; cfg.compute_dtype = 'bf16'
cfg.max_generation_steps = T
torch.manual_seed(0)
G = gmt.Generator(cfg).to(DEV).eval()
g = torch.Generator().manual_seed(1)
s = torch.rand(B, 2, generator=g).to(DEV); e = torch.rand(B, 2, generator=g).to(DEV); z = torch.randn(B, cfg.latent_dim, generator=g).to(DEV)
with torch.no_grad():
    for _ in range(2):
        out = G._rollout(s, e, z, None, None, T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = G._rollout(s, e, z, None, None, T)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
g_tok = 8 * H * (2 * H + 7) + 16 * H * H + H * H + 3 * H
print(f'B={B} T={T}: {dt * 1e3:.2f} ms per rollout = {dt / T * 1e6:.1f} us per step, {B / dt:.0f} traj/s, {g_tok * T * B / dt / 1e12:.0f} TFLOP/s algorithmic (F_G)')
