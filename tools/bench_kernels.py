#!/usr/bin/env python3
"""Per-kernel roofline of the HBM-bound kernels (SURVEY.md section 8d byte counts) and the config-5
pipeline (generator-only sampling + telemetry recompute), measured with events on the launch stream.

    python tools/bench_kernels.py [--samples 131072] > profiles/<round>_per_kernel_roofline.txt
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_HBM = 8000.0      # GB/s, MI355X_MICROARCH.md
PEAK_F32 = 157.3       # TFLOP/s


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=131072, help='config-5 trajectories per launch (1M = 8 such launches)')
    args = ap.parse_args()
    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd import ops, telemetry as TEL
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    print(f'# device: {torch.cuda.get_device_name(0)}')

    # --- kinematics (mode A, fp32): 48 B/point forward, 60 B/point backward
    B, L = 4096, 129
    pos = (torch.rand(1, B, 2, device=dev) + torch.cumsum(torch.randn(L, B, 2, device=dev) * 0.01, 0)).requires_grad_(True)
    dts = (torch.rand(L, B, device=dev) * 0.02 + 0.004).requires_grad_(True)
    t = timed(lambda: ops.kinematics_tm(pos.detach(), dts.detach()), reps=50)
    print(f'kinematics fwd  B={B} L={L}: {t * 1e6:8.1f} us  {48 * B * L / t / 1e9:8.1f} GB/s  ({48 * B * L / t / 1e9 / PEAK_HBM:.1%} of 8 TB/s; 25 MB working set: launch-bound)')
    Bb = 1 << 18
    posb = torch.rand(L, Bb, 2, device=dev); dtsb = torch.rand(L, Bb, device=dev) * 0.02 + 0.004
    t = timed(lambda: ops.kinematics_tm(posb, dtsb), reps=10)
    print(f'kinematics fwd  B={Bb} L={L}: {t * 1e6:8.1f} us  {48 * Bb * L / t / 1e9:8.1f} GB/s  ({48 * Bb * L / t / 1e9 / PEAK_HBM:.1%})')
    from generative_mouse_trajectories_amd import _lib as L_
    gfeat = torch.randn(L, Bb, 9, device=dev); gpos = torch.empty(L, Bb, 2, device=dev); gdts = torch.empty(L, Bb, device=dev)
    t = timed(lambda: L_.call('gmt_kinematics_bwd', L_.ptr(posb), L_.ptr(dtsb), L_.ptr(gfeat), L_.ptr(gpos), L_.ptr(gdts),
                              L, Bb, 1e-8, L_.stream()), reps=10)
    print(f'kinematics bwd  B={Bb} L={L}: {t * 1e6:8.1f} us  {60 * Bb * L / t / 1e9:8.1f} GB/s  ({60 * Bb * L / t / 1e9 / PEAK_HBM:.1%})')
    pb = db = f = g = None
    del gfeat, gpos, gdts
    del posb, dtsb, pb, db, f, g

    # --- Adam: 28 B/parameter
    for n in (1020808, 1 << 26):
        p, gr, m, v = (torch.randn(n, device=dev) for _ in range(4))
        v.abs_()
        t = timed(lambda: ops.adam_step_(p, gr, m, v, 1e-4, (0.5, 0.9), 1e-8, 3), reps=50)
        print(f'adam            n={n}: {t * 1e6:8.1f} us  {28 * n / t / 1e9:8.1f} GB/s  ({28 * n / t / 1e9 / PEAK_HBM:.1%})')
    del p, gr, m, v

    # --- config 5: generator-only sampling (all max steps, no early-stop shortcut) + telemetry recompute
    cfg = gmt.Config()
    cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 64, 128, 128
    cfg.max_generation_steps = 256
    G = gmt.Generator(cfg).to(dev).eval()
    N, T = args.samples, 256
    start = torch.rand(N, 2, device=dev); end = torch.rand(N, 2, device=dev)
    z = torch.randn(N, cfg.latent_dim, device=dev)
    with torch.no_grad():
        t_gen = timed(lambda: G._rollout(start, end, z, None, None, T), reps=3, warm=1)
        out = G._rollout(start, end, z, None, None, T)              # [T, N, 3]
    Hg = cfg.generator_hidden_dim
    g_flops = 8 * Hg * (2 * Hg + 7) + 16 * Hg * Hg + Hg * Hg + 3 * Hg
    print(f'G sampling      N={N} T={T}: {t_gen * 1e3:8.2f} ms  {N / t_gen:10.0f} traj/s  '
          f'{N * T * g_flops / t_gen / 1e12:6.1f} TFLOP/s algorithmic ({N * T * g_flops / t_gen / 1e12 / PEAK_F32:.1%} of f32 MFMA)')
    x = (start[:, 0].unsqueeze(0) + torch.cumsum(out[:, :, 0], 0)).double().contiguous()
    y = (start[:, 1].unsqueeze(0) + torch.cumsum(out[:, :, 1], 0)).double().contiguous()
    dt = out[:, :, 2].double().contiguous()
    kin = torch.empty(6, T, N, device=dev, dtype=torch.float64)
    t_tel = timed(lambda: TEL.telemetry_kinematics(x, y, dt, None, time_is_dt=True, out=kin), reps=5, warm=1)
    print(f'telemetry (f64) N={N} T={T}: {t_tel * 1e3:8.2f} ms  {72 * N * T / t_tel / 1e9:8.1f} GB/s  ({72 * N * T / t_tel / 1e9 / PEAK_HBM:.1%} of 8 TB/s)')
    tot = t_gen + t_tel
    print(f'config 5 (scaled to 1M x 256): {tot * (1 << 20) / N:8.2f} s per 1M trajectories = {N / tot:10.0f} trajectories/s sampled + re-measured')


if __name__ == '__main__':
    main()
