#!/usr/bin/env python3
"""The gradient-penalty pass in isolation at the bench shape (1024 rows of B = 4096, L = 129, H = 128):
compute_gradient_penalty + backward, wall time per pass (events) and the labelled kernels' share.

    python tools/bench_gp.py [--rows 1024] [--no-hoist]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1024)
    ap.add_argument('--no-hoist', action='store_true')
    ap.add_argument('--no-tile16', action='store_true')
    ap.add_argument('--reps', type=int, default=6)
    args = ap.parse_args()
    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd import _lib, ops
    from generative_mouse_trajectories_amd.trainer import compute_gradient_penalty
    ops.GP_HOIST_XPROJ = not args.no_hoist
    ops.GP_TILE16 = not args.no_tile16
    dev = torch.device('cuda:0')
    cfg = gmt.Config()
    cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 64, 128, 128
    torch.manual_seed(0)
    D = gmt.Discriminator(cfg).to(dev).train()
    b, L = args.rows, 129
    g = torch.Generator().manual_seed(1)
    mk = lambda: (torch.rand(b, 1, 2, generator=g) + torch.cumsum(torch.randn(b, L, 2, generator=g) * 0.01, 1)).to(dev)
    real, fake = mk(), mk()
    rd = (torch.rand(b, L, generator=g) * 0.02 + 0.004).to(dev)
    fd = (torch.rand(b, L, generator=g) * 0.02 + 0.004).to(dev)
    lens = torch.full((b,), L, dtype=torch.long, device=dev)
    alpha = torch.rand(b, generator=g).to(dev)

    def one():
        for p in D.parameters():
            p.grad = None
        gp = compute_gradient_penalty(D, real, fake, rd, fd, lens, dev, alpha)
        (10.0 * gp).backward()
        return gp

    for _ in range(3):
        v = one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        one()
    e1.record()
    torch.cuda.synchronize()
    print(f'gradient penalty pass, {b} rows, hoist={ops.GP_HOIST_XPROJ} tile16={ops.GP_TILE16}: {e0.elapsed_time(e1) / args.reps:.2f} ms per pass; gp = {float(v):.6f}')
    _lib.profile_start()
    for _ in range(args.reps):
        one()
    torch.cuda.synchronize()
    prof = _lib.profile_stop()
    tot = 0.0
    for k, (n, ms, fl, by) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print(f'  {k:28s} {n / args.reps:5.1f} launches  {ms / args.reps:7.3f} ms  {fl / (ms * 1e-3) / 1e12 if fl else 0:7.1f} TF/s')
        tot += ms / args.reps
    print(f'  labelled kernels: {tot:.2f} ms per pass')
    gsum = sum(float(p.grad.double().abs().sum()) for p in D.parameters() if p.grad is not None)
    print(f'  sum |grad| = {gsum:.6e}')


if __name__ == '__main__':
    main()
