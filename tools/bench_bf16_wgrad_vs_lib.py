#!/usr/bin/env python3
"""gmt_bf16_wgrad_tn against torch.mm(dA^T, X, out_dtype=float32) on config 3's weight-gradient shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from generative_mouse_trajectories_amd import ops_bf16 as OB

dev = 'cuda:0'
BF = torch.bfloat16
M = 257 * 4096


def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (N, I) in [(2048, 512), (4096, 1024), (4096, 512)]:
    dA = (torch.randn(M, N, device=dev) * 0.05).to(BF)
    X = (torch.randn(M, I, device=dev) * 0.5).to(BF)
    t_own = timeit(lambda: OB.wgrad_tn(dA, X))
    fl = 2.0 * M * N * I
    line = f'N={N} I={I}: own {t_own:.2f} ms = {fl / t_own / 1e9:.0f} TF'
    try:
        t_lib = timeit(lambda: torch.mm(dA.t(), X, out_dtype=torch.float32))
        c0 = OB.wgrad_tn(dA, X)
        c1 = torch.mm(dA.t(), X, out_dtype=torch.float32)
        line += f'   library {t_lib:.2f} ms = {fl / t_lib / 1e9:.0f} TF   rel diff {float((c0 - c1).abs().max() / c0.abs().max()):.2e}'
    except Exception as e:
        line += f'   library: {type(e).__name__}: {str(e)[:120]}'
    print(line, flush=True)
    del dA, X
