#!/usr/bin/env python3
"""gmt_bf16_gemm_nt against the library GEMM torch dispatches to (hipBLASLt / rocBLAS) on config 3's hoisted shapes."""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from generative_mouse_trajectories_amd import ops_bf16 as OB

dev = 'cuda:0'
BF = torch.bfloat16
M = 257 * 4096


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (N, K, bias) in [(4096, 512, True), (4096, 1024, True), (1024, 4096, False), (512, 4096, False), (512, 512, True)]:
    A = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
    b32 = torch.randn(N, device=dev) * 0.1 if bias else None
    b16 = b32.to(BF) if bias else None
    t_own = timeit(lambda: OB.gemm_nt(A, W, b32, None, 0, 0.0))
    t_lib = timeit(lambda: F.linear(A, W, b16))
    c0 = OB.gemm_nt(A[:4096], W, b32, None, 0, 0.0).float()
    c1 = F.linear(A[:4096], W, b16).float()
    fl = 2.0 * M * N * K
    print(f'N={N} K={K} bias={bias}: own {t_own:.2f} ms = {fl / t_own / 1e9:.0f} TF   library {t_lib:.2f} ms = {fl / t_lib / 1e9:.0f} TF   '
          f'max diff {float((c0 - c1).abs().max()):.3e}', flush=True)
    del A, W
