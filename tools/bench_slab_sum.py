#!/usr/bin/env python3
"""gmt_slab_sum on the slab shapes of the step's weight-gradient contractions (after a write of the slabs, like the real call)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from generative_mouse_trajectories_amd import _lib as L_

dev = 'cuda:0'
for (k, n) in [(256, 512 * 128), (128, 512 * 256), (256, 128 * 128), (256, 512)]:
    slab = torch.randn(k, n, device=dev)
    out = torch.empty(n, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for it in range(30):
        slab.mul_(1.0001)                      # the producer's write (the slabs sit where wgrad_kernel left them)
        a.record()
        L_.call('gmt_slab_sum', slab.data_ptr(), out.data_ptr(), n, k, 0, L_.stream(), label='skip')
        b.record()
        torch.cuda.synchronize()
        if it >= 5:
            tot += a.elapsed_time(b)
    ref = slab.sum(0)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f'k={k} n={n}: {tot / 25 * 1e3:.1f} us  ({k * n * 4 / (tot / 25 * 1e-3) / 1e12:.2f} TB/s)  rel err {err:.1e}', flush=True)
