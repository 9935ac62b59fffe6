#!/usr/bin/env python3
"""Launches each hand-written contraction kernel of the fp32 step a few times at the bench shape (B = 4096, L = 129,
H = 128), for counter passes:

    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES \
              SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d <dir> -- python3 tools/run_hot_kernels.py
    python3 tools/pmc_table.py <dir>
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import generative_mouse_trajectories_amd as gmt
from generative_mouse_trajectories_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)
B, L, H = 4096, 129, 128
R = L * B
cfg = gmt.Config()
cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 64, H, H
D = gmt.Discriminator(cfg).to(dev).train()
pos = torch.rand(1, B, 2, device=dev) + torch.cumsum(torch.randn(L, B, 2, device=dev) * 0.01, 0)
dts = torch.rand(L, B, device=dev) * 0.02 + 0.004
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
for _ in range(2):                                   # critic forward + backward: encoder_fwd, lstm_fwd / bwd, dx_gemm_res, wgrad
    D.zero_grad()
    p = pos.clone().requires_grad_(True)
    s = D.forward_tm(p, dts, lens)
    s.sum().backward()
# the hoisted input projection and the resident dx kernels on their own
x = torch.randn(R, 2 * H, device=dev)
lw = D.lstm.layer_weights(1)
w_ih = [lw[0], lw[4]]
w_pk = ops.pack_forward(w_ih, H, 2 * H)
for _ in range(2):
    ops.lstm_xproj(x, w_pk, None, 2, H, 'xproj')
dA = torch.randn(2, R, 4 * H, device=dev)
for _ in range(2):
    ops.dx_from_da(dA, w_ih, R)
torch.cuda.synchronize()
print('done')
