"""The data-parallel exchange on the GPU: a one-rank RCCL group (all a one-GPU box can hold) drives the same
code the N-GPU run uses - autograd hooks fill the flat gradient buckets and start asynchronous all-reduces
during the backward pass - and the step must come out identical to the plain single-GPU step.  Collected
last (file name), so a box without a working RCCL does not hide the other tests."""
import copy
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def test_overlapped_bucketed_allreduce_is_the_same_step(tmp_path):
    import generative_mouse_trajectories_amd as gmt
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    port = 23000 + os.getpid() % 4000
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=DEV)
    try:
        cfg = gmt.Config()
        cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 16, 64, 64
        cfg.lstm_dropout, cfg.head_dropout = 0.0, 0.0
        torch.manual_seed(3)
        G, D = gmt.Generator(cfg), gmt.Discriminator(cfg)
        G2, D2 = copy.deepcopy(G), copy.deepcopy(D)
        tr1 = gmt.Trainer(G, D, cfg, DEV, log_dir=str(tmp_path / 'a'))
        tr2 = gmt.Trainer(G2, D2, cfg, DEV, log_dir=str(tmp_path / 'b'), world_size=2, process_group=dist.group.WORLD)
        tr2.g_optimizer.grad_scale = tr2.d_optimizer.grad_scale = 1.0      # the "sum" runs over one rank
        assert [len(b['idx']) for b in tr2._d_sync.buckets] == [4, 8, 8, 6]
        assert [len(b['idx']) for b in tr2._g_sync.buckets] == [9, 12]
        B, T = 64, 12
        g = torch.Generator().manual_seed(11)
        deltas = torch.randn(B, T, 2, generator=g) * 0.01
        batch = {'starts': torch.rand(B, 2, generator=g), 'sequences': torch.cat([deltas, torch.rand(B, T, 1, generator=g) * 0.02 + 0.004], -1),
                 'lengths': torch.full((B,), T, dtype=torch.long)}
        batch['ends'] = batch['starts'] + deltas.sum(1)
        for step in range(2):                           # d_step 0 has the gradient penalty, d_step 1 has not
            noise_d = {'z': torch.randn(B, cfg.latent_dim, generator=g), 'gp_idx': torch.randperm(B, generator=g),
                       'alpha': torch.rand(16, 1, 1, generator=g)}
            noise_g = {'z': torch.randn(B, cfg.latent_dim, generator=g), 'tf_mask': [False] * T}
            m1 = (tr1.train_discriminator_step(batch, noise_d), tr1.train_generator_step(batch, noise_g))
            m2 = (tr2.train_discriminator_step(batch, noise_d), tr2.train_generator_step(batch, noise_g))
            for k1, k2 in zip(m1, m2):
                for key in k1:
                    assert abs(k1[key] - k2[key]) <= 1e-5 * max(1.0, abs(k1[key])), (step, key, k1[key], k2[key])
            assert all(tr2._d_sync.launched) and all(tr2._g_sync.launched)
        for nm, a, b in (('G params', tr1.g_flat.flat, tr2.g_flat.flat), ('D params', tr1.d_flat.flat, tr2.d_flat.flat),
                         ('G grads', tr1.g_flat.grad, tr2.g_flat.grad), ('D grads', tr1.d_flat.grad, tr2.d_flat.grad)):
            # (not bitwise: a few library GEMMs of the step accumulate split-K partial sums with atomics, so two
            # runs of the SAME trainer already differ in the last bit of a handful of elements)
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (nm, float((a - b).abs().max()), int((a != b).sum()), a.numel())
    finally:
        dist.destroy_process_group()
