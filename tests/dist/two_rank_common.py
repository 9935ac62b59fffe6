"""Shapes, data and noise shared by the two-rank children and the single-process reference run."""
import numpy as np
import torch

B_LOCAL, WORLD, T, ITERS, EPOCH = 64, 2, 16, 2, 50      # epoch 50: teacher-forcing ratio 0.5


def config(gmt):
    cfg = gmt.Config()
    cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 16, 64, 64
    cfg.lstm_dropout, cfg.head_dropout = 0.0, 0.0        # dropout masks are indexed by the LOCAL row: off for exactness
    return cfg


def full_batch():
    g = torch.Generator().manual_seed(7)
    B = B_LOCAL * WORLD
    lengths = torch.randint(3, T + 1, (B,), generator=g)
    lengths[0] = T; lengths[B_LOCAL] = T - 5              # ragged, and the two shards' mask sums differ
    mask = (torch.arange(T)[None, :] < lengths[:, None]).float()
    deltas = torch.randn(B, T, 2, generator=g) * 0.01 * mask[:, :, None]
    dts = (torch.rand(B, T, generator=g) * 0.02 + 0.004) * mask
    starts = torch.rand(B, 2, generator=g)
    return {'starts': starts, 'ends': starts + deltas.sum(1) + 0.03, 'sequences': torch.cat([deltas, dts.unsqueeze(-1)], -1),
            'lengths': lengths}


def noise(it, rank):
    g = torch.Generator().manual_seed(1000 + 10 * it + rank)
    nb = -(-B_LOCAL // 4)
    nd = {'z': torch.randn(B_LOCAL, 16, generator=g), 'gp_idx': torch.randperm(B_LOCAL, generator=g), 'alpha': torch.rand(nb, generator=g)}
    ng = {'z': torch.randn(B_LOCAL, 16, generator=g)}
    return nd, ng


def global_noise(it):
    """The same draws arranged for ONE process training the concatenated batch."""
    per = [noise(it, r) for r in range(WORLD)]
    nb = -(-B_LOCAL // 4)
    sel = torch.cat([per[r][0]['gp_idx'][:nb] + r * B_LOCAL for r in range(WORLD)])
    rest = torch.tensor([i for i in range(B_LOCAL * WORLD) if i not in set(sel.tolist())], dtype=torch.long)
    nd = {'z': torch.cat([p[0]['z'] for p in per]), 'gp_idx': torch.cat([sel, rest]),
          'alpha': torch.cat([p[0]['alpha'] for p in per])}
    ng = {'z': torch.cat([p[1]['z'] for p in per])}
    return nd, ng
