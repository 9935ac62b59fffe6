#!/usr/bin/env python3
"""Headline benchmark: trained GAN trajectories/s at B=4096 per GPU, T=128, H_g=H_d=128, fp32
(BASELINE.json configs[1]).

One "step" = one train_discriminator_step + one train_generator_step on a synthetic batch
(n_critic = 1, gradient penalty every 4th D step on 25 % of the batch: reference defaults).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run, one rank per GPU (RCCL); every rank trains its own
B=4096 shard (weak scaling) and the flat G / D gradient buffers are all-reduced once per step.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_LOCAL, T_STEPS, HID, LATENT = 4096, 128, 128, 64
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0
PMC_TRAFFIC_BYTES = {('lstm_fwd_fused_I2', 4096): 4.839e9}    # algorithmic: 4.328e9


def flops_per_trajectory(T=T_STEPS, Hg=HID, Hd=HID):
    """SURVEY.md section 8d algorithmic FLOPs (2*MAC, matmuls only)."""
    g = 8 * Hg * (2 * Hg + 7) + 16 * Hg * Hg + Hg * Hg + 3 * Hg
    F_G = T * g + 2 * (70 * Hg + Hg * Hg)
    d = 2 * (9 * Hd + Hd * Hd) + 32 * Hd * Hd + 48 * Hd * Hd
    F_D = (T + 1) * d + 2 * (2 * Hd * Hd + Hd * Hd / 2 + Hd / 2)
    return 4 * F_G + 8.375 * F_D, F_G, F_D


def synthetic_batch(B, T, seed, device):
    """SURVEY.md section 8d generator: starts~U(0,1)^2, deltas~N(0,0.01^2), dts~U(0.004,0.024)."""
    g = torch.Generator().manual_seed(seed)
    starts = torch.rand(B, 2, generator=g)
    deltas = torch.randn(B, T, 2, generator=g) * 0.01
    dts = torch.rand(B, T, generator=g) * 0.02 + 0.004
    ends = starts + deltas.sum(1)
    seq = torch.cat([deltas, dts.unsqueeze(-1)], -1)
    b = {'starts': starts, 'ends': ends, 'sequences': seq, 'lengths': torch.full((B,), T, dtype=torch.long)}
    return {k: v.to(device) for k, v in b.items()}


def cpu_baseline(threads: int, B: int = 256):
    """The CPU oracle (oracle/gan_oracle.py, pinned to the reference by tests/golden) timed on
    this host, SURVEY.md section 8d: same synthetic batch at a reduced B, dropout on, one warm-up
    iteration, then 4 iterations (D steps 0..3: the first carries the gradient penalty) so the GP
    cost is amortised exactly as in the GPU metric."""
    from oracle import gan_oracle as O
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import paramgen as PG
    torch.set_num_threads(threads)
    T = T_STEPS
    cfg = O.OracleConfig(latent_dim=LATENT, generator_hidden_dim=HID, discriminator_hidden_dim=HID)
    Pg = {k: torch.from_numpy(v) for k, v in PG.make_generator_params(cfg, 1).items()}
    pd, bd = PG.make_discriminator_params(cfg, 2)
    Pd = {k: torch.from_numpy(v) for k, v in pd.items()}
    Bd = {k: torch.from_numpy(v) for k, v in bd.items()}
    batch = synthetic_batch(B, T, 1234, 'cpu')
    gen = torch.Generator().manual_seed(0)
    b = max(1, -(-B // 4))
    og, od = {}, {}

    def iteration(d_step):
        z = torch.randn(B, LATENT, generator=gen)
        noise = {'z': z, 'gp_idx': torch.randperm(B, generator=gen), 'alpha': torch.rand(b, generator=gen)}
        O.discriminator_step(Pg, Pd, Bd, od, batch, noise, cfg, d_step, dropout=True, gen=gen)
        O.generator_step(Pg, Pd, Bd, og, batch, {'z': torch.randn(B, LATENT, generator=gen)}, cfg, [False] * T, 0.0,
                         dropout=True, gen=gen)

    iteration(1)                                    # warm-up (no GP)
    t0 = time.perf_counter()
    for d_step in range(4):
        iteration(d_step)
    dt = time.perf_counter() - t0
    return {'value': 4 * B / dt, 'unit': 'trajectories/s', 'cores': threads, 'kind': 'port',
            'sample': f'oracle (plain PyTorch CPU restatement of RecurrentGAN.py:913-1118), B={B}, T={T}, H=128/128, dropout on: '
                      f'1 warm-up + 4 iterations (D step + G step; GP on the first) in {dt:.1f}s'}


def main():
    if os.environ.get('GMT_DEBUG_HANG'):              # diagnosis aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['GMT_DEBUG_HANG']), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--batch', type=int, default=B_LOCAL)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-overlap', action='store_true', help='run the two generator rollouts of an iteration one after the other')
    ap.add_argument('--cpu-threads', type=int, default=16)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank %= max(1, torch.cuda.device_count())       # (rehearsing N ranks on a one-GPU box: they share the card)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    pg = None
    force_dist = bool(os.environ.get('GMT_FORCE_DIST'))      # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        backend = os.environ.get('GMT_DIST_BACKEND', 'nccl')   # 'gloo' to rehearse the multi-rank path on one GPU
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
        pg = dist.group.WORLD

    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd import _lib

    cfg = gmt.Config()
    cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = LATENT, HID, HID
    cfg.batch_size = args.batch
    torch.manual_seed(0)        # identical replicas on every rank
    G, D = gmt.Generator(cfg), gmt.Discriminator(cfg)
    tr = gmt.Trainer(G, D, cfg, dev, log_dir='/tmp/gmt_runs', world_size=world, process_group=pg)
    if force_dist and world == 1:
        tr.world_size = 2                   # take the all-reduce branch (a sum over one rank)
        tr._enable_overlapped_sync()
        tr.g_optimizer.grad_scale = tr.d_optimizer.grad_scale = 1.0
    tr.epoch = cfg.teacher_forcing_decay_epochs     # steady state: teacher-forcing ratio 0 (fully autoregressive)
    tr.overlap_g_forward = not args.no_overlap      # G-step rollout launched beside the D-step rollout (2 x 128 workgroups)
    G.train(); D.train()
    torch.manual_seed(1234 + rank)                  # per-rank noise
    batch = synthetic_batch(args.batch, T_STEPS, 1234 + rank, dev)
    tf_mask = [False] * T_STEPS                     # same host decisions on every rank

    def one_step():
        tr.train_discriminator_step(batch)
        tr.train_generator_step(batch, {'tf_mask': tf_mask})

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    _lib.profile_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    prof = _lib.profile_stop()
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        f_iter, F_G, F_D = flops_per_trajectory()
        ms = dt / args.steps * 1e3
        value = world * args.batch / (dt / args.steps)
        # dominant kernel: the one with the largest summed duration among the HIP kernels
        kern_flops = {  # algorithmic FLOPs per launch (2*MAC), rows = 32-padded batch of that launch
            'lstm_fwd_fused_I1': lambda rows: 2 * 2 * rows * (T_STEPS + 1) * (4 * HID * HID + 4 * HID * HID),
            'lstm_fwd_fused_I2': lambda rows: 2 * 2 * rows * (T_STEPS + 1) * (4 * HID * 2 * HID + 4 * HID * HID),
            'lstm_fwd_pregates': lambda rows: 2 * 2 * rows * (T_STEPS + 1) * (4 * HID * HID),
            'lstm_bwd': lambda rows: 2 * 2 * rows * (T_STEPS + 1) * (4 * HID * HID),
        }
        roof = None
        if prof:
            cand = {k: v for k, v in prof.items() if k in kern_flops}
            if cand:
                name = max(cand, key=lambda k: cand[k][1])
                n, tot = cand[name]
                # launches mix full-batch calls and (GP-free) half calls; use the full-batch FLOPs and the
                # duration of an average launch of this kernel at the bench batch
                avg_ms = tot / n
                fl = kern_flops[name](args.batch)
                ach = fl / (avg_ms * 1e-3) / 1e12
                # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 on gfx950 +
                # WRITE_SIZE, separate rocprofv3 runs): only known for the kernel / shape that was counted
                traffic = PMC_TRAFFIC_BYTES.get((name, args.batch))
                roof = {'bound': 'mfma', 'kernel': name, 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                        'traffic_unit': 'bytes/launch (profiles/r1_06_pmc_lstm_fwd_I2.txt)' if traffic else None,
                        'avg_launch_ms': avg_ms, 'launches': n,
                        'all_kernels_ms_per_step': {k: v[1] / args.steps for k, v in prof.items()}}
        out = {
            'metric': 'GAN trajectories/sec trained (B=4096,T=128)', 'value': value, 'unit': 'trajectories/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'WGAN-GP train step (D step + G step), B={args.batch}/GPU, T={T_STEPS}, '
                                   f'H_g=H_d={HID}, latent {LATENT}, n_critic 1, GP every 4th D step on 25% of the batch, '
                                   'teacher-forcing ratio 0',
                       'global_batch': world * args.batch, 'seq_len': T_STEPS, 'parallelism': f'dp{world}',
                       'lstm_dropout': cfg.lstm_dropout, 'head_dropout': cfg.head_dropout},
            'model_flops_per_trajectory': f_iter,
            'step_tflops': value * f_iter / 1e12,
            'frac_of_f32_mfma_peak': value * f_iter / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
            'roofline': roof,
            'hbm_peak_allocated_gb': torch.cuda.max_memory_allocated() / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(min(args.cpu_threads, os.cpu_count() or 1))
            except Exception as e:                                  # the baseline must never break the bench line
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
