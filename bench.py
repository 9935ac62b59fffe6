#!/usr/bin/env python3
"""Headline benchmark: trained GAN trajectories/s at B=4096 per GPU, T=128, H_g=H_d=128, fp32
(BASELINE.json configs[1]).

One "step" = one train_discriminator_step + one train_generator_step on a synthetic batch
(n_critic = 1, gradient penalty every 4th D step on 25 % of the batch: reference defaults).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run, one rank per GPU (RCCL); every rank trains its own
B=4096 shard (weak scaling) and the flat G / D gradient buffers are all-reduced once per step.
Rank 0 prints ONE JSON line.

    python bench.py --config 5 [--steps K]

measures BASELINE.json configs[4] instead: generator-only sampling of 1 048 576 trajectories x 256 steps (8 launch
groups of 131 072, one "step" each) + the f64 telemetry metric recompute of every generated point, trajectories/s.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 matrix peak (the headline 5 PF figure is 2:1 sparse)
# HBM bytes per launch from the PMC passes under profiles/ (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3
# runs): known only for the kernel / shape that was counted
PMC_TRAFFIC_BYTES = {('lstm_fwd_fused_I2', 4096): (4.86e9, 'profiles/r3_08_pmc_hbm.txt'),
                     ('lstm_bwd_fx', 4096): (7.85e9, 'profiles/r3_08_pmc_hbm.txt'),
                     ('lstm_fwd_fused_I1', 4096): (3.95e9, 'profiles/r3_08_pmc_hbm.txt'),
                     # config 3: a labelled launch is a whole layer sweep of 257 step launches (183.6 MB read + 61.0 MB written each)
                     ('bf16_lstm_bwd', 4096): (257 * 244.6e6, 'profiles/r2_06_pmc_lstm_steps.txt: per step launch x 257'),
                     ('bf16_lstm_fwd', 4096): (257 * 147.9e6, 'profiles/r2_06_pmc_lstm_steps.txt: per step launch x 257')}


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


def physical_cores() -> int:
    """Physical cores of this host from /proc/cpuinfo (distinct (physical id, core id) pairs); 0 if unknown."""
    try:
        seen, phys, core = set(), None, None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen)
    except OSError:
        return 0


def _cpu_steps(threads: int, B: int, iters: int):
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
    for d_step in range(iters):
        iteration(d_step)
    return time.perf_counter() - t0


def cpu_baseline(threads: int, B: int = 256):
    """The CPU oracle (oracle/gan_oracle.py, pinned to the reference by tests/golden) timed on
    this host, SURVEY.md section 8d: same synthetic batch at a reduced B, dropout on, one warm-up
    iteration, then 4 iterations (D steps 0..3: the first carries the gradient penalty) so the GP
    cost is amortised exactly as in the GPU metric.  Also a single-thread figure on a smaller sample."""
    dt = _cpu_steps(threads, B, 4)
    b1 = 32
    dt1 = _cpu_steps(1, b1, 4)
    torch.set_num_threads(threads)
    return {'value': 4 * B / dt, 'unit': 'trajectories/s', 'cores': threads, 'kind': 'port',
            'sample': f'oracle (plain PyTorch CPU restatement of RecurrentGAN.py:913-1118), B={B}, T={T_STEPS}, H=128/128, dropout on: '
                      f'1 warm-up + 4 iterations (D step + G step; GP on the first) in {dt:.1f}s on {threads} threads',
            'single_thread': {'value': 4 * b1 / dt1, 'cores': 1,
                              'sample': f'same, B={b1}: 1 warm-up + 4 iterations in {dt1:.1f}s'},
            'host_physical_cores': physical_cores(), 'host_logical_cpus': os.cpu_count(),
            'note': 'cores = torch intra-op threads used (the GPU box grants this job 16 CPUs of the host)'}


def kernel_table(prof, steps, peak_tflops=PEAK_F32_MFMA_TFLOPS):
    """Per labelled kernel: launches and ms per step, achieved TFLOP/s or GB/s from the algorithmic work each call
    declared, fraction of the matching peak."""
    rows = {}
    for k, (n, ms, fl, by) in prof.items():
        r = {'launches_per_step': n / steps, 'ms_per_step': ms / steps, 'avg_launch_ms': ms / n}
        if fl > 0 and by > 0:
            # a call that declares both (the bf16 recurrence steps: as much memory traffic as matrix work) is bounded by the
            # roof it is nearer to; the other fraction is reported beside it
            f_m, f_h = fl / (ms * 1e-3) / 1e12 / peak_tflops, by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS
            if f_h >= f_m:
                r.update(bound='hbm', achieved=by / (ms * 1e-3) / 1e9, unit='GB/s', frac=f_h, frac_of_mfma_peak=f_m)
            else:
                r.update(bound='mfma', achieved=fl / (ms * 1e-3) / 1e12, unit='TFLOP/s', frac=f_m, frac_of_hbm_peak=f_h)
        elif fl > 0:
            r.update(bound='mfma', achieved=fl / (ms * 1e-3) / 1e12, unit='TFLOP/s')
            r['frac'] = r['achieved'] / peak_tflops
        elif by > 0:
            r.update(bound='hbm', achieved=by / (ms * 1e-3) / 1e9, unit='GB/s')
            r['frac'] = r['achieved'] / PEAK_HBM_GBS
        rows[k] = r
    return rows


def roofline_of_dominant(prof, steps, batch, peak_tflops=PEAK_F32_MFMA_TFLOPS):
    """The dominant kernel = the labelled HIP kernel with the largest summed duration in the timed region (ALL of
    them are candidates: recurrent sweeps, weight-gradient and input-gradient contractions, rollouts)."""
    rows = kernel_table(prof, steps, peak_tflops)
    cand = {k: r for k, r in rows.items() if 'achieved' in r}
    if not cand:
        return None, rows
    name = max(cand, key=lambda k: cand[k]['ms_per_step'])
    r = cand[name]
    n, ms, fl, by = prof[name]
    pmc = PMC_TRAFFIC_BYTES.get((name, batch))
    roof = {'bound': r['bound'], 'kernel': name, 'achieved': r['achieved'],
            'peak': peak_tflops if r['bound'] == 'mfma' else PEAK_HBM_GBS, 'unit': r['unit'], 'frac': r['frac'],
            'traffic': pmc[0] if pmc else None, 'traffic_unit': f'bytes/launch ({pmc[1]})' if pmc else None,
            'avg_launch_ms': ms / n, 'launches': n,
            'algorithmic_work_per_launch': (by if r['bound'] == 'hbm' else fl) / n,
            'all_kernels_ms_per_step': {k: v['ms_per_step'] for k, v in rows.items()}}
    return roof, rows


def bench_config5(args, dev):
    """BASELINE.json configs[4]: 1 048 576 x 256 sampled in launch groups of 131 072 + the f64 metric recompute."""
    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd import _lib, sampling
    cfg = gmt.Config()
    cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = LATENT, HID, HID
    cfg.max_generation_steps = 256
    torch.manual_seed(0)
    G = gmt.Generator(cfg).to(dev).eval()
    N, T = args.samples, 256
    groups = args.steps if args.steps_given else (1 << 20) // N
    g = torch.Generator().manual_seed(1234)
    start = torch.rand(N, 2, generator=g).to(dev)
    end = torch.rand(N, 2, generator=g).to(dev)
    z = torch.randn(N, cfg.latent_dim, generator=g).to(dev)
    kin = torch.empty(6, T + 1, N, device=dev, dtype=torch.float64)
    warm = max(1, args.warmup if args.warmup_given else 1)
    for _ in range(warm):
        sampling.sample_and_measure(G, start, end, z, T, kin)
    torch.cuda.synchronize()
    _lib.profile_start()
    t0 = time.perf_counter()
    for _ in range(groups):
        sampling.sample_and_measure(G, start, end, z, T, kin)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = _lib.profile_stop()
    roof, rows = roofline_of_dominant(prof, groups, N)
    out = {'metric': 'generator-only sampling + telemetry metric recompute, trajectories/s (256 steps each)',
           'value': groups * N / dt, 'unit': 'trajectories/s', 'n_gpus': 1, 'steps': groups, 'warmup': warm,
           'ms_per_step': dt / groups * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32 rollout / f64 metrics', 'data': 'synthetic',
           'config': {'workload': f'BASELINE configs[4]: {groups} launch groups x {N} trajectories x {T} steps of Generator.generate '
                                  f'(no early-stop shortcut, H_g={HID}) + mouse-telemetry kinematics in f64 on every point',
                      'trajectories_total': groups * N, 'seq_len': T},
           'seconds_per_1M_trajectories': dt / (groups * N) * (1 << 20),
           'roofline': roof, 'kernels': rows,
           'note': 'roofline.achieved of the rollout counts EXECUTED matrix FLOPs (417 152 per row-step at H=128: the '
                   'condition term of SURVEY.md\'s 548 224 is hoisted out of the loop)',
           'hbm_peak_allocated_gb': torch.cuda.max_memory_allocated() / 1e9}
    return out


def bench_config3(args, dev):
    """BASELINE.json configs[2]: B=4096, T=256, H_g = H_d = 512, bf16 operands / fp32 accumulation (state, parameter gradients,
    master weights and Adam in fp32).  The same step as config 2 - one D step + one G step, gradient penalty every 4th D step
    on 25 % of the batch, reference dropout on - on the bf16 kernel family (ops_bf16: hoisted GEMMs + one launch per timestep).
    F_iter = 4 F_G + 8.375 F_D = 55.19 GFLOP per trained trajectory; peak = dense bf16 2.5 PFLOP/s."""
    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd import _lib
    B, T, Hd = args.batch, 256, 512
    cfg = gmt.Config()
    cfg.discriminator_hidden_dim, cfg.generator_hidden_dim, cfg.latent_dim = Hd, Hd, LATENT
    cfg.compute_dtype = 'bf16'
    torch.manual_seed(0)
    G, D = gmt.Generator(cfg).to(dev), gmt.Discriminator(cfg).to(dev)
    tr = gmt.Trainer(G, D, cfg, dev, log_dir=os.path.join('/tmp', 'gmt_bench_runs'))
    tr.epoch = cfg.teacher_forcing_decay_epochs     # steady state: teacher-forcing ratio 0
    tr.overlap_g_forward = False
    G.train(); D.train()
    batch = synthetic_batch(B, T, 1234, dev)
    tf_mask = [False] * T

    def step():
        tr.train_discriminator_step(batch)
        tr.train_generator_step(batch, {'tf_mask': tf_mask})

    steps = args.steps if args.steps_given else 4
    warm = args.warmup if args.warmup_given else 4
    for _ in range(max(1, warm - 4)):
        step()
    torch.cuda.synchronize()
    _lib.profile_start()
    for _ in range(4):                                             # the per-kernel table: one gradient-penalty period
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_stop()
    roof, table = roofline_of_dominant(prof, 4, B, PEAK_BF16_MFMA_TFLOPS)
    _lib.profile_start(only=[roof['kernel']])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    roof_timed, _ = roofline_of_dominant(_lib.profile_stop(), steps, B, PEAK_BF16_MFMA_TFLOPS)
    roof_timed['all_kernels_ms_per_step'] = roof['all_kernels_ms_per_step']
    fl = flops_per_trajectory(T, Hd, Hd)[0]
    out = {'metric': 'GAN trajectories/sec trained (B=4096,T=256,H=512, bf16 compute / fp32 accum)', 'value': B / dt,
           'unit': 'trajectories/s', 'n_gpus': 1, 'steps': steps, 'warmup': warm, 'ms_per_step': dt * 1e3,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'bf16 operands / f32 accumulate, state, parameter gradients and optimizer', 'data': 'synthetic',
           'config': {'workload': f'BASELINE configs[2]: WGAN-GP train step (D step + G step), B={B}, T={T}, H_g=H_d={Hd}, latent {LATENT}, '
                                  'n_critic 1, GP every 4th D step on 25% of the batch, '
                                  'teacher-forcing ratio 0', 'global_batch': B, 'seq_len': T, 'parallelism': 'dp1',
                      'lstm_dropout': cfg.lstm_dropout, 'head_dropout': cfg.head_dropout},
           'model_flops_per_trajectory': fl, 'step_tflops': fl * B / dt / 1e12,
           'frac_of_bf16_mfma_peak': fl * B / dt / 1e12 / PEAK_BF16_MFMA_TFLOPS,
           'roofline': roof_timed, 'kernels': table, 'hbm_peak_allocated_gb': torch.cuda.max_memory_allocated() / 1e9}
    return out


def extra_configs(args, dev):
    """Config 3 (B=4096, T=256, H=512, bf16) and config 5 (1 M x 256 sampling + f64 recompute) with their default
    step counts (the caller has released the config-2 trainer).  A failure is reported in place, never raised."""
    import copy
    import gc
    from generative_mouse_trajectories_amd import ops
    gc.collect()
    ops.weights_changed()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    res = {}
    for name, fn in (('config5', bench_config5), ('config3', bench_config3)):
        a = copy.copy(args)
        a.steps_given = a.warmup_given = False
        t0 = time.perf_counter()
        try:
            torch.cuda.reset_peak_memory_stats()
            r = fn(a, dev)
            r['wall_s_incl_setup'] = time.perf_counter() - t0
            res[name] = r
        except Exception as e:
            res[name] = {'value': None, 'error': repr(e)}
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return res


def main():
    if os.environ.get('GMT_DEBUG_HANG'):              # diagnosis aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['GMT_DEBUG_HANG']), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--batch', type=int, default=B_LOCAL)
    ap.add_argument('--config', type=int, default=2, choices=(2, 3, 5),
                    help='BASELINE.json configs[] entry (1-based): 2 = train step, 5 = sampling + recompute')
    ap.add_argument('--samples', type=int, default=131072, help='config 5: trajectories per launch group')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true',
                    help='config 2 at one GPU only: skip the config 3 and config 5 measurements that follow the headline timed region')
    ap.add_argument('--no-overlap', action='store_true', help='run the two generator rollouts of an iteration one after the other')
    ap.add_argument('--cpu-threads', type=int, default=16)
    ap.add_argument('--ops-set', action='append', default=[], metavar='NAME=VALUE',
                    help='A/B: set a module-level switch of generative_mouse_trajectories_amd.ops (e.g. GEN_BWD_SUMS_G0=0)')
    ap.add_argument('--no-mlp-kernels', action='store_true',
                    help='A/B: condition encoder and critic head as torch ops instead of csrc/mlp.hip')
    ap.add_argument('--no-real-overlap', action='store_true',
                    help='do not launch the next D step\'s real-sample critic forward beside the G step\'s BPTT kernel')
    ap.add_argument('--sync-metrics', action='store_true',
                    help='block on the metrics read-back inside every step function (the reference\'s .item() behaviour)')
    args = ap.parse_args()
    args.steps_given, args.warmup_given = args.steps is not None, args.warmup is not None
    if args.steps is None:
        args.steps = 8
    if args.warmup is None:
        args.warmup = 8

    for kv in args.ops_set:                                    # NAME=V (ops) or ops_bf16.NAME=V
        import importlib
        k, v = kv.split('=')
        mod, _, k = k.rpartition('.')
        _m = importlib.import_module('generative_mouse_trajectories_amd.' + (mod or 'ops'))
        assert hasattr(_m, k), k
        setattr(_m, k, type(getattr(_m, k))(int(v)))
    if args.no_mlp_kernels:
        from generative_mouse_trajectories_amd import ops as _ops
        _ops.MLP_KERNELS = False
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > max(1, torch.cuda.device_count()) and not args.no_real_overlap:
        # Rehearsal of N ranks on fewer GPUs (several processes on one card): the G step's real-forward overlap is left
        # out.  Two processes that each run a half-machine BPTT kernel beside a full-machine persistent forward on a
        # second stream stalled on a shared MI355X at the bench size (round 3: both ranks' queues stop, Python stacks in
        # the host-staged all-reduce after the G step; `HIP_LAUNCH_BLOCKING=1` or this switch complete; one rank per GPU
        # - also with an RCCL group - completes: DESIGN.md section 6).  One rank per GPU, the real launch, keeps it.
        args.no_real_overlap = True
    local_rank %= max(1, torch.cuda.device_count())       # (rehearsing N ranks on a one-GPU box: they share the card)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if args.config == 5:
        return print(json.dumps(bench_config5(args, dev)), flush=True)
    if args.config == 3:
        return print(json.dumps(bench_config3(args, dev)), flush=True)
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

    host = {'d_enqueue': 0.0, 'd_wait': 0.0, 'g_enqueue': 0.0, 'g_wait': 0.0}
    if args.sync_metrics:
        tr.max_steps_in_flight = 0                  # every step function blocks on its own metrics like the reference's .item()
    metrics = []

    def one_step():
        # host-side split of a step: time to ENQUEUE it (Python + launches) vs time blocked inside the step function
        # after that (the metrics read-back with --sync-metrics; else only the bound on steps in flight)
        tr.host_clock = clk = []
        t0 = time.perf_counter()
        metrics.append(tr.train_discriminator_step(batch))
        t1 = time.perf_counter()
        metrics.append(tr.train_generator_step(batch, {'tf_mask': tf_mask}, next_batch=None if args.no_real_overlap else batch))
        t2 = time.perf_counter()
        host['d_enqueue'] += clk[0][1] - t0; host['d_wait'] += t1 - clk[0][1]
        host['g_enqueue'] += clk[1][1] - t1; host['g_wait'] += t2 - clk[1][1]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # Kernel timing with HIP events perturbs what it measures (an event pair around each of the ~300 labelled calls of
    # an iteration costs 8 % of the step), so: the last 4 warm-up iterations (one gradient-penalty period) are timed
    # call by call - that gives the per-kernel table and names the dominant kernel - and the TIMED region carries
    # events around the dominant kernel's launches only (roofline.achieved is measured there, as the contract asks).
    n_tab = min(4, args.warmup)
    for _ in range(args.warmup - n_tab):
        one_step()
    rows, dominant = {}, 'lstm_fwd_fused_I2'
    if n_tab:
        barrier()
        _lib.profile_start()
        for _ in range(n_tab):
            one_step()
        barrier()
        r0, rows = roofline_of_dominant(_lib.profile_stop(), n_tab, args.batch)
        if r0:
            dominant = r0['kernel']
    barrier()
    for k in host:
        host[k] = 0.0
    _lib.profile_start(only=[dominant])
    t0 = time.perf_counter()
    del metrics[:]
    for _ in range(args.steps):
        one_step()
    barrier()
    # every step's metrics are READ inside the timed region (what a training loop does with them at the latest)
    last = [dict(m) for m in metrics][-2:]
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
        roof, _ = roofline_of_dominant(prof, args.steps, args.batch)
        if roof:
            roof['all_kernels_ms_per_step'] = {k: v['ms_per_step'] for k, v in rows.items()}
            roof['note'] = ('achieved / avg_launch_ms: HIP events around this kernel only, inside the timed region; '
                            'all_kernels_ms_per_step and "kernels": events around every labelled call during the last '
                            f'{n_tab} warm-up iterations (timing every call slows the step by ~8 %)')
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
            'kernels': rows,
            'host_ms_per_step': {k: v / args.steps * 1e3 for k, v in host.items()},
            'metrics_readback': 'blocking per step (--sync-metrics)' if args.sync_metrics else
                                f'asynchronous (StepMetrics; at most {tr.max_steps_in_flight} step calls in flight), all read inside the timed region',
            'last_step_metrics': {k: v for m in last for k, v in m.items()},
            'hbm_peak_allocated_gb': torch.cuda.max_memory_allocated() / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(min(args.cpu_threads, os.cpu_count() or 1))
            except Exception as e:                                  # the baseline must never break the bench line
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}
        if world == 1 and not force_dist and not args.no_extra and args.batch == B_LOCAL:
            # BASELINE.json configs[2] and configs[4], measured AFTER the headline timed region so that the one line the
            # driver records carries them too (each is the same code path as `--config 3` / `--config 5`)
            G = D = batch = None                                    # (rebinds the cells one_step closes over)
            del metrics[:]
            tr.__dict__.clear()
            out['extra_configs'] = extra_configs(args, dev)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
