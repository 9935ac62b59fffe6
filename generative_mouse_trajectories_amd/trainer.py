"""Trainer with the surface of the reference's Trainer (RecurrentGAN.py:845-1303) on top of
the HIP kernels: same step functions, same returned metric keys, same checkpoint keys.

Differences that are deliberate (documented in DESIGN.md):
  * parameters, gradients and Adam moments of each model live in ONE flat buffer each, so
    the optimizer step is a single fused launch (and the data-parallel all-reduce one call);
  * the step scalars come back in one device-to-host copy instead of 4-5 `.item()` syncs;
  * the discriminator weight gradients that the reference computes (and throws away) inside
    the generator step are not computed;
  * noise (z, teacher-forcing coins, GP indices, alpha) is drawn on the GPU generator unless
    passed explicitly (`noise=`), which is how the parity tests replay the reference.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim.lr_scheduler import ReduceLROnPlateau

from . import _lib as L_
from . import ops
from . import ops_bf16
from .config import Config
from .models import Discriminator, Generator, _pad_batch, trajectory_to_absolute


# ------------------------------------------------------------------------------------------
# flat parameter storage + fused Adam
# ------------------------------------------------------------------------------------------
class FlatParams:
    """Re-homes every parameter of a module into one contiguous fp32 buffer (16-byte aligned
    segments) with a matching flat gradient buffer; p.data / p.grad become views."""

    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        dev = params[0].device
        self.params = params
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views_g: List[torch.Tensor] = []
        for p, o in zip(params, offs):
            seg = self.flat[o:o + p.numel()].view_as(p)
            seg.copy_(p.data)
            p.data = seg
            g = self.grad[o:o + p.numel()].view_as(p)
            p.grad = g
            self.views_g.append(g)
        self.offsets = offs
        self.total = total
        ops.weights_changed()

    def zero_grad(self):
        """Gradients start as None: autograd then hands each parameter its gradient tensor as is
        instead of launching one `grad += g` kernel per parameter into a zeroed buffer."""
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """After backward: the per-parameter gradients -> the flat buffer (one multi-tensor copy),
        p.grad re-pointed at its view so that callers, the all-reduce and the fused Adam all see the
        same memory.  Alignment gaps between segments stay zero from construction."""
        views, grads = [], []
        for p, v in zip(self.params, self.views_g):
            if p.grad is None:
                v.zero_()                    # unused in this step's graph
            elif p.grad.data_ptr() != v.data_ptr():
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if views:
            torch._foreach_copy_(views, grads)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (RG.py:864-873) as one gmt_adam_step launch over the flat
    buffer.  Exposes param_groups / state like torch.optim.Adam so schedulers and checkpoints
    work unchanged."""

    def __init__(self, flat: FlatParams, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
        super().__init__(flat.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.flat = flat
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0
        self.grad_scale = grad_scale
        for p, o in zip(flat.params, flat.offsets):
            self.state[p] = {'step': torch.tensor(0.0),
                             'exp_avg': self.exp_avg[o:o + p.numel()].view_as(p),
                             'exp_avg_sq': self.exp_avg_sq[o:o + p.numel()].view_as(p)}

    def zero_grad(self, set_to_none: bool = True):
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step_(self.flat.flat, self.flat.grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'], g['eps'],
                       self.step_count, self.grad_scale)
        ops.weights_changed(self.flat.params)    # packed / permuted copies of THESE parameters are stale now
        for st in self.state.values():
            st['step'] = torch.tensor(float(self.step_count))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # re-home loaded moments into the flat buffers
        # (a parameter that never received a gradient has no entry in a torch.optim.Adam checkpoint: zero moments.
        # One step count serves the whole flat buffer: the largest one found - in the reference every parameter
        # of a model gets a gradient at every step, so they all agree.)
        steps = [int(st['step']) for st in self.state.values() if 'step' in st]
        self.step_count = max(steps) if steps else 0
        for p, o in zip(self.flat.params, self.flat.offsets):
            st = self.state[p]
            for key, buf in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
                view = buf[o:o + p.numel()].view_as(p)
                if key in st:
                    view.copy_(st[key])
                else:
                    view.zero_()
                st[key] = view
            st['step'] = torch.tensor(float(self.step_count))


def _host_backend(group=None) -> bool:
    """True when the process group's collectives run on the host (gloo: the CPU tests and the rehearsal of N ranks
    on one GPU).  ProcessGroupGloo does accept device tensors, staging them through pinned memory on high-priority
    pool streams of its own - and with a deep queue of persistent kernels ahead of it that staging never gets its
    turn (measured on MI355X, two ranks sharing the card at B = 4096: the GPU sits at 100 % busy and the event
    recorded before the FIRST collective is never reached, on either rank, while the same step with the collective
    on an explicit host copy runs at full speed; tools/two_rank_diag.py).  So device tensors
    are never handed to gloo: all_reduce_sum / broadcast_ stage them through the host on the caller's own stream."""
    import torch.distributed as dist
    return dist.get_backend(group) == 'gloo'


def all_reduce_sum(t: torch.Tensor, group=None, async_op: bool = False):
    """Sum all-reduce of `t` in place: RCCL on the device buffer (asynchronous if asked), or - host backend - on a
    host copy (synchronous; returns None)."""
    import torch.distributed as dist
    if t.is_cuda and _host_backend(group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast_(t: torch.Tensor, src: int, group=None) -> None:
    import torch.distributed as dist
    if t.is_cuda and _host_backend(group):
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def allreduce_flat_grad(flat: FlatParams, world_size: int, group=None) -> None:
    """Data-parallel gradient exchange: ONE sum all-reduce (RCCL over xGMI on the GPUs, gloo in
    the CPU tests) of the model's flat gradient buffer; the 1/world_size factor is applied by
    the fused Adam kernel (grad_scale)."""
    if world_size > 1:
        all_reduce_sum(flat.grad, group)


class OverlappedGradSync:
    """Bucketed gradient exchange overlapped with the backward pass: the parameters
    are cut into consecutive buckets that become final together (a BPTT sweep produces all weight gradients
    of its layer at once), and as soon as autograd has delivered the last gradient of a bucket its slice of
    the flat gradient buffer is filled (one multi-tensor copy) and an asynchronous sum all-reduce is started
    on it - RCCL's own stream, so it runs under the sweeps that are still to come.  finish() waits for the
    handles before the fused Adam step.  bucket_key(name) labels a parameter; consecutive parameters with
    the same label form a bucket (the flat buffer follows module.parameters() order, so a bucket is one
    contiguous slice)."""

    def __init__(self, flat: FlatParams, names: List[str], bucket_key, world_size: int, group=None):
        self.flat, self.world_size, self.group = flat, world_size, group
        self.bucket_of: List[int] = []
        self.buckets: List[Dict] = []
        last = object()
        for i, (n, p, o) in enumerate(zip(names, flat.params, flat.offsets)):
            k = bucket_key(n)
            if k != last:
                self.buckets.append({'idx': [], 'lo': o, 'hi': o})
                last = k
            b = self.buckets[-1]
            b['idx'].append(i)
            b['hi'] = o + (p.numel() + 3) // 4 * 4
            self.bucket_of.append(len(self.buckets) - 1)
        self.buckets[-1]['hi'] = flat.total
        self.enabled = False
        self.host_backend = flat.grad.is_cuda and _host_backend(group)
        self.pending: List[int] = []
        self.next = -1
        self.launched: List[bool] = []
        self.works: List = []
        self._hooks = [p.register_post_accumulate_grad_hook(lambda _p, i=i: self._ready(i))
                       for i, p in enumerate(flat.params)]

    def begin(self) -> None:
        """Call right before backward (after zero_grad)."""
        self.pending = [len(b['idx']) for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.next = len(self.buckets) - 1
        self.works = []
        self.enabled = True

    def _ready(self, i: int) -> None:
        if not self.enabled:
            return
        self.pending[self.bucket_of[i]] -= 1
        # Collectives must be issued in the same order on every rank: buckets start strictly last-to-first (the
        # order the backward pass finishes them); one that completes early waits for its turn.
        while self.next >= 0 and self.pending[self.next] == 0:
            self._launch(self.next, async_op=True)
            self.next -= 1

    def _launch(self, b: int, async_op: bool) -> None:
        import torch.distributed as dist
        bk = self.buckets[b]
        views, grads = [], []
        for i in bk['idx']:
            p, v = self.flat.params[i], self.flat.views_g[i]
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                views.append(v)
                grads.append(p.grad)
            p.grad = v
        if views:
            torch._foreach_copy_(views, grads)
        self.launched[b] = True
        if self.host_backend:
            return                       # exchanged by finish(): one host-staged all-reduce of the whole buffer
        w = dist.all_reduce(self.flat.grad[bk['lo']:bk['hi']], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            self.works.append(w)

    def finish(self) -> None:
        """Call right after backward: buckets autograd never completed (a parameter unused in this step) are
        exchanged now, then every outstanding all-reduce is waited for (the current stream waits; no host sync
        with RCCL)."""
        self.enabled = False
        while self.next >= 0:                            # same last-to-first order as from the hooks
            self._launch(self.next, async_op=True)
            self.next -= 1
        for w in self.works:
            w.wait()
        self.works = []
        if self.host_backend:
            all_reduce_sum(self.flat.grad, self.group)


def discriminator_bucket_key(name: str) -> str:
    """feature_encoder | lstm layer 0 | lstm layer 1 | output_layers: the order the backward finishes them, reversed."""
    top = name.split('.')[0]
    return 'lstm' + name.split('_l')[-1][0] if top == 'lstm' else top


def generator_bucket_key(name: str) -> str:
    """(initial_input, condition encoder) | (lstm, output head): the rollout's BPTT launch finishes the second
    group in one go, the condition encoder follows."""
    return 'cond' if name.split('.')[0] in ('initial_input', 'condition_encoder') else 'rollout'


def dp_tf_coins(tf_seed: int, g_step: int, T: int, tf_ratio: float) -> List[bool]:
    """Teacher-forcing coins of generator step number g_step under data parallelism: a pure function of
    (tf_seed, g_step), so every rank decides alike; a shorter T is a prefix of a longer one."""
    g = torch.Generator().manual_seed((int(tf_seed) * 1000003 + int(g_step)) & 0x7FFFFFFF)
    return [bool(v) for v in (torch.rand(T, generator=g) < tf_ratio).tolist()]


_SIDE_STREAMS: Dict = {}


def side_stream(device: torch.device) -> 'torch.cuda.Stream':
    """The ONE high-priority side stream of a device, shared by every Trainer of the process: a process has few hardware
    queues (GPU_MAX_HW_QUEUES = 4) and every stream ever created keeps a share of one, so no Trainer makes its own."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        # high priority: once RCCL has created its own streams a default-priority side stream lands on the main stream's
        # hardware queue and the two rollouts serialise (measured: +4 ms/step); priority streams come from a separate pool
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=-1)
    return st


class StepMetrics(dict):
    """The dict of floats a step function returns (RG.py:1024-1029, 1112-1118) - same keys, same values - whose
    numbers are still on their way from the GPU when the function returns: the step's scalars are copied into
    pinned host memory by the stream (one asynchronous copy, one event) and the dict fills itself in at the first
    READ of any value.  The reference ends every step in 4-5 blocking `.item()` calls; here the host goes on to
    enqueue the next step while the GPU finishes this one, so the command queue never runs dry at a step boundary
    (profiles/r3_01_gap_table.txt: the blocking read-back left the GPU idle for the head of every step).
    Reading is what synchronises: `m['d_loss']`, `m.items()`, `dict(m)`, `json.dumps(m)`, `print(m)` all wait for
    the step's event first.  Keys, len() and `in` are known at once."""

    def __init__(self, keys, host_buf: torch.Tensor, event, extra: Optional[Dict[str, float]] = None):
        super().__init__({k: None for k in keys})
        if extra:
            super().update(extra)
        self._pending = (tuple(keys), host_buf, event)

    @property
    def ready(self) -> bool:
        return self._pending is None or self._pending[2].query()

    def wait(self) -> 'StepMetrics':
        if self._pending is not None:
            keys, buf, ev = self._pending
            ev.synchronize()
            self._pending = None
            for k, v in zip(keys, buf.tolist()):
                super().__setitem__(k, v)
        return self

    # every way of reading a value goes through wait(); __iter__ / keys are overridden as well so that dict(m),
    # {**m} and json.dumps take the generic mapping path instead of copying the raw table
    def __getitem__(self, k): return dict.__getitem__(self.wait(), k)
    def get(self, k, default=None): return dict.get(self.wait(), k, default)
    def items(self): return dict.items(self.wait())
    def values(self): return dict.values(self.wait())
    def keys(self): return dict.keys(self.wait())
    def __iter__(self): return dict.__iter__(self.wait())
    def __repr__(self): return dict.__repr__(self.wait())
    def __eq__(self, other): return dict.__eq__(self.wait(), other.wait() if isinstance(other, StepMetrics) else other)
    def __ne__(self, other): return dict.__ne__(self.wait(), other.wait() if isinstance(other, StepMetrics) else other)
    def copy(self): return dict(self.wait())
    def pop(self, *a): return dict.pop(self.wait(), *a)
    def setdefault(self, *a): return dict.setdefault(self.wait(), *a)
    def __reduce__(self): return (dict, (dict(self.wait()),))
    __hash__ = None


class _NullWriter:
    def add_scalar(self, *a, **k): pass
    def add_figure(self, *a, **k): pass
    def close(self): pass


def _make_writer(log_dir: str):
    try:
        from datetime import datetime
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(os.path.join(log_dir, 'trajectory_gan_' + datetime.now().strftime('%Y%m%d_%H%M%S')))
    except Exception:
        return _NullWriter()


def load_checkpoint_file(path: str, map_location=None) -> Dict:
    """torch.load(weights_only=True) of a checkpoint written by this Trainer OR by the reference.
    The reference stores its Config *instance* (RG.py:1219), pickled by qualified name - `__main__.Config`
    when RecurrentGAN.py ran as a script - and loads with weights_only=False (RG.py:1231).  Nothing of the
    file is executed here: the restricted unpickler is only allowed to build an inert stand-in object for
    that one name, whose attributes come back as a plain dict under 'config'."""
    stand_ins = []
    for mod in ('__main__', 'RecurrentGAN', 'learning.RecurrentGAN', 'python.learning.RecurrentGAN'):
        cls = type('Config', (), {})
        cls.__module__, cls.__qualname__ = mod, 'Config'
        stand_ins.append(cls)
    with torch.serialization.safe_globals(stand_ins):
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    cfg = ckpt.get('config')
    if cfg is not None and not isinstance(cfg, dict):
        ckpt['config'] = dict(vars(cfg))         # instance attributes only: class-level defaults are not pickled
    return ckpt


# ------------------------------------------------------------------------------------------
# gradient penalty (RecurrentGAN.py:790-842)
# ------------------------------------------------------------------------------------------
def compute_gradient_penalty(discriminator: Discriminator, real_trajectories, fake_trajectories, real_dts, fake_dts,
                             lengths, device, alpha: Optional[torch.Tensor] = None) -> torch.Tensor:
    from .composite import discriminator_forward_gp
    batch_size = real_trajectories.shape[0]
    min_len = min(real_trajectories.shape[1], fake_trajectories.shape[1])
    real_trajectories, fake_trajectories = real_trajectories[:, :min_len], fake_trajectories[:, :min_len]
    real_dts, fake_dts = real_dts[:, :min_len], fake_dts[:, :min_len]
    lengths = lengths.clamp(max=min_len)
    if alpha is None:
        alpha = torch.rand(batch_size, 1, 1, device=real_trajectories.device)
    alpha = alpha.reshape(batch_size, 1, 1)
    alpha_dt = alpha.squeeze(-1)
    interpolated = (alpha * real_trajectories + (1 - alpha) * fake_trajectories).detach().requires_grad_(True)
    interpolated_dts = alpha_dt * real_dts + (1 - alpha_dt) * fake_dts
    d_interpolated = discriminator_forward_gp(discriminator, interpolated, interpolated_dts, lengths)
    gradients = torch.autograd.grad(outputs=d_interpolated, inputs=interpolated,
                                    grad_outputs=torch.ones_like(d_interpolated),
                                    create_graph=True, retain_graph=True)[0]
    gradient_norm = gradients.reshape(batch_size, -1).norm(2, dim=1)
    return ((gradient_norm - 1) ** 2).mean()


# ------------------------------------------------------------------------------------------
# Trainer
# ------------------------------------------------------------------------------------------
class Trainer:
    def __init__(self, generator: Generator, discriminator: Discriminator, config: Config, device: torch.device,
                 log_dir: str = 'runs', world_size: int = 1, process_group=None):
        device = torch.device(device)
        if device.type != 'cuda':
            raise L_.GmtError('Trainer runs on an MI355X (device "cuda"); there is no CPU path')
        L_.lib()
        self.generator = generator.to(device)
        self.discriminator = discriminator.to(device)
        self.config = config
        self.device = device
        self.use_amp = False          # fp32 parity configuration
        self._bf16 = getattr(config, 'compute_dtype', 'fp32') == 'bf16'
        self._ragged_key = self._ragged_val = self._ragged_ref = None
        self.scaler = None
        self.world_size = world_size
        self.process_group = process_group
        self.g_flat = FlatParams(self.generator)
        self.d_flat = FlatParams(self.discriminator)
        gs = 1.0 / world_size
        self.g_optimizer = FusedAdam(self.g_flat, lr=config.learning_rate, betas=config.betas, grad_scale=gs)
        self.d_optimizer = FusedAdam(self.d_flat, lr=config.learning_rate, betas=config.betas, grad_scale=gs)
        self.g_scheduler = ReduceLROnPlateau(self.g_optimizer, mode='min', factor=config.lr_factor,
                                             patience=config.lr_patience, min_lr=config.min_lr)
        self.d_scheduler = ReduceLROnPlateau(self.d_optimizer, mode='min', factor=config.lr_factor,
                                             patience=config.lr_patience, min_lr=config.min_lr)
        self._g_sync = self._d_sync = None
        # data parallel: the per-step teacher-forcing coins are host decisions that every rank must take alike
        # (the rollout kernels of all ranks follow the same path): they come from this
        # generator, re-seeded per generator step, never from the rank's own global RNG
        self.tf_seed = 0
        self._g_step = 0
        if world_size > 1:
            self._enable_overlapped_sync()
        self.writer = _make_writer(log_dir)
        self.best_loss = float('inf')
        self.patience_counter = 0
        self.epoch = 0
        self.d_step = 0
        # launch the next generator step's forward rollout beside the D step's (see _launch_speculative_g_forward);
        # off by default so that the step functions stay independent calls with the reference's RNG draw order
        self.overlap_g_forward = False
        self._side = None
        self._spec = None
        self._g_version = 0
        self._d_version = 0
        self._spec_real = None         # see _launch_speculative_real_forward
        self._main_stream = None
        # two streams feed one set of parameters on purpose (the speculative launches above)
        _quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if _quiet is not None:
            _quiet(False)
        self.host_clock = None         # a list: the step functions append (tag, perf_counter()) when a step is fully enqueued
        # step functions return StepMetrics whose floats arrive asynchronously; the host may run at most this many
        # step-function calls ahead of the GPU (0: every step function blocks on its own read-back like the reference)
        self.max_steps_in_flight = 4
        self._in_flight: List[StepMetrics] = []
        self.sync_replicas()

    # ---- helpers ---------------------------------------------------------------------
    def get_teacher_forcing_ratio(self) -> float:
        c = self.config
        if self.epoch >= c.teacher_forcing_decay_epochs:
            return c.teacher_forcing_end
        progress = self.epoch / c.teacher_forcing_decay_epochs
        return c.teacher_forcing_start - progress * (c.teacher_forcing_start - c.teacher_forcing_end)

    def _enable_overlapped_sync(self) -> None:
        pick = lambda m, flat: [n for n, p in m.named_parameters() if p.requires_grad]
        self._g_sync = OverlappedGradSync(self.g_flat, pick(self.generator, self.g_flat), generator_bucket_key,
                                          self.world_size, self.process_group)
        self._d_sync = OverlappedGradSync(self.d_flat, pick(self.discriminator, self.d_flat), discriminator_bucket_key,
                                          self.world_size, self.process_group)

    def sync_replicas(self) -> None:
        """Data parallel: every rank takes rank 0's parameters, Adam moments and spectral-norm buffers, so
        replicas are identical whatever RNG state each process built its models from (construction, and again
        after load_checkpoint)."""
        if self.world_size <= 1:
            return
        import torch.distributed as dist
        src = dist.get_global_rank(self.process_group, 0) if self.process_group is not None else 0
        steps = torch.tensor([self.g_optimizer.step_count, self.d_optimizer.step_count, self.d_step, self._g_step],
                             device=self.device, dtype=torch.int64)
        ts = [self.g_flat.flat, self.d_flat.flat, self.g_optimizer.exp_avg, self.g_optimizer.exp_avg_sq,
              self.d_optimizer.exp_avg, self.d_optimizer.exp_avg_sq, steps]
        ts += [b for m in (self.generator, self.discriminator) for b in m.buffers()]
        for t_ in ts:
            broadcast_(t_, src, self.process_group)
        ops.weights_changed()
        self.g_optimizer.step_count, self.d_optimizer.step_count, self.d_step, self._g_step = (int(v) for v in steps.tolist())

    def _tf_coins(self, T: int, tf_ratio: float) -> List[bool]:
        """The per-step teacher-forcing coins of RG.py:588.  One process: torch.rand(1) per step from the global
        CPU generator, exactly like the reference.  Data parallel: from a generator seeded by (tf_seed, index of
        this generator step) - the same sequence on every rank (ranks whose batches differ in T share the prefix)."""
        if self.world_size <= 1:
            return [torch.rand(1).item() < tf_ratio for _ in range(T)]
        return dp_tf_coins(self.tf_seed, self._g_step, T, tf_ratio)

    def _backward_and_sync(self, loss: torch.Tensor, flat: FlatParams, sync: Optional[OverlappedGradSync]) -> None:
        """backward + gradients into the flat buffer + (data parallel) their sum over the ranks: bucketed
        all-reduces that start while the backward is still running, or nothing at world size 1."""
        if sync is not None and self.world_size > 1:
            sync.begin()
            loss.backward()
            sync.finish()
        else:
            loss.backward()
            flat.gather_grads()

    def _metrics(self, keys, scalars: List[torch.Tensor], extra: Optional[Dict[str, float]] = None) -> StepMetrics:
        """The step's device scalars -> one pinned-memory copy on the stream + an event; see StepMetrics."""
        # (entries may be vectors: the loss kernels hand several metrics over as one tensor)
        dev_vals = scalars[0].detach().float().reshape(-1) if len(scalars) == 1 else \
            torch.cat([s_.detach().float().reshape(-1) for s_ in scalars])
        host = torch.empty(dev_vals.numel(), dtype=torch.float32, pin_memory=True)
        host.copy_(dev_vals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        m = StepMetrics(keys, host, ev, extra)
        self._in_flight.append(m)
        while len(self._in_flight) > max(0, int(self.max_steps_in_flight)):
            self._in_flight.pop(0).wait()
        return m

    def _zero_scalar(self) -> torch.Tensor:
        """A constant 0-d zero on the device (the d_gp metric of the steps without a penalty): made once, never written."""
        z = getattr(self, '_zero0', None)
        if z is None or z.device != self.device:
            z = self._zero0 = torch.zeros((), device=self.device)
        return z

    def _to_dev(self, batch: Dict):
        d = self.device
        return (batch['starts'].to(d, non_blocking=True), batch['ends'].to(d, non_blocking=True),
                batch['sequences'].to(d, non_blocking=True), batch['lengths'].to(d, non_blocking=True))

    # ---- discriminator step (RecurrentGAN.py:913-1029) ------------------------------
    def _launch_speculative_g_forward(self, real_batch, starts, ends, sequences, lengths, tf_mask=None) -> None:
        """The generator step that follows this D step runs G with the weights G has NOW (only D is
        updated in between), on its own noise.  At B = 4096 a rollout is 128 workgroups - half the CUs -
        so that forward is launched here on a second stream, beside the D step's no-grad rollout, and
        train_generator_step picks it up if nothing it depends on changed (same batch object, same G
        version, same teacher-forcing decisions, no injected z)."""
        if self._side is None:
            self._side = side_stream(self.device)
        main = torch.cuda.current_stream()
        self.generator.prepack()             # the cached weight packs are built on the main stream, both streams read them
        self._side.wait_stream(main)
        T = sequences.shape[1]
        tf_ratio = self.get_teacher_forcing_ratio()
        if tf_mask is None:
            tf_mask = self._tf_coins(T, tf_ratio)
        tf_mask = [bool(b) for b in tf_mask]
        with torch.cuda.stream(self._side):
            z = torch.randn(sequences.shape[0], self.config.latent_dim, device=self.device)
            out = self.generator.forward_tm(starts, ends, z, sequences, tf_mask)
        for t_ in (starts, ends, sequences, lengths):
            t_.record_stream(self._side)
        self._spec = {'batch': real_batch, 'g_version': self._g_version, 'tf_ratio': tf_ratio, 'tf_mask': tf_mask, 'out': out,
                      'z': z}

    def _launch_speculative_real_forward(self, next_batch: Dict, before_bptt: 'torch.cuda.Event') -> None:
        """Called from inside the generator step's backward pass right after the BPTT kernel was launched
        (ops.AFTER_GEN_BWD_LAUNCH).  That kernel is B / 32 workgroups - half the CUs at B = 4096 - and nothing else of
        the G step can run beside it.  The NEXT discriminator step's real-sample forward (RG.py:957) can: it depends
        on D's weights and spectral-norm buffers as the G step's own critic call left them (D is not updated by a G
        step) and on the next batch, nothing else.  It is enqueued here on the stream the BPTT kernel is NOT on,
        ordered after `before_bptt`; train_discriminator_step(next_batch) picks it up.  Same arithmetic, same order of
        power iterations on u / v as the reference (G step's critic call, then the D step's real call)."""
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = side_stream(self.device)
        other = self._main_stream if cur == self._side else self._side
        other.wait_event(before_bptt)
        D = self.discriminator
        with torch.enable_grad(), torch.cuda.stream(other):
            starts, ends, sequences, lengths = self._to_dev(next_batch)
            batch_size = sequences.shape[0]
            Bp = ops.pad_rows(batch_size, ops.row_tile(self.config))
            real_pos, real_dts = ops.traj_from_seq(sequences.float(), starts, False, batch_size, Bp)
            lens_p1 = self._critic_inputs(lengths, Bp, next_batch['lengths'], sequences.shape[1])
            real_score = D.forward_tm(real_pos, real_dts, lens_p1)               # padded [Bp, 1]
        self._spec_real = {'batch': next_batch, 'd_version': self._d_version, 'training': D.training, 'stream': other,
                           'real_pos': real_pos, 'real_dts': real_dts, 'lens_p1': lens_p1, 'real_score': real_score}

    def _critic_inputs(self, lengths: torch.Tensor, Bp: int, host_lengths: Optional[torch.Tensor] = None,
                       T: Optional[int] = None) -> torch.Tensor:
        """lengths + 1 (positions incl. the start point, RG.py:957 / 960 / 1064) as the kernels' int32 [Bp]
        (clamped to >= 1 like RG.py:766; padding rows have length 1).  The bf16 critic picks its kernel variant by whether
        any row is shorter than the batch's length: answered here from the batch's own lengths - on the host when the
        batch came from a DataLoader, once per lengths tensor otherwise - and left on the result (`gmt_ragged`)."""
        out = _pad_batch((lengths + 1).to(dtype=torch.int32).clamp(min=1), Bp, 0, 1).contiguous()
        if self._bf16 and host_lengths is not None and T is not None:
            key = (id(host_lengths), host_lengths._version, T, Bp)
            if self._ragged_key != key:
                short = bool((host_lengths.min() < T).item()) or Bp != host_lengths.numel()    # (padding rows have length 1)
                self._ragged_key, self._ragged_val = key, short
                self._ragged_ref = host_lengths              # keeps the id from being reused while cached
            out.gmt_ragged = self._ragged_val
        return out

    def train_discriminator_step(self, real_batch: Dict, noise: Optional[Dict] = None) -> Dict[str, float]:
        noise = noise or {}
        cfg = self.config
        self.d_optimizer.zero_grad()
        if self._bf16:
            ops_bf16.apply_row_group_policy()
        starts, ends, sequences, lengths = self._to_dev(real_batch)
        batch_size, T = sequences.shape[0], sequences.shape[1]
        Bp = ops.pad_rows(batch_size, ops.row_tile(self.config))
        G, D = self.generator, self.discriminator
        spec_r, self._spec_real = self._spec_real, None
        if not (spec_r is not None and spec_r['batch'] is real_batch and spec_r['d_version'] == self._d_version
                and spec_r['training'] == D.training and spec_r['real_pos'].shape[1] == Bp):
            spec_r = None
        if spec_r is not None:
            # this step's real-sample critic forward was launched beside the previous G step's BPTT kernel
            main = torch.cuda.current_stream()
            if spec_r['stream'] != main:
                main.wait_stream(spec_r['stream'])
                for t_ in (spec_r['real_pos'], spec_r['real_dts'], spec_r['lens_p1'], spec_r['real_score']):
                    t_.record_stream(main)
            real_pos, real_dts = spec_r['real_pos'], spec_r['real_dts']
        else:
            # real_trajectories / real_dts_padded of RG.py:931-934, straight into the kernels' time-major padded layout
            real_pos, real_dts = ops.traj_from_seq(sequences.float(), starts, False, batch_size, Bp)
        z = noise.get('z')
        z = torch.randn(batch_size, cfg.latent_dim, device=self.device) if z is None else z.to(self.device)
        if self.overlap_g_forward and self.device.type == 'cuda' and (self.d_step + 1) % max(1, cfg.n_critic) == 0:
            self._launch_speculative_g_forward(real_batch, starts, ends, sequences, lengths, noise.get('g_tf_mask'))
        with torch.no_grad():
            fake_tm = G.forward_tm(starts, ends, z, sequences, [False] * T)              # (T,Bp,3), RG.py:938-944
            fake_pos, fake_dts = ops.traj_from_seq(fake_tm, starts, True, batch_size, Bp)     # RG.py:946-949
        lens_p1 = self._critic_inputs(lengths, Bp, real_batch['lengths'], T)
        real_score_p = spec_r['real_score'] if spec_r is not None else D.forward_tm(real_pos, real_dts, lens_p1)
        fake_score_p = D.forward_tm(fake_pos, fake_dts, lens_p1)                         # fake_lengths == lengths (RG.py:599-600)
        if ops.STEP_LOSS_KERNELS:
            # mean(fake) - mean(real) over the valid rows and both score means: one launch (one more for the backward)
            d_loss, score_means = ops.critic_loss(real_score_p, fake_score_p, batch_size)
        else:
            real_score, fake_score = real_score_p[:batch_size], fake_score_p[:batch_size]
            d_loss = fake_score.mean() - real_score.mean()
            score_means = torch.stack([real_score.detach().mean(), fake_score.detach().mean()])
        gp = self._zero_scalar()
        do_gp = (self.d_step % max(1, cfg.gp_every_n) == 0)
        self.d_step += 1
        if do_gp:
            frac = min(max(cfg.gp_batch_frac, 0.0), 1.0)
            gp_bs = max(1, int(math.ceil(batch_size * frac)))
            if gp_bs < batch_size:
                idx = noise.get('gp_idx')
                idx = torch.randperm(batch_size, device=self.device) if idx is None else idx.to(self.device)
                idx = idx[:gp_bs]
            else:
                idx = torch.arange(batch_size, device=self.device)
            # the penalty's sub-batch, batch-first as compute_gradient_penalty takes it (RG.py:978-1003)
            sel = lambda t_tm: t_tm.index_select(1, idx).transpose(0, 1)
            alpha = noise.get('alpha')
            gp = compute_gradient_penalty(D, sel(real_pos), sel(fake_pos), sel(real_dts), sel(fake_dts),
                                          (lengths + 1).index_select(0, idx), self.device,
                                          None if alpha is None else alpha.to(self.device))
        total_loss = d_loss + cfg.gradient_penalty_weight * gp if do_gp else d_loss
        self._backward_and_sync(total_loss, self.d_flat, self._d_sync)
        self.d_optimizer.step()
        self._d_version += 1
        if self.host_clock is not None:
            self.host_clock.append(('d_enqueued', __import__('time').perf_counter()))
        return self._metrics(('d_loss', 'd_gp', 'd_real_score', 'd_fake_score'), [d_loss, gp, score_means])

    # ---- generator step (RecurrentGAN.py:1031-1118) ---------------------------------
    def train_generator_step(self, real_batch: Dict, noise: Optional[Dict] = None,
                             next_batch: Optional[Dict] = None) -> Dict[str, float]:
        """next_batch (optional, not in the reference): the batch the NEXT train_discriminator_step will be given.
        With it the real-sample critic forward of that step is launched beside this step's half-machine BPTT kernel
        (_launch_speculative_real_forward).  Only pass it when that call is certain to follow: the forward advances
        the spectral-norm power iteration like any critic call."""
        noise = noise or {}
        cfg = self.config
        self.g_optimizer.zero_grad()
        starts, ends, sequences, lengths = self._to_dev(real_batch)
        batch_size, T = sequences.shape[0], sequences.shape[1]
        Bp = ops.pad_rows(batch_size, ops.row_tile(self.config))
        G, D = self.generator, self.discriminator
        tf_ratio = self.get_teacher_forcing_ratio()
        tf_mask = noise.get('tf_mask')
        spec, self._spec = self._spec, None
        if (spec is not None and spec['batch'] is real_batch and spec['g_version'] == self._g_version
                and 'z' not in noise and spec['tf_ratio'] == tf_ratio
                and (tf_mask is None or [bool(b) for b in tf_mask] == spec['tf_mask'])):
            # the rollout of this step was launched beside the D step's no-grad rollout (same G weights)
            torch.cuda.current_stream().wait_stream(self._side)
            fake_tm = spec['out']
            fake_tm.record_stream(torch.cuda.current_stream())
        else:
            z = noise.get('z')
            z = torch.randn(batch_size, cfg.latent_dim, device=self.device) if z is None else z.to(self.device)
            if tf_mask is None:
                tf_mask = self._tf_coins(T, tf_ratio)
            fake_tm = G.forward_tm(starts, ends, z, sequences, [bool(b) for b in tf_mask])     # RG.py:1051-1056
        self._g_step += 1
        fake_pos, fake_dts = ops.traj_from_seq(fake_tm, starts, True, batch_size, Bp)           # RG.py:1058-1061
        # the critic is a fixed function here: its weight gradients (which the reference computes
        # and discards at the next d_optimizer.zero_grad()) are skipped
        d_params = self.d_flat.params
        for p in d_params:
            p.requires_grad_(False)
        try:
            fake_score_p = D.forward_tm(fake_pos, fake_dts, self._critic_inputs(lengths, Bp, real_batch['lengths'], T))   # RG.py:1064
        finally:
            for p in d_params:
                p.requires_grad_(True)
        if ops.STEP_LOSS_KERNELS:
            # RG.py:1069-1097 - adversarial, endpoint and direction terms, their weighted sum and (backward) the adjoints of
            # the rollout and of the score vector - in the loss kernels; a data-parallel run sums the direction term's
            # numerator and denominator over the ranks in between (RG.py:1095 divides by the mask sum of the WHOLE batch)
            reduce = None
            if self.world_size > 1:
                import torch.distributed as dist
                reduce = lambda nd: (all_reduce_sum(nd, self.process_group), dist.get_world_size(self.process_group))[1]
            total_loss, parts = ops.gen_step_loss(fake_tm, fake_score_p, starts, ends, lengths.to(torch.int32),
                                                  cfg.endpoint_loss_weight, cfg.direction_loss_weight, reduce)
            return self._finish_generator_step(total_loss, [parts], next_batch, tf_ratio)
        fake_score = fake_score_p[:batch_size]
        g_loss = -fake_score.mean()                                                              # RG.py:1069
        # endpoint and direction losses (RG.py:1072-1095): every reduction in one kernel, gradient in one more
        scal = ops.gen_aux_losses(fake_tm, starts, ends, lengths.to(torch.int32))
        endpoint_loss = scal[0] / (2.0 * batch_size)                                             # F.mse_loss over (B, 2)
        dir_num, dir_den = scal[1], scal[2]
        direction_loss = dir_num / dir_den
        dir_for_grad = direction_loss
        if self.world_size > 1:
            # RG.py:1095 divides by the mask sum of the WHOLE batch: with ragged lengths the shards' mask sums differ,
            # so numerator and denominator are summed over the ranks (one 2-element all-reduce) and the local term is
            # scaled so that the 1/world of the gradient average gives d(sum num / sum den) exactly
            import torch.distributed as dist
            nd = torch.stack([dir_num.detach(), dir_den.detach()])
            all_reduce_sum(nd, self.process_group)
            direction_loss = nd[0] / nd[1]
            dir_for_grad = dir_num * (float(dist.get_world_size(self.process_group)) / nd[1])
        total_loss = g_loss + cfg.endpoint_loss_weight * endpoint_loss + cfg.direction_loss_weight * dir_for_grad
        return self._finish_generator_step(total_loss, [g_loss, endpoint_loss, direction_loss, fake_score.detach().mean()],
                                           next_batch, tf_ratio)

    def _finish_generator_step(self, total_loss, metric_tensors, next_batch, tf_ratio) -> Dict[str, float]:
        if next_batch is not None and self.discriminator.training:
            self._main_stream = torch.cuda.current_stream()
            ops.AFTER_GEN_BWD_LAUNCH[:] = [lambda ev, nb=next_batch: self._launch_speculative_real_forward(nb, ev)]
        try:
            self._backward_and_sync(total_loss, self.g_flat, self._g_sync)
        finally:
            del ops.AFTER_GEN_BWD_LAUNCH[:]
        self.g_optimizer.step()
        self._g_version += 1
        if self.host_clock is not None:
            self.host_clock.append(('g_enqueued', __import__('time').perf_counter()))
        return self._metrics(('g_loss', 'g_endpoint_loss', 'g_direction_loss', 'g_fake_score'), metric_tensors,
                             {'teacher_forcing_ratio': tf_ratio})

    # ---- epoch / run loop (RecurrentGAN.py:1120-1303) -------------------------------
    def train_epoch(self, dataloader: Iterable) -> Dict[str, float]:
        self.generator.train()
        self.discriminator.train()
        keys = ['d_loss', 'd_gp', 'd_real_score', 'd_fake_score', 'g_loss', 'g_endpoint_loss',
                'g_direction_loss', 'g_fake_score']
        epoch_metrics: Dict[str, List[float]] = {k: [] for k in keys}
        steps: List[Dict[str, float]] = []
        it = iter(dataloader)
        batch = next(it, None)
        while batch is not None:
            nxt = next(it, None)                       # one batch of look-ahead: the G step starts the next D step's real forward
            for _ in range(self.config.n_critic):
                steps.append(self.train_discriminator_step(batch))
            steps.append(self.train_generator_step(batch, next_batch=nxt if self.overlap_g_forward else None))
            batch = nxt
        for m in steps:                                # read after the loop: the step functions never wait on the GPU
            for k, v in m.items():
                if k in epoch_metrics:
                    epoch_metrics[k].append(v)
        avg = {k: float(np.mean(v)) for k, v in epoch_metrics.items() if v}
        avg['teacher_forcing_ratio'] = self.get_teacher_forcing_ratio()
        return avg

    def _rank(self) -> int:
        if self.world_size <= 1:
            return 0
        import torch.distributed as dist
        return dist.get_rank(self.process_group)

    def _reduce_metrics(self, metrics: Dict[str, float]) -> Dict[str, float]:
        """Data parallel: the epoch metrics drive ReduceLROnPlateau, the best-loss test and early stopping, so every
        rank must see the same numbers (else learning rates diverge and one rank leaves the loop while the others
        wait in an all-reduce): their mean over the ranks, one small all-reduce per epoch."""
        if self.world_size <= 1:
            return metrics
        import torch.distributed as dist
        keys = sorted(metrics)
        t_ = torch.tensor([metrics[k] for k in keys], dtype=torch.float64)
        if not _host_backend(self.process_group):
            t_ = t_.to(self.device)
        dist.all_reduce(t_, op=dist.ReduceOp.SUM, group=self.process_group)
        return {k: float(v) / self.world_size for k, v in zip(keys, t_.tolist())}

    def log_metrics(self, metrics: Dict[str, float], step: int) -> None:
        for name, value in metrics.items():
            self.writer.add_scalar(f'train/{name}', value, step)
        self.writer.add_scalar('train/g_lr', self.g_optimizer.param_groups[0]['lr'], step)
        self.writer.add_scalar('train/d_lr', self.d_optimizer.param_groups[0]['lr'], step)

    def log_trajectories(self, dataloader, step: int) -> None:
        """Observability only (RG.py:1167-1209); figures need matplotlib + tensorboard and are skipped without them."""
        return None

    def save_checkpoint(self, path: str, is_best: bool = False) -> None:
        checkpoint = {'epoch': self.epoch, 'generator_state_dict': self.generator.state_dict(),
                      'discriminator_state_dict': self.discriminator.state_dict(),
                      'g_optimizer_state_dict': self.g_optimizer.state_dict(),
                      'd_optimizer_state_dict': self.d_optimizer.state_dict(),
                      'config': {k: getattr(self.config, k) for k in dir(self.config) if not k.startswith('_')},
                      'best_loss': self.best_loss}
        torch.save(checkpoint, path)
        if is_best:
            torch.save(checkpoint, path.replace('.pt', '_best.pt'))

    def load_checkpoint(self, path: str) -> None:
        checkpoint = load_checkpoint_file(path, self.device)
        self.epoch = checkpoint['epoch']
        self.generator.load_state_dict(checkpoint['generator_state_dict'])
        self._g_version += 1
        self._spec = None
        self.discriminator.load_state_dict(checkpoint['discriminator_state_dict'])
        self._d_version += 1
        self._spec_real = None
        self.g_optimizer.load_state_dict(checkpoint['g_optimizer_state_dict'])
        self.d_optimizer.load_state_dict(checkpoint['d_optimizer_state_dict'])
        self.best_loss = checkpoint.get('best_loss', float('inf'))
        # the data-parallel teacher-forcing coins are a function of (tf_seed, generator step index): the index is not
        # a checkpoint key (the file layout is the reference's), the restored Adam state carries it
        self._g_step = self.g_optimizer.step_count
        ops.weights_changed()
        self.sync_replicas()
        print(f'Loaded checkpoint from epoch {self.epoch}')

    def train(self, dataloader, num_epochs: int, checkpoint_dir: str = 'checkpoints') -> None:
        os.makedirs(checkpoint_dir, exist_ok=True)
        print(f'Starting training for {num_epochs} epochs')
        print(f'Device: {self.device}')
        print(f'Batch size: {self.config.batch_size}')
        for epoch in range(self.epoch, num_epochs):
            self.epoch = epoch
            metrics = self._reduce_metrics(self.train_epoch(dataloader))
            self.log_metrics(metrics, epoch)
            if epoch % 10 == 0:
                self.log_trajectories(dataloader, epoch)
            combined_loss = metrics['d_loss'] + metrics['g_loss']
            self.g_scheduler.step(metrics['g_loss'])
            self.d_scheduler.step(metrics['d_loss'])
            if combined_loss < self.best_loss:
                self.best_loss = combined_loss
                self.patience_counter = 0
                is_best = True
            else:
                self.patience_counter += 1
                is_best = False
            if epoch % 10 == 0 or is_best:
                if self._rank() == 0:
                    self.save_checkpoint(os.path.join(checkpoint_dir, f'checkpoint_epoch_{epoch}.pt'), is_best)
                if self.world_size > 1:
                    import torch.distributed as dist
                    dist.barrier(group=self.process_group)       # nobody runs ahead of the file being complete
            print(f"Epoch {epoch}/{num_epochs} | D Loss: {metrics['d_loss']:.4f} | G Loss: {metrics['g_loss']:.4f} | "
                  f"Endpoint: {metrics['g_endpoint_loss']:.4f} | TF: {metrics['teacher_forcing_ratio']:.2f}")
            if self.patience_counter >= self.config.patience:
                print(f'Early stopping at epoch {epoch}')
                break
        self.writer.close()
        print('Training complete!')
