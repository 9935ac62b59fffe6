"""CPU oracle for the WGAN-GP trajectory train step (TEST INFRASTRUCTURE ONLY).

This file is a from-scratch CPU restatement, in explicit functional PyTorch-CPU
math, of the reference hot path `python/learning/RecurrentGAN.py` ("RG.py"),
lines 315-1118.  It exists so that the HIP path can be checked; it is NOT part
of the product.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real RG.py in
the build container and stores its outputs (losses, scores, gradients, updated
parameters, Adam moments, spectral-norm vectors) under `tests/golden/*.npz`;
`tests/test_oracle_vs_reference.py` checks every function below against them.

Design notes (how this differs from RG.py on purpose):
  * parameters live in plain dicts keyed by the reference's state-dict names,
    there are no nn.Modules, no nn.LSTM, no packed sequences: the LSTMs are
    explicit per-step cells with per-row length masks, which is also exactly
    the semantics the HIP kernels implement;
  * every random draw of the reference (z, teacher-forcing coins, randperm
    indices, alpha) is an explicit input, in the order RG.py consumes them
    (SURVEY.md section 8c);
  * `dtype` may be float64 to measure the true error of the fp32 GPU path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------
# hyper-parameters (RG.py:44-91); a plain attribute bag with the same names
# --------------------------------------------------------------------------
class OracleConfig:
    latent_dim = 64
    generator_hidden_dim = 256
    generator_num_layers = 2
    discriminator_hidden_dim = 128
    discriminator_num_layers = 2
    learning_rate = 1e-4
    betas = (0.5, 0.9)
    gradient_penalty_weight = 10.0
    gp_every_n = 4
    gp_batch_frac = 0.25
    endpoint_loss_weight = 50.0
    direction_loss_weight = 10.0
    teacher_forcing_start = 1.0
    teacher_forcing_end = 0.0
    teacher_forcing_decay_epochs = 100
    max_generation_steps = 200
    distance_threshold = 0.02
    # dropout probabilities hard-coded in RG.py:456, 726, 733
    lstm_dropout = 0.1
    head_dropout = 0.3

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


# --------------------------------------------------------------------------
# a1  trajectory_to_absolute                                   RG.py:392-417
# --------------------------------------------------------------------------
def to_absolute(start: Tensor, deltas: Tensor) -> Tensor:
    run = torch.cumsum(deltas, dim=1) + start[:, None, :]
    return torch.cat([start[:, None, :], run], dim=1)


# --------------------------------------------------------------------------
# a2  compute_kinematics ("mode A")                            RG.py:315-389
# --------------------------------------------------------------------------
def kinematics(pos: Tensor, dts: Tensor, eps: float = 1e-8) -> Tensor:
    """pos (B,L,2), dts (B,L) -> (B,L,9) = x,y,dx,dy,dt,vel,acc,jerk,curv."""
    x, y = pos[..., 0], pos[..., 1]
    zcol = torch.zeros_like(x[:, :1])
    dx = torch.cat([zcol, x[:, 1:] - x[:, :-1]], dim=1)           # RG.py:335-338
    dy = torch.cat([zcol, y[:, 1:] - y[:, :-1]], dim=1)
    tden = dts + eps
    vel = torch.sqrt(dx * dx + dy * dy + eps) / tden              # RG.py:341-342
    vx, vy = dx / tden, dy / tden                                 # RG.py:345-346
    dt_mid = (dts[:, 1:] + dts[:, :-1]) / 2 + eps                 # RG.py:351
    acc = torch.cat([zcol, (vel[:, 1:] - vel[:, :-1]) / dt_mid], dim=1)      # 349-352
    if pos.shape[1] >= 3:
        jk = (acc[:, 2:] - acc[:, 1:-1]) / (dt_mid[:, 1:] + eps)  # RG.py:355-357 (eps twice)
        jerk = torch.cat([zcol, zcol, jk], dim=1)
    else:
        jerk = torch.zeros_like(acc)
    ax = torch.cat([zcol, (vx[:, 1:] - vx[:, :-1]) / dt_mid], dim=1)         # 361-364
    ay = torch.cat([zcol, (vy[:, 1:] - vy[:, :-1]) / dt_mid], dim=1)
    curv = (vx * ay - vy * ax).abs() / (vx * vx + vy * vy + eps) ** 1.5      # 366-368
    vel = vel.clamp(0, 100)                                       # RG.py:371-374
    acc = acc.clamp(-1000, 1000)
    jerk = jerk.clamp(-10000, 10000)
    curv = curv.clamp(0, 100)
    return torch.stack([x, y, dx, dy, dts, vel, acc, jerk, curv], dim=-1)


# --------------------------------------------------------------------------
# LSTM cell, gate order i,f,g,o                (torch.nn.LSTM semantics)
# --------------------------------------------------------------------------
def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    a = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    H = h.shape[-1]
    i = torch.sigmoid(a[:, 0 * H:1 * H])
    f = torch.sigmoid(a[:, 1 * H:2 * H])
    g = torch.tanh(a[:, 2 * H:3 * H])
    o = torch.sigmoid(a[:, 3 * H:4 * H])
    c2 = f * c + i * g
    h2 = o * torch.tanh(c2)
    return h2, c2


def _dropout(x: Tensor, p: float, gen: Optional[torch.Generator]) -> Tensor:
    if p <= 0.0:
        return x
    keep = (torch.rand(x.shape, generator=gen, dtype=x.dtype) >= p).to(x.dtype)
    return x * keep / (1.0 - p)


# --------------------------------------------------------------------------
# a6  legacy torch.nn.utils.spectral_norm, one power iteration per forward
#     (torch/nn/utils/spectral_norm.py; used at RG.py:713,715,731,734,736)
# --------------------------------------------------------------------------
def sn_weight(w_orig: Tensor, u: Tensor, v: Tensor, training: bool,
              eps: float = 1e-12) -> Tensor:
    """Returns W_orig / sigma.  In training mode u, v are advanced IN PLACE."""
    if training:
        with torch.no_grad():
            v_new = F.normalize(torch.mv(w_orig.t(), u), dim=0, eps=eps)
            u_new = F.normalize(torch.mv(w_orig, v_new), dim=0, eps=eps)
            v.copy_(v_new)
            u.copy_(u_new)
    uc, vc = u.detach().clone(), v.detach().clone()
    sigma = torch.dot(uc, torch.mv(w_orig, vc))
    return w_orig / sigma


# --------------------------------------------------------------------------
# a7-a9  Generator                                             RG.py:424-604
# --------------------------------------------------------------------------
def g_condition(P: Params, start, end, z):
    diff = end - start
    dist = torch.sqrt((diff ** 2).sum(-1, keepdim=True) + 1e-8)   # RG.py:478
    ang = torch.atan2(diff[:, 1:2], diff[:, 0:1])                 # RG.py:479
    x = torch.cat([start, end, dist, ang, z], dim=-1)
    Hg = P['condition_encoder.0.weight'].shape[0]
    x = F.linear(x, P['condition_encoder.0.weight'], P['condition_encoder.0.bias'])
    x = F.layer_norm(x, (Hg,), P['condition_encoder.1.weight'], P['condition_encoder.1.bias'])
    x = F.leaky_relu(x, 0.2)
    x = F.linear(x, P['condition_encoder.3.weight'], P['condition_encoder.3.bias'])
    x = F.layer_norm(x, (Hg,), P['condition_encoder.4.weight'], P['condition_encoder.4.bias'])
    return F.leaky_relu(x, 0.2)


def g_dynamic(pos, end):
    rem = end - pos                                               # RG.py:502
    dist = torch.sqrt((rem ** 2).sum(-1, keepdim=True) + 1e-8)    # RG.py:503
    ang = torch.atan2(rem[:, 1:2], rem[:, 0:1])                   # RG.py:504
    return torch.cat([rem, dist, ang], dim=-1)


def _g_head(P: Params, h):
    y = F.leaky_relu(F.linear(h, P['output_layer.0.weight'], P['output_layer.0.bias']), 0.2)
    o = F.linear(y, P['output_layer.2.weight'], P['output_layer.2.bias'])
    return torch.cat([o[:, :2], F.softplus(o[:, 2:3]) + 0.001], dim=-1)      # RG.py:577-580


def generator_forward(P: Params, start, end, z, target: Tensor, tf_mask: Sequence[bool],
                      dropout_p: float = 0.0, gen: Optional[torch.Generator] = None,
                      num_layers: int = 2, dropout_mask: Optional[Tensor] = None) -> Tensor:
    """RG.py:508-604 with the per-step teacher-forcing coin (RG.py:588) given as
    tf_mask[t] == (coin_t < teacher_forcing_ratio).  Returns (B,T,3)."""
    B, T = start.shape[0], target.shape[1]
    cond = g_condition(P, start, end, z)
    Hg = cond.shape[1]
    h = [torch.zeros(B, Hg, dtype=start.dtype) for _ in range(num_layers)]
    c = [torch.zeros(B, Hg, dtype=start.dtype) for _ in range(num_layers)]
    cur = P['initial_input'].reshape(1, 3).expand(B, 3)
    pos = start.clone()
    tf_pos = start.clone()
    outs = []
    for t in range(T):
        x = torch.cat([cur, cond, g_dynamic(pos, end)], dim=-1)   # RG.py:561-568
        for l in range(num_layers):
            h[l], c[l] = lstm_cell(x, h[l], c[l], P[f'lstm.weight_ih_l{l}'], P[f'lstm.weight_hh_l{l}'],
                                   P[f'lstm.bias_ih_l{l}'], P[f'lstm.bias_hh_l{l}'])
            x = h[l]
            if l + 1 < num_layers:
                # dropout_mask (T,B,H), values 0 or 1/(1-p): an explicit inter-layer mask (tests)
                x = x * dropout_mask[t] if dropout_mask is not None else _dropout(x, dropout_p, gen)
        out = _g_head(P, x)
        outs.append(out)
        pos = pos + out[:, :2]                                    # RG.py:585
        tf_pos = tf_pos + target[:, t, :2]                        # running form of RG.py:591-593
        if tf_mask[t]:
            cur = target[:, t, :]
            pos = tf_pos
        else:
            cur = out
    return torch.stack(outs, dim=1)


def generator_generate(P: Params, start, end, z, cfg, num_layers: int = 2):
    """RG.py:606-690 (eval mode: no dropout).  Returns (outputs (B,t_end,3), lengths)."""
    B = start.shape[0]
    cond = g_condition(P, start, end, z)
    Hg = cond.shape[1]
    h = [torch.zeros(B, Hg, dtype=start.dtype) for _ in range(num_layers)]
    c = [torch.zeros(B, Hg, dtype=start.dtype) for _ in range(num_layers)]
    cur = P['initial_input'].reshape(1, 3).expand(B, 3)
    pos = start.clone()
    done = torch.zeros(B, dtype=torch.bool)
    lengths = torch.zeros(B, dtype=torch.long)
    outs = []
    t = 0
    for t in range(cfg.max_generation_steps):
        x = torch.cat([cur, cond, g_dynamic(pos, end)], dim=-1)
        for l in range(num_layers):
            h[l], c[l] = lstm_cell(x, h[l], c[l], P[f'lstm.weight_ih_l{l}'], P[f'lstm.weight_hh_l{l}'],
                                   P[f'lstm.bias_ih_l{l}'], P[f'lstm.bias_hh_l{l}'])
            x = h[l]
        out = _g_head(P, x)
        outs.append(out)
        cur = out
        pos = pos + out[:, :2]
        newly = torch.sqrt(((pos - end) ** 2).sum(-1)) < cfg.distance_threshold   # RG.py:671-672
        lengths = torch.where(newly & ~done, torch.tensor(t + 1), lengths)
        done = done | newly
        if bool(done.all()):
            break
    lengths = torch.where(lengths == 0, torch.tensor(t + 1), lengths)
    return torch.stack(outs, dim=1), lengths


# --------------------------------------------------------------------------
# a3-a5  Discriminator                                         RG.py:697-783
# --------------------------------------------------------------------------
SN_LAYERS = ('feature_encoder.0', 'feature_encoder.2', 'output_layers.0',
             'output_layers.3', 'output_layers.5')


def _masked_lstm_dir(x_seq: Tensor, lengths: Tensor, w_ih, w_hh, b_ih, b_hh, reverse: bool):
    """One direction of a packed-sequence LSTM layer.  x_seq (B,L,I).  A row is
    updated at step t only when t < len; outputs at padded steps are zero
    (pack_padded_sequence / pad_packed_sequence semantics, RG.py:764-772)."""
    B, L, _ = x_seq.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H, dtype=x_seq.dtype)
    c = torch.zeros(B, H, dtype=x_seq.dtype)
    outs: List[Optional[Tensor]] = [None] * L
    order = range(L - 1, -1, -1) if reverse else range(L)
    for t in order:
        act = (t < lengths).to(x_seq.dtype)[:, None]
        h2, c2 = lstm_cell(x_seq[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
        h = act * h2 + (1 - act) * h
        c = act * c2 + (1 - act) * c
        outs[t] = act * h2
    return torch.stack(outs, dim=1), h


def discriminator_forward(P: Params, bufs: Params, traj, dts, lengths, training: bool = True,
                          lstm_dropout: float = 0.0, head_dropout: float = 0.0,
                          gen: Optional[torch.Generator] = None, num_layers: int = 2,
                          return_intermediates: bool = False, lstm_dropout_masks=None):
    """RG.py:739-783.  `bufs` holds weight_u / weight_v and is advanced in place
    when training (one power iteration per call).  lstm_dropout_masks: optional explicit
    multiplicative masks [B,L,2H] (0 or 1/(1-p)), one per layer boundary, in place of drawn ones
    (the tests hand over the mask the HIP kernels derive from their counter-based hash)."""
    def snw(name):
        return sn_weight(P[name + '.weight_orig'], bufs[name + '.weight_u'],
                         bufs[name + '.weight_v'], training)

    feats = kinematics(traj, dts)
    e = F.leaky_relu(F.linear(feats, snw('feature_encoder.0'), P['feature_encoder.0.bias']), 0.2)
    e = F.leaky_relu(F.linear(e, snw('feature_encoder.2'), P['feature_encoder.2.bias']), 0.2)
    lens = lengths.clamp(min=1)                                   # RG.py:766
    x = e
    hf = hb = None
    for l in range(num_layers):
        of, hf = _masked_lstm_dir(x, lens, P[f'lstm.weight_ih_l{l}'], P[f'lstm.weight_hh_l{l}'],
                                  P[f'lstm.bias_ih_l{l}'], P[f'lstm.bias_hh_l{l}'], False)
        ob, hb = _masked_lstm_dir(x, lens, P[f'lstm.weight_ih_l{l}_reverse'], P[f'lstm.weight_hh_l{l}_reverse'],
                                  P[f'lstm.bias_ih_l{l}_reverse'], P[f'lstm.bias_hh_l{l}_reverse'], True)
        x = torch.cat([of, ob], dim=-1)
        if l + 1 < num_layers:
            x = x * lstm_dropout_masks[l].to(x.dtype) if lstm_dropout_masks is not None else _dropout(x, lstm_dropout, gen)
    hcomb = torch.cat([hf, hb], dim=-1)                           # RG.py:776-778
    y = F.leaky_relu(F.linear(hcomb, snw('output_layers.0'), P['output_layers.0.bias']), 0.2)
    y = _dropout(y, head_dropout, gen)
    y = F.leaky_relu(F.linear(y, snw('output_layers.3'), P['output_layers.3.bias']), 0.2)
    score = F.linear(y, snw('output_layers.5'), P['output_layers.5.bias'])
    if return_intermediates:
        return score, {'features': feats, 'encoded': e, 'h_combined': hcomb}
    return score


# --------------------------------------------------------------------------
# a10  gradient penalty                                        RG.py:790-842
# --------------------------------------------------------------------------
def gradient_penalty(P_d: Params, bufs_d: Params, real_traj, fake_traj, real_dts, fake_dts,
                     lengths, alpha: Tensor, return_grads: bool = False, **dkw):
    b = real_traj.shape[0]
    L = min(real_traj.shape[1], fake_traj.shape[1])
    real_traj, fake_traj = real_traj[:, :L], fake_traj[:, :L]
    real_dts, fake_dts = real_dts[:, :L], fake_dts[:, :L]
    lengths = lengths.clamp(max=L)
    a3 = alpha.reshape(b, 1, 1)
    a2 = alpha.reshape(b, 1)
    x = (a3 * real_traj + (1 - a3) * fake_traj).detach().requires_grad_(True)   # RG.py:816-818
    xd = a2 * real_dts + (1 - a2) * fake_dts
    s = discriminator_forward(P_d, bufs_d, x, xd, lengths, **dkw)
    (g,) = torch.autograd.grad(s, x, torch.ones_like(s), create_graph=True, retain_graph=True)
    gp = ((g.reshape(b, -1).norm(2, dim=1) - 1) ** 2).mean()      # RG.py:838-840
    if return_grads:
        return gp, g
    return gp


# --------------------------------------------------------------------------
# a13  Adam, torch.optim.Adam defaults with lr/betas of RG.py:864-873
# --------------------------------------------------------------------------
def adam_update(P: Params, grads: Dict[str, Optional[Tensor]], state: Dict, lr: float,
                betas=(0.5, 0.9), eps: float = 1e-8) -> None:
    b1, b2 = betas
    with torch.no_grad():
        for k, p in P.items():
            g = grads.get(k)
            if g is None:
                continue
            st = state.setdefault(k, {'step': 0, 'exp_avg': torch.zeros_like(p),
                                      'exp_avg_sq': torch.zeros_like(p)})
            st['step'] += 1
            st['exp_avg'].lerp_(g, 1 - b1)
            st['exp_avg_sq'].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** st['step']
            bc2 = 1 - b2 ** st['step']
            denom = (st['exp_avg_sq'].sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(st['exp_avg'], denom, value=-(lr / bc1))


def teacher_forcing_ratio(epoch: int, cfg) -> float:
    """RG.py:903-911."""
    if epoch >= cfg.teacher_forcing_decay_epochs:
        return cfg.teacher_forcing_end
    prog = epoch / cfg.teacher_forcing_decay_epochs
    return cfg.teacher_forcing_start - prog * (cfg.teacher_forcing_start - cfg.teacher_forcing_end)


# --------------------------------------------------------------------------
# a11  train_discriminator_step                                RG.py:913-1029
# --------------------------------------------------------------------------
def _req(P: Params) -> Params:
    return {k: v.detach().requires_grad_(True) for k, v in P.items()}


def discriminator_step(P_g: Params, P_d: Params, bufs_d: Params, opt_d: Dict, batch: Dict, noise: Dict,
                       cfg, d_step: int, dropout: bool = False, gen: Optional[torch.Generator] = None,
                       apply_update: bool = True):
    """noise: {'z': (B,latent), 'gp_idx': LongTensor or None, 'alpha': (b,)}.
    Returns (metrics, grads).  P_d / bufs_d / opt_d are updated in place."""
    starts, ends, seq, lengths = batch['starts'], batch['ends'], batch['sequences'], batch['lengths']
    B, T = seq.shape[0], seq.shape[1]
    lp = cfg.lstm_dropout if dropout else 0.0
    hp = cfg.head_dropout if dropout else 0.0
    real_traj = to_absolute(starts, seq[:, :, :2])
    real_dts = F.pad(seq[:, :, 2], (1, 0), value=0.001)           # RG.py:934
    with torch.no_grad():
        fake_seq = generator_forward(P_g, starts, ends, noise['z'], seq, [False] * T,
                                     dropout_p=lp, gen=gen)
    fake_traj = to_absolute(starts, fake_seq[:, :, :2])
    fake_dts = F.pad(fake_seq[:, :, 2], (1, 0), value=0.001)
    Pd = _req(P_d)
    dkw = dict(training=True, lstm_dropout=lp, head_dropout=hp, gen=gen)
    real_score = discriminator_forward(Pd, bufs_d, real_traj, real_dts, lengths + 1, **dkw)
    fake_score = discriminator_forward(Pd, bufs_d, fake_traj, fake_dts, lengths + 1, **dkw)
    d_loss = fake_score.mean() - real_score.mean()                # RG.py:965
    gp = torch.zeros((), dtype=seq.dtype)
    if d_step % max(1, cfg.gp_every_n) == 0:                      # RG.py:969-989
        frac = min(max(cfg.gp_batch_frac, 0.0), 1.0)
        b = max(1, int(math.ceil(B * frac)))
        if b < B:
            idx = noise['gp_idx'][:b]
            sel = lambda t: t.index_select(0, idx)
        else:
            sel = lambda t: t
        gp = gradient_penalty(Pd, bufs_d, sel(real_traj), sel(fake_traj), sel(real_dts), sel(fake_dts),
                              sel(lengths + 1), noise['alpha'], **dkw)
    total = d_loss + cfg.gradient_penalty_weight * gp
    names = list(Pd.keys())
    gl = torch.autograd.grad(total, [Pd[k] for k in names], allow_unused=True)
    grads = dict(zip(names, gl))
    if apply_update:
        adam_update(P_d, grads, opt_d, cfg.learning_rate, cfg.betas)
    metrics = {'d_loss': float(d_loss.detach()), 'd_gp': float(gp.detach()),
               'd_real_score': float(real_score.mean()), 'd_fake_score': float(fake_score.mean())}
    aux = {'fake_sequences': fake_seq, 'real_score': real_score.detach(), 'fake_score': fake_score.detach()}
    return metrics, grads, aux


# --------------------------------------------------------------------------
# a12  train_generator_step                                    RG.py:1031-1118
# --------------------------------------------------------------------------
def generator_losses(fake_seq, fake_score, starts, ends, lengths):
    """RG.py:1069-1095.  Returns (g_loss, endpoint_loss, direction_loss)."""
    B, T = fake_seq.shape[0], fake_seq.shape[1]
    fake_deltas = fake_seq[:, :, :2]
    fake_traj = to_absolute(starts, fake_deltas)
    g_loss = -fake_score.mean()
    final_pos = fake_traj[torch.arange(B), lengths]
    endpoint = F.mse_loss(final_pos, ends)
    tdir = ends - starts
    tdir = tdir / (torch.sqrt((tdir ** 2).sum(-1, keepdim=True)) + 1e-8)       # eps outside sqrt
    mdir = fake_deltas / torch.sqrt((fake_deltas ** 2).sum(-1, keepdim=True) + 1e-8)   # eps inside
    cos = (mdir * tdir[:, None, :]).sum(-1)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).to(fake_seq.dtype)
    direction = ((1 - cos) * mask).sum() / mask.sum()
    return g_loss, endpoint, direction


def generator_step(P_g: Params, P_d: Params, bufs_d: Params, opt_g: Dict, batch: Dict, noise: Dict,
                   cfg, tf_mask: Sequence[bool], tf_ratio: float = 0.0, dropout: bool = False,
                   gen: Optional[torch.Generator] = None, apply_update: bool = True):
    """noise: {'z': (B,latent)}; tf_mask[t] is the outcome of the coin at RG.py:588."""
    starts, ends, seq, lengths = batch['starts'], batch['ends'], batch['sequences'], batch['lengths']
    lp = cfg.lstm_dropout if dropout else 0.0
    hp = cfg.head_dropout if dropout else 0.0
    Pg = _req(P_g)
    Pd = _req(P_d)   # the reference also back-propagates into D here; those grads are discarded
    fake_seq = generator_forward(Pg, starts, ends, noise['z'], seq, tf_mask, dropout_p=lp, gen=gen)
    fake_traj = to_absolute(starts, fake_seq[:, :, :2])
    fake_dts = F.pad(fake_seq[:, :, 2], (1, 0), value=0.001)
    fake_score = discriminator_forward(Pd, bufs_d, fake_traj, fake_dts, lengths + 1, training=True,
                                       lstm_dropout=lp, head_dropout=hp, gen=gen)
    g_loss, endpoint, direction = generator_losses(fake_seq, fake_score, starts, ends, lengths)
    total = g_loss + cfg.endpoint_loss_weight * endpoint + cfg.direction_loss_weight * direction
    names = list(Pg.keys())
    gl = torch.autograd.grad(total, [Pg[k] for k in names] + [fake_seq], allow_unused=True)
    grads = dict(zip(names, gl[:-1]))
    if apply_update:
        adam_update(P_g, grads, opt_g, cfg.learning_rate, cfg.betas)
    metrics = {'g_loss': float(g_loss), 'g_endpoint_loss': float(endpoint),
               'g_direction_loss': float(direction), 'g_fake_score': float(fake_score.mean()),
               'teacher_forcing_ratio': tf_ratio}
    aux = {'fake_sequences': fake_seq.detach(), 'd_fake_sequences': gl[-1]}
    return metrics, grads, aux


# --------------------------------------------------------------------------
# parameter construction helpers (shapes of RG.py:424-467, 704-737)
# --------------------------------------------------------------------------
def generator_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    H, Z = cfg.generator_hidden_dim, cfg.latent_dim
    s = {'initial_input': (1, 1, 3),
         'condition_encoder.0.weight': (H, 6 + Z), 'condition_encoder.0.bias': (H,),
         'condition_encoder.1.weight': (H,), 'condition_encoder.1.bias': (H,),
         'condition_encoder.3.weight': (H, H), 'condition_encoder.3.bias': (H,),
         'condition_encoder.4.weight': (H,), 'condition_encoder.4.bias': (H,)}
    for l in range(cfg.generator_num_layers):
        I = 3 + H + 4 if l == 0 else H
        s[f'lstm.weight_ih_l{l}'] = (4 * H, I)
        s[f'lstm.weight_hh_l{l}'] = (4 * H, H)
        s[f'lstm.bias_ih_l{l}'] = (4 * H,)
        s[f'lstm.bias_hh_l{l}'] = (4 * H,)
    s['output_layer.0.weight'] = (H // 2, H)
    s['output_layer.0.bias'] = (H // 2,)
    s['output_layer.2.weight'] = (3, H // 2)
    s['output_layer.2.bias'] = (3,)
    return s


def discriminator_shapes(cfg):
    """Returns (param shapes, buffer shapes) in the reference's state-dict key order."""
    H = cfg.discriminator_hidden_dim
    p: Dict[str, Tuple[int, ...]] = {}
    b: Dict[str, Tuple[int, ...]] = {}

    def sn(name, o, i):
        p[name + '.bias'] = (o,)
        p[name + '.weight_orig'] = (o, i)
        b[name + '.weight_u'] = (o,)
        b[name + '.weight_v'] = (i,)

    sn('feature_encoder.0', H, 9)
    sn('feature_encoder.2', H, H)
    for l in range(cfg.discriminator_num_layers):
        I = H if l == 0 else 2 * H
        for sfx in ('', '_reverse'):
            p[f'lstm.weight_ih_l{l}{sfx}'] = (4 * H, I)
            p[f'lstm.weight_hh_l{l}{sfx}'] = (4 * H, H)
            p[f'lstm.bias_ih_l{l}{sfx}'] = (4 * H,)
            p[f'lstm.bias_hh_l{l}{sfx}'] = (4 * H,)
    sn('output_layers.0', H, 2 * H)
    sn('output_layers.3', H // 2, H)
    sn('output_layers.5', 1, H // 2)
    return p, b
