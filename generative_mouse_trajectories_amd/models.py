"""Generator / Discriminator with the class surface and state-dict keys of the reference
(python/learning/RecurrentGAN.py:424-783), computed by the HIP kernels of libgmt_hip.so.

Both modules run on the GPU only.  Internally everything is time-major ([L, B, ...]) and
the batch is padded to a multiple of 32 rows (one workgroup tile); the public methods
take and return the reference's batch-first tensors.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L_
from . import ops
from . import ops_bf16
from .config import Config


# ------------------------------------------------------------------------------------------
# functional API of the reference
# ------------------------------------------------------------------------------------------
def compute_kinematics(positions: torch.Tensor, dts: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """RecurrentGAN.py:315-389.  positions (B,L,2), dts (B,L) -> (B,L,9)."""
    L_.require_cuda(positions, dts)
    feat = ops.kinematics_tm(positions.transpose(0, 1), dts.transpose(0, 1), eps)
    return feat.transpose(0, 1)


def trajectory_to_absolute(start: torch.Tensor, deltas: torch.Tensor) -> torch.Tensor:
    """RecurrentGAN.py:392-417.  start (B,2), deltas (B,L,2) -> (B,L+1,2)."""
    run = torch.cumsum(deltas, dim=1) + start.unsqueeze(1)
    return torch.cat([start.unsqueeze(1), run], dim=1)


def _pad_batch(t: torch.Tensor, Bp: int, dim: int, value: float = 0.0) -> torch.Tensor:
    B = t.shape[dim]
    if B == Bp:
        return t
    pad = [0, 0] * (t.dim() - dim - 1) + [0, Bp - B]
    return F.pad(t, pad, value=value)


# ------------------------------------------------------------------------------------------
# spectral-norm linear: parameters / buffers named as torch.nn.utils.spectral_norm names them
# ------------------------------------------------------------------------------------------
class SNLinear(nn.Module):
    """nn.Linear wrapped by the legacy torch.nn.utils.spectral_norm (RG.py:713-736): state
    `bias`, `weight_orig`, buffers `weight_u`, `weight_v`; one power iteration per
    training-mode forward, W = W_orig / (u^T W_orig v)."""

    def __init__(self, in_features: int, out_features: int, eps: float = 1e-12):
        super().__init__()
        lin = nn.Linear(in_features, out_features)               # same init draws as the reference
        self.in_features, self.out_features, self.eps = in_features, out_features, eps
        self.bias = nn.Parameter(lin.bias.detach().clone())
        self.weight_orig = nn.Parameter(lin.weight.detach().clone())
        u = F.normalize(torch.empty(out_features).normal_(0, 1), dim=0, eps=eps)
        v = F.normalize(torch.empty(in_features).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer('weight_u', u)
        self.register_buffer('weight_v', v)

    def compute_weight(self) -> torch.Tensor:
        """One power iteration (training mode, in place on weight_u / weight_v) and W_orig / sigma:
        gmt_spectral_norm_fwd, one launch (Discriminator batches its five layers into one call)."""
        return ops.spectral_norm_weights([self.weight_orig], [self.weight_u], [self.weight_v], self.training, self.eps)[0]

    def forward(self, x: torch.Tensor, weight: Optional[torch.Tensor] = None) -> torch.Tensor:
        w = self.compute_weight() if weight is None else weight
        if x.is_cuda and x.dim() == 3:          # per-timestep use: weight gradient over L*B rows on the split-K kernel
            return ops.linear(x, w, self.bias)
        return F.linear(x, w, self.bias)


class _LSTMParams(nn.Module):
    """Holds LSTM parameters under torch.nn.LSTM's names (weight_ih_l0, ..., _reverse) and
    with its default initialisation; the math runs in the HIP kernels."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int, bidirectional: bool):
        super().__init__()
        ref = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                      batch_first=True, bidirectional=bidirectional)
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.bidirectional = num_layers, bidirectional
        for name, p in ref.named_parameters():
            self.register_parameter(name, nn.Parameter(p.detach().clone()))

    def layer_weights(self, layer: int) -> List[torch.Tensor]:
        out = []
        for sfx in (('', '_reverse') if self.bidirectional else ('',)):
            out += [getattr(self, f'weight_ih_l{layer}{sfx}'), getattr(self, f'weight_hh_l{layer}{sfx}'),
                    getattr(self, f'bias_ih_l{layer}{sfx}'), getattr(self, f'bias_hh_l{layer}{sfx}')]
        return out


# ------------------------------------------------------------------------------------------
# Generator (RecurrentGAN.py:424-690)
# ------------------------------------------------------------------------------------------
class Generator(nn.Module):
    def __init__(self, config: Config):
        super().__init__()
        self.config = config
        H = config.generator_hidden_dim
        if config.generator_num_layers != 2:
            raise NotImplementedError('the fused rollout kernel implements the reference default of 2 LSTM layers')
        self.bf16 = getattr(config, 'compute_dtype', 'fp32') == 'bf16' and H > 256
        if self.bf16:
            if H != 512:
                raise NotImplementedError('compute_dtype bf16: the generator rollout is built for generator_hidden_dim = 512')
        elif H not in (32, 64, 128, 256):
            raise NotImplementedError("generator_hidden_dim must be 32, 64, 128 or 256 - the widths the rollout kernels are "
                                      "instantiated for (512: compute_dtype = 'bf16')")
        self.condition_encoder = nn.Sequential(
            nn.Linear(6 + config.latent_dim, H), nn.LayerNorm(H), nn.LeakyReLU(0.2),
            nn.Linear(H, H), nn.LayerNorm(H), nn.LeakyReLU(0.2))
        self.lstm = _LSTMParams(3 + H + 4, H, config.generator_num_layers, False)
        self.output_layer = nn.Sequential(nn.Linear(H, H // 2), nn.LeakyReLU(0.2), nn.Linear(H // 2, 3))
        self.initial_input = nn.Parameter(torch.zeros(1, 1, 3))

    def _compute_condition(self, start, end, z):
        diff = end - start
        distance = torch.sqrt((diff ** 2).sum(dim=-1, keepdim=True) + 1e-8)
        angle = torch.atan2(diff[:, 1:2], diff[:, 0:1])
        return self.condition_encoder(torch.cat([start, end, distance, angle, z], dim=-1))

    def prepack(self) -> None:
        """Builds the cached fragment-major copies of the rollout's weights on the CURRENT stream (Trainer calls
        this before it launches a rollout on a second stream, so that both streams find them complete)."""
        lp = self.lstm
        if self.bf16:
            # H = 512: the bf16 / slab-ordered copies the no-grad rollout reads (the fp32 packs are never used there)
            ops_bf16.gen_weights(lp.weight_ih_l0, lp.weight_hh_l0, lp.bias_ih_l0, lp.bias_hh_l0, lp.weight_ih_l1,
                                 lp.weight_hh_l1, lp.bias_ih_l1, lp.bias_hh_l1, self.output_layer[0].weight)
            return
        ops.gen_forward_packs(lp.weight_hh_l0, lp.weight_ih_l1, lp.weight_hh_l1, self.output_layer[0].weight)

    def _rollout(self, start, end, z, target_tm: Optional[torch.Tensor], tf_mask: Optional[torch.Tensor], T: int,
                 dropout_seed: Optional[int] = None, stop_distance: float = 0.0, padded: bool = False):
        """start/end (B,2), z (B,latent), target_tm (T,B,3) -> out (T,B,3); pads B to 32 (padded=True: returns
        the padded (T,Bp,3) buffer itself, rows >= B are the rollout of zero inputs).
        stop_distance > 0 (sampling): also returns the per-row lengths the kernel found (int32 [B])."""
        L_.require_cuda(start, end, z)
        H = self.config.generator_hidden_dim
        B = start.shape[0]
        if self.bf16:
            return self._rollout_bf16(start, end, z, target_tm, tf_mask, T, stop_distance, padded, dropout_seed)
        # (with a bf16 critic beside this fp32 rollout - compute_dtype 'bf16', H <= 256 here - the batch is padded to the
        # critic's 256-row tile, so that the step functions' time-major buffers all have one row count)
        Bp = ops.pad_rows(B, ops.row_tile(self.config))
        lp = self.lstm
        w_ih0 = lp.weight_ih_l0
        if ops.cond_encoder_supported(self):
            # condition features, both Linear / LayerNorm / LeakyReLU stages and the hoisted gate term in one launch, already
            # padded to Bp rows and in the unit-major column order of the rollout's gate buffers (csrc/mlp.hip)
            _, g0 = ops.cond_encoder(self, start, end, z, Bp, unit_major=True, want_g0=True)
        else:                                            # the A/B switch ops.MLP_COND off: torch ops
            cond = self._compute_condition(start, end, z)
            g0 = F.linear(cond, w_ih0[:, 3:3 + H], lp.bias_ih_l0 + lp.bias_hh_l0)
            g0 = _pad_batch(g0, Bp, 0)
            g0 = g0.view(Bp, 4, H).transpose(1, 2).reshape(Bp, 4 * H)
        w_u = torch.cat([w_ih0[:, :3], w_ih0[:, 3 + H:]], dim=1)
        bias1 = lp.bias_ih_l1 + lp.bias_hh_l1
        s_p, e_p = _pad_batch(start, Bp, 0), _pad_batch(end, Bp, 0)
        if target_tm is not None:
            target_tm = _pad_batch(target_tm, Bp, 1).contiguous()
        # nn.LSTM(dropout=0.1) of RG.py:451-457: inter-layer dropout in training mode, fused into the kernel
        p_drop = float(self.config.lstm_dropout) if self.training else 0.0
        if dropout_seed is None:
            dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0
        lens = torch.empty(Bp, device=start.device, dtype=torch.int32) if stop_distance > 0 else None
        out = ops.GenRolloutFn.apply(g0, w_u, lp.weight_hh_l0, lp.weight_ih_l1, lp.weight_hh_l1, bias1,
                                     self.output_layer[0].weight, self.output_layer[0].bias,
                                     self.output_layer[2].weight, self.output_layer[2].bias,
                                     self.initial_input, s_p, e_p, target_tm, tf_mask, T,
                                     torch.is_grad_enabled(), p_drop, dropout_seed, float(stop_distance), B, lens)
        if stop_distance > 0:
            return out[:, :B], lens[:B]
        return out if padded else out[:, :B]

    def _rollout_bf16(self, start, end, z, target_tm, tf_mask, T: int, stop_distance: float, padded: bool, dropout_seed=None):
        """H = 512 (BASELINE configs[2]): the rollout on the bf16 kernels, three launches per timestep
        (ops_bf16.gen_rollout under no_grad, ops_bf16.GenRolloutBf16Fn with BPTT otherwise)."""
        if stop_distance > 0:
            raise NotImplementedError('compute_dtype bf16: the in-kernel early stop is not built; use generate(early_exit=False)')
        B = start.shape[0]
        H = self.config.generator_hidden_dim
        Bp = (B + ops_bf16.ROW_TILE - 1) // ops_bf16.ROW_TILE * ops_bf16.ROW_TILE
        lp = self.lstm
        training_rollout = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        g0 = None
        if ops.cond_encoder_supported(self):
            cond, g0 = ops.cond_encoder(self, start, end, z, Bp, unit_major=False, want_g0=training_rollout)
        else:
            cond = _pad_batch(self._compute_condition(start, end, z), Bp, 0)
        if target_tm is not None:
            target_tm = _pad_batch(target_tm, Bp, 1).contiguous()
        p_drop = float(self.config.lstm_dropout) if self.training else 0.0
        if dropout_seed is None:
            dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0
        drop = (p_drop, dropout_seed)
        if training_rollout:
            # training rollout: every step's state is kept, BPTT in ops_bf16.GenRolloutBf16Fn
            w_ih0 = lp.weight_ih_l0
            if g0 is None:
                g0 = F.linear(cond, w_ih0[:, 3:3 + H], lp.bias_ih_l0 + lp.bias_hh_l0)
            w_u = torch.cat([w_ih0[:, :3], w_ih0[:, 3 + H:]], dim=1)
            # (the host's copy of the coins when the caller kept one - forward_tm / forward do: no device read-back)
            tf_list = getattr(tf_mask, 'gmt_host_list', None) if tf_mask is not None else None
            if tf_mask is not None and tf_list is None:
                tf_list = [bool(v) for v in tf_mask.tolist()]
            out = ops_bf16.GenRolloutBf16Fn.apply(g0, w_u, lp.weight_hh_l0, lp.weight_ih_l1, lp.weight_hh_l1,
                                                  lp.bias_ih_l1 + lp.bias_hh_l1, self.output_layer[0].weight,
                                                  self.output_layer[0].bias, self.output_layer[2].weight,
                                                  self.output_layer[2].bias, self.initial_input.reshape(3),
                                                  _pad_batch(start, Bp, 0), _pad_batch(end, Bp, 0), target_tm, tf_list, T, drop)
            return out if padded else out[:, :B]
        out = ops_bf16.gen_rollout(cond, (lp.weight_ih_l0, lp.weight_hh_l0, lp.bias_ih_l0, lp.bias_hh_l0,
                                          lp.weight_ih_l1, lp.weight_hh_l1, lp.bias_ih_l1, lp.bias_hh_l1),
                                   (self.output_layer[0].weight, self.output_layer[0].bias,
                                    self.output_layer[2].weight, self.output_layer[2].bias),
                                   self.initial_input.reshape(3), _pad_batch(start, Bp, 0), _pad_batch(end, Bp, 0),
                                   target_tm, tf_mask, T, drop)
        return out if padded else out[:, :B]

    def forward_tm(self, start, end, z, target_sequences: torch.Tensor, tf_mask: Sequence[bool],
                   dropout_seed: Optional[int] = None) -> torch.Tensor:
        """The training rollout in the kernels' own layout: out (T,Bp,3), time-major, batch padded to 32 rows -
        what ops.traj_from_seq / ops.gen_aux_losses / Discriminator.forward_tm consume without a transpose."""
        T = target_sequences.shape[1]
        tm = None
        if any(tf_mask):
            tm = torch.tensor([1 if b else 0 for b in tf_mask], dtype=torch.uint8).to(start.device)
            tm.gmt_host_list = [bool(b) for b in tf_mask]
        return self._rollout(start, end, z, target_sequences.transpose(0, 1), tm, T, dropout_seed, padded=True)

    def forward(self, start, end, z, target_sequences: Optional[torch.Tensor] = None,
                target_lengths: Optional[torch.Tensor] = None, teacher_forcing_ratio: float = 0.0,
                tf_mask: Optional[Sequence[bool]] = None,
                dropout_seed: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """RecurrentGAN.py:508-604.  `tf_mask` (optional, length T) replaces the per-step coin
        `torch.rand(1) < teacher_forcing_ratio` (RG.py:588) with explicit outcomes."""
        B = start.shape[0]
        if target_sequences is not None:
            T = target_sequences.shape[1]
            if tf_mask is None:
                # one host coin per step for the whole batch, drawn every step like the reference
                tf_mask = [torch.rand(1).item() < teacher_forcing_ratio for _ in range(T)]
            tm = None
            if any(tf_mask):
                tm = torch.tensor([1 if b else 0 for b in tf_mask], dtype=torch.uint8).to(start.device)
                tm.gmt_host_list = [bool(b) for b in tf_mask]
            out = self._rollout(start, end, z, target_sequences.transpose(0, 1), tm, T, dropout_seed)
        else:
            T = self.config.max_generation_steps
            out = self._rollout(start, end, z, None, None, T)
        outputs = out.transpose(0, 1)
        if target_lengths is not None:
            lengths = target_lengths
        else:
            lengths = torch.full((B,), T, device=start.device)
        return outputs, lengths

    @torch.no_grad()
    def generate(self, start, end, z: Optional[torch.Tensor] = None, early_exit: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """RecurrentGAN.py:606-690: sampling with per-sample early stop at distance < threshold.
        Default: the rollout runs all max_generation_steps in one launch (rows are independent); the stop
        step of every row and the common truncation point are then read off the positions - the outputs
        equal the reference's everywhere, also beyond a row's length.
        early_exit=True: the kernel tracks the per-row stop step itself and a 32-row tile leaves the time
        loop as soon as all its rows are done (large sampling jobs); lengths are the same, outputs agree
        up to each row's length and are zero where a finished tile stopped generating."""
        B = start.shape[0]
        if z is None:
            z = torch.randn(B, self.config.latent_dim, device=start.device)
        was_training = self.training
        self.eval()
        T = self.config.max_generation_steps
        if early_exit:
            out, lens = self._rollout(start, end, z, None, None, T, stop_distance=float(self.config.distance_threshold))
            if was_training:
                self.train()
            lengths = lens.long()
            t_end = int(lengths.max().item())
            return out.transpose(0, 1)[:, :t_end].contiguous(), lengths
        out = self._rollout(start, end, z, None, None, T).transpose(0, 1)          # (B,T,3)
        if was_training:
            self.train()
        pos = start.unsqueeze(1) + torch.cumsum(out[:, :, :2], dim=1)
        dist = torch.sqrt(((pos - end.unsqueeze(1)) ** 2).sum(-1))
        hit = dist < self.config.distance_threshold                                   # (B,T)
        steps = torch.arange(1, T + 1, device=start.device).unsqueeze(0).expand(B, T)
        first = torch.where(hit, steps, torch.full_like(steps, T + 1)).min(dim=1).values
        t_end = int(min(T, int(first.max().item())))       # the reference breaks once every row is done
        lengths = torch.where(first <= t_end, first, torch.full_like(first, t_end))
        return out[:, :t_end].contiguous(), lengths


# ------------------------------------------------------------------------------------------
# Discriminator (RecurrentGAN.py:697-783)
# ------------------------------------------------------------------------------------------
class Discriminator(nn.Module):
    def __init__(self, config: Config):
        super().__init__()
        self.config = config
        H = config.discriminator_hidden_dim
        self.bf16 = getattr(config, 'compute_dtype', 'fp32') == 'bf16'
        if self.bf16:
            if H % 256 != 0 or H > 1024:
                raise NotImplementedError('compute_dtype bf16: discriminator_hidden_dim must be a multiple of 256, at most 1024')
        elif H not in (32, 64, 128, 256):
            raise NotImplementedError("discriminator_hidden_dim must be 32, 64, 128 or 256 - the widths the layer kernels are "
                                      "instantiated for (larger: compute_dtype = 'bf16')")
        self.feature_encoder = nn.Sequential(SNLinear(9, H), nn.LeakyReLU(0.2), SNLinear(H, H), nn.LeakyReLU(0.2))
        self.lstm = _LSTMParams(H, H, config.discriminator_num_layers, True)
        self.output_layers = nn.Sequential(SNLinear(2 * H, H), nn.LeakyReLU(0.2), nn.Dropout(config.head_dropout),
                                           SNLinear(H, H // 2), nn.LeakyReLU(0.2), SNLinear(H // 2, 1))
        self.fuse_input_gemm = True

    def spectral_weights(self):
        """The normalised weights of the five SN-Linears for ONE critic call (one power iteration each in
        training mode), in module order, from a single launch."""
        ls = [m for m in list(self.feature_encoder) + list(self.output_layers) if isinstance(m, SNLinear)]
        return list(ops.spectral_norm_weights([m.weight_orig for m in ls], [m.weight_u for m in ls],
                                              [m.weight_v for m in ls], self.training, ls[0].eps))

    @staticmethod
    def run_sequential(seq: nn.Sequential, x: torch.Tensor, weights: List[torch.Tensor]) -> torch.Tensor:
        """seq(x) with the SN-Linears taking their (precomputed) weights from the front of `weights`."""
        for m in seq:
            x = m(x, weights.pop(0)) if isinstance(m, SNLinear) else m(x)
        return x

    def forward_tm(self, pos_tm: torch.Tensor, dts_tm: torch.Tensor, lengths_i32: torch.Tensor,
                   dropout_seed: Optional[int] = None) -> torch.Tensor:
        """Time-major core: pos [L,Bp,2], dts [L,Bp], lengths int32 [Bp] (Bp % 32 == 0) -> [Bp,1].
        The inter-layer dropout of nn.LSTM (RG.py:725) is a counter-based mask inside the kernels of the
        two layers it sits between (ops.LSTMLayerFn in_drop / out_drop; mask of layer l = seed + l,
        ops.dropout_keep_mask reproduces it) when the layer shapes allow, else F.dropout."""
        sn_w = self.spectral_weights()
        H = self.config.discriminator_hidden_dim
        # kinematic features + both encoder Linears + both LeakyReLUs: one launch (ops.EncoderFn); the LSTM layer
        # that consumes x applies LeakyReLU'(x) to the gradient it returns (x_act_slope), the encoder's backward
        # expects that
        fe = self.feature_encoder
        slope = float(fe[1].negative_slope)
        if self.bf16:
            # config 3: hoisted bf16 GEMMs + one launch per timestep (ops_bf16)
            p = float(self.config.lstm_dropout) if self.training else 0.0
            if p > 0 and dropout_seed is None:
                dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
            w1, w2 = sn_w.pop(0), sn_w.pop(0)
            lw = [w for l in range(self.config.discriminator_num_layers) for w in self.lstm.layer_weights(l)]
            h_final = ops_bf16.critic_body(pos_tm, dts_tm, lengths_i32, slope, 1e-8, self.config.discriminator_num_layers,
                                           w1, fe[0].bias, w2, fe[2].bias, lw, (p, dropout_seed or 0))
            return self.head(h_final, sn_w, dropout_seed)
        fused_enc = self.fuse_input_gemm and fe[3].negative_slope == fe[1].negative_slope
        if fused_enc:
            w1, w2 = sn_w.pop(0), sn_w.pop(0)
            x = ops.encoder(pos_tm, dts_tm, w1, fe[0].bias, w2, fe[2].bias, slope, 1e-8, True)
        else:
            x = self.run_sequential(self.feature_encoder, ops.kinematics_tm(pos_tm, dts_tm), sn_w)
        nl = self.config.discriminator_num_layers
        p = float(self.config.lstm_dropout) if self.training else 0.0
        fused = p > 0 and self.fuse_input_gemm and (2 * H) % 128 == 0
        if fused and dropout_seed is None:
            dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        none = (0.0, 0)
        h_final = None
        if nl == 2 and fused_enc and self.fuse_input_gemm and (fused or p == 0) and ops.lstm_stack2_supported(H, 2):
            # H = 128: both layers as one node whose backward contracts the top layer's input gradient inside its sweep
            h_final = ops.lstm_stack2(x, lengths_i32, self.lstm.layer_weights(0), self.lstm.layer_weights(1),
                                      (p, dropout_seed + 1) if fused else none, slope)
            return self.head(h_final, sn_w, dropout_seed)
        for l in range(nl):
            h_all, h_final = ops.lstm_layer(x, lengths_i32, self.lstm.layer_weights(l), fuse_x=self.fuse_input_gemm,
                                            in_drop=(p, dropout_seed + l) if fused and l > 0 else none,
                                            out_drop=(p, dropout_seed + l + 1) if fused and l + 1 < nl else none,
                                            x_act_slope=slope if (fused_enc and l == 0) else -1.0)
            x = h_all
            if l + 1 < nl and p > 0 and not fused:
                x = F.dropout(x, p, True)
        return self.head(h_final, sn_w, dropout_seed)

    def head(self, h_final: torch.Tensor, sn_w: List[torch.Tensor], dropout_seed: Optional[int] = None) -> torch.Tensor:
        """output_layers (RG.py:730-737) on the final hidden states [Bp, 2H] with the three normalised weights left in
        `sn_w`: one launch (ops.CriticHeadFn) with the head's Dropout as a counter-based mask (seed derived from
        `dropout_seed`, else drawn on the host), torch ops for widths outside the kernel's table."""
        ol = self.output_layers
        H = self.config.discriminator_hidden_dim
        standard = (len(ol) == 6 and isinstance(ol[0], SNLinear) and isinstance(ol[2], nn.Dropout) and len(sn_w) == 3)
        if (standard and h_final.dtype == torch.float32 and ops.critic_head_supported(H) and h_final.shape[0] % ops.TM == 0
                and ol[1].negative_slope == ol[4].negative_slope):
            p = float(ol[2].p) if self.training else 0.0
            seed = 0
            if p > 0:
                seed = dropout_seed + 7919 if dropout_seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
            w1, w2, w3 = sn_w
            return ops.critic_head(h_final, w1, ol[0].bias, w2, ol[3].bias, w3, ol[5].bias, float(ol[1].negative_slope), p, seed)
        return self.run_sequential(self.output_layers, h_final, sn_w)

    def forward(self, trajectories: torch.Tensor, dts: torch.Tensor, lengths: torch.Tensor,
                dropout_seed: Optional[int] = None) -> torch.Tensor:
        """trajectories (B,L,2), dts (B,L), lengths (B,) -> critic scores (B,1)."""
        L_.require_cuda(trajectories, dts)
        B = trajectories.shape[0]
        Bp = ops.pad_rows(B)
        if self.bf16:
            Bp = (B + ops_bf16.ROW_TILE - 1) // ops_bf16.ROW_TILE * ops_bf16.ROW_TILE
        pos_tm = _pad_batch(trajectories.transpose(0, 1), Bp, 1).contiguous()
        dts_tm = _pad_batch(dts.transpose(0, 1), Bp, 1).contiguous()
        lens = _pad_batch(lengths.to(device=trajectories.device, dtype=torch.int32).clamp(min=1), Bp, 0, 1)
        return self.forward_tm(pos_tm, dts_tm, lens.contiguous(), dropout_seed)[:B]
