"""Host side of the bf16-operand / fp32-accumulate path (BASELINE.json configs[2]: H = 512, T = 256).

At H = 512 the recurrent weights of one LSTM direction are 2 MB in bf16 - they fit neither the register file nor the LDS
of a CU, so the persistent one-workgroup-per-32-rows kernels of `ops` do not apply.  This path instead
  * hoists everything that is not on the recurrence into dense bf16 GEMMs over all L*B row-steps
    (x W_ih^T + b of both directions, the feature encoder, every dX and every dW): `gmt_bf16_gemm_nt`, `gmt_bf16_wgrad_tn`;
  * runs the recurrence as one launch per timestep (`gmt_bf16_lstm_layer_fwd/bwd`), the kernel boundary being the
    grid-wide hand-off of h_t;
  * keeps activations and weights in bf16, accumulators / cell state / parameter gradients / master weights in fp32.
The critic body (kinematics -> feature encoder -> BiLSTM stack -> final hidden states, RG.py:757-778) is ONE autograd
Function so that the big buffers are reused in place (gates -> d(pre-activation)).

The generator (RG.py:551-719) is the same idea with three launches per step (`gmt_bf16_gen_rollout_fwd/bwd`,
`GenRolloutBf16Fn`), and the gradient penalty's double backward through the BiLSTM stack is `LSTMStackGPFn` /
`LSTMStackBwdGPFn` over `gmt_bf16_lstm_layer_bwd_gp/_bwd2`.  Limits (refused loudly): generator
H = 512 only, no in-kernel early stop."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F

from . import _lib as L_
from . import ops

ROW_TILE = 256          # rows per workgroup tile of every bf16 kernel: batches are padded to a multiple of it
BF16 = torch.bfloat16

# ----------------------------------------------------------------------------------------------------------------------
# row groups (include/gmt_hip.h, gmt_bf16_set_row_groups): the library's second chain of step launches lives on a stream
# of its own, and a process has few hardware queues (GPU_MAX_HW_QUEUES = 4).  Round 2 met a run that never completed once
# default-priority streams had filled them.  The default of two groups therefore only stands while
# the process is known to hold few streams: as soon as a device-side process group (RCCL: communicator streams of its
# own) is alive, layer calls run as ONE chain on the caller's stream, whatever count was asked for.
# ----------------------------------------------------------------------------------------------------------------------
import os as _os

_ROW_GROUPS_WANTED = None        # what set_row_groups was last asked for (None: the library's default / GMT_BF16_ROW_GROUPS)
_ROW_GROUPS_POLICY = None        # what apply_row_group_policy last told the library (None: nothing yet)


def _device_group_alive() -> bool:
    import torch.distributed as dist
    return bool(dist.is_available() and dist.is_initialized() and dist.group.WORLD is not None
                and 'nccl' in str(dist.get_backend(dist.group.WORLD)))


def set_row_groups(n: int) -> int:
    """Ask for n (1 or 2) row groups.  Returns what the library was told: 1 while a device-side process group is alive,
    whatever was asked - round 3 ran exactly that combination (bf16 Trainer, one-rank RCCL group, overlap_g_forward, two
    groups) at the end of the full GPU suite and the first D step never completed (profiles/r3_03_stream_count_stall_stacks.txt);
    GMT_BF16_ROW_GROUPS_FORCE=1 lifts the guard for experiments."""
    global _ROW_GROUPS_WANTED, _ROW_GROUPS_POLICY
    _ROW_GROUPS_WANTED = int(n)
    _ROW_GROUPS_POLICY = None
    return apply_row_group_policy()


def apply_row_group_policy() -> int:
    """Called by the Trainer before every bf16 step (and by set_row_groups): tells the library ONE row group while a
    device-side process group (RCCL: communicator streams of its own) is alive, else the number asked for (default 2).
    Returns the number in force."""
    global _ROW_GROUPS_POLICY
    want = _ROW_GROUPS_WANTED
    if want is None:
        want = int(_os.environ.get('GMT_BF16_ROW_GROUPS', '2'))
    want = 1 if want < 1 else (2 if want > 2 else want)
    if _device_group_alive() and not _os.environ.get('GMT_BF16_ROW_GROUPS_FORCE'):
        want = 1
    if want != _ROW_GROUPS_POLICY:
        L_.call('gmt_bf16_set_row_groups', want)
        _ROW_GROUPS_POLICY = want
    return want

_PERMS: dict = {}


def slab_perm(H: int, device) -> torch.Tensor:
    """Row order of a [4H] gate axis in the kernels' slab layout: position s*128 + g*32 + u  <-  torch row g*H + 32 s + u."""
    key = (H, str(device))
    if key not in _PERMS:
        c = torch.arange(4 * H, device=device)
        s, g, u = c // 128, (c % 128) // 32, c % 32
        _PERMS[key] = g * H + s * 32 + u
    return _PERMS[key]


def slab_unperm_rows(w: torch.Tensor, H: int) -> torch.Tensor:
    """Inverse of indexing the first axis with slab_perm: slab-ordered rows -> torch's gate-major rows."""
    out = torch.empty_like(w)
    out[slab_perm(H, w.device)] = w
    return out


# The PLAIN hoisted GEMMs (no activation epilogue: the layers' input projections x W_ih^T + b and the input gradients
# dA W_ih over all L*B row-steps) go to the library GEMM torch dispatches to (hipBLASLt): measured on the config-3 shapes
# (tools/bench_bf16_gemm_vs_lib.py, M = 1.05 M rows): N4096 K512 763 -> 835 TF/s, N4096 K1024 953 -> 1262, N1024 K4096
# 1101 -> 1445, N512 K4096 1024 -> 1340.  GEMMs with a fused LeakyReLU / x LeakyReLU' epilogue stay on gmt_bf16_gemm_nt
# (a separate element-wise pass over [M, N] would cost more than the library gains).  The library adds the bias in bf16
# (one rounding of the bias more than the kernel's fp32 add: inside the bf16 path's tolerances).  False: everything on
# the hand-written kernel (A/B switch).
LIBRARY_PLAIN_GEMM = True


def gemm_nt(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, act: Optional[torch.Tensor] = None,
            epilogue: int = 0, slope: float = 0.2, label: Optional[str] = None) -> torch.Tensor:
    """C = epilogue(A W^T + bias): A [M, K] bf16 (row stride >= K), W [N, K] bf16 -> [M, N] bf16."""
    M, K = A.shape
    N = W.shape[0]
    L_.require_cuda(A, W)
    assert A.dtype == BF16 and W.dtype == BF16 and A.stride(1) == 1 and W.stride(1) == 1 and W.shape[1] == K
    assert act is None or (act.dtype == BF16 and act.stride(1) == 1 and act.shape == (M, N))
    if LIBRARY_PLAIN_GEMM and epilogue == 0 and act is None and A.is_contiguous() and W.is_contiguous():
        with L_.timed(label or f'bf16_gemm_N{N}_K{K}', flops=2.0 * M * N * K):
            return F.linear(A, W, None if bias is None else bias.to(BF16))
    C = torch.empty(M, N, device=A.device, dtype=BF16)
    L_.call('gmt_bf16_gemm_nt', A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), L_.ptr(C), N,
            L_.ptr(bias) if bias is not None else None, act.data_ptr() if act is not None else None,
            act.stride(0) if act is not None else 0, int(epilogue), float(slope), M, N, K, L_.stream(),
            label=label or f'bf16_gemm_N{N}_K{K}', flops=2.0 * M * N * K)
    return C


def wgrad_tn(dA: torch.Tensor, X: torch.Tensor, label: Optional[str] = None) -> torch.Tensor:
    """dA^T X in fp32: dA [M, N], X [M, I] bf16 views (row strides free) -> [N, I]."""
    M, N = dA.shape
    I = X.shape[1]
    assert dA.dtype == BF16 and X.dtype == BF16 and dA.stride(1) == 1 and X.stride(1) == 1 and X.shape[0] == M
    if LIBRARY_PLAIN_GEMM and N >= 4096 and I >= 1024:
        # the one weight-gradient shape of config 3 on which the library's fp32-output GEMM beats the kernel (layer 1's dW_ih,
        # N = 8H, I = 2H: 932 -> 1 089 TF/s; N2048 I512 704 vs 517 and N4096 I512 879 vs 817 stay here; tools/bench_bf16_wgrad_vs_lib.py)
        with L_.timed(label or f'bf16_wgrad_N{N}_I{I}', flops=2.0 * M * N * I):
            return torch.mm(dA.t(), X, out_dtype=torch.float32)
    tiles = (N // 256) * (I // 256)
    n_slabs = max(1, min(64, (768 + tiles - 1) // tiles, M // 4096 if M >= 4096 else 1))
    slabs = torch.empty(n_slabs, N, I, device=dA.device, dtype=torch.float32)
    L_.call('gmt_bf16_wgrad_tn', dA.data_ptr(), dA.stride(0), X.data_ptr(), X.stride(0), L_.ptr(slabs), M, N, I, n_slabs,
            L_.stream(), label=label or f'bf16_wgrad_N{N}_I{I}', flops=2.0 * M * N * I)
    return slabs[0] if n_slabs == 1 else slabs.sum(0)


def lstm_layer_weights(weights: List[torch.Tensor], H: int, nd: int):
    """bf16 forms of one layer's parameters (cached until the parameters change, ops.derived):
    wih [nd*4H, I] (slab-ordered rows, directions stacked), bias [nd*4H] fp32 (b_ih + b_hh), whh [nd, 4H, H],
    whhT [nd, H, 4H], wihT [I, nd*4H] (for dX = dA wihT^T)."""
    def build():
        p = slab_perm(H, weights[0].device)
        wih = torch.cat([weights[4 * d][p] for d in range(nd)], 0)
        bias = torch.cat([(weights[4 * d + 2] + weights[4 * d + 3])[p] for d in range(nd)], 0).float().contiguous()
        whh = torch.stack([weights[4 * d + 1][p] for d in range(nd)], 0)
        return (wih.to(BF16).contiguous(), bias, whh.to(BF16).contiguous(), whh.transpose(1, 2).to(BF16).contiguous(),
                wih.t().to(BF16).contiguous())
    return ops.derived('bf16_lstm_layer', [w for w in weights], build)


class CriticBodyBf16Fn(torch.autograd.Function):
    """pos [L,B,2], dts [L,B] fp32 (time-major, B % 256 == 0), lengths int32 [B], the two (spectrally normalised) encoder
    Linears and the LSTM parameters (per layer and direction: w_ih, w_hh, b_ih, b_hh) -> h_final [B, 2H] fp32
    (final hidden state of the last layer, forward | reverse: RG.py:776-778)."""

    @staticmethod
    def forward(ctx, pos, dts, lengths, slope, eps, n_layers, drop, w1, b1, w2, b2, *lw):
        L_.require_cuda(pos, dts, w1, w2)
        pos, dts = pos.contiguous().float(), dts.contiguous().float()
        L, B = dts.shape
        H = w1.shape[0]
        nd = 2
        if B % ROW_TILE or H % 256 or w1.shape != (H, 9) or w2.shape != (H, H) or len(lw) != 4 * nd * n_layers:
            raise L_.GmtError(f'bf16 critic body: B={B} must be a multiple of {ROW_TILE}, H={H} of 256')
        dev = pos.device
        M = L * B
        lengths = lengths.contiguous()
        # whether any row is shorter than L selects the kernel variant: a HOST decision.  Whoever made `lengths` may have
        # left the answer on the tensor (Trainer._critic_inputs, from the batch's lengths before they went to the device:
        # no synchronisation); otherwise it is read back here (one blocking device-to-host copy per call)
        ragged = getattr(lengths, 'gmt_ragged', None)
        if ragged is None:
            ragged = bool((lengths.min() < L).item())
        need_grad = any(ctx.needs_input_grad)
        # kinematic features, zero-padded to one 64-wide K tile
        feat = torch.empty(L, B, 9, device=dev, dtype=torch.float32)
        L_.call('gmt_kinematics_fwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(feat), L, B, float(eps), L_.stream(), nbytes=48.0 * M)
        # The first Linear's K is 9, its K tile 64: the spare slots carry the bf16 round-off of the features and of the
        # weights (x = x_hi + x_lo, w = w_hi + w_lo; slots 0-8 x_hi w_hi, 9-17 x_lo w_hi, 18-26 x_hi w_lo), so that layer
        # is computed to ~16 bits at no cost - the raw features span 1e-4 ... 1e4 (clamped jerk), too wide for 8 bits.
        # (The buffer is 256 wide - one weight-gradient tile - with a column of ones at 27: dW1 and db1 then come out of ONE
        # gmt_bf16_wgrad_tn call in the backward pass; the forward GEMM reads the first 64 columns through the row stride.)
        f2 = feat.view(M, 9)
        f_hi = f2.to(BF16)
        featw = torch.zeros(M, 256, device=dev, dtype=BF16)
        featw[:, 0:9] = f_hi
        featw[:, 9:18] = (f2 - f_hi.float()).to(BF16)
        featw[:, 18:27] = f_hi
        featw[:, 27] = 1.0
        feat64 = featw[:, :64]
        w1d = w1.detach().float()
        w_hi = w1d.to(BF16)
        w1p = torch.zeros(H, 64, device=dev, dtype=BF16)
        w1p[:, 0:9] = w_hi
        w1p[:, 9:18] = w_hi
        w1p[:, 18:27] = (w1d - w_hi.float()).to(BF16)          # (column 27 stays zero: the ones column does not reach h1)
        w2b = w2.detach().to(BF16).contiguous()
        h1 = gemm_nt(feat64, w1p, b1.detach().float().contiguous(), None, 1, slope, label='bf16_enc1')
        x = gemm_nt(h1, w2b, b2.detach().float().contiguous(), None, 1, slope, label='bf16_enc2')
        saved_layers = []
        for l in range(n_layers):
            ws = list(lw[4 * nd * l: 4 * nd * (l + 1)])
            wih, bias, whh, whhT, wihT = lstm_layer_weights(ws, H, nd)
            G = gemm_nt(x, wih, bias, None, 0, 0.0, label=f'bf16_pre_l{l}')                  # [M, nd*4H]
            c_all = torch.empty(M, nd * H, device=dev, dtype=torch.float32)
            h_all = torch.empty(M, nd * H, device=dev, dtype=BF16)
            # nn.LSTM's inter-layer dropout (RG.py:725): layer l also writes the masked copy of its output that layer l+1 reads
            p_l = float(drop[0]) if l + 1 < n_layers else 0.0
            h_drop = torch.empty(M, nd * H, device=dev, dtype=BF16) if p_l > 0 else None
            L_.call('gmt_bf16_lstm_layer_fwd', L_.ptr(G), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_drop), L_.ptr(whh),
                    L_.ptr(lengths), int(ragged), L, B, H, nd, p_l, (int(drop[1]) + l + 1) & 0xFFFFFFFF, L_.stream(),
                    label='bf16_lstm_fwd', flops=2.0 * (L - 1) * B * nd * 4 * H * H,
                    # per row-step-direction: pre 8H + c_prev 4H + h_prev 2H read, gates 8H + c 4H + h 2H (+ dropped h 2H) written
                    nbytes=float(L) * B * nd * H * (28 + (2 if p_l > 0 else 0)))
            saved_layers.append((x, G, c_all, h_all))
            x = h_drop if p_l > 0 else h_all
        x = h_all
        h_final = torch.cat([x[(L - 1) * B:, :H], x[:B, H:]], 1).float()
        if need_grad:
            ctx.save_for_backward(pos, dts, lengths, feat, w1, w2, *lw)
            ctx.bufs = (featw, h1, saved_layers)
        ctx.cfg = (float(slope), float(eps), int(n_layers), ragged, L, B, H, nd, (float(drop[0]), int(drop[1])))
        return h_final

    @staticmethod
    @torch.autograd.function.once_differentiable      # a create_graph=True pass (the gradient penalty) must fail loudly here
    def backward(ctx, dh_final):
        if ctx.bufs is None:
            raise L_.GmtError('bf16 critic body: backward called twice (the saved gate buffers were consumed in place)')
        pos, dts, lengths, feat, w1, w2, *lw = ctx.saved_tensors
        featw, h1, saved_layers = ctx.bufs
        ctx.bufs = None
        slope, eps, n_layers, ragged, L, B, H, nd, drop = ctx.cfg
        dev = pos.device
        M = L * B
        need = ctx.needs_input_grad
        need_in = need[0] or need[1]
        need_w1, need_w2 = need[7] or need[8], need[9] or need[10]
        need_lw = any(need[11:])
        dh_final = dh_final.contiguous().float()
        dh_fin = torch.stack([dh_final[:, :H], dh_final[:, H:]], 0).contiguous()
        dcc = torch.empty(nd, B, H, device=dev, dtype=torch.float32)
        dhc = torch.empty(nd, B, H, device=dev, dtype=torch.float32) if ragged else None
        grads_lw: List[Optional[torch.Tensor]] = [None] * len(lw)
        dh_out = None
        for l in range(n_layers - 1, -1, -1):
            x, G, c_all, h_all = saved_layers[l]
            ws = list(lw[4 * nd * l: 4 * nd * (l + 1)])
            wih, bias, whh, whhT, wihT = lstm_layer_weights(ws, H, nd)
            p_l = drop[0] if l + 1 < n_layers else 0.0       # the mask layer l's output went through on its way up
            L_.call('gmt_bf16_lstm_layer_bwd', L_.ptr(G), L_.ptr(c_all), L_.ptr(dh_out), L_.ptr(dh_fin), L_.ptr(dcc),
                    L_.ptr(dhc), L_.ptr(whhT), L_.ptr(lengths), int(ragged), L, B, H, nd, p_l,
                    (drop[1] + l + 1) & 0xFFFFFFFF, L_.stream(),
                    label='bf16_lstm_bwd', flops=2.0 * (L - 1) * B * nd * 4 * H * H,
                    # per row-step-direction: gates 8H + dA_next 8H + c 4H + c_prev 4H + dc 4H (+ dh_out 2H) read, dA 8H + dc 4H written
                    nbytes=float(L) * B * nd * H * (40 + (2 if dh_out is not None else 0)))
            dA = G                                                                             # [M, nd*4H] d(pre-activation)
            if need_lw:
                p = slab_perm(H, dev)
                d_wih = wgrad_tn(dA, x, label=f'bf16_wgrad_ih_l{l}')                           # [nd*4H, I]
                db = dA.sum(0, dtype=torch.float32)
                for d in range(nd):
                    cols = slice(d * 4 * H, (d + 1) * 4 * H)
                    if d == 0:      # h_prev of row-step r is h_all[r - B]
                        d_whh = wgrad_tn(dA[B:, cols], h_all[:M - B, :H], label='bf16_wgrad_hh')
                    else:           # reverse direction: h_prev is h_all[r + B]
                        d_whh = wgrad_tn(dA[:M - B, cols], h_all[B:, H:], label='bf16_wgrad_hh')
                    k = 4 * nd * l + 4 * d
                    grads_lw[k] = slab_unperm_rows(d_wih[cols], H)
                    grads_lw[k + 1] = slab_unperm_rows(d_whh, H)
                    grads_lw[k + 2] = slab_unperm_rows(db[cols], H)
                    grads_lw[k + 3] = grads_lw[k + 2]
            if l > 0:
                dh_out = gemm_nt(dA, wihT, None, None, 0, 0.0, label=f'bf16_dx_l{l}')         # [M, nd*H]
                dh_fin = torch.zeros_like(dh_fin)            # only the last layer's final states reach the head
            saved_layers[l] = None
        g_w1 = g_b1 = g_w2 = g_b2 = gpos = gdts = None
        if need_in or need_w1 or need_w2:
            e = x                                                                              # layer 0's input
            d_pre2 = gemm_nt(dA, wihT, None, e, 2, slope, label='bf16_dx_l0')                 # [M, H], LeakyReLU'(e) applied
            del dA, G
            if need_w2:
                g_w2 = wgrad_tn(d_pre2, h1, label='bf16_wgrad_enc2')
                g_b2 = d_pre2.sum(0, dtype=torch.float32)
            if need_in or need_w1:
                w2T = w2.detach().t().to(BF16).contiguous()
                d_pre1 = gemm_nt(d_pre2, w2T, None, h1, 2, slope, label='bf16_dx_enc2')       # [M, H]
                if need_w1:
                    gw = wgrad_tn(d_pre1, featw, label='bf16_wgrad_enc1')                      # [H, 256]
                    g_w1 = (gw[:, :9] + gw[:, 9:18]).contiguous()                              # x_hi + x_lo
                    g_b1 = gw[:, 27].contiguous()                                              # the ones column
                if need_in:
                    gfeat = torch.mm(d_pre1, w1.detach().to(BF16)).float().contiguous()        # [M, 9]
                    gpos, gdts = torch.empty_like(pos), torch.empty_like(dts)
                    L_.call('gmt_kinematics_bwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(gfeat), L_.ptr(gpos), L_.ptr(gdts),
                            L, B, eps, L_.stream(), nbytes=60.0 * M)
        return (gpos, gdts, None, None, None, None, None, g_w1, g_b1, g_w2, g_b2, *grads_lw)


def critic_body(pos, dts, lengths_i32, slope, eps, n_layers, w1, b1, w2, b2, lstm_weights, drop=(0.0, 0)):
    """drop = (p, seed): nn.LSTM's inter-layer dropout; the mask between layers l and l+1 is
    ops.dropout_keep_mask(L, B, 2H, p, seed + l + 1, pair16=True)."""
    return CriticBodyBf16Fn.apply(pos, dts, lengths_i32, slope, eps, n_layers, (float(drop[0]), int(drop[1])),
                                  w1, b1, w2, b2, *lstm_weights)


# ----------------------------------------------------------------------------------------------------------------------
# generator rollout at H = 512, no-grad form (csrc/gen_bf16.hip)
# ----------------------------------------------------------------------------------------------------------------------
def gen_weights(w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, wh1):
    """bf16 / slab-ordered forms of the generator's recurrent and head weights (cached until the parameters change)."""
    H = w_hh0.shape[1]

    def build():
        p = slab_perm(H, w_hh0.device)
        w_u = torch.cat([w_ih0[:, :3], w_ih0[:, 3 + H:]], 1)[p]                       # [4H, 7]: cur (3), dyn (4)
        w0 = torch.zeros(4 * H, H + 64, device=w_hh0.device, dtype=torch.float32)
        w0[:, :H] = w_hh0[p]
        w0[:, H:H + 7] = w_u
        w0[:, H + 7:H + 14] = w_u                                                     # the inputs' bf16 round-off rides here
        w1 = torch.cat([w_ih1[p], w_hh1[p]], 1)
        w_cond = w_ih0[:, 3:3 + H][p]
        return (w0.to(BF16).contiguous(), w1.to(BF16).contiguous(), (b_ih1 + b_hh1)[p].float().contiguous(),
                wh1.to(BF16).contiguous(), w_cond.to(BF16).contiguous(), (b_ih0 + b_hh0)[p].float().contiguous())
    return ops.derived('bf16_gen', [w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, wh1], build)


@torch.no_grad()
def gen_rollout(cond, lstm_params, head, init_in, starts, ends, target_tm, tf_mask, T: int, drop=(0.0, 0)) -> torch.Tensor:
    """cond [B, H] fp32 (condition encoder output), lstm_params = (w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1),
    head = (W1 [H/2, H], b1, W2 [3, H/2], b2), init_in [3], starts / ends [B, 2], target_tm [T, B, 3] or None, tf_mask uint8
    [T] (device) or None -> out [T, B, 3] fp32 (dx, dy, dt).  B % 256 == 0, H == 512."""
    from ._lib import GenBf16Args
    import ctypes
    B, H = cond.shape
    if H != 512 or B % ROW_TILE:
        raise L_.GmtError(f'bf16 generator rollout: H={H} must be 512 and B={B} a multiple of {ROW_TILE}')
    dev = cond.device
    w0, w1, bias1, wh1, w_cond, bias0 = gen_weights(*lstm_params, head[0])
    cond_pre = gemm_nt(cond.to(BF16).contiguous(), w_cond, bias0, None, 0, 0.0, label='bf16_gen_cond')       # [B, 4H]
    x0 = torch.zeros(2, B, H + 64, device=dev, dtype=BF16)
    rem = ends - starts
    u0 = torch.cat([init_in.reshape(1, 3).expand(B, 3), rem, torch.sqrt((rem ** 2).sum(-1, keepdim=True) + 1e-8),
                    torch.atan2(rem[:, 1:2], rem[:, 0:1])], 1).float()
    u_hi = u0.to(BF16)
    x0[0, :, H:H + 7] = u_hi
    x0[0, :, H + 7:H + 14] = (u0 - u_hi.float()).to(BF16)
    x1 = torch.zeros(2, B, 2 * H, device=dev, dtype=BF16)
    c0 = torch.zeros(2, B, H, device=dev, dtype=torch.float32)
    c1 = torch.zeros(2, B, H, device=dev, dtype=torch.float32)
    pos = torch.stack([starts, starts], 0).float().contiguous()
    out = torch.empty(T, B, 3, device=dev, dtype=torch.float32)
    ends_c = ends.float().contiguous()
    a = GenBf16Args()
    keep = [cond_pre, w0, w1, bias1, wh1, head[1].detach().float().contiguous(), head[2].detach().float().contiguous(),
            head[3].detach().float().contiguous(), ends_c, None if target_tm is None else target_tm.float().contiguous(), tf_mask]
    (a.cond_pre, a.w0, a.w1, a.bias1, a.wh1, a.bh1, a.wh2, a.bh2, a.ends, a.target, a.tf_mask) = [L_.ptr(t) for t in keep]
    a.out_seq, a.x0, a.x1, a.c0, a.c1, a.pos = (L_.ptr(t) for t in (out, x0, x1, c0, c1, pos))
    a.T, a.B, a.H = T, B, H
    a.dropout_p, a.seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFF
    g_exec = 2.0 * (4 * H * (H + 7) + 4 * H * 2 * H + (H // 2) * H + 3 * (H // 2))
    L_.call('gmt_bf16_gen_rollout_fwd', ctypes.byref(a), L_.stream(), label='bf16_gen_rollout', flops=g_exec * T * B)
    return out


def _slab_unperm_cols(g: torch.Tensor, H: int) -> torch.Tensor:
    out = torch.empty_like(g)
    out[:, slab_perm(H, g.device)] = g
    return out


class GenRolloutBf16Fn(torch.autograd.Function):
    """The training rollout at H = 512: g0 [B, 4H] fp32 (condition term incl. layer-0 biases, torch gate order), w_u [4H, 7],
    w_hh0, w_ih1, w_hh1 [4H, H], bias1 [4H], head (W1 [H/2, H], b1, W2 [3, H/2], b2), init_in [3] -> out [T, B, 3] fp32.
    Forward keeps every step's operands and state (gmt_bf16_gen_rollout_fwd, training form), backward is the three-launch
    BPTT sweep (gmt_bf16_gen_rollout_bwd) followed by the weight-gradient GEMMs over all T*B row-steps."""

    @staticmethod
    def forward(ctx, g0, w_u, w_hh0, w_ih1, w_hh1, bias1, wh1, bh1, wh2, bh2, init_in, starts, ends, target_tm, tf_list, T, drop):
        from ._lib import GenBf16Args
        import ctypes
        B = g0.shape[0]
        H = w_hh0.shape[1]
        if H != 512 or B % ROW_TILE:
            raise L_.GmtError(f'bf16 generator rollout: H={H} must be 512 and B={B} a multiple of {ROW_TILE}')
        dev = g0.device
        XW = 768
        p = slab_perm(H, dev)
        with torch.no_grad():
            w0 = torch.zeros(4 * H, H + 64, device=dev, dtype=torch.float32)
            w0[:, :H] = w_hh0[p]
            w0[:, H:H + 7] = w_u[p]
            w0[:, H + 7:H + 14] = w_u[p]
            w0 = w0.to(BF16)
            w1 = torch.cat([w_ih1[p], w_hh1[p]], 1).to(BF16).contiguous()
            bias1_s = bias1[p].float().contiguous()
            wh1_b = wh1.to(BF16).contiguous()
            cond_pre = g0[:, p].to(BF16).contiguous()
            x0 = torch.zeros(T + 1, B, XW, device=dev, dtype=BF16)
            rem = ends - starts
            u0 = torch.cat([init_in.reshape(1, 3).expand(B, 3), rem, torch.sqrt((rem ** 2).sum(-1, keepdim=True) + 1e-8),
                            torch.atan2(rem[:, 1:2], rem[:, 0:1])], 1).float()
            u_hi = u0.to(BF16)
            x0[0, :, H:H + 7] = u_hi
            x0[0, :, H + 7:H + 14] = (u0 - u_hi.float()).to(BF16)
            x1 = torch.zeros(T + 1, B, 2 * H, device=dev, dtype=BF16)
            c0 = torch.zeros(T + 1, B, H, device=dev, dtype=torch.float32)
            c1 = torch.zeros(T + 1, B, H, device=dev, dtype=torch.float32)
            gates0 = torch.empty(T, B, 4 * H, device=dev, dtype=BF16)
            gates1 = torch.empty(T, B, 4 * H, device=dev, dtype=BF16)
            y_all = torch.empty(T, B, H // 2, device=dev, dtype=BF16)
            oraw = torch.empty(T, B, device=dev, dtype=torch.float32)
            pos = torch.stack([starts, starts], 0).float().contiguous()
            out = torch.empty(T, B, 3, device=dev, dtype=torch.float32)
            tf_dev = None
            if tf_list is not None and any(tf_list):
                tf_dev = torch.tensor([1 if b else 0 for b in tf_list], dtype=torch.uint8).to(dev)
            wh2_c, bh1_c, bh2_c = wh2.detach().float().contiguous(), bh1.detach().float().contiguous(), bh2.detach().float().contiguous()
            ends_c = ends.float().contiguous()
            tgt = None if target_tm is None else target_tm.float().contiguous()
            a = GenBf16Args()
            (a.cond_pre, a.w0, a.w1, a.bias1, a.wh1, a.bh1, a.wh2, a.bh2, a.ends, a.target, a.tf_mask) = [
                L_.ptr(t) for t in (cond_pre, w0, w1, bias1_s, wh1_b, bh1_c, wh2_c, bh2_c, ends_c, tgt, tf_dev)]
            a.out_seq, a.x0, a.x1, a.c0, a.c1, a.pos = (L_.ptr(t) for t in (out, x0, x1, c0, c1, pos))
            a.gates0, a.gates1, a.y_all, a.oraw_all = (L_.ptr(t) for t in (gates0, gates1, y_all, oraw))
            a.T, a.B, a.H = T, B, H
            a.dropout_p, a.seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFF
            g_exec = 2.0 * (4 * H * (H + 7) + 4 * H * 2 * H + (H // 2) * H + 3 * (H // 2))
            L_.call('gmt_bf16_gen_rollout_fwd', ctypes.byref(a), L_.stream(), label='bf16_gen_rollout_train', flops=g_exec * T * B)
        ctx.save_for_backward(w_u, w_hh0, w_ih1, w_hh1, wh1, wh2)
        ctx.bufs = (x0, x1, c0, c1, gates0, gates1, y_all, oraw)
        ctx.cfg = (T, B, H, [bool(b) for b in tf_list] if tf_list is not None else [False] * T, (float(drop[0]), int(drop[1]) & 0xFFFFFFFF))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        from ._lib import GenBf16BwdArgs
        import ctypes
        if ctx.bufs is None:
            raise L_.GmtError('bf16 generator rollout: backward called twice (the saved gate buffers were consumed in place)')
        w_u, w_hh0, w_ih1, w_hh1, wh1, wh2 = ctx.saved_tensors
        x0, x1, c0, c1, gates0, gates1, y_all, oraw = ctx.bufs
        ctx.bufs = None
        T, B, H, tf, drop = ctx.cfg
        dev = d_out.device
        p = slab_perm(H, dev)
        HH = H // 2
        w1T = wh1.detach().t().to(BF16).contiguous()                                   # [H, HH]
        whh1T = w_hh1.detach()[p].t().to(BF16).contiguous()                            # [H, 4H]
        wih1T = w_ih1.detach()[p].t().to(BF16).contiguous()
        whh0T = w_hh0.detach()[p].t().to(BF16).contiguous()
        wu8 = torch.zeros(4 * H, 8, device=dev, dtype=BF16)
        wu8[:, :7] = w_u.detach()[p]
        wh2_c = wh2.detach().float().contiguous()
        dypre = torch.empty(T, B, HH, device=dev, dtype=BF16)
        d_o = torch.empty(T, B, 3, device=dev, dtype=torch.float32)
        d_in0 = torch.empty(B, 3, device=dev, dtype=torch.float32)
        dc0 = torch.empty(B, H, device=dev, dtype=torch.float32)
        dc1 = torch.empty(B, H, device=dev, dtype=torch.float32)
        pc = torch.zeros(B, 2, device=dev, dtype=torch.float32)
        tf_host = (ctypes.c_ubyte * T)(*[1 if b else 0 for b in tf])
        d_out = d_out.contiguous().float()
        a = GenBf16BwdArgs()
        (a.gates0, a.gates1, a.c0, a.c1, a.x0, a.y_all, a.oraw_all, a.d_out, a.w1T, a.whh1T, a.wih1T, a.whh0T, a.w_u, a.wh2,
         a.dypre_all, a.d_o, a.d_in0, a.dc0, a.dc1, a.p_carry) = [L_.ptr(t) for t in (
             gates0, gates1, c0, c1, x0, y_all, oraw, d_out, w1T, whh1T, wih1T, whh0T, wu8, wh2_c, dypre, d_o, d_in0, dc0, dc1, pc)]
        a.tf_mask_host = ctypes.cast(tf_host, ctypes.c_void_p)
        a.T, a.B, a.H = T, B, H
        a.dropout_p, a.seed = drop
        g_exec = 2.0 * (4 * H * (H + 7) + 4 * H * 2 * H + (H // 2) * H + 3 * (H // 2))
        L_.call('gmt_bf16_gen_rollout_bwd', ctypes.byref(a), L_.stream(), label='bf16_gen_bptt', flops=g_exec * T * B)
        M = T * B
        dA0, dA1 = gates0.view(M, 4 * H), gates1.view(M, 4 * H)
        gw0 = wgrad_tn(dA0, x0[:T].view(M, x0.shape[2]), label='bf16_gen_wgrad0')      # [4H, 768]
        gw1 = wgrad_tn(dA1, x1[:T].view(M, 2 * H), label='bf16_gen_wgrad1')            # [4H, 2H]
        d_w_hh0 = slab_unperm_rows(gw0[:, :H].contiguous(), H)
        d_w_u = slab_unperm_rows((gw0[:, H:H + 7] + gw0[:, H + 7:H + 14]).contiguous(), H)
        d_w_ih1 = slab_unperm_rows(gw1[:, :H].contiguous(), H)
        d_w_hh1 = slab_unperm_rows(gw1[:, H:].contiguous(), H)
        d_bias1 = slab_unperm_rows(dA1.sum(0, dtype=torch.float32), H)
        d_g0 = _slab_unperm_cols(gates0.sum(0, dtype=torch.float32), H)
        h1_all = x1[1:].view(M, 2 * H)[:, H:]                                          # h1_t, row stride 2H
        d_wh1 = wgrad_tn(dypre.view(M, HH), h1_all, label='bf16_gen_wgrad_head')       # [HH, H]
        d_bh1 = dypre.view(M, HH).sum(0, dtype=torch.float32)
        d_wh2 = d_o.view(M, 3).t() @ y_all.view(M, HH).float()
        d_bh2 = d_o.view(M, 3).sum(0)
        return (d_g0, d_w_u, d_w_hh0, d_w_ih1, d_w_hh1, d_bias1, d_wh1, d_bh1, d_wh2, d_bh2, d_in0.sum(0),
                None, None, None, None, None, None)


# ----------------------------------------------------------------------------------------------------------------------
# gradient penalty at H = 512 (compute_gradient_penalty, RG.py:790-842): the BiLSTM stack as a twice-differentiable op
# ----------------------------------------------------------------------------------------------------------------------
class _Saved:
    """What the stack's forward keeps for its two backward forms (a plain Python holder: not a tensor argument)."""
    def __init__(self):
        self.layers, self.dims, self.first = [], None, None
        self.valid, self.last = None, None      # ragged batches: [M] 0/1 per row-step (bf16), row index of (len-1, b) per trajectory
        self.drop = []                          # inter-layer dropout masks [M, nd*H] fp32 (0 or 1/(1-p)), one per layer boundary


def gp_drop_mask(L: int, B: int, width: int, p: float, device) -> torch.Tensor:
    """nn.LSTM's inter-layer dropout mask for the penalty's critic pass (RG.py:725 under D.train(), RG.py:822): a fresh draw, as
    in the reference; [L*B, width] fp32 of 0 / 1/(1-p).  (The tests replace this function to replay a known mask.)"""
    return (torch.rand(L * B, width, device=device) >= p).float() / (1.0 - p)


def _drop(x: torch.Tensor, mk: Optional[torch.Tensor]) -> torch.Tensor:
    return x if mk is None else (x.float() * mk).to(x.dtype)


def _mask_g_(X: torch.Tensor, valid: Optional[torch.Tensor], H: int, nd: int = 2) -> torch.Tensor:
    """Ragged batches on the full-length step kernels.  X [M, nd*4H] lives in pre-activation space (slab column order).  The
    backward direction of a packed sequence (RG.py:764-772) starts at t = len-1 from zero state; the full-length kernels start
    at t = L-1.  Zeroing the HOISTED part of gate g's pre-activation at the row-steps t >= len of direction 1 keeps that
    direction's state exactly zero there (h = 0 -> the recurrent part of a_g is 0 -> g = 0 -> c = i g = 0 -> h = 0, by
    induction from the zero initial state), whatever the parameters and inputs: the masked full-length layer IS the packed
    layer as a function, so its derivatives of every order are too.  The mask is the linear map pre -> M * pre between the
    GEMMs and the sweeps, so it is applied wherever a quantity crosses that boundary (pre, d pre, the adjoint sweep's hoisted
    term).  Direction 0 needs no mask: steps t >= len come after everything that is read (final state = h at t = len-1)."""
    if valid is not None:
        v = X.view(X.shape[0], nd, H // 32, 4, 32)
        v[:, 1, :, 2, :] *= valid.to(X.dtype)[:, None, None]
    return X


class LSTMStackGPFn(torch.autograd.Function):
    """e [L, B, H] fp32 (the feature encoder's output), lengths [B] or None, the LSTM parameters -> h_final [B, 2H] fp32.
    Ragged batches run on the same full-length kernels behind a mask (_mask_g_).
    Under create_graph=True (the penalty's autograd.grad w.r.t. the interpolated positions) its backward is the differentiable
    op LSTMStackBwdGPFn; the penalty's own backward then runs the second-order sweep.  Inter-layer dropout (drop_p > 0) is a
    fresh mask per call (gp_drop_mask), a linear map between the layers that every pass applies at the same place."""

    @staticmethod
    def forward(ctx, e, lengths, drop_p, n_layers, *lw):
        L, B, H = e.shape
        nd = 2
        if B % ROW_TILE or H % 256:
            raise L_.GmtError(f'bf16 gradient penalty: B={B} must be a multiple of {ROW_TILE}, H={H} of 256')
        dev = e.device
        M = L * B
        hold = _Saved()
        hold.dims = (L, B, H, nd, int(n_layers))
        if lengths is not None and int(lengths.min().item()) < L:
            ln = lengths.to(device=dev, dtype=torch.int64).clamp(1, L)
            hold.valid = (torch.arange(L, device=dev)[:, None] < ln[None, :]).reshape(M).to(BF16)
            hold.last = (ln - 1) * B + torch.arange(B, device=dev)
        x = e.detach().reshape(M, H).to(BF16).contiguous()
        for l in range(n_layers):
            ws = list(lw[4 * nd * l: 4 * nd * (l + 1)])
            wih, bias, whh, whhT, wihT = lstm_layer_weights(ws, H, nd)
            G = _mask_g_(gemm_nt(x, wih, bias, None, 0, 0.0, label=f'bf16_gp_pre_l{l}'), hold.valid, H, nd)
            c_all = torch.empty(M, nd * H, device=dev, dtype=torch.float32)
            h_all = torch.empty(M, nd * H, device=dev, dtype=BF16)
            L_.call('gmt_bf16_lstm_layer_fwd', L_.ptr(G), L_.ptr(c_all), L_.ptr(h_all), None, L_.ptr(whh), None, 0, L, B, H, nd,
                    0.0, 0, L_.stream(), label='bf16_gp_lstm_fwd', flops=2.0 * (L - 1) * B * nd * 4 * H * H)
            hold.layers.append((x, G, c_all, h_all))
            mk = gp_drop_mask(L, B, nd * H, float(drop_p), dev) if (drop_p and l + 1 < n_layers) else None
            hold.drop.append(mk)
            x = _drop(h_all, mk)                # the next layer reads h through the mask; the recurrence and the output do not
        ctx.hold = hold
        ctx.save_for_backward(e, *lw)
        ctx.set_materialize_grads(False)
        x = hold.layers[-1][3]
        h_fwd = x[(L - 1) * B:, :H] if hold.last is None else x[hold.last, :H]      # direction 0: h at t = len-1
        return torch.cat([h_fwd, x[:B, H:]], 1).float()

    @staticmethod
    def backward(ctx, dh_final):
        e, *lw = ctx.saved_tensors
        if torch.is_grad_enabled():
            # first backward under create_graph=True (the penalty's autograd.grad w.r.t. the interpolated positions): only the
            # input gradient is asked for, and it is produced by a differentiable op
            de = LSTMStackBwdGPFn.apply(dh_final, e, ctx.hold, *lw)
            return (de, None, None, None) + (None,) * len(lw)
        # The penalty's own backward reaches this node once more, through the head's double backward (the adjoint of h_final
        # as a saved activation of the head).  The head is piecewise linear, so that adjoint is identically zero: nothing to add
        # to what LSTMStackBwdGPFn.backward already returned.  Anything else is not the gradient penalty.
        hold = ctx.hold
        hold.layers, hold.first = [], None
        if dh_final is None or not bool(dh_final.any().item()):
            return (None, None, None, None) + (None,) * len(lw)
        raise L_.GmtError('LSTMStackGPFn serves the gradient penalty only; the ordinary critic pass is ops_bf16.critic_body')


class LSTMStackBwdGPFn(torch.autograd.Function):
    """The first backward sweep of the stack as a differentiable op: (dh_final, e, parameters) -> de.  Its backward is the
    second-order pass: per layer the hoisted xhat W_ih^T GEMM + gmt_bf16_lstm_layer_bwd2, the explicit weight terms
    (da^T xhat, da^T hhat), then an ordinary backward sweep that takes the adjoints of the saved gates / cell states as extra
    inputs and yields the remaining parameter gradients and the gradient w.r.t. e."""

    @staticmethod
    def forward(ctx, dh_final, e, hold, *lw):
        L, B, H, nd, n_layers = hold.dims
        dev = e.device
        M = L * B
        dh_fin = torch.stack([dh_final[:, :H], dh_final[:, H:]], 0).float().contiguous()
        dcc = torch.empty(nd, B, H, device=dev, dtype=torch.float32)
        first = []
        dh_out = None
        if hold.last is not None:
            # ragged: direction 0's final state was read at t = len-1, so its gradient enters there, as a gradient of h_all
            dh_out = torch.zeros(M, nd * H, device=dev, dtype=BF16)
            dh_out[hold.last, :H] = dh_final[:, :H].to(BF16)
            dh_fin[0].zero_()
        for l in range(n_layers - 1, -1, -1):
            x, G, c_all, h_all = hold.layers[l]
            ws = list(lw[4 * nd * l: 4 * nd * (l + 1)])
            wih, bias, whh, whhT, wihT = lstm_layer_weights(ws, H, nd)
            da = torch.empty_like(G)
            dct = torch.empty(M, nd * H, device=dev, dtype=torch.float32)
            dht = torch.empty(M, nd * H, device=dev, dtype=torch.float32)
            L_.call('gmt_bf16_lstm_layer_bwd_gp', L_.ptr(G), L_.ptr(da), L_.ptr(c_all), L_.ptr(dh_out), L_.ptr(dh_fin), None, None,
                    L_.ptr(dct), L_.ptr(dht), L_.ptr(dcc), L_.ptr(whhT), L, B, H, nd, L_.stream(), label='bf16_gp_lstm_bwd',
                    flops=2.0 * (L - 1) * B * nd * 4 * H * H)
            # da is the gradient of the TOTAL pre-activation (what the sweep contracts with W_hh); what the GEMM side sees
            # (dX, and later the W_ih terms) is M * da
            da_m = da if hold.valid is None else _mask_g_(da.clone(), hold.valid, H, nd)
            first.insert(0, (da, dct, dht, da_m))
            dh_out = gemm_nt(da_m, wihT, None, None, 0, 0.0, label=f'bf16_gp_dx_l{l}')
            if l > 0:
                dh_out = _drop(dh_out, hold.drop[l - 1])     # gradient of layer l-1's h_all: through the inter-layer mask
            dh_fin = torch.zeros_like(dh_fin)
        hold.first = first
        ctx.hold = hold
        ctx.save_for_backward(*lw)
        return dh_out.float().view(L, B, H)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, xhat):
        lw = ctx.saved_tensors
        hold = ctx.hold
        L, B, H, nd, n_layers = hold.dims
        dev = xhat.device
        M = L * B
        p = slab_perm(H, dev)
        g_lw = [None] * len(lw)
        xh = xhat.reshape(M, H).to(BF16).contiguous()
        second = []
        hhat_fin = None
        for l in range(n_layers):                      # the adjoint chain runs bottom-up: layer 0's sweep produced de
            x, G, c_all, h_all = hold.layers[l]
            da, dct, dht, da_m = hold.first[l]
            ws = list(lw[4 * nd * l: 4 * nd * (l + 1)])
            wih, bias, whh, whhT, wihT = lstm_layer_weights(ws, H, nd)
            PH = _mask_g_(gemm_nt(xh, wih, None, None, 0, 0.0, label=f'bf16_gp_prehat_l{l}'), hold.valid, H, nd)
            hhat = torch.empty(M, nd * H, device=dev, dtype=BF16)
            hhat_fin = torch.empty(nd, B, H, device=dev, dtype=torch.float32)
            gg = torch.empty(M, nd * 4 * H, device=dev, dtype=torch.float32)
            gc = torch.empty(M, nd * H, device=dev, dtype=torch.float32)
            chat = torch.empty(nd, B, H, device=dev, dtype=torch.float32)
            L_.call('gmt_bf16_lstm_layer_bwd2', L_.ptr(PH), L_.ptr(G), L_.ptr(c_all), L_.ptr(dct), L_.ptr(dht), L_.ptr(hhat),
                    L_.ptr(hhat_fin), L_.ptr(gg), L_.ptr(gc), L_.ptr(chat), L_.ptr(whh), L, B, H, nd, L_.stream(),
                    label='bf16_gp_lstm_bwd2', flops=2.0 * (L - 1) * B * nd * 4 * H * H)
            # explicit weight terms of the first backward sweep: dX = da W_ih, dh_t = dh_out[t] + da_{t+-1} W_hh
            e_wih = wgrad_tn(da_m, xh, label='bf16_gp_wgrad_ih2')
            for d in range(nd):
                cols = slice(d * 4 * H, (d + 1) * 4 * H)
                if d == 0:
                    e_whh = wgrad_tn(da[B:, cols], hhat[:M - B, :H], label='bf16_gp_wgrad_hh2')
                else:
                    e_whh = wgrad_tn(da[:M - B, cols], hhat[B:, H:], label='bf16_gp_wgrad_hh2')
                k = 4 * nd * l + 4 * d
                g_lw[k] = slab_unperm_rows(e_wih[cols], H)
                g_lw[k + 1] = slab_unperm_rows(e_whh, H)
            second.append((gg, gc))
            top_hhat = hhat
            xh = _drop(hhat, hold.drop[l])                   # the adjoint of the next layer's dX: the same linear map
        xh = top_hhat
        # adjoint of dh_final: direction 0's entered the sweep through dh_out at t = len-1 when the batch is ragged
        hhat_fwd = hhat_fin[0] if hold.last is None else xh[hold.last, :H].float()
        hhat_final = torch.cat([hhat_fwd, hhat_fin[1]], 1)
        # ordinary backward sweep through the forward pass, fed by the adjoints of its saved activations
        zero_fin = torch.zeros(nd, B, H, device=dev, dtype=torch.float32)
        dcc = torch.empty(nd, B, H, device=dev, dtype=torch.float32)
        dh_out = None
        g_e = None
        for l in range(n_layers - 1, -1, -1):
            x, G, c_all, h_all = hold.layers[l]
            gg, gc = second[l]
            ws = list(lw[4 * nd * l: 4 * nd * (l + 1)])
            wih, bias, whh, whhT, wihT = lstm_layer_weights(ws, H, nd)
            da2 = torch.empty_like(G)
            L_.call('gmt_bf16_lstm_layer_bwd_gp', L_.ptr(G), L_.ptr(da2), L_.ptr(c_all), L_.ptr(dh_out), L_.ptr(zero_fin), L_.ptr(gg),
                    L_.ptr(gc), None, None, L_.ptr(dcc), L_.ptr(whhT), L, B, H, nd, L_.stream(), label='bf16_gp_lstm_bwd_x',
                    flops=2.0 * (L - 1) * B * nd * 4 * H * H)
            d_whh = [wgrad_tn(da2[B:, :4 * H], h_all[:M - B, :H], label='bf16_gp_wgrad_hh'),      # the W_hh terms: total d(pre-activation)
                     wgrad_tn(da2[:M - B, 4 * H:], h_all[B:, H:], label='bf16_gp_wgrad_hh')]
            _mask_g_(da2, hold.valid, H, nd)                                                   # the GEMM side: M * da2
            d_wih = wgrad_tn(da2, x, label='bf16_gp_wgrad_ih')
            db = da2.sum(0, dtype=torch.float32)
            for d in range(nd):
                cols = slice(d * 4 * H, (d + 1) * 4 * H)
                k = 4 * nd * l + 4 * d
                g_lw[k] = g_lw[k] + slab_unperm_rows(d_wih[cols], H)
                g_lw[k + 1] = g_lw[k + 1] + slab_unperm_rows(d_whh[d], H)
                g_lw[k + 2] = slab_unperm_rows(db[cols], H)
                g_lw[k + 3] = g_lw[k + 2]
            dx = gemm_nt(da2, wihT, None, None, 0, 0.0, label=f'bf16_gp_dx2_l{l}')
            if l > 0:
                dh_out = _drop(dx, hold.drop[l - 1])
            else:
                g_e = dx.float().view(L, B, H)
        return (hhat_final, g_e, None, *g_lw)


def lstm_stack_gp(e, n_layers, lstm_weights, lengths=None, drop_p=0.0):
    """lengths [B] (valid steps per row, 1..L) or None for full-length rows; drop_p: nn.LSTM's inter-layer dropout."""
    return LSTMStackGPFn.apply(e, lengths, float(drop_p), n_layers, *lstm_weights)
