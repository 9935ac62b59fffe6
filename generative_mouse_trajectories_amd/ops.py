"""torch.autograd glue around the HIP kernels of libgmt_hip.so.

Every Function here is a thin host wrapper: it allocates the output / saved buffers with
torch (PyTorch owns device memory and streams), calls the C ABI on the current stream and
returns.  The large time-parallel contractions that are plain GEMMs (input projections,
weight gradients over K = L*B) go to rocBLAS through torch.mm; everything recurrent,
element-wise or reducing is hand-written HIP.  GPU only - there is no CPU path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _lib as L_

TM = 32  # trajectories per workgroup tile (csrc/lstm_tile.h GMT_TM)


def pad_rows(n: int) -> int:
    return (n + TM - 1) // TM * TM


# ----------------------------------------------------------------------------------------
# kinematic features (RecurrentGAN.py:315-389), time-major
# ----------------------------------------------------------------------------------------
class KinematicsFn(torch.autograd.Function):
    """pos [L,B,2], dts [L,B] -> feat [L,B,9]."""

    @staticmethod
    def forward(ctx, pos: torch.Tensor, dts: torch.Tensor, eps: float = 1e-8):
        L_.require_cuda(pos, dts)
        pos = pos.contiguous().float()
        dts = dts.contiguous().float()
        L, B = dts.shape
        feat = torch.empty(L, B, 9, device=pos.device, dtype=torch.float32)
        L_.call('gmt_kinematics_fwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(feat), L, B, eps, L_.stream())
        ctx.save_for_backward(pos, dts)
        ctx.eps = eps
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        pos, dts = ctx.saved_tensors
        L, B = dts.shape
        gfeat = gfeat.contiguous()
        gpos = torch.empty_like(pos)
        gdts = torch.empty_like(dts)
        L_.call('gmt_kinematics_bwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(gfeat), L_.ptr(gpos), L_.ptr(gdts),
                L, B, ctx.eps, L_.stream())
        return gpos, gdts, None


def kinematics_tm(pos: torch.Tensor, dts: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    return KinematicsFn.apply(pos, dts, eps)


# ----------------------------------------------------------------------------------------
# weight packing (fragment-major order of csrc/lstm_tile.h)
# ----------------------------------------------------------------------------------------
def pack_forward(ws: List[torch.Tensor], H: int, K: int, col0: int = 0, k_valid: Optional[int] = None) -> torch.Tensor:
    """ws: per-direction [4H, ld] row-major weights; packs columns [col0, col0+k_valid) as
    the B operand of out = A W^T.  Returns [ndir, 4H*K]."""
    nd = len(ws)
    out = torch.empty(nd, 4 * H * K, device=ws[0].device, dtype=torch.float32)
    kv = K if k_valid is None else k_valid
    for d, w in enumerate(ws):
        w = w.detach()
        assert w.is_contiguous() and w.shape[0] == 4 * H
        src = w.data_ptr() + 4 * col0
        L_.call('gmt_pack_weight', src, out[d].data_ptr(), K, kv, H // 32, 4, H, 1, w.shape[1], L_.stream())
    return out


def pack_transposed(ws: List[torch.Tensor], n_out: int, K: int, col0: int = 0, n_tiles_per_wave: int = 1) -> torch.Tensor:
    """ws: per-direction [K, ld] row-major; packs B[k][n] = w[k][col0 + n], n < n_out (the
    operand of dX = dA W).  Returns [ndir, K*n_out]."""
    nd = len(ws)
    out = torch.empty(nd, K * n_out, device=ws[0].device, dtype=torch.float32)
    for d, w in enumerate(ws):
        w = w.detach()
        assert w.is_contiguous() and w.shape[0] == K
        src = w.data_ptr() + 4 * col0
        L_.call('gmt_pack_weight', src, out[d].data_ptr(), K, K, n_out // 32, 1, 0, w.shape[1], 1, L_.stream())
    return out


# ----------------------------------------------------------------------------------------
# packed bidirectional LSTM layer (RecurrentGAN.py:720-727, 764-778)
# ----------------------------------------------------------------------------------------
class LSTMLayerFn(torch.autograd.Function):
    """x [L,B,I] (B % 32 == 0), lengths int32 [B] -> (h_all [L,B,nd*H], h_final [B,nd*H]).
    weights: per direction (w_ih [4H,I], w_hh [4H,H], b_ih [4H], b_hh [4H])."""

    @staticmethod
    def forward(ctx, x, lengths, fuse_x, need_h_all, *weights):
        L_.require_cuda(x)
        nd = len(weights) // 4
        w_ih = [weights[4 * d] for d in range(nd)]
        w_hh = [weights[4 * d + 1] for d in range(nd)]
        H = w_hh[0].shape[1]
        L, B, I = x.shape
        assert B % TM == 0, 'pad the batch to a multiple of 32 rows'
        x = x.contiguous()
        dev = x.device
        bias = torch.stack([weights[4 * d + 2].detach() + weights[4 * d + 3].detach() for d in range(nd)]).contiguous()
        w_hh_pk = pack_forward(w_hh, H, H)
        gates = torch.empty(nd, L, B, 4 * H, device=dev, dtype=torch.float32)
        c_all = torch.empty(nd, L, B, H, device=dev, dtype=torch.float32)
        h_all = torch.empty(L, B, nd * H, device=dev, dtype=torch.float32)
        h_final = torch.empty(B, nd * H, device=dev, dtype=torch.float32)
        if fuse_x:
            w_ih_pk = pack_forward(w_ih, H, I)
            L_.call('gmt_lstm_layer_fwd', L_.ptr(x), None, L_.ptr(w_ih_pk), L_.ptr(w_hh_pk), L_.ptr(bias),
                    L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                    L, B, I, H, nd, L_.stream())
        else:
            x2 = x.view(L * B, I)
            for d in range(nd):
                torch.addmm(bias[d], x2, w_ih[d].detach().t(), out=gates[d].view(L * B, 4 * H))
            L_.call('gmt_lstm_layer_fwd', None, L_.ptr(gates), None, L_.ptr(w_hh_pk), None,
                    L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                    L, B, I, H, nd, L_.stream())
        ctx.save_for_backward(x, lengths, gates, c_all, h_all, *weights)
        ctx.dims = (L, B, I, H, nd)
        ctx.set_materialize_grads(False)
        return h_all, h_final

    @staticmethod
    def backward(ctx, dh_all, dh_final):
        x, lengths, gates, c_all, h_all, *weights = ctx.saved_tensors
        L, B, I, H, nd = ctx.dims
        w_ih = [weights[4 * d] for d in range(nd)]
        w_hh = [weights[4 * d + 1] for d in range(nd)]
        if dh_all is not None:
            dh_all = dh_all.contiguous()
        if dh_final is not None:
            dh_final = dh_final.contiguous()
        w_hh_pkT = pack_transposed(w_hh, H, 4 * H)
        # `gates` is overwritten in place with d(pre-activation): the graph can be back-propagated once
        L_.call('gmt_lstm_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(w_hh_pkT), L_.ptr(lengths),
                L_.ptr(dh_all), L_.ptr(dh_final), None, None, L, B, H, nd, L_.stream())
        da = gates
        x2 = x.view(L * B, I)
        grads: List[Optional[torch.Tensor]] = []
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(da[0].view(L * B, 4 * H), w_ih[0])
            for d in range(1, nd):
                dx.addmm_(da[d].view(L * B, 4 * H), w_ih[d])
            dx = dx.view(L, B, I)
        need_w = any(ctx.needs_input_grad[4:])
        for d in range(nd):
            if not need_w:
                grads += [None, None, None, None]
                continue
            da2 = da[d].view(L * B, 4 * H)
            d_wih = torch.mm(da2.t(), x2)
            hv = h_all.view(L * B, nd * H)[:, d * H:(d + 1) * H]
            if L > 1:
                if d == 0:      # forward direction: h_prev(t) = h[t-1]
                    d_whh = torch.mm(da2[B:].t(), hv[:-B])
                else:           # reverse direction: h_prev(t) = h[t+1] (zero beyond the row's length)
                    d_whh = torch.mm(da2[:-B].t(), hv[B:])
            else:
                d_whh = torch.zeros_like(w_hh[d])
            db = da2.sum(0)
            grads += [d_wih, d_whh, db, db]
        return (dx, None, None, None, *grads)


def lstm_layer(x, lengths_i32, weights, fuse_x: bool = True, need_h_all: bool = True):
    return LSTMLayerFn.apply(x, lengths_i32, fuse_x, need_h_all, *weights)


# ----------------------------------------------------------------------------------------
# fused Adam over a flat buffer
# ----------------------------------------------------------------------------------------
def adam_step_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
               lr: float, betas: Tuple[float, float], eps: float, step: int, grad_scale: float = 1.0) -> None:
    L_.require_cuda(param, grad, exp_avg, exp_avg_sq)
    n = param.numel()
    assert grad.numel() == n and exp_avg.numel() == n and exp_avg_sq.numel() == n
    L_.call('gmt_adam_step', L_.ptr(param), L_.ptr(grad), L_.ptr(exp_avg), L_.ptr(exp_avg_sq), n,
            lr, betas[0], betas[1], eps, step, grad_scale, L_.stream())
