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
import torch.nn.functional as F

from . import _lib as L_

TM = 32  # trajectories per workgroup tile (csrc/lstm_tile.h GMT_TM)


def pad_rows(n: int) -> int:
    return (n + TM - 1) // TM * TM


# ----------------------------------------------------------------------------------------
# kinematic features (RecurrentGAN.py:315-389), time-major
# ----------------------------------------------------------------------------------------
def gate_perm(w: torch.Tensor) -> torch.Tensor:
    """[4H, ...] in torch's gate-major order (row g*H + unit) -> the unit-major order of the kernels'
    saved-gate buffers (row 4*unit + g); see csrc/lstm_tile.h."""
    H = w.shape[0] // 4
    return w.reshape(4, H, -1).transpose(0, 1).reshape(w.shape).contiguous()


def gate_unperm(w: torch.Tensor) -> torch.Tensor:
    """Inverse of gate_perm along dim 0."""
    H = w.shape[0] // 4
    return w.reshape(H, 4, -1).transpose(0, 1).reshape(w.shape).contiguous()


def wgrad_gates(dA: torch.Tensor, X: torch.Tensor, want_db: bool = False, x_drop: Tuple[float, int] = (0.0, 0)):
    """wgrad for a d(pre-activation) buffer (unit-major columns): results in torch's gate-major order."""
    dW, db = wgrad(dA, X, want_db, x_drop)
    return gate_unperm(dW), (None if db is None else gate_unperm(db))


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
        L_.call('gmt_pack_weight', src, out[d].data_ptr(), K, kv, 1 << 30, H // 32, 4, H, 1, w.shape[1], L_.stream())
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
        L_.call('gmt_pack_weight', src, out[d].data_ptr(), K, K, 1 << 30, n_out // 32, 1, 0, w.shape[1], 1, L_.stream())
    return out


# ----------------------------------------------------------------------------------------
# weight-gradient contraction over time (split-K MFMA kernel, csrc/wgrad.hip)
# ----------------------------------------------------------------------------------------
def wgrad(dA: torch.Tensor, X: torch.Tensor, want_db: bool = False, x_drop: Tuple[float, int] = (0.0, 0)):
    """dW = dA^T X over the rows (dA [R,N] contiguous, X [R,I] a row-strided view); optionally
    db = dA.sum(0).  Kernel shapes (csrc/wgrad.hip): N % 4 == 0 and either I % 128 == 0 with a
    16-byte aligned X, or a skinny X (I <= 32, any row pitch); R % 16 == 0.  Anything else goes
    to rocBLAS.  x_drop = (p, seed): X is read through the fused inter-layer dropout mask
    (counter r*ldx + i; wide X only - lstm_layer never asks for it otherwise)."""
    R, N = dA.shape
    I = X.shape[1]
    wide = I % 128 == 0 and X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0
    ok = (dA.is_cuda and N % 4 == 0 and R % 16 == 0 and R >= 16 and dA.is_contiguous() and dA.data_ptr() % 16 == 0
          and X.stride(1) == 1 and (wide or I <= 32) and dA.dtype == torch.float32 and X.dtype == torch.float32)
    if not ok:
        if x_drop[0] > 0:
            raise L_.GmtError('wgrad: the fused dropout mask needs the split-K kernel (I % 128 == 0, R % 16 == 0)')
        dW = torch.mm(dA.t(), X)
        return dW, (dA.sum(0) if want_db else None)
    import ctypes
    ks, need = ctypes.c_int(0), ctypes.c_longlong(0)
    L_.call('gmt_wgrad_workspace_floats', N, I, R, ctypes.addressof(ks), ctypes.addressof(need), label='skip')
    ws = torch.empty(need.value, device=dA.device, dtype=torch.float32)
    dW = torch.empty(N, I, device=dA.device, dtype=torch.float32)
    db = torch.empty(N, device=dA.device, dtype=torch.float32) if want_db else None
    L_.call('gmt_wgrad', dA.data_ptr(), N, X.data_ptr(), X.stride(0), R, N, I, dW.data_ptr(),
            None if db is None else db.data_ptr(), 0, ws.data_ptr(), need.value, float(x_drop[0]), int(x_drop[1]),
            L_.stream(), label=f'wgrad_N{N}_I{I}')
    return dW, db


# ----------------------------------------------------------------------------------------
# library GEMMs: which library, per shape
# ----------------------------------------------------------------------------------------
_BLAS_CHOICE: dict = {}


def _time_mm(a, b, out, reps=3):
    torch.mm(a, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.mm(a, b, out=out)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)


def mm_library(a: torch.Tensor, b: torch.Tensor):
    """Which BLAS library torch should use for a @ b of this shape: rocBLAS and hipBLASLt each win some
    of the tall-skinny fp32 shapes of this step (measured on MI355X, [528 384 x 512] x [512 x N]: N = 256
    rocBLAS 121 vs 113 TFLOP/s, N = 128 hipBLASLt 114 vs 65).  Timed once per (M, N, K) and cached;
    GMT_BLAS_AUTOTUNE=0 keeps torch's default."""
    import os
    key = (a.shape[0], b.shape[1], a.shape[1], a.device.index)
    lib = _BLAS_CHOICE.get(key)
    if lib is None:
        lib = 'default'
        if os.environ.get('GMT_BLAS_AUTOTUNE', '1') != '0' and a.shape[0] >= 65536:
            cur = torch.backends.cuda.preferred_blas_library()
            out = torch.empty(a.shape[0], b.shape[1], device=a.device, dtype=a.dtype)
            best = None
            for cand in ('cublaslt', 'cublas'):
                try:
                    torch.backends.cuda.preferred_blas_library(cand)
                    ms = _time_mm(a, b, out)
                except RuntimeError:
                    continue
                if best is None or ms < best[0]:
                    best = (ms, cand)
            torch.backends.cuda.preferred_blas_library(cur)
            if best is not None:
                lib = best[1]
        _BLAS_CHOICE[key] = lib
    return lib


class _use_blas:
    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        if self.lib != 'default':
            self.prev = torch.backends.cuda.preferred_blas_library()
            torch.backends.cuda.preferred_blas_library(self.lib)

    def __exit__(self, *exc):
        if self.lib != 'default':
            torch.backends.cuda.preferred_blas_library(self.prev)


def dx_from_da(da: torch.Tensor, w_ih: List[torch.Tensor], rows: int) -> torch.Tensor:
    """dx [rows, I] = sum over directions of da[d] [rows, 4H] @ W_ih[d] (4H axis in the buffers' unit-major
    order): the one large library GEMM of a layer's backward."""
    ws = [gate_perm(w.detach()) for w in w_ih]
    a0 = da[0].view(rows, -1)
    with _use_blas(mm_library(a0, ws[0])):
        dx = torch.mm(a0, ws[0])
        for d in range(1, len(ws)):
            dx.addmm_(da[d].view(rows, -1), ws[d])
    return dx


class LinearFn(torch.autograd.Function):
    """y = x W^T + b over the rows of x [..., I] (the per-timestep Linear layers of the feature
    encoder, RG.py:713-718, applied to L*B rows).  Forward and dx are library GEMMs; the weight and
    bias gradients - a [N x R] x [R x I] contraction with R ~ 5e5 - run on the split-K kernel."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        with _use_blas(mm_library(x.reshape(-1, x.shape[-1]), w.detach().t())):
            return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            with _use_blas(mm_library(dy2, w)):
                dx = (dy2 @ w).view(x.shape)
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            dW, db = wgrad(dy2, x2, want_db=ctx.has_b)
        return dx, dW, db


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    return LinearFn.apply(x, w, b)


# ----------------------------------------------------------------------------------------
# packed bidirectional LSTM layer (RecurrentGAN.py:720-727, 764-778)
# ----------------------------------------------------------------------------------------
class LSTMLayerFn(torch.autograd.Function):
    """x [L,B,I] (B % 32 == 0), lengths int32 [B] -> (h_all [L,B,nd*H], h_final [B,nd*H]).
    weights: per direction (w_ih [4H,I], w_hh [4H,H], b_ih [4H], b_hh [4H]).
    in_drop / out_drop = (p, seed): nn.LSTM's inter-layer dropout (RG.py:725) without a mask tensor.
    The layer with in_drop reads its input through the mask (forward and dW_ih) and returns dx
    w.r.t. the DROPPED input; the layer below it is given the same pair as out_drop and applies
    the mask to that adjoint.  The two must be used together (models.Discriminator does)."""

    @staticmethod
    def forward(ctx, x, lengths, fuse_x, need_h_all, in_drop, out_drop, *weights):
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
        if in_drop[0] > 0 and not (fuse_x and I % 128 == 0 and (L * B) % 16 == 0):
            raise L_.GmtError('fused input dropout needs the fused input contraction and I % 128 == 0')
        if fuse_x:
            w_ih_pk = pack_forward(w_ih, H, I)
            L_.call('gmt_lstm_layer_fwd', L_.ptr(x), None, L_.ptr(w_ih_pk), L_.ptr(w_hh_pk), L_.ptr(bias),
                    L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                    L, B, I, H, nd, float(in_drop[0]), int(in_drop[1]), L_.stream(), label=f'lstm_fwd_fused_I{I // H}')
        else:
            x2 = x.view(L * B, I)
            for d in range(nd):
                torch.addmm(gate_perm(bias[d]), x2, gate_perm(w_ih[d].detach()).t(), out=gates[d].view(L * B, 4 * H))
            L_.call('gmt_lstm_layer_fwd', None, L_.ptr(gates), None, L_.ptr(w_hh_pk), None,
                    L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                    L, B, I, H, nd, 0.0, 0, L_.stream(), label='lstm_fwd_pregates')
        ctx.save_for_backward(x, lengths, gates, c_all, h_all, *weights)
        ctx.dims = (L, B, I, H, nd)
        ctx.drops = (in_drop, out_drop)
        ctx.set_materialize_grads(False)
        return h_all, h_final

    @staticmethod
    def backward(ctx, dh_all, dh_final):
        x, lengths, gates, c_all, h_all, *weights = ctx.saved_tensors
        L, B, I, H, nd = ctx.dims
        in_drop, out_drop = ctx.drops
        w_ih = [weights[4 * d] for d in range(nd)]
        w_hh = [weights[4 * d + 1] for d in range(nd)]
        if dh_all is not None:
            dh_all = dh_all.contiguous()
        if dh_final is not None:
            dh_final = dh_final.contiguous()
        w_hh_pkT = pack_transposed(w_hh, H, 4 * H)
        # `gates` is overwritten in place with d(pre-activation): the graph can be back-propagated once
        L_.call('gmt_lstm_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(w_hh_pkT), L_.ptr(lengths),
                L_.ptr(dh_all), L_.ptr(dh_final), None, None, None, None, None, L, B, H, nd,
                float(out_drop[0]), int(out_drop[1]), L_.stream(), label='lstm_bwd')
        da = gates
        x2 = x.view(L * B, I)
        grads: List[Optional[torch.Tensor]] = []
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dx_from_da(da, w_ih, L * B).view(L, B, I)
        need_w = any(ctx.needs_input_grad[6:])
        if need_w:
            grads = _lstm_weight_grads(da, x2, h_all, w_hh, L, B, H, nd, in_drop)
        else:
            grads = [None] * (4 * nd)
        return (dx, None, None, None, None, None, *grads)


def lstm_layer(x, lengths_i32, weights, fuse_x: bool = True, need_h_all: bool = True,
               in_drop: Tuple[float, int] = (0.0, 0), out_drop: Tuple[float, int] = (0.0, 0)):
    return LSTMLayerFn.apply(x, lengths_i32, fuse_x, need_h_all, in_drop, out_drop, *weights)


def _lstm_weight_grads(da, x2, h_all, w_hh, L, B, H, nd, x_drop=(0.0, 0)):
    """dW_ih, dW_hh, db per direction from d(pre-activation) [nd,L,B,4H] (one contraction each)."""
    grads = []
    for d in range(nd):
        da2 = da[d].view(L * B, 4 * H)
        d_wih, db = wgrad_gates(da2, x2, want_db=True, x_drop=x_drop)
        hv = h_all.view(L * B, nd * H)[:, d * H:(d + 1) * H]
        if L > 1:
            # forward direction: h_prev(t) = h[t-1]; reverse: h_prev(t) = h[t+1] (zero beyond the row's length)
            d_whh, _ = wgrad_gates(da2[B:], hv[:-B]) if d == 0 else wgrad_gates(da2[:-B], hv[B:])
        else:
            d_whh = torch.zeros_like(w_hh[d])
        grads += [d_wih, d_whh, db, db]
    return grads


class LSTMLayerBwdGPFn(torch.autograd.Function):
    """The first backward sweep of a layer as a differentiable op (gradient penalty only):
    (dh_all, dh_final, gates, c_all, w_ih.., w_hh..) -> dx.  Its backward is the second-order
    kernel gmt_lstm_layer_bwd2."""

    @staticmethod
    def forward(ctx, dh_all, dh_final, gates, c_all, lengths, I, *w):
        nd = len(w) // 2
        w_ih, w_hh = list(w[:nd]), list(w[nd:])
        _, L, B, H4 = gates.shape
        H = H4 // 4
        dev = gates.device
        if dh_all is not None:
            dh_all = dh_all.contiguous()
        if dh_final is not None:
            dh_final = dh_final.contiguous()
        da = torch.empty_like(gates)
        dct = torch.empty(nd, L, B, H, device=dev, dtype=torch.float32)
        dht = torch.empty(nd, L, B, H, device=dev, dtype=torch.float32)
        w_hh_pkT = pack_transposed(w_hh, H, 4 * H)
        L_.call('gmt_lstm_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(w_hh_pkT), L_.ptr(lengths),
                L_.ptr(dh_all), L_.ptr(dh_final), None, None, L_.ptr(da), L_.ptr(dct), L_.ptr(dht),
                L, B, H, nd, 0.0, 0, L_.stream(), label='lstm_bwd_gp')
        dx = dx_from_da(da, w_ih, L * B)
        ctx.save_for_backward(gates, c_all, lengths, da, dct, dht, *w)
        ctx.dims = (L, B, I, H, nd)
        ctx.has = (dh_all is not None, dh_final is not None)
        return dx.view(L, B, I)

    @staticmethod
    def backward(ctx, xhat):
        gates, c_all, lengths, da, dct, dht, *w = ctx.saved_tensors
        L, B, I, H, nd = ctx.dims
        w_ih, w_hh = list(w[:nd]), list(w[nd:])
        dev = xhat.device
        xhat = xhat.contiguous()
        hhat_all = torch.empty(L, B, nd * H, device=dev, dtype=torch.float32)
        hhat_final = torch.empty(B, nd * H, device=dev, dtype=torch.float32)
        g_gates = torch.empty_like(gates)
        g_c = torch.empty_like(c_all)
        w_ih_pk = pack_forward(w_ih, H, I)
        w_hh_pk = pack_forward(w_hh, H, H)
        L_.call('gmt_lstm_layer_bwd2', L_.ptr(xhat), L_.ptr(w_ih_pk), L_.ptr(w_hh_pk), L_.ptr(lengths),
                L_.ptr(gates), L_.ptr(c_all), L_.ptr(dct), L_.ptr(dht), L_.ptr(hhat_all), L_.ptr(hhat_final),
                L_.ptr(g_gates), L_.ptr(g_c), L, B, I, H, nd, L_.stream(), label='lstm_bwd2')
        xh2 = xhat.view(L * B, I)
        g_wih, g_whh = [], []
        for d in range(nd):
            da2 = da[d].view(L * B, 4 * H)
            g_wih.append(wgrad_gates(da2, xh2)[0])
            hv = hhat_all.view(L * B, nd * H)[:, d * H:(d + 1) * H]
            if L > 1:
                g_whh.append(wgrad_gates(da2[B:], hv[:-B])[0] if d == 0 else wgrad_gates(da2[:-B], hv[B:])[0])
            else:
                g_whh.append(torch.zeros_like(w_hh[d]))
        return (hhat_all if ctx.has[0] else None, hhat_final if ctx.has[1] else None, g_gates, g_c, None, None,
                *g_wih, *g_whh)


class LSTMLayerGPFn(torch.autograd.Function):
    """Twice-differentiable variant of LSTMLayerFn for the gradient penalty: the saved
    activations are outputs, so that the adjoints the second-order sweep produces for them
    flow back into this layer's (ordinary) backward kernel as extra adjoints."""

    @staticmethod
    def forward(ctx, x, lengths, *weights):
        nd = len(weights) // 4
        w_ih = [weights[4 * d] for d in range(nd)]
        w_hh = [weights[4 * d + 1] for d in range(nd)]
        H = w_hh[0].shape[1]
        L, B, I = x.shape
        assert B % TM == 0
        x = x.contiguous()
        dev = x.device
        bias = torch.stack([weights[4 * d + 2].detach() + weights[4 * d + 3].detach() for d in range(nd)]).contiguous()
        gates = torch.empty(nd, L, B, 4 * H, device=dev, dtype=torch.float32)
        c_all = torch.empty(nd, L, B, H, device=dev, dtype=torch.float32)
        h_all = torch.empty(L, B, nd * H, device=dev, dtype=torch.float32)
        h_final = torch.empty(B, nd * H, device=dev, dtype=torch.float32)
        w_ih_pk = pack_forward(w_ih, H, I)      # keep both alive until the launch is enqueued
        w_hh_pk = pack_forward(w_hh, H, H)
        L_.call('gmt_lstm_layer_fwd', L_.ptr(x), None, L_.ptr(w_ih_pk), L_.ptr(w_hh_pk),
                L_.ptr(bias), L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                L, B, I, H, nd, 0.0, 0, L_.stream(), label=f'lstm_fwd_gp_I{I // H}')
        ctx.save_for_backward(x, lengths, gates, c_all, h_all, *weights)
        ctx.dims = (L, B, I, H, nd)
        ctx.set_materialize_grads(False)
        return h_all, h_final, gates, c_all

    @staticmethod
    def backward(ctx, dh_all, dh_final, dgates, dc):
        x, lengths, gates, c_all, h_all, *weights = ctx.saved_tensors
        L, B, I, H, nd = ctx.dims
        w_ih = [weights[4 * d] for d in range(nd)]
        w_hh = [weights[4 * d + 1] for d in range(nd)]
        if torch.is_grad_enabled():
            # first backward under create_graph=True: only the input gradient is produced (this is
            # what torch.autograd.grad(inputs=interpolated) of the penalty asks for), as a
            # differentiable op
            dx = LSTMLayerBwdGPFn.apply(dh_all, dh_final, gates, c_all, lengths, I, *w_ih, *w_hh)
            return (dx, None) + (None,) * len(weights)
        cont = lambda t: None if t is None else t.contiguous()
        dh_all, dh_final, dgates, dc = cont(dh_all), cont(dh_final), cont(dgates), cont(dc)
        da = torch.empty_like(gates)
        w_hh_pkT = pack_transposed(w_hh, H, 4 * H)
        L_.call('gmt_lstm_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(w_hh_pkT), L_.ptr(lengths),
                L_.ptr(dh_all), L_.ptr(dh_final), L_.ptr(dgates), L_.ptr(dc), L_.ptr(da), None, None,
                L, B, H, nd, 0.0, 0, L_.stream(), label='lstm_bwd_gp2')
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dx_from_da(da, w_ih, L * B).view(L, B, I)
        grads = _lstm_weight_grads(da, x.view(L * B, I), h_all, w_hh, L, B, H, nd)
        return (dx, None, *grads)


# ----------------------------------------------------------------------------------------
# spectral normalisation of the critic's Linear weights (csrc/spectral_norm.hip)
# ----------------------------------------------------------------------------------------
class SpectralNormFn(torch.autograd.Function):
    """(training, eps, n, W_1..W_n, u_1..u_n, v_1..v_n) -> (W_1/sigma_1, ..., W_n/sigma_n): what the
    torch.nn.utils.spectral_norm hooks of RG.py:713-736 compute at every forward, all layers of a critic
    call in one launch (one more for the backward).  In training mode the u / v buffers advance by one
    power iteration, in place (they are buffers: autograd never sees them)."""

    @staticmethod
    def forward(ctx, training: bool, eps: float, n: int, *t):
        ws, us, vs = t[:n], t[n:2 * n], t[2 * n:3 * n]
        L_.require_cuda(*ws, *us, *vs)
        assert 1 <= n <= L_.SN_MAX_LAYERS
        dev = ws[0].device
        outs = [torch.empty_like(w, memory_format=torch.contiguous_format) for w in ws]
        tot = sum(w.shape[0] + w.shape[1] for w in ws)
        saved = torch.empty(tot + n, device=dev, dtype=torch.float32)      # per layer: u, v; then the n sigmas
        tab = (L_.SnFwdLayer * n)()
        offs, o = [], 0
        for k, (w, u, v) in enumerate(zip(ws, us, vs)):
            assert w.dim() == 2 and w.is_contiguous() and u.is_contiguous() and v.is_contiguous()
            assert u.numel() == w.shape[0] and v.numel() == w.shape[1]
            tab[k] = L_.SnFwdLayer(w.data_ptr(), u.data_ptr(), v.data_ptr(), outs[k].data_ptr(),
                                   saved.data_ptr() + 4 * (tot + k), saved.data_ptr() + 4 * o, w.shape[0], w.shape[1])
            offs.append(o)
            o += w.shape[0] + w.shape[1]
        L_.call('gmt_spectral_norm_fwd', ctypes_byref_array(tab), n, float(eps), int(bool(training)), L_.stream())
        ctx.save_for_backward(saved, *ws)
        ctx.meta = (n, offs, tot)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        saved, *ws = ctx.saved_tensors
        n, offs, tot = ctx.meta
        live = [k for k in range(n) if gs[k] is not None and ctx.needs_input_grad[3 + k]]
        grads: List[Optional[torch.Tensor]] = [None] * n
        if live:
            tab = (L_.SnBwdLayer * len(live))()
            keep = []
            for q, k in enumerate(live):
                g = gs[k].contiguous()
                keep.append(g)
                grads[k] = torch.empty_like(ws[k], memory_format=torch.contiguous_format)
                tab[q] = L_.SnBwdLayer(g.data_ptr(), ws[k].data_ptr(), saved.data_ptr() + 4 * offs[k],
                                       saved.data_ptr() + 4 * (tot + k), grads[k].data_ptr(), ws[k].shape[0], ws[k].shape[1])
            L_.call('gmt_spectral_norm_bwd', ctypes_byref_array(tab), len(live), L_.stream())
        return (None, None, None, *grads, *([None] * (2 * n)))


def spectral_norm_weights(ws, us, vs, training: bool, eps: float = 1e-12):
    return SpectralNormFn.apply(training, eps, len(ws), *ws, *us, *vs)


def ctypes_byref_array(arr):
    import ctypes
    return ctypes.cast(arr, ctypes.c_void_p)


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


# ----------------------------------------------------------------------------------------
# generator rollout (RecurrentGAN.py:508-604)
# ----------------------------------------------------------------------------------------
def _pack1(w: torch.Tensor, K: int, k_valid: int, n_valid: int, n_waves: int, n_gates: int, gate_stride: int,
           stride_k: int, stride_n: int) -> torch.Tensor:
    w = w.detach().contiguous()
    out = torch.empty(n_waves * n_gates * 32 * K, device=w.device, dtype=torch.float32)
    L_.call('gmt_pack_weight', w.data_ptr(), out.data_ptr(), K, k_valid, n_valid, n_waves, n_gates, gate_stride,
            stride_k, stride_n, L_.stream())
    return out


class GenRolloutFn(torch.autograd.Function):
    """Inputs (all fp32, cuda):
      g0 [B,4H] (unit-major columns 4*unit + gate), w_u [4H,7], w_hh0 [4H,H], w_ih1 [4H,H], w_hh1 [4H,H], bias1 [4H],
      w1 [H/2,H], b1 [H/2], w2 [3,H/2], b2 [3], init_in [3], starts [B,2], ends [B,2],
      target [T,B,3] or None, tf_mask uint8 [T] (device) or None, T.
    Output: out_seq [T,B,3].  B % 32 == 0."""

    @staticmethod
    def forward(ctx, g0, w_u, w_hh0, w_ih1, w_hh1, bias1, w1, b1, w2, b2, init_in, starts, ends, target, tf_mask, T,
                save_for_bwd=True, dropout_p=0.0, seed=0, stop_distance=0.0, rows_valid=0, lengths_out=None):
        L_.require_cuda(g0)
        B, H4 = g0.shape
        H = H4 // 4
        HH = H // 2
        NW = H // 32
        dev = g0.device
        assert B % TM == 0
        need_grad = any(ctx.needs_input_grad) and save_for_bwd
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        a = L_.GenFwdArgs()
        keep = []

        def P(t):
            if t is None:
                return None
            t = t.detach().contiguous()
            keep.append(t)
            return t.data_ptr()

        a.g0 = P(g0)
        a.w_u_pk = P(_pack1(w_u, 8, 7, 1 << 30, NW, 4, H, 1, 7))
        a.w_hh0_pk = P(_pack1(w_hh0, H, H, 1 << 30, NW, 4, H, 1, H))
        a.w_ih1_pk = P(_pack1(w_ih1, H, H, 1 << 30, NW, 4, H, 1, H))
        a.w_hh1_pk = P(_pack1(w_hh1, H, H, 1 << 30, NW, 4, H, 1, H))
        a.bias1 = P(bias1)
        a.w1_pk = P(_pack1(w1, H, H, HH, max(1, H // 64), 1, 0, 1, H))
        a.b1 = P(b1); a.w2 = P(w2); a.b2 = P(b2)
        a.starts = P(starts); a.ends = P(ends); a.init_in = P(init_in.reshape(3))
        a.target = P(target); a.tf_mask = P(tf_mask)
        out_seq = f(T, B, 3)
        a.out_seq = out_seq.data_ptr()
        saved = None
        if need_grad:
            saved = dict(gates0=f(T, B, 4 * H), c0=f(T, B, H), h0=f(T, B, H), gates1=f(T, B, 4 * H), c1=f(T, B, H),
                         h1=f(T, B, H), u=f(T, B, 8), y=f(T, B, HH), oraw=f(T, B))
            a.gates0 = saved['gates0'].data_ptr(); a.c0_all = saved['c0'].data_ptr(); a.h0_all = saved['h0'].data_ptr()
            a.gates1 = saved['gates1'].data_ptr(); a.c1_all = saved['c1'].data_ptr(); a.h1_all = saved['h1'].data_ptr()
            a.u_all = saved['u'].data_ptr(); a.y_all = saved['y'].data_ptr(); a.oraw_all = saved['oraw'].data_ptr()
        if need_grad and dropout_p > 0:
            saved['h0d'] = f(T, B, H)
            a.h0d_all = saved['h0d'].data_ptr()
        a.T, a.B, a.H = T, B, H
        a.dropout_p, a.seed = float(dropout_p), int(seed) & 0xFFFFFFFF
        if stop_distance > 0:                      # sampling with per-row early stop (no-grad rollout only)
            assert not need_grad and target is None and lengths_out.dtype == torch.int32 and lengths_out.numel() == B
            a.stop_distance, a.rows_valid, a.lengths_out = float(stop_distance), int(rows_valid), lengths_out.data_ptr()
        ctx.drop = (float(dropout_p), int(seed) & 0xFFFFFFFF)
        L_.call('gmt_gen_rollout_fwd', ctypes_byref(a), L_.stream(),
                label='gen_fwd_train' if need_grad else 'gen_fwd_nograd')
        if need_grad:
            ctx.save_for_backward(w_u, w_hh0, w_ih1, w_hh1, w1, w2, tf_mask if tf_mask is not None else torch.empty(0),
                                  *saved.values())
            ctx.dims = (T, B, H)
            ctx.has_tf = tf_mask is not None
            ctx.init_shape = tuple(init_in.shape)
        return out_seq

    @staticmethod
    def backward(ctx, d_out):
        (w_u, w_hh0, w_ih1, w_hh1, w1, w2, tf_mask, gates0, c0, h0, gates1, c1, h1, u, y, oraw, *rest) = ctx.saved_tensors
        h0d = rest[0] if rest else h0                  # layer 1's input (after dropout when it is on)
        T, B, H = ctx.dims
        HH, NW = H // 2, H // 32
        dev = d_out.device
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        d_out = d_out.contiguous()
        a = L_.GenBwdArgs()
        w1T = _pack1(w1, HH, HH, 1 << 30, NW, 1, 0, H, 1)
        wl1 = torch.cat([w_hh1.detach(), w_ih1.detach()], dim=1).contiguous()        # [4H, 2H]
        wl1T = _pack1(wl1, 4 * H, 4 * H, 1 << 30, NW, 2, H, 2 * H, 1)
        whh0T = _pack1(w_hh0, 4 * H, 4 * H, 1 << 30, NW, 1, 0, H, 1)
        w_uP = torch.zeros(4 * H, 8, device=dev, dtype=torch.float32)
        w_uP[:, :7] = w_u.detach()
        w2c = w2.detach().contiguous()
        d_ypre = f(T, B, HH); d_o = f(T, B, 3); d_in0 = f(B, 3)
        a.gates0 = gates0.data_ptr(); a.c0_all = c0.data_ptr(); a.gates1 = gates1.data_ptr(); a.c1_all = c1.data_ptr()
        a.y_all = y.data_ptr(); a.oraw_all = oraw.data_ptr(); a.u_all = u.data_ptr(); a.d_out = d_out.data_ptr()
        a.w1T_pk = w1T.data_ptr(); a.w_l1T_pk = wl1T.data_ptr(); a.w_hh0T_pk = whh0T.data_ptr()
        a.w_uP = w_uP.data_ptr(); a.w2 = w2c.data_ptr()
        a.tf_mask = tf_mask.data_ptr() if ctx.has_tf else None
        a.d_ypre = d_ypre.data_ptr(); a.d_o = d_o.data_ptr(); a.d_in0 = d_in0.data_ptr()
        a.T, a.B, a.H = T, B, H
        a.dropout_p, a.seed = ctx.drop
        L_.call('gmt_gen_rollout_bwd', ctypes_byref(a), L_.stream())
        da0 = gates0.view(T * B, 4 * H)
        da1 = gates1.view(T * B, 4 * H)
        h0v, h1v = h0.view(T * B, H), h1.view(T * B, H)
        d_g0 = gates0.sum(0)                                            # time-invariant input: sum over t (unit-major, like g0)
        d_w_u = wgrad_gates(da0, u.view(T * B, 8))[0][:, :7]
        d_w_hh0 = wgrad_gates(da0[B:], h0v[:-B])[0] if T > 1 else torch.zeros_like(w_hh0)
        d_w_ih1, d_bias1 = wgrad_gates(da1, h0d.view(T * B, H), want_db=True)
        d_w_hh1 = wgrad_gates(da1[B:], h1v[:-B])[0] if T > 1 else torch.zeros_like(w_hh1)
        dyp = d_ypre.view(T * B, HH)
        d_w1, d_b1 = wgrad(dyp, h1v, want_db=True)
        dov = d_o.view(T * B, 3)
        d_w2 = wgrad(y.view(T * B, HH), dov)[0].t()                    # [3,HH] via its transpose: N = HH, skinny X = d_o
        d_b2 = dov.sum(0)
        d_init = d_in0.sum(0).reshape(ctx.init_shape)
        return (d_g0, d_w_u, d_w_hh0, d_w_ih1, d_w_hh1, d_bias1, d_w1, d_b1, d_w2, d_b2, d_init,
                None, None, None, None, None, None, None, None, None, None, None)


def ctypes_byref(a):
    import ctypes
    return ctypes.cast(ctypes.pointer(a), ctypes.c_void_p)


def dropout_keep_mask(T: int, B: int, H: int, p: float, seed: int, pair16: bool = False) -> torch.Tensor:
    """Host replica of the kernels' counter-based dropout masks (csrc/lstm_tile.h): [T,B,H] float mask
    of 0 / 1/(1-p).  pair16=False: gmt_keep (generator rollout, one 24-bit draw per element);
    pair16=True: gmt_keep16 (critic inter-layer mask, two 16-bit draws per hash).  Used by the tests
    to feed the oracle the same mask."""
    import numpy as np
    u = np.uint64
    idx = np.arange(T * B * H, dtype=np.uint64)
    ctr = idx >> u(1) if pair16 else idx
    x = (ctr + (u(seed & 0xFFFFFFFF) * u(0x9E3779B1))) & u(0xFFFFFFFF)
    x = (x * u(747796405) + u(2891336453)) & u(0xFFFFFFFF)
    w = (((x >> ((x >> u(28)) + u(4))) ^ x) * u(277803737)) & u(0xFFFFFFFF)
    h = ((w >> u(22)) ^ w) & u(0xFFFFFFFF)
    if pair16:
        half = np.where((idx & u(1)) == u(1), h >> u(16), h & u(0xFFFF))
        keep = half >= u(int(np.float32(p) * np.float32(65536.0)))
    else:
        keep = (h >> u(8)) >= u(int(np.float32(p) * np.float32(16777216.0)))
    return torch.from_numpy((keep.astype(np.float32) / np.float32(1.0 - p)).reshape(T, B, H))
