"""torch.autograd glue around the HIP kernels of libgmt_hip.so.

Every Function here is a thin host wrapper: it allocates the output / saved buffers with
torch (PyTorch owns device memory and streams), calls the C ABI on the current stream and
returns.  Every contraction of a train step - recurrent, input-gradient (csrc/dx_gemm.hip), weight-gradient
over K = L*B rows (csrc/wgrad.hip), hoisted input projection (csrc/xproj.hip) - is hand-written HIP, and so is
everything element-wise or reducing on the hot path; torch.mm is the fallback for shapes outside the kernels'
tables only.  GPU only - there is no CPU path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L_

TM = 32  # trajectories per workgroup tile (csrc/lstm_tile.h GMT_TM)


def pad_rows(n: int, tile: int = TM) -> int:
    return (n + tile - 1) // tile * tile


def row_tile(config) -> int:
    """Rows every batch is padded to a multiple of: 32 for the fp32 kernels, 256 for the bf16 family (ops_bf16.ROW_TILE)."""
    return 256 if getattr(config, 'compute_dtype', 'fp32') == 'bf16' else TM


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
    """wgrad for a d(pre-activation) buffer (unit-major columns): results in torch's gate-major order - straight from the
    kernel's slab reduction when the shape allows (gmt_wgrad_gates), else by a permutation pass."""
    N = dA.shape[1]
    if N % 4 == 0 and X.shape[1] % 4 == 0:
        out = wgrad(dA, X, want_db, x_drop, unit_major_h=N // 4)
        if out is not None:
            return out
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
        L_.call('gmt_kinematics_fwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(feat), L, B, eps, L_.stream(), nbytes=48.0 * L * B)
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
                L, B, ctx.eps, L_.stream(), nbytes=60.0 * L * B)
        return gpos, gdts, None


def kinematics_tm(pos: torch.Tensor, dts: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    return KinematicsFn.apply(pos, dts, eps)


# ----------------------------------------------------------------------------------------
# derived-weight cache: packed / permuted / summed copies of the parameters are rebuilt once per
# optimizer step, not once per call (include/gmt_hip.h: "Run once per optimizer step per weight")
# ----------------------------------------------------------------------------------------
_WEIGHTS_EPOCH = 0
_DERIVED: dict = {}


def weights_changed(params=None) -> None:
    """Called by whoever writes parameters behind autograd's back (the fused Adam kernel, flat-buffer re-homing,
    checkpoint loads): cached derived weights of `params` (default: of every parameter) are stale from here on.
    Selective, so that an optimizer step of one network does not throw away the other network's packs."""
    global _WEIGHTS_EPOCH
    if params is None:
        _WEIGHTS_EPOCH += 1
        _DERIVED.clear()
        return
    ids = {id(p) for p in params}
    for key in [k for k in _DERIVED if not ids.isdisjoint(k[1:])]:
        del _DERIVED[key]


def _record_stream(value, stream) -> None:
    if isinstance(value, torch.Tensor):
        if value.is_cuda:
            value.record_stream(stream)
    elif isinstance(value, (tuple, list)):
        for v in value:
            _record_stream(v, stream)


def derived(kind: str, tensors, build):
    """build() -> derived tensor(s) of the PARAMETERS `tensors`, cached until those change.  An entry is valid only
    for the very same tensor objects (weak references: a new tensor that happens to get a dead one's address or
    id never matches), at the same autograd version (in-place torch updates bump it) and the same weights epoch
    (the fused Adam kernel and the load paths, which write behind autograd's back, call weights_changed()).
    Never pass temporaries (e.g. spectrally normalised weights, which are new tensors every call).
    An entry remembers the stream build() ran on (an event recorded right after it): the first hit from ANOTHER stream
    makes that stream wait for the event, so two streams may share the cache in either order (the critic's main-stream
    calls and the gradient-penalty pass on the side stream; Generator.prepack still builds the rollout's packs up front)."""
    import weakref
    key = (kind,) + tuple(id(t) for t in tensors)
    hit = _DERIVED.get(key)
    cur = torch.cuda.current_stream() if tensors[0].is_cuda else None
    if hit is not None:
        refs, versions, epoch, value, built = hit
        if epoch == _WEIGHTS_EPOCH and all(r() is t for r, t in zip(refs, tensors)) \
                and versions == tuple(t._version for t in tensors):
            if built is not None and cur is not None and cur.cuda_stream not in built[1]:
                # first use on another stream than the one build() ran on: ordered behind the build, once
                cur.wait_event(built[0])
                built[1].add(cur.cuda_stream)
                _record_stream(value, cur)           # (the allocator must not hand the block back to the building stream early)
            return value
    if len(_DERIVED) > 512:                          # parameters updated by plain torch optimizers: bounded
        _DERIVED.clear()
    with torch.no_grad():
        value = build()
    built = None
    if cur is not None:
        ev = torch.cuda.Event()
        ev.record(cur)
        built = (ev, {cur.cuda_stream})
    _DERIVED[key] = (tuple(weakref.ref(t) for t in tensors), tuple(t._version for t in tensors), _WEIGHTS_EPOCH, value, built)
    return value


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
def wgrad(dA: torch.Tensor, X: torch.Tensor, want_db: bool = False, x_drop: Tuple[float, int] = (0.0, 0),
          unit_major_h: int = 0, x_padded: bool = False):
    """dW = dA^T X over the rows (dA [R,N] contiguous, X [R,I] a row-strided view); optionally
    db = dA.sum(0).  Kernel shapes (csrc/wgrad.hip): N % 4 == 0 and either I % 128 == 0 with a
    16-byte aligned X, or a skinny X (I <= 32, any row pitch); R % 16 == 0.  Anything else goes
    to rocBLAS.  x_drop = (p, seed): X is read through the fused inter-layer dropout mask
    (counter r*ldx + i; wide X only - lstm_layer never asks for it otherwise).
    x_padded: the caller vouches that X is a column slice of a buffer whose rows are readable and ZERO up to the next
    multiple of 128 columns (the saved activations of csrc/mlp.hip): 32 < I, I % 128 != 0 then runs on the wide tiles."""
    R, N = dA.shape
    I = X.shape[1]
    wide = X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0 and \
        (I % 128 == 0 or (x_padded and I > 32 and X.stride(0) >= (I + 127) // 128 * 128))
    ok = (dA.is_cuda and N % 4 == 0 and R % 16 == 0 and R >= 16 and dA.is_contiguous() and dA.data_ptr() % 16 == 0
          and X.stride(1) == 1 and (wide or I <= 32) and dA.dtype == torch.float32 and X.dtype == torch.float32)
    if not ok:
        if unit_major_h:
            return None                      # the caller permutes the fallback's result itself
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
    L_.call('gmt_wgrad_gates', dA.data_ptr(), N, X.data_ptr(), X.stride(0), R, N, I, dW.data_ptr(),
            None if db is None else db.data_ptr(), 0, ws.data_ptr(), need.value, float(x_drop[0]), int(x_drop[1]),
            int(unit_major_h), L_.stream(), label=f'wgrad_N{N}_I{I}', flops=2.0 * R * N * I)
    return dW, db


# ----------------------------------------------------------------------------------------
# input-gradient contraction (csrc/dx_gemm.hip)
# ----------------------------------------------------------------------------------------
def _dx_shape_ok(K: int, N: int) -> bool:
    return N in (32, 64, 128, 256, 512) and K in (N, 2 * N, 4 * N) and K <= 1024 and not (N == 512 and K > 1024)


def pack_dx_weight(w_kn: torch.Tensor) -> torch.Tensor:
    """w_kn [K, N] row-major = the B operand of dX = dA . W; fragment-major for gmt_dx_gemm."""
    K, N = w_kn.shape
    if L_.lib().gmt_dx_gemm_resident(K, N):           # the resident-weight kernel: one 32-column block per wave index
        return _pack1(w_kn, K, K, 1 << 30, N // 32, 1, 0, N, 1)
    nw = min(4, N // 32)
    ng = N // (32 * nw)
    return _pack1(w_kn, K, K, 1 << 30, nw, ng, 32 * nw, N, 1)


def dx_gemm(dA: torch.Tensor, w_pk: torch.Tensor, K: int, N: int, act: Optional[torch.Tensor] = None,
            slope: float = 0.2) -> torch.Tensor:
    """dA [nd, R, K] (or [R, K]) contiguous, w_pk [nd] packed by pack_dx_weight -> dX [R, N]
    (times LeakyReLU'(act) when act [R, N] is given)."""
    if dA.dim() == 2:
        dA = dA.unsqueeze(0)
    nd, R, _ = dA.shape
    dX = torch.empty(R, N, device=dA.device, dtype=torch.float32)
    L_.call('gmt_dx_gemm', L_.ptr(dA), L_.ptr(w_pk), L_.ptr(dX), L_.ptr(act), float(slope), R, K, N, nd, L_.stream(),
            label=f'dx_gemm_K{K}_N{N}' + ('_act' if act is not None else ''), flops=2.0 * nd * R * K * N)
    return dX


def dx_from_da(da: torch.Tensor, w_ih: List[torch.Tensor], rows: int, act: Optional[torch.Tensor] = None,
               slope: float = 0.2) -> torch.Tensor:
    """dx [rows, I] = sum over directions of da[d] [rows, 4H] @ W_ih[d] (4H axis in the buffers' unit-major
    order), optionally times LeakyReLU'(act): one gmt_dx_gemm launch, both directions accumulated in registers.
    Shapes outside the kernel's table (I not in {32 .. 512} or I not in {H, 2H, 4H}) use torch.mm."""
    K, I = w_ih[0].shape
    if _dx_shape_ok(K, I) and rows % TM == 0 and da.is_contiguous():
        w_pk = derived('dx_pk', w_ih, lambda: torch.stack([pack_dx_weight(gate_perm(w.detach())) for w in w_ih]))
        return dx_gemm(da.view(len(w_ih), rows, K), w_pk, K, I, act, slope)
    ws = derived('dx_perm', w_ih, lambda: [gate_perm(w.detach()) for w in w_ih])
    dx = torch.mm(da[0].view(rows, -1), ws[0])
    for d in range(1, len(ws)):
        dx.addmm_(da[d].view(rows, -1), ws[d])
    if act is not None:
        dx = dx * torch.where(act > 0, 1.0, slope)
    return dx


class LinearFn(torch.autograd.Function):
    """y = x W^T + b over the rows of x [..., I]: the per-timestep Linear layers on the twice-differentiable
    gradient-penalty path (composite.py; the ordinary critic call uses the fused encoder kernel).  Forward and
    dx are torch ops (so that autograd can differentiate the backward once more); the weight and bias gradients
    - a [N x R] x [R x I] contraction - run on the split-K kernel, and only when they are asked for."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
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
            dx = (dy2 @ w).view(x.shape)
        dW = db = None
        # under create_graph=True (the penalty's inner autograd.grad) only dx is asked for: no wasted contraction
        if not torch.is_grad_enabled() and (ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2])):
            dW, db = wgrad(dy2.detach(), x2.detach(), want_db=ctx.has_b)
        return dx, dW, db


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    return LinearFn.apply(x, w, b)


# ----------------------------------------------------------------------------------------
# packed bidirectional LSTM layer (RecurrentGAN.py:720-727, 764-778)
# ----------------------------------------------------------------------------------------
def lstm_bias(weights, nd: int) -> torch.Tensor:
    """[nd, 4H] b_ih + b_hh per direction (cached per optimizer step)."""
    bs = [weights[4 * d + k] for d in range(nd) for k in (2, 3)]
    return derived('lstm_bias', bs, lambda: torch.stack([weights[4 * d + 2].detach() + weights[4 * d + 3].detach()
                                                         for d in range(nd)]).contiguous())


class LSTMLayerFn(torch.autograd.Function):
    """x [L,B,I] (B % 32 == 0), lengths int32 [B] -> (h_all [L,B,nd*H], h_final [B,nd*H]).
    weights: per direction (w_ih [4H,I], w_hh [4H,H], b_ih [4H], b_hh [4H]) - nn.Parameters (their packed forms
    are cached per optimizer step).
    in_drop / out_drop = (p, seed): nn.LSTM's inter-layer dropout (RG.py:725) without a mask tensor.
    The layer with in_drop reads its input through the mask (forward and dW_ih) and returns dx
    w.r.t. the DROPPED input; the layer below it is given the same pair as out_drop and applies
    the mask to that adjoint.  The two must be used together (models.Discriminator does).
    x_act_slope >= 0: x is the output of a LeakyReLU with that slope (the feature encoder's, RG.py:717) and the
    gradient returned for x is already multiplied by LeakyReLU'(x), i.e. it is the gradient of the
    PRE-activation (the consumer, ops.EncoderFn, expects exactly that): one pass over [L*B, I] less."""

    @staticmethod
    def forward(ctx, x, lengths, fuse_x, need_h_all, in_drop, out_drop, x_act_slope, *weights):
        L_.require_cuda(x)
        nd = len(weights) // 4
        w_ih = [weights[4 * d] for d in range(nd)]
        w_hh = [weights[4 * d + 1] for d in range(nd)]
        H = w_hh[0].shape[1]
        L, B, I = x.shape
        assert B % TM == 0, 'pad the batch to a multiple of 32 rows'
        x = x.contiguous()
        dev = x.device
        bias = lstm_bias(weights, nd)
        w_hh_pk = derived('hh_pk', w_hh, lambda: pack_forward(w_hh, H, H))
        gates = torch.empty(nd, L, B, 4 * H, device=dev, dtype=torch.float32)
        c_all = torch.empty(nd, L, B, H, device=dev, dtype=torch.float32)
        h_all = torch.empty(L, B, nd * H, device=dev, dtype=torch.float32)
        h_final = torch.empty(B, nd * H, device=dev, dtype=torch.float32)
        if in_drop[0] > 0 and not (fuse_x and I % 128 == 0 and (L * B) % 16 == 0):
            raise L_.GmtError('fused input dropout needs the fused input contraction and I % 128 == 0')
        if fuse_x:
            w_ih_pk = derived('ih_pk', w_ih, lambda: pack_forward(w_ih, H, I))
            L_.call('gmt_lstm_layer_fwd', L_.ptr(x), None, L_.ptr(w_ih_pk), L_.ptr(w_hh_pk), L_.ptr(bias),
                    L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                    L, B, I, H, nd, float(in_drop[0]), int(in_drop[1]), L_.stream(), label=f'lstm_fwd_fused_I{I // H}',
                    flops=2.0 * nd * L * B * 4 * H * (I + H))
        else:
            x2 = x.view(L * B, I)
            for d in range(nd):
                torch.addmm(gate_perm(bias[d]), x2, gate_perm(w_ih[d].detach()).t(), out=gates[d].view(L * B, 4 * H))
            L_.call('gmt_lstm_layer_fwd', None, L_.ptr(gates), None, L_.ptr(w_hh_pk), None,
                    L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                    L, B, I, H, nd, 0.0, 0, L_.stream(), label='lstm_fwd_pregates', flops=2.0 * nd * L * B * 4 * H * H)
        ctx.save_for_backward(x, lengths, gates, c_all, h_all, *weights)
        ctx.dims = (L, B, I, H, nd)
        ctx.drops = (in_drop, out_drop)
        ctx.x_act_slope = x_act_slope
        ctx.consumed = False
        ctx.set_materialize_grads(False)
        return h_all, h_final

    @staticmethod
    def backward(ctx, dh_all, dh_final):
        if ctx.consumed:
            raise L_.GmtError('this LSTM layer was already back-propagated: its saved gate buffer now holds '
                              'd(pre-activation) (in-place backward); run the forward again instead of retain_graph')
        ctx.consumed = True
        x, lengths, gates, c_all, h_all, *weights = ctx.saved_tensors
        L, B, I, H, nd = ctx.dims
        in_drop, out_drop = ctx.drops
        w_ih = [weights[4 * d] for d in range(nd)]
        w_hh = [weights[4 * d + 1] for d in range(nd)]
        if dh_all is not None:
            dh_all = dh_all.contiguous()
        if dh_final is not None:
            dh_final = dh_final.contiguous()
        w_hh_pkT = derived('hh_pkT', w_hh, lambda: pack_transposed(w_hh, H, 4 * H))
        # `gates` is overwritten in place with d(pre-activation): the graph can be back-propagated once
        L_.call('gmt_lstm_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(w_hh_pkT), L_.ptr(lengths),
                L_.ptr(dh_all), L_.ptr(dh_final), None, None, None, None, None, L, B, H, nd,
                float(out_drop[0]), int(out_drop[1]), L_.stream(), label='lstm_bwd', flops=2.0 * nd * L * B * 4 * H * H)
        da = gates
        x2 = x.view(L * B, I)
        grads: List[Optional[torch.Tensor]] = []
        dx = None
        if ctx.needs_input_grad[0]:
            act = x2 if ctx.x_act_slope >= 0 else None
            dx = dx_from_da(da, w_ih, L * B, act, max(ctx.x_act_slope, 0.0)).view(L, B, I)
        need_w = any(ctx.needs_input_grad[7:])
        if need_w:
            grads = _lstm_weight_grads(da, x2, h_all, w_hh, L, B, H, nd, in_drop)
        else:
            grads = [None] * (4 * nd)
        return (dx, None, None, None, None, None, None, *grads)


def lstm_layer(x, lengths_i32, weights, fuse_x: bool = True, need_h_all: bool = True,
               in_drop: Tuple[float, int] = (0.0, 0), out_drop: Tuple[float, int] = (0.0, 0), x_act_slope: float = -1.0):
    return LSTMLayerFn.apply(x, lengths_i32, fuse_x, need_h_all, in_drop, out_drop, x_act_slope, *weights)


LSTM_STACK_FUSED_DX = True       # False: the two layers as separate nodes with a separate input-gradient contraction (A/B switch)


def lstm_stack2_supported(H: int, nd: int) -> bool:
    return LSTM_STACK_FUSED_DX and H == 128 and nd == 2


class LSTMStack2Fn(torch.autograd.Function):
    """The critic's two-layer bidirectional LSTM (RG.py:720-727 applied at RG.py:772) as ONE autograd node, H = 128:
    (x [L,B,H], lengths int32 [B], drop = (p, seed) of the inter-layer dropout, x_act_slope, layer-0 weights, layer-1
    weights) -> h_final [B, 2H] of the top layer (all the critic head reads, RG.py:774-781).
    Forward = the two persistent layer launches of LSTMLayerFn.  Backward: the top layer's sweep also contracts its INPUT
    gradient (gmt_lstm_layer_bwd_fx: dx_t = da_t W_ih inside the time loop, written as per-direction pairs), the bottom
    layer's sweep adds the pairs while it reads them - no separate dx contraction, no second pass over the top layer's
    d(pre-activation), and the element-wise phase of the top sweep amortised over three times the matrix work.  One node,
    because the pair buffer is not a gradient autograd could be handed."""

    @staticmethod
    def forward(ctx, x, lengths, drop, x_act_slope, *weights):
        L_.require_cuda(x)
        nd = 2
        w0, w1 = weights[:4 * nd], weights[4 * nd:]
        H = w0[1].shape[1]
        L, B, I = x.shape
        assert B % TM == 0 and I == H and len(w1) == 4 * nd
        x = x.contiguous()
        dev = x.device
        f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
        saved = []
        inp, I_l = x, I
        for l, w in enumerate((w0, w1)):
            w_ih = [w[4 * d] for d in range(nd)]
            w_hh = [w[4 * d + 1] for d in range(nd)]
            bias = lstm_bias(w, nd)
            w_hh_pk = derived('hh_pk', w_hh, lambda: pack_forward(w_hh, H, H))
            w_ih_pk = derived('ih_pk', w_ih, lambda: pack_forward(w_ih, H, I_l))
            gates, c_all, h_all, h_final = f(nd, L, B, 4 * H), f(nd, L, B, H), f(L, B, nd * H), f(B, nd * H)
            in_drop = drop if l == 1 else (0.0, 0)
            L_.call('gmt_lstm_layer_fwd', L_.ptr(inp), None, L_.ptr(w_ih_pk), L_.ptr(w_hh_pk), L_.ptr(bias),
                    L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                    L, B, I_l, H, nd, float(in_drop[0]), int(in_drop[1]), L_.stream(), label=f'lstm_fwd_fused_I{I_l // H}',
                    flops=2.0 * nd * L * B * 4 * H * (I_l + H))
            saved += [gates, c_all, h_all]
            inp, I_l = h_all, nd * H
        ctx.save_for_backward(x, lengths, *saved, *weights)
        ctx.dims = (L, B, H, nd)
        ctx.drop = drop
        ctx.x_act_slope = x_act_slope
        ctx.consumed = False
        return h_final

    @staticmethod
    def backward(ctx, dh_final):
        if ctx.consumed:
            raise L_.GmtError('this LSTM stack was already back-propagated: its saved gate buffers now hold '
                              'd(pre-activation) (in-place backward); run the forward again instead of retain_graph')
        if torch.is_grad_enabled():
            raise NotImplementedError('LSTMStack2Fn: first-order backward only (the penalty pass uses LSTMLayerGPFn)')
        ctx.consumed = True
        x, lengths, g0, c0, h0, g1, c1, h1, *weights = ctx.saved_tensors
        L, B, H, nd = ctx.dims
        drop = ctx.drop
        w0, w1 = weights[:4 * nd], weights[4 * nd:]
        w_ih0, w_hh0 = [w0[4 * d] for d in range(nd)], [w0[4 * d + 1] for d in range(nd)]
        w_ih1, w_hh1 = [w1[4 * d] for d in range(nd)], [w1[4 * d + 1] for d in range(nd)]
        R, I1 = L * B, nd * H
        dh_final = dh_final.contiguous()
        need_w = any(ctx.needs_input_grad[4:])
        # ---- top layer: sweep + input gradient in one launch
        w_hh1_pkT = derived('hh_pkT', w_hh1, lambda: pack_transposed(w_hh1, H, 4 * H))
        # W_ih as the B operand of dx = da W_ih with K = 4H in the LDS tile's gate-major order: the weight as it is
        w_ih1_fx = derived('ih_fx_pk', w_ih1, lambda: torch.stack(
            [_pack1(w, 4 * H, 4 * H, 1 << 30, 4, I1 // 128, 128, I1, 1) for w in w_ih1]))
        dxp = torch.empty(R, I1, 2, device=x.device, dtype=torch.float32)
        L_.call('gmt_lstm_layer_bwd_fx', L_.ptr(g1), L_.ptr(c1), L_.ptr(w_hh1_pkT), L_.ptr(lengths), None, L_.ptr(dh_final),
                L_.ptr(w_ih1_fx), L_.ptr(dxp), L, B, H, I1, nd, 0.0, 0, L_.stream(), label='lstm_bwd_fx',
                flops=2.0 * nd * R * 4 * H * (H + I1))
        grads1 = _lstm_weight_grads(g1, h0.view(R, I1), h1, w_hh1, L, B, H, nd, drop) if need_w else [None] * (4 * nd)
        # ---- bottom layer: reads the pairs (through the inter-layer dropout mask)
        w_hh0_pkT = derived('hh_pkT', w_hh0, lambda: pack_transposed(w_hh0, H, 4 * H))
        L_.call('gmt_lstm_layer_bwd_fx', L_.ptr(g0), L_.ptr(c0), L_.ptr(w_hh0_pkT), L_.ptr(lengths), L_.ptr(dxp), None,
                None, None, L, B, H, H, nd, float(drop[0]), int(drop[1]), L_.stream(), label='lstm_bwd',
                flops=2.0 * nd * R * 4 * H * H)
        x2 = x.view(R, H)
        dx = None
        if ctx.needs_input_grad[0]:
            act = x2 if ctx.x_act_slope >= 0 else None
            dx = dx_from_da(g0, w_ih0, R, act, max(ctx.x_act_slope, 0.0)).view(L, B, H)
        grads0 = _lstm_weight_grads(g0, x2, h0, w_hh0, L, B, H, nd) if need_w else [None] * (4 * nd)
        return (dx, None, None, None, *grads0, *grads1)


def lstm_stack2(x, lengths_i32, w0, w1, drop: Tuple[float, int] = (0.0, 0), x_act_slope: float = -1.0):
    return LSTMStack2Fn.apply(x, lengths_i32, drop, x_act_slope, *w0, *w1)


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


# The gradient-penalty sub-batch is a quarter of the batch (RG.py:969-989): its persistent kernels occupy a quarter of
# the CUs, so the time-parallel input contractions of that path are hoisted out of the time loops and run on the whole
# machine (csrc/xproj.hip); the full-batch path keeps them fused (same MFMA work either way, one pass less over x).
GP_HOIST_XPROJ = True


# ... and with 16-row tiles (csrc/lstm_gp16.hip) whenever 32-row tiles would leave CUs empty: twice the workgroups
GP_TILE16 = True


def _gp16(H: int, B: int, nd: int) -> bool:
    return (GP_TILE16 and GP_HOIST_XPROJ and (B // TM) * nd < 256 and bool(L_.lib().gmt_lstm16_supported(H, B)))


def _hh_raw(w_hh) -> torch.Tensor:
    """[nd, 4H, H]: the layer's recurrent weights as they are, stacked over directions (cached per optimizer step)."""
    return derived('hh_raw', w_hh, lambda: torch.stack([w.detach() for w in w_hh]).contiguous())


def lstm_xproj(x2: torch.Tensor, w_ih_pk: torch.Tensor, bias: Optional[torch.Tensor], nd: int, H: int, label: str) -> torch.Tensor:
    """x2 [R, I] -> [nd, R, 4H] = x W_ih^T (+ bias [nd, 4H]) in the gate buffers' unit-major column order."""
    R, I = x2.shape
    pre = torch.empty(nd, R, 4 * H, device=x2.device, dtype=torch.float32)
    L_.call('gmt_lstm_xproj', L_.ptr(x2), L_.ptr(w_ih_pk), L_.ptr(bias), L_.ptr(pre), R, I, H, nd, L_.stream(),
            label=label, flops=2.0 * nd * R * 4 * H * I)
    return pre


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
        if _gp16(H, B, nd):
            L_.call('gmt_lstm16_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(_hh_raw(w_hh)), L_.ptr(lengths),
                    L_.ptr(dh_all), L_.ptr(dh_final), None, None, L_.ptr(da), L_.ptr(dct), L_.ptr(dht),
                    L, B, H, nd, L_.stream(), label='lstm16_bwd_gp', flops=2.0 * nd * L * B * 4 * H * H)
        else:
            w_hh_pkT = derived('hh_pkT', w_hh, lambda: pack_transposed(w_hh, H, 4 * H))
            L_.call('gmt_lstm_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(w_hh_pkT), L_.ptr(lengths),
                    L_.ptr(dh_all), L_.ptr(dh_final), None, None, L_.ptr(da), L_.ptr(dct), L_.ptr(dht),
                    L, B, H, nd, 0.0, 0, L_.stream(), label='lstm_bwd_gp', flops=2.0 * nd * L * B * 4 * H * H)
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
        w_ih_pk = derived('ih_pk', w_ih, lambda: pack_forward(w_ih, H, I))
        w_hh_pk = derived('hh_pk', w_hh, lambda: pack_forward(w_hh, H, H))
        if _gp16(H, B, nd):
            pre = lstm_xproj(xhat.view(L * B, I), w_ih_pk, None, nd, H, f'xproj_gp_I{I // H}')
            L_.call('gmt_lstm16_layer_bwd2', L_.ptr(pre), L_.ptr(_hh_raw(w_hh)), L_.ptr(lengths),
                    L_.ptr(gates), L_.ptr(c_all), L_.ptr(dct), L_.ptr(dht), L_.ptr(hhat_all), L_.ptr(hhat_final),
                    L_.ptr(g_gates), L_.ptr(g_c), L, B, H, nd, L_.stream(), label='lstm16_bwd2',
                    flops=2.0 * nd * L * B * 4 * H * H)
            del pre
        elif GP_HOIST_XPROJ:
            pre = lstm_xproj(xhat.view(L * B, I), w_ih_pk, None, nd, H, f'xproj_gp_I{I // H}')
            L_.call('gmt_lstm_layer_bwd2', L_.ptr(pre), None, L_.ptr(w_hh_pk), L_.ptr(lengths),
                    L_.ptr(gates), L_.ptr(c_all), L_.ptr(dct), L_.ptr(dht), L_.ptr(hhat_all), L_.ptr(hhat_final),
                    L_.ptr(g_gates), L_.ptr(g_c), L, B, I, H, nd, L_.stream(), label='lstm_bwd2_h',
                    flops=2.0 * nd * L * B * 4 * H * H)
            del pre
        else:
            L_.call('gmt_lstm_layer_bwd2', L_.ptr(xhat), L_.ptr(w_ih_pk), L_.ptr(w_hh_pk), L_.ptr(lengths),
                    L_.ptr(gates), L_.ptr(c_all), L_.ptr(dct), L_.ptr(dht), L_.ptr(hhat_all), L_.ptr(hhat_final),
                    L_.ptr(g_gates), L_.ptr(g_c), L, B, I, H, nd, L_.stream(), label='lstm_bwd2',
                    flops=2.0 * nd * L * B * 4 * H * (I + H))
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
        bias = lstm_bias(weights, nd)
        gates = torch.empty(nd, L, B, 4 * H, device=dev, dtype=torch.float32)
        c_all = torch.empty(nd, L, B, H, device=dev, dtype=torch.float32)
        h_all = torch.empty(L, B, nd * H, device=dev, dtype=torch.float32)
        h_final = torch.empty(B, nd * H, device=dev, dtype=torch.float32)
        w_ih_pk = derived('ih_pk', w_ih, lambda: pack_forward(w_ih, H, I))
        w_hh_pk = derived('hh_pk', w_hh, lambda: pack_forward(w_hh, H, H))
        if _gp16(H, B, nd):
            L_.call('gmt_lstm_xproj', L_.ptr(x), L_.ptr(w_ih_pk), L_.ptr(bias), L_.ptr(gates), L * B, I, H, nd, L_.stream(),
                    label=f'xproj_gp_I{I // H}', flops=2.0 * nd * L * B * 4 * H * I)
            L_.call('gmt_lstm16_layer_fwd', L_.ptr(gates), L_.ptr(_hh_raw(w_hh)), L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all),
                    L_.ptr(h_all), L_.ptr(h_final), L, B, H, nd, L_.stream(), label='lstm16_fwd_gp',
                    flops=2.0 * nd * L * B * 4 * H * H)
        elif GP_HOIST_XPROJ:
            # x W_ih^T + b straight into the gate buffer (a full-machine launch); the persistent kernel reads its
            # step's slice from there a step ahead and overwrites it with the activated gates
            L_.call('gmt_lstm_xproj', L_.ptr(x), L_.ptr(w_ih_pk), L_.ptr(bias), L_.ptr(gates), L * B, I, H, nd, L_.stream(),
                    label=f'xproj_gp_I{I // H}', flops=2.0 * nd * L * B * 4 * H * I)
            L_.call('gmt_lstm_layer_fwd', None, L_.ptr(gates), None, L_.ptr(w_hh_pk), None, L_.ptr(lengths),
                    L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final), L, B, I, H, nd, 0.0, 0, L_.stream(),
                    label='lstm_fwd_gp_h', flops=2.0 * nd * L * B * 4 * H * H)
        else:
            L_.call('gmt_lstm_layer_fwd', L_.ptr(x), None, L_.ptr(w_ih_pk), L_.ptr(w_hh_pk),
                    L_.ptr(bias), L_.ptr(lengths), L_.ptr(gates), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(h_final),
                    L, B, I, H, nd, 0.0, 0, L_.stream(), label=f'lstm_fwd_gp_I{I // H}', flops=2.0 * nd * L * B * 4 * H * (I + H))
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
        if _gp16(H, B, nd) and (dgates is None) == (dc is None):
            L_.call('gmt_lstm16_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(_hh_raw(w_hh)), L_.ptr(lengths),
                    L_.ptr(dh_all), L_.ptr(dh_final), L_.ptr(dgates), L_.ptr(dc), L_.ptr(da), None, None,
                    L, B, H, nd, L_.stream(), label='lstm16_bwd_gp2', flops=2.0 * nd * L * B * 4 * H * H)
        else:
            w_hh_pkT = derived('hh_pkT', w_hh, lambda: pack_transposed(w_hh, H, 4 * H))
            L_.call('gmt_lstm_layer_bwd', L_.ptr(gates), L_.ptr(c_all), L_.ptr(w_hh_pkT), L_.ptr(lengths),
                    L_.ptr(dh_all), L_.ptr(dh_final), L_.ptr(dgates), L_.ptr(dc), L_.ptr(da), None, None,
                    L, B, H, nd, 0.0, 0, L_.stream(), label='lstm_bwd_gp2', flops=2.0 * nd * L * B * 4 * H * H)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dx_from_da(da, w_ih, L * B).view(L, B, I)
        grads = _lstm_weight_grads(da, x.view(L * B, I), h_all, w_hh, L, B, H, nd)
        return (dx, None, *grads)


# ----------------------------------------------------------------------------------------
# fused critic front end: kinematics + feature encoder (csrc/encoder.hip, RG.py:315-389, 712-717, 757-761)
# ----------------------------------------------------------------------------------------
class EncoderFn(torch.autograd.Function):
    """pos [L,B,2], dts [L,B] (time-major, B % 32 == 0), the two (spectrally normalised) Linears ->
    e [L,B,H] = lrelu(W2 lrelu(W1 feat(pos, dts) + b1) + b2) in ONE launch.
    Backward: the incoming gradient is that of the second PRE-activation when grad_is_preact (the consuming LSTM
    layer applied LeakyReLU'(e) in its dx kernel, ops.LSTMLayerFn x_act_slope); weight gradients on gmt_wgrad,
    the Linear-2 input gradient with LeakyReLU'(h1) fused on gmt_dx_gemm, the 9-wide feature gradient by one
    small library GEMM, kinematics backward on gmt_kinematics_bwd."""

    @staticmethod
    def forward(ctx, pos, dts, w1, b1, w2, b2, slope, eps, grad_is_preact):
        L_.require_cuda(pos, dts, w1, w2)
        pos, dts = pos.contiguous().float(), dts.contiguous().float()
        L, B = dts.shape
        H = w1.shape[0]
        assert B % TM == 0 and w1.shape == (H, 9) and w2.shape == (H, H)
        dev = pos.device
        NW = H // 32
        need_grad = any(ctx.needs_input_grad)
        w1_pk = _pack1(w1, 16, 9, 1 << 30, NW, 1, 0, 1, 9)          # (normalised weights are new tensors every call)
        w2_pk = _pack1(w2, H, H, 1 << 30, NW, 1, 0, 1, H)
        e = torch.empty(L, B, H, device=dev, dtype=torch.float32)
        feat = torch.empty(L, B, 9, device=dev, dtype=torch.float32) if need_grad else None
        h1 = torch.empty(L, B, H, device=dev, dtype=torch.float32) if need_grad else None
        L_.call('gmt_encoder_fwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(w1_pk), L_.ptr(b1.detach().contiguous()),
                L_.ptr(w2_pk), L_.ptr(b2.detach().contiguous()), L_.ptr(feat), L_.ptr(h1), L_.ptr(e), L, B, H,
                float(eps), float(slope), L_.stream(), label='encoder_fwd', flops=2.0 * L * B * (9 * H + H * H))
        if need_grad:
            ctx.save_for_backward(pos, dts, feat, h1, e, w1, w2)
        ctx.cfg = (float(slope), float(eps), bool(grad_is_preact))
        return e

    @staticmethod
    def backward(ctx, g):
        pos, dts, feat, h1, e, w1, w2 = ctx.saved_tensors
        slope, eps, grad_is_preact = ctx.cfg
        L, B = dts.shape
        H = w1.shape[0]
        R = L * B
        g = g.contiguous().view(R, H)
        if not grad_is_preact:
            g = g * torch.where(e.view(R, H) > 0, 1.0, slope)
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_w1 = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        need_w2 = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        h1v = h1.view(R, H)
        d_w2 = d_b2 = d_w1 = d_b1 = gpos = gdts = None
        if need_w2:
            d_w2, d_b2 = wgrad(g, h1v, want_db=True)
        if need_in or need_w1:
            if _dx_shape_ok(H, H):
                d_pre1 = dx_gemm(g, pack_dx_weight(w2.detach().contiguous()), H, H, h1v, slope)
            else:
                d_pre1 = (g @ w2.detach()) * torch.where(h1v > 0, 1.0, slope)
            if need_w1:
                d_w1, d_b1 = wgrad(d_pre1, feat.view(R, 9), want_db=True)
            if need_in:
                gfeat = torch.mm(d_pre1, w1.detach())                   # [R, 9]: 0.02 % of the step's FLOPs
                gpos = torch.empty_like(pos)
                gdts = torch.empty_like(dts)
                L_.call('gmt_kinematics_bwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(gfeat), L_.ptr(gpos), L_.ptr(gdts),
                        L, B, eps, L_.stream(), nbytes=60.0 * L * B)
        return gpos, gdts, d_w1, d_b1, d_w2, d_b2, None, None, None


def lrelu_mask_mul(g: torch.Tensor, act: torch.Tensor, slope: float) -> torch.Tensor:
    """g * LeakyReLU'(act), one launch (csrc/kinematics.hip)."""
    g, act = g.contiguous(), act.contiguous()
    out = torch.empty_like(g)
    L_.call('gmt_lrelu_mask_mul', L_.ptr(g), L_.ptr(act), L_.ptr(out), float(slope), g.numel(), L_.stream(), label='skip')
    return out


class EncoderBwdGPFn(torch.autograd.Function):
    """The FIRST backward of the critic's front end as a differentiable op (gradient penalty, RG.py:829-835):
    (g = adjoint of e, the forward's saved pos / dts / h1 / e, the two normalised weights) -> gpos.
        d_pre2 = g * lrelu'(e);  d_pre1 = (d_pre2 W2) * lrelu'(h1);  gfeat = d_pre1 W1;  gpos = J^T gfeat
    Every map is linear in g, so its backward - the front end's share of the second-order sweep - is the same chain
    transposed, all on existing kernels plus the kinematics JVP:
        fhat = J phat (gmt_kinematics_jvp);  a1 = (fhat W1^T) * lrelu'(h1);  ghat = (a1 W2^T) * lrelu'(e)  (gmt_dx_gemm)
        W1hat = d_pre1^T fhat,  W2hat = d_pre2^T a1  (gmt_wgrad).
    The interpolated positions are a leaf no parameter feeds and the masks are piecewise constant: neither gets an
    adjoint."""

    @staticmethod
    def forward(ctx, g, pos, dts, h1, e, w1, w2, slope, eps):
        L, B = dts.shape
        H = w1.shape[0]
        R = L * B
        d_pre2 = lrelu_mask_mul(g.reshape(R, H), e.view(R, H), slope)
        d_pre1 = dx_gemm(d_pre2, pack_dx_weight(w2.detach().contiguous()), H, H, h1.view(R, H), slope)
        gfeat = torch.mm(d_pre1, w1.detach())                       # [R, 9]
        gpos = torch.empty_like(pos)
        gdts = torch.empty_like(dts)
        L_.call('gmt_kinematics_bwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(gfeat), L_.ptr(gpos), L_.ptr(gdts),
                L, B, eps, L_.stream(), nbytes=60.0 * L * B)
        ctx.save_for_backward(pos, dts, h1, e, w1, w2, d_pre1, d_pre2)
        ctx.cfg = (float(slope), float(eps))
        return gpos

    @staticmethod
    def backward(ctx, phat):
        pos, dts, h1, e, w1, w2, d_pre1, d_pre2 = ctx.saved_tensors
        slope, eps = ctx.cfg
        L, B = dts.shape
        H = w1.shape[0]
        R = L * B
        phat = phat.contiguous()
        fhat = torch.empty(L, B, 9, device=pos.device, dtype=torch.float32)
        L_.call('gmt_kinematics_jvp', L_.ptr(pos), L_.ptr(dts), L_.ptr(phat), L_.ptr(fhat), L, B, eps, L_.stream(),
                nbytes=56.0 * L * B)
        fh = fhat.view(R, 9)
        a1 = lrelu_mask_mul(torch.mm(fh, w1.detach().t()), h1.view(R, H), slope)
        ghat = dx_gemm(a1, pack_dx_weight(w2.detach().t().contiguous()), H, H, e.view(R, H), slope)
        w1hat = w2hat = None
        if ctx.needs_input_grad[5]:
            w1hat = wgrad(d_pre1, fh)[0]
        if ctx.needs_input_grad[6]:
            w2hat = wgrad(d_pre2, a1)[0]
        return ghat.view(L, B, H), None, None, None, None, w1hat, w2hat, None, None


class EncoderGPFn(torch.autograd.Function):
    """Twice-differentiable critic front end of the gradient-penalty pass: the same launch as EncoderFn (kinematics +
    both spectral-norm Linears + both LeakyReLUs, gmt_encoder_fwd).  Its backward under create_graph=True (the
    penalty's autograd.grad w.r.t. the interpolated positions) is EncoderBwdGPFn; the ordinary backward that follows
    the second-order sweep yields the weight / bias gradients only (the positions are a leaf of the inner grad)."""

    @staticmethod
    def forward(ctx, pos, dts, w1, b1, w2, b2, slope, eps):
        L_.require_cuda(pos, dts, w1, w2)
        pos, dts = pos.contiguous().float(), dts.contiguous().float()
        L, B = dts.shape
        H = w1.shape[0]
        assert B % TM == 0 and w1.shape == (H, 9) and w2.shape == (H, H)
        dev = pos.device
        NW = H // 32
        w1_pk = _pack1(w1, 16, 9, 1 << 30, NW, 1, 0, 1, 9)
        w2_pk = _pack1(w2, H, H, 1 << 30, NW, 1, 0, 1, H)
        e = torch.empty(L, B, H, device=dev, dtype=torch.float32)
        feat = torch.empty(L, B, 9, device=dev, dtype=torch.float32)
        h1 = torch.empty(L, B, H, device=dev, dtype=torch.float32)
        L_.call('gmt_encoder_fwd', L_.ptr(pos), L_.ptr(dts), L_.ptr(w1_pk), L_.ptr(b1.detach().contiguous()),
                L_.ptr(w2_pk), L_.ptr(b2.detach().contiguous()), L_.ptr(feat), L_.ptr(h1), L_.ptr(e), L, B, H,
                float(eps), float(slope), L_.stream(), label='encoder_fwd_gp', flops=2.0 * L * B * (9 * H + H * H))
        ctx.save_for_backward(pos, dts, feat, h1, e, w1, w2)
        ctx.cfg = (float(slope), float(eps))
        return e

    @staticmethod
    def backward(ctx, g):
        pos, dts, feat, h1, e, w1, w2 = ctx.saved_tensors
        slope, eps = ctx.cfg
        L, B = dts.shape
        H = w1.shape[0]
        R = L * B
        if torch.is_grad_enabled():
            gpos = EncoderBwdGPFn.apply(g, pos, dts, h1, e, w1, w2, slope, eps)
            return gpos, None, None, None, None, None, None, None
        gq = lrelu_mask_mul(g.reshape(R, H), e.view(R, H), slope)
        h1v = h1.view(R, H)
        d_w2 = d_b2 = d_w1 = d_b1 = None
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            d_w2, d_b2 = wgrad(gq, h1v, want_db=True)
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            d_pre1 = dx_gemm(gq, pack_dx_weight(w2.detach().contiguous()), H, H, h1v, slope)
            d_w1, d_b1 = wgrad(d_pre1, feat.view(R, 9), want_db=True)
        return None, None, d_w1, d_b1, d_w2, d_b2, None, None


def encoder_gp(pos, dts, w1, b1, w2, b2, slope: float = 0.2, eps: float = 1e-8):
    return EncoderGPFn.apply(pos, dts, w1, b1, w2, b2, slope, eps)


def encoder_gp_supported(H: int) -> bool:
    """Shapes the kernelised penalty front end covers: gmt_encoder_fwd's widths and gmt_dx_gemm's (H, H) table."""
    return H % 32 == 0 and H <= 256 and _dx_shape_ok(H, H)


def encoder(pos, dts, w1, b1, w2, b2, slope: float = 0.2, eps: float = 1e-8, grad_is_preact: bool = False):
    return EncoderFn.apply(pos, dts, w1, b1, w2, b2, slope, eps, grad_is_preact)


# ----------------------------------------------------------------------------------------
# trajectory bookkeeping and the generator's auxiliary losses (csrc/loss.hip)
# ----------------------------------------------------------------------------------------
class TrajFromSeqFn(torch.autograd.Function):
    """(seq, starts) -> (pos_tm [T+1,Bp,2], dts_tm [T+1,Bp]): trajectory_to_absolute (RG.py:392-417) + the
    F.pad(dt, (1, 0), 0.001) of RG.py:934 / 949 / 1061 + the move to the kernels' time-major, 32-row padded layout
    in one launch.  seq is either batch-first [B,T,3] (a data batch) or time-major [T,Bp,3] (a rollout output)."""

    @staticmethod
    def forward(ctx, seq, starts, time_major, n_valid, Bp, first_dt):
        L_.require_cuda(seq, starts)
        seq = seq.contiguous()
        starts = starts.contiguous().float()
        if time_major:
            T, bs, _ = seq.shape
            assert bs == Bp
            sb, st = 3, 3 * Bp
        else:
            bs, T, _ = seq.shape
            assert bs == n_valid
            sb, st = 3 * T, 3
        pos = torch.empty(T + 1, Bp, 2, device=seq.device, dtype=torch.float32)
        dts = torch.empty(T + 1, Bp, device=seq.device, dtype=torch.float32)
        L_.call('gmt_traj_from_seq', L_.ptr(seq), sb, st, L_.ptr(starts), L_.ptr(pos), L_.ptr(dts), T, Bp, n_valid,
                float(first_dt), L_.stream())
        ctx.meta = (tuple(seq.shape), sb, st, T, Bp, n_valid, Bp if time_major else n_valid)
        return pos, dts

    @staticmethod
    def backward(ctx, gpos, gdts):
        shape, sb, st, T, Bp, n_valid, n_rows_out = ctx.meta
        dev = (gpos if gpos is not None else gdts).device
        gpos = torch.zeros(T + 1, Bp, 2, device=dev) if gpos is None else gpos.contiguous()
        gdts = torch.zeros(T + 1, Bp, device=dev) if gdts is None else gdts.contiguous()
        gseq = torch.empty(shape, device=dev, dtype=torch.float32)
        L_.call('gmt_traj_from_seq_bwd', L_.ptr(gpos), L_.ptr(gdts), L_.ptr(gseq), sb, st, T, Bp, n_valid, n_rows_out,
                L_.stream())
        return gseq, None, None, None, None, None


def traj_from_seq(seq, starts, time_major: bool, n_valid: int, Bp: int, first_dt: float = 0.001):
    return TrajFromSeqFn.apply(seq, starts, time_major, n_valid, Bp, first_dt)


class GenAuxLossFn(torch.autograd.Function):
    """Every reduction of train_generator_step's auxiliary losses (RG.py:1072-1095) in one kernel:
    seq_tm [T,Bp,3] (rollout output), starts / ends [B,2], lengths int32 [B] ->
    scal [4] = (sum_b |final_b - end_b|^2, sum (1 - cos) mask, sum mask, 0)."""

    @staticmethod
    def forward(ctx, seq_tm, starts, ends, lengths):
        L_.require_cuda(seq_tm, starts, ends, lengths)
        import ctypes
        seq_tm = seq_tm.contiguous()
        T, Bp, _ = seq_tm.shape
        B = starts.shape[0]
        starts, ends = starts.contiguous().float(), ends.contiguous().float()
        assert lengths.dtype == torch.int32 and lengths.numel() == B
        need = ctypes.c_longlong(0)
        L_.call('gmt_gen_loss_workspace_floats', B, ctypes.addressof(need), label='skip')
        ws = torch.empty(need.value, device=seq_tm.device, dtype=torch.float32)
        final = torch.empty(B, 2, device=seq_tm.device, dtype=torch.float32)
        scal = torch.empty(4, device=seq_tm.device, dtype=torch.float32)
        L_.call('gmt_gen_loss_fwd', L_.ptr(seq_tm), 3, 3 * Bp, L_.ptr(starts), L_.ptr(ends), L_.ptr(lengths), None,
                L_.ptr(final), L_.ptr(scal), L_.ptr(ws), T, B, L_.stream())
        ctx.save_for_backward(seq_tm, starts, ends, lengths, final)
        return scal

    @staticmethod
    def backward(ctx, gscal):
        seq_tm, starts, ends, lengths, final = ctx.saved_tensors
        T, Bp, _ = seq_tm.shape
        B = starts.shape[0]
        gscal = gscal.contiguous().float()
        gseq = torch.empty_like(seq_tm)
        L_.call('gmt_gen_loss_bwd', L_.ptr(seq_tm), 3, 3 * Bp, L_.ptr(starts), L_.ptr(ends), L_.ptr(lengths), L_.ptr(final),
                L_.ptr(gscal), L_.ptr(gseq), 3, 3 * Bp, T, Bp, B, L_.stream())
        return gseq, None, None, None


def gen_aux_losses(seq_tm, starts, ends, lengths_i32):
    return GenAuxLossFn.apply(seq_tm, starts, ends, lengths_i32)


STEP_LOSS_KERNELS = True        # False: the step functions' scalar loss arithmetic as torch ops (A/B switch)


class GenStepLossFn(torch.autograd.Function):
    """The whole loss of train_generator_step (RG.py:1069-1097) from the rollout and the critic's padded score vector:
    (seq_tm [T,Bp,3], score [Bp,1], starts [B,2], ends [B,2], lengths int32 [B], w_endpoint, w_direction, reduce) ->
    total [] (differentiable), parts [4] = (g_loss, endpoint_loss, direction_loss, mean score) (not differentiable).
    Three launches forward (reductions, their fixed-order finish, the combination), one backward (d total / d seq and
    d total / d score together).  `reduce` (data-parallel runs): callable that sums a [2] tensor (direction numerator and
    denominator) over the ranks in place and returns the number of ranks - RG.py:1095 divides by the mask sum of the
    WHOLE batch, and the ranks' gradients are averaged afterwards, hence the factor world / den in the coefficients."""

    @staticmethod
    def forward(ctx, seq_tm, score, starts, ends, lengths, w_e, w_d, reduce):
        L_.require_cuda(seq_tm, score, starts, ends, lengths)
        import ctypes
        seq_tm = seq_tm.contiguous()
        T, Bp, _ = seq_tm.shape
        B = starts.shape[0]
        score = score.detach().contiguous()
        assert score.numel() == Bp and lengths.dtype == torch.int32 and lengths.numel() == B
        starts, ends = starts.contiguous().float(), ends.contiguous().float()
        dev = seq_tm.device
        f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
        need = ctypes.c_longlong(0)
        L_.call('gmt_gen_loss_workspace_floats', B, ctypes.addressof(need), label='skip')
        ws, final, scal = f(need.value), f(B, 2), f(4)
        L_.call('gmt_gen_loss_fwd', L_.ptr(seq_tm), 3, 3 * Bp, L_.ptr(starts), L_.ptr(ends), L_.ptr(lengths), L_.ptr(score),
                L_.ptr(final), L_.ptr(scal), L_.ptr(ws), T, B, L_.stream())
        nd, world = None, 1.0
        if reduce is not None:
            nd = scal[1:3].clone()
            world = float(reduce(nd))
        total, parts, coef = f(()), f(4), f(3)
        L_.call('gmt_gen_step_loss_finish', L_.ptr(scal), L_.ptr(nd), world, B, float(w_e), float(w_d), L_.ptr(total),
                L_.ptr(parts), L_.ptr(coef), L_.stream())
        ctx.save_for_backward(seq_tm, starts, ends, lengths, final, coef)
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        seq_tm, starts, ends, lengths, final, coef = ctx.saved_tensors
        T, Bp, _ = seq_tm.shape
        B = starts.shape[0]
        g = g_total.contiguous().float()
        gseq = torch.empty_like(seq_tm)
        d_score = torch.empty(Bp, 1, device=seq_tm.device, dtype=torch.float32)
        L_.call('gmt_gen_step_loss_bwd', L_.ptr(seq_tm), 3, 3 * Bp, L_.ptr(starts), L_.ptr(ends), L_.ptr(lengths), L_.ptr(final),
                L_.ptr(coef), L_.ptr(g), L_.ptr(gseq), 3, 3 * Bp, L_.ptr(d_score), T, Bp, B, L_.stream())
        return gseq, d_score, None, None, None, None, None, None


def gen_step_loss(seq_tm, score_padded, starts, ends, lengths_i32, w_endpoint: float, w_direction: float, reduce=None):
    return GenStepLossFn.apply(seq_tm, score_padded, starts, ends, lengths_i32, w_endpoint, w_direction, reduce)


class CriticLossFn(torch.autograd.Function):
    """d_loss = mean(fake) - mean(real) (RG.py:963) over the first B rows of two padded score vectors [Bp,1]:
    -> loss [] (differentiable), parts [2] = (mean real, mean fake); one launch each way (zeros on the padding rows)."""

    @staticmethod
    def forward(ctx, real, fake, B):
        L_.require_cuda(real, fake)
        real, fake = real.detach().contiguous(), fake.detach().contiguous()
        assert real.numel() == fake.numel() and real.numel() >= B
        loss = torch.empty((), device=real.device, dtype=torch.float32)
        parts = torch.empty(2, device=real.device, dtype=torch.float32)
        L_.call('gmt_critic_loss_fwd', L_.ptr(real), L_.ptr(fake), int(B), L_.ptr(loss), L_.ptr(parts), L_.stream())
        ctx.meta = (int(B), real.numel(), tuple(real.shape))
        ctx.mark_non_differentiable(parts)
        return loss, parts

    @staticmethod
    def backward(ctx, g, _gp):
        B, Bp, shape = ctx.meta
        g = g.contiguous().float()
        d_real = torch.empty(shape, device=g.device, dtype=torch.float32)
        d_fake = torch.empty(shape, device=g.device, dtype=torch.float32)
        L_.call('gmt_critic_loss_bwd', L_.ptr(g), B, Bp, L_.ptr(d_real), L_.ptr(d_fake), L_.stream())
        return d_real, d_fake, None


def critic_loss(real_padded, fake_padded, B: int):
    return CriticLossFn.apply(real_padded, fake_padded, B)


# ----------------------------------------------------------------------------------------
# the two small dense stacks (csrc/mlp.hip): generator condition encoder, critic output head
# ----------------------------------------------------------------------------------------
MLP_WIDTHS = (32, 64, 128, 256, 512)
MLP_KERNELS = True       # False: condition encoder / critic head as torch ops (A/B switch, bench.py --no-mlp-kernels)
MLP_COND = True          # the two halves of that switch
MLP_HEAD = True


def _pitch128(n: int) -> int:
    """Row pitch of an activation that gmt_wgrad will read as X: n itself when the kernel takes it as it is
    (n % 128 == 0 or skinny), else the next multiple of 128 (zero-filled padding, wgrad(x_padded=True))."""
    return n if (n % 128 == 0 or n <= 32) else (n + 127) // 128 * 128


def _rows(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return t if t.stride(-1) == 1 else t.contiguous()


class CondEncoderFn(torch.autograd.Function):
    """Generator._compute_condition + condition_encoder (RG.py:469-484, 440-447) and the hoisted gate term
    g0 = cond . W_ih_l0[:, 3:3+H]^T + b_ih_l0 + b_hh_l0, one launch (gmt_cond_encoder_fwd); backward = one launch for the
    data path (gmt_cond_encoder_bwd) + the weight-gradient contractions on gmt_wgrad.
    (start [B,2], end [B,2], z [B,latent], w1, b1, ln1_g, ln1_b, w2, b2, ln2_g, ln2_b, w_ih0 [4H, 3+H+4], b_ih0, b_hh0,
     Bp, unit_major, want_g0, slope, ln_eps) -> cond [Bp,H], g0 [Bp,4H] (empty tensor when want_g0 is False).
    Rows >= B of both are zeros.  No gradient for start / end / z (network inputs in the reference's training loop)."""

    @staticmethod
    def forward(ctx, start, end, z, w1, b1, g1, be1, w2, b2, g2, be2, w_ih0, b_ih0, b_hh0, Bp, unit_major, want_g0,
                slope, ln_eps):
        L_.require_cuda(start, end, z, w1, w_ih0)
        B, latent = z.shape
        H = w2.shape[0]
        assert w1.shape == (H, 6 + latent) and w_ih0.shape == (4 * H, 3 + H + 4) and Bp % TM == 0 and Bp >= B
        dev = z.device
        f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
        need = any(ctx.needs_input_grad[3:14])
        start, end, z = _rows(start), _rows(end), _rows(z)
        ws = [t.detach().contiguous() for t in (w1, b1, g1, be1, w2, b2, g2, be2, w_ih0, b_ih0, b_hh0)]
        a = L_.CondFwdArgs()
        a.start, a.end, a.z = start.data_ptr(), end.data_ptr(), z.data_ptr()
        a.ld_start, a.ld_end, a.ld_z = start.stride(0), end.stride(0), z.stride(0)
        (a.w1, a.b1, a.ln1_g, a.ln1_b, a.w2, a.b2, a.ln2_g, a.ln2_b) = [t.data_ptr() for t in ws[:8]]
        cond = f(Bp, H)
        g0 = f(Bp, 4 * H) if want_g0 else f(0)
        a.cond = cond.data_ptr()
        if want_g0:
            a.w_g, a.ld_wg = ws[8].data_ptr() + 4 * 3, ws[8].shape[1]
            a.bias_a, a.bias_b, a.g0 = ws[9].data_ptr(), ws[10].data_ptr(), g0.data_ptr()
        saved = ()
        cp = _pitch128(6 + latent)
        if need:
            saved = (f(Bp, cp), f(Bp, H), f(Bp), f(Bp, H), f(Bp, H), f(Bp))      # c_in, xhat1, rstd1, a1, xhat2, rstd2
            a.c_in, a.xhat1, a.rstd1, a.a1, a.xhat2, a.rstd2 = [t.data_ptr() for t in saved]
        a.B, a.Bp, a.H, a.latent, a.c_pitch, a.g0_unit_major = B, Bp, H, latent, cp, int(bool(unit_major))
        a.slope, a.ln_eps, a.dist_eps = float(slope), float(ln_eps), 1e-8
        L_.call('gmt_cond_encoder_fwd', ctypes_byref(a), L_.stream(), label='cond_encoder_fwd',
                flops=2.0 * Bp * H * (6 + latent + H + (4 * H if want_g0 else 0)))
        if need:
            ctx.save_for_backward(cond, g1, w2, g2, w_ih0, *saved)
            ctx.meta = (B, Bp, H, latent, cp, bool(unit_major), bool(want_g0), float(slope))
        return cond, g0

    @staticmethod
    def backward(ctx, d_cond, d_g0):
        if torch.is_grad_enabled():
            raise NotImplementedError('CondEncoderFn: first-order backward only (the generator is never differentiated twice)')
        cond, g1, w2, g2, w_ih0, c_in, xhat1, rstd1, a1, xhat2, rstd2 = ctx.saved_tensors
        B, Bp, H, latent, cp, unit_major, want_g0, slope = ctx.meta
        dev = cond.device
        f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
        if not want_g0:
            d_g0 = None
        if d_g0 is not None:
            d_g0 = d_g0.contiguous() if B == Bp else d_g0.clone(memory_format=torch.contiguous_format)   # pad rows are zeroed in place
        if d_cond is not None:
            d_cond = d_cond.contiguous()
        if d_g0 is None and d_cond is None:
            return (None,) * 19
        w_ih0c, w2c = w_ih0.detach().contiguous(), w2.detach().contiguous()
        g1c, g2c = g1.detach().contiguous(), g2.detach().contiguous()
        dy2, dy1, ln_part = f(Bp, H), f(Bp, H), f(Bp // TM, 4, H)
        a = L_.CondBwdArgs()
        a.d_g0, a.d_cond = L_.ptr(d_g0), L_.ptr(d_cond)
        a.w_g, a.ld_wg = w_ih0c.data_ptr() + 4 * 3, w_ih0c.shape[1]
        a.cond, a.xhat2, a.rstd2, a.ln2_g = cond.data_ptr(), xhat2.data_ptr(), rstd2.data_ptr(), g2c.data_ptr()
        a.w2, a.a1, a.xhat1, a.rstd1, a.ln1_g = w2c.data_ptr(), a1.data_ptr(), xhat1.data_ptr(), rstd1.data_ptr(), g1c.data_ptr()
        a.dy2, a.dy1, a.ln_part = dy2.data_ptr(), dy1.data_ptr(), ln_part.data_ptr()
        a.B, a.Bp, a.H, a.g0_unit_major, a.slope = B, Bp, H, int(unit_major), slope
        L_.call('gmt_cond_encoder_bwd', ctypes_byref(a), L_.stream(), label='cond_encoder_bwd',
                flops=2.0 * Bp * H * (H + (4 * H if d_g0 is not None else 0)))
        d_w_ih0 = d_bg = None
        if d_g0 is not None:
            d_wg, d_bg = wgrad_gates(d_g0, cond, want_db=True) if unit_major else wgrad(d_g0, cond, want_db=True)
            d_w_ih0 = F.pad(d_wg, (3, 4))                   # the condition columns of W_ih_l0 [4H, 3 + H + 4]
        d_w2, d_b2 = wgrad(dy2, a1, want_db=True)
        d_w1, d_b1 = wgrad(dy1, c_in[:, :6 + latent], want_db=True, x_padded=True)
        ln = f(4, H)
        L_.call('gmt_slab_sum', ln_part.data_ptr(), ln.data_ptr(), 4 * H, Bp // TM, 0, L_.stream(), label='skip')
        return (None, None, None, d_w1, d_b1, ln[2], ln[3], d_w2, d_b2, ln[0], ln[1], d_w_ih0, d_bg, d_bg,
                None, None, None, None, None)


def cond_encoder_supported(G) -> bool:
    ce = G.condition_encoder
    return (MLP_KERNELS and MLP_COND and G.config.generator_hidden_dim in MLP_WIDTHS and ce[2].negative_slope == ce[5].negative_slope
            and ce[1].eps == ce[4].eps and ce[0].weight.dtype == torch.float32)


def cond_encoder(G, start, end, z, Bp: int, unit_major: bool, want_g0: bool):
    """cond [Bp,H] (+ g0 [Bp,4H]) of Generator `G` for the batch (start, end, z): see CondEncoderFn."""
    ce, lp = G.condition_encoder, G.lstm
    cond, g0 = CondEncoderFn.apply(start, end, z, ce[0].weight, ce[0].bias, ce[1].weight, ce[1].bias, ce[3].weight, ce[3].bias,
                                   ce[4].weight, ce[4].bias, lp.weight_ih_l0, lp.bias_ih_l0, lp.bias_hh_l0, Bp, unit_major,
                                   want_g0, float(ce[2].negative_slope), float(ce[1].eps))
    return cond, (g0 if want_g0 else None)


class CriticHeadFn(torch.autograd.Function):
    """Discriminator.output_layers (RG.py:730-737 at RG.py:781) on the final hidden states, one launch each way
    (gmt_critic_head_fwd / _bwd) + the weight-gradient contractions on gmt_wgrad.
    (x [Bp,2H], w1 [H,2H], b1, w2 [H/2,H], b2, w3 [1,H/2], b3, slope, p, seed) -> score [Bp,1]; w1..w3 are the spectrally
    normalised weights.  Dropout(p) is the counter-based mask gmt_keep(row * H + column, seed): dropout_keep_mask(1, Bp, H,
    p, seed) is its host replica.  First-order backward only (the penalty pass keeps the torch composite)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, slope, p, seed):
        L_.require_cuda(x, w1)
        Bp, H2 = x.shape
        H, N2 = H2 // 2, H2 // 4
        assert Bp % TM == 0 and w1.shape == (H, H2) and w2.shape == (N2, H) and w3.shape == (1, N2)
        dev = x.device
        f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
        need = any(ctx.needs_input_grad[:7])
        xc = x.detach().contiguous()
        ws = [t.detach().contiguous() for t in (w1, b1, w2, b2, w3, b3)]
        a2p = _pitch128(N2)
        score = f(Bp, 1)
        a = L_.HeadFwdArgs()
        a.x = xc.data_ptr()
        a.w1, a.b1, a.w2, a.b2, a.w3, a.b3 = [t.data_ptr() for t in ws]
        a.score = score.data_ptr()
        if need:
            a1, a2 = f(Bp, H), f(Bp, a2p)
            a.a1, a.a2 = a1.data_ptr(), a2.data_ptr()
        a.Bp, a.H, a.a2_pitch = Bp, H, a2p
        a.slope, a.dropout_p, a.seed = float(slope), float(p), int(seed) & 0xFFFFFFFF
        L_.call('gmt_critic_head_fwd', ctypes_byref(a), L_.stream(), label='critic_head_fwd',
                flops=2.0 * Bp * (H2 * H + H * N2 + N2))
        if need:
            ctx.save_for_backward(xc, w1, w2, w3, a1, a2)
            ctx.meta = (Bp, H, a2p, float(slope), float(p), int(seed) & 0xFFFFFFFF)
        return score

    @staticmethod
    def backward(ctx, d_score):
        x, w1, w2, w3, a1, a2 = ctx.saved_tensors
        Bp, H, a2p, slope, p, seed = ctx.meta
        N2 = H // 2
        if torch.is_grad_enabled():
            # create_graph=True: the same backward out of torch ops, which autograd can differentiate once more (the
            # masks are constants: LeakyReLU'' = 0).  Not a hot path - the penalty uses composite.discriminator_forward_gp.
            m2 = torch.where(a2[:, :N2] > 0, 1.0, slope)
            m1 = torch.where(a1 > 0, 1.0, slope) * ((a1 != 0).float() / (1.0 - p) if p > 0 else 1.0)
            dy2 = (d_score @ w3) * m2
            dy1 = (dy2 @ w2) * m1
            return (dy1 @ w1, dy1.t() @ x, dy1.sum(0), dy2.t() @ a1, dy2.sum(0), d_score.t() @ a2[:, :N2], d_score.sum(0),
                    None, None, None)
        dev = x.device
        f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
        ds = d_score.contiguous()
        w1c, w2c, w3c = w1.detach().contiguous(), w2.detach().contiguous(), w3.detach().contiguous()
        dx, dy1, dy2, ds4 = f(Bp, 2 * H), f(Bp, H), f(Bp, N2), f(Bp, 4)
        a = L_.HeadBwdArgs()
        a.d_score, a.a1, a.a2 = ds.data_ptr(), a1.data_ptr(), a2.data_ptr()
        a.w1, a.w2, a.w3 = w1c.data_ptr(), w2c.data_ptr(), w3c.data_ptr()
        a.dx, a.dy1, a.dy2, a.ds4 = dx.data_ptr(), dy1.data_ptr(), dy2.data_ptr(), ds4.data_ptr()
        a.Bp, a.H, a.a2_pitch, a.slope, a.dropout_p, a.seed = Bp, H, a2p, slope, p, seed
        L_.call('gmt_critic_head_bwd', ctypes_byref(a), L_.stream(), label='critic_head_bwd',
                flops=2.0 * Bp * (2 * H * H + H * N2 + N2))
        d_w1, d_b1 = wgrad(dy1, x, want_db=True)
        d_w2, d_b2 = wgrad(dy2, a1, want_db=True)
        d_w3, d_b3 = wgrad(ds4, a2[:, :N2], want_db=True, x_padded=a2p != N2)
        return dx, d_w1, d_b1, d_w2, d_b2, d_w3[:1], d_b3[:1], None, None, None


def _head_fwd_launch(x, w1, b1, w2, b2, w3, b3, slope, p, seed, a1, a2, a2p, m1=None, m2=None, label='critic_head_fwd'):
    Bp, H2 = x.shape
    H = H2 // 2
    score = torch.empty(Bp, 1, device=x.device, dtype=torch.float32)
    a = L_.HeadFwdArgs()
    a.x = x.data_ptr()
    a.w1, a.w2, a.w3 = w1.data_ptr(), w2.data_ptr(), w3.data_ptr()
    a.b1, a.b2, a.b3 = L_.ptr(b1), L_.ptr(b2), L_.ptr(b3)
    a.score, a.a1, a.a2 = score.data_ptr(), L_.ptr(a1), L_.ptr(a2)
    a.m1, a.m2 = L_.ptr(m1), L_.ptr(m2)
    a.Bp, a.H, a.a2_pitch = Bp, H, a2p
    a.slope, a.dropout_p, a.seed = float(slope), float(p), int(seed) & 0xFFFFFFFF
    L_.call('gmt_critic_head_fwd', ctypes_byref(a), L_.stream(), label=label, flops=2.0 * Bp * (H2 * H + H * (H // 2) + H // 2))
    return score


class CriticHeadBwdGPFn(torch.autograd.Function):
    """The head's first backward as a differentiable function of the score adjoint and the three weights (the saved
    activations only enter through LeakyReLU' / dropout factors, which are piecewise constant):
    (d_score [Bp,1], w1, w2, w3, a1, a2, meta) -> dx [Bp,2H] = ((d_score w3 * f2) W2 * f1) W1   (gmt_critic_head_bwd).
    Its backward - the head's piece of the penalty's second backward (RG.py:829-835) - is the head forward kernel in
    mask mode for the adjoint of d_score plus three contractions on gmt_wgrad for the adjoints of the weights."""

    @staticmethod
    def forward(ctx, d_score, w1, w2, w3, a1, a2, meta):
        Bp, H, a2p, slope, p, seed = meta
        N2 = H // 2
        dev = a1.device
        f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
        ds = d_score.detach().contiguous()
        w1c, w2c, w3c = w1.detach().contiguous(), w2.detach().contiguous(), w3.detach().contiguous()
        dx, dy1, dy2, ds4 = f(Bp, 2 * H), f(Bp, H), f(Bp, N2), f(Bp, 4)
        a = L_.HeadBwdArgs()
        a.d_score, a.a1, a.a2 = ds.data_ptr(), a1.data_ptr(), a2.data_ptr()
        a.w1, a.w2, a.w3 = w1c.data_ptr(), w2c.data_ptr(), w3c.data_ptr()
        a.dx, a.dy1, a.dy2, a.ds4 = dx.data_ptr(), dy1.data_ptr(), dy2.data_ptr(), ds4.data_ptr()
        a.Bp, a.H, a.a2_pitch, a.slope, a.dropout_p, a.seed = Bp, H, a2p, slope, p, seed
        L_.call('gmt_critic_head_bwd', ctypes_byref(a), L_.stream(), label='critic_head_bwd',
                flops=2.0 * Bp * (2 * H * H + H * N2 + N2))
        ctx.save_for_backward(w1c, w2c, w3c, a1, a2, dy1, dy2, ds4)
        ctx.meta = meta
        return dx

    @staticmethod
    def backward(ctx, xhat):
        if torch.is_grad_enabled():
            raise NotImplementedError('CriticHeadBwdGPFn: differentiable once (the penalty needs no third derivative)')
        w1, w2, w3, a1, a2, dy1, dy2, ds4 = ctx.saved_tensors
        Bp, H, a2p, slope, p, seed = ctx.meta
        N2 = H // 2
        dev = a1.device
        xh = xhat.contiguous()
        u1 = torch.empty(Bp, H, device=dev, dtype=torch.float32)
        u2 = torch.empty(Bp, a2p, device=dev, dtype=torch.float32)
        dds = _head_fwd_launch(xh, w1, None, w2, None, w3, None, slope, p, seed, u1, u2, a2p, a1, a2, label='critic_head_bwd2')
        d_w1 = d_w2 = d_w3 = None
        if ctx.needs_input_grad[1]:
            d_w1 = wgrad(dy1, xh)[0]
        if ctx.needs_input_grad[2]:
            d_w2 = wgrad(dy2, u1)[0]
        if ctx.needs_input_grad[3]:
            d_w3 = wgrad(ds4, u2[:, :N2], x_padded=a2p != N2)[0][:1]
        return (dds if ctx.needs_input_grad[0] else None), d_w1, d_w2, d_w3, None, None, None


class CriticHeadGPFn(torch.autograd.Function):
    """CriticHeadFn for the gradient-penalty pass: same forward launch; under create_graph=True its backward hands the
    input gradient over as CriticHeadBwdGPFn (differentiable once more); called without a graph it is CriticHeadFn's."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, slope, p, seed):
        L_.require_cuda(x, w1)
        Bp, H2 = x.shape
        H, N2 = H2 // 2, H2 // 4
        assert Bp % TM == 0 and w1.shape == (H, H2) and w2.shape == (N2, H) and w3.shape == (1, N2)
        f = lambda *sh: torch.empty(*sh, device=x.device, dtype=torch.float32)
        xc = x.detach().contiguous()
        ws = [t.detach().contiguous() for t in (w1, b1, w2, b2, w3, b3)]
        a2p = _pitch128(N2)
        a1, a2 = f(Bp, H), f(Bp, a2p)
        score = _head_fwd_launch(xc, ws[0], ws[1], ws[2], ws[3], ws[4], ws[5], slope, p, seed, a1, a2, a2p)
        ctx.save_for_backward(xc, w1, w2, w3, a1, a2)
        ctx.meta = (Bp, H, a2p, float(slope), float(p), int(seed) & 0xFFFFFFFF)
        return score

    @staticmethod
    def backward(ctx, d_score):
        x, w1, w2, w3, a1, a2 = ctx.saved_tensors
        meta = ctx.meta
        Bp, H, a2p, slope, p, seed = meta
        N2 = H // 2
        if torch.is_grad_enabled():
            # the penalty's inner autograd.grad: only the input gradient is asked for (RG.py:829-835)
            dx = CriticHeadBwdGPFn.apply(d_score, w1, w2, w3, a1, a2, meta)
            return dx, None, None, None, None, None, None, None, None, None
        ds = d_score.contiguous()
        f = lambda *sh: torch.empty(*sh, device=x.device, dtype=torch.float32)
        w1c, w2c, w3c = w1.detach().contiguous(), w2.detach().contiguous(), w3.detach().contiguous()
        dx, dy1, dy2, ds4 = f(Bp, 2 * H), f(Bp, H), f(Bp, N2), f(Bp, 4)
        a = L_.HeadBwdArgs()
        a.d_score, a.a1, a.a2 = ds.data_ptr(), a1.data_ptr(), a2.data_ptr()
        a.w1, a.w2, a.w3 = w1c.data_ptr(), w2c.data_ptr(), w3c.data_ptr()
        a.dx, a.dy1, a.dy2, a.ds4 = dx.data_ptr(), dy1.data_ptr(), dy2.data_ptr(), ds4.data_ptr()
        a.Bp, a.H, a.a2_pitch, a.slope, a.dropout_p, a.seed = Bp, H, a2p, slope, p, seed
        L_.call('gmt_critic_head_bwd', ctypes_byref(a), L_.stream(), label='critic_head_bwd',
                flops=2.0 * Bp * (2 * H * H + H * N2 + N2))
        d_w1, d_b1 = wgrad(dy1, x, want_db=True)
        d_w2, d_b2 = wgrad(dy2, a1, want_db=True)
        d_w3, d_b3 = wgrad(ds4, a2[:, :N2], want_db=True, x_padded=a2p != N2)
        return dx, d_w1, d_b1, d_w2, d_b2, d_w3[:1], d_b3[:1], None, None, None


def critic_head_gp(x, w1, b1, w2, b2, w3, b3, slope: float = 0.2, p: float = 0.0, seed: int = 0):
    return CriticHeadGPFn.apply(x, w1, b1, w2, b2, w3, b3, slope, p, seed)


def critic_head_supported(H: int) -> bool:
    return MLP_KERNELS and MLP_HEAD and H in MLP_WIDTHS


def critic_head(x, w1, b1, w2, b2, w3, b3, slope: float = 0.2, p: float = 0.0, seed: int = 0):
    return CriticHeadFn.apply(x, w1, b1, w2, b2, w3, b3, slope, p, seed)


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
            lr, betas[0], betas[1], eps, step, grad_scale, L_.stream(), nbytes=28.0 * n)


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


def gen_forward_packs(w_hh0, w_ih1, w_hh1, w1):
    """Fragment-major copies of the generator's recurrent / head weights for gmt_gen_rollout_fwd (cached per
    optimizer step; all four are nn.Parameters)."""
    H = w_hh0.shape[1]
    NW, HH = H // 32, H // 2
    return derived('gen_fwd_pk', [w_hh0, w_ih1, w_hh1, w1], lambda: (
        _pack1(w_hh0, H, H, 1 << 30, NW, 4, H, 1, H), _pack1(w_ih1, H, H, 1 << 30, NW, 4, H, 1, H),
        _pack1(w_hh1, H, H, 1 << 30, NW, 4, H, 1, H), _pack1(w1, H, H, HH, max(1, H // 64), 1, 0, 1, H)))


def gen_backward_packs(w_hh0, w_ih1, w_hh1, w1):
    H = w_hh0.shape[1]
    NW, HH = H // 32, H // 2

    def build():
        wl1 = torch.cat([w_hh1.detach(), w_ih1.detach()], dim=1).contiguous()        # [4H, 2H]
        return (_pack1(w1, HH, HH, 1 << 30, NW, 1, 0, H, 1), _pack1(wl1, 4 * H, 4 * H, 1 << 30, NW, 2, H, 2 * H, 1),
                _pack1(w_hh0, 4 * H, 4 * H, 1 << 30, NW, 1, 0, H, 1))
    return derived('gen_bwd_pk', [w_hh0, w_ih1, w_hh1, w1], build)


GEN_BWD_SUMS_G0 = True                # False: d g0 by a separate reduction pass over gates0 (A/B switch)
AFTER_GEN_BWD_LAUNCH: list = []       # at most one callable(event recorded just before the BPTT launch); one-shot


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
        pk_hh0, pk_ih1, pk_hh1, pk_w1 = gen_forward_packs(w_hh0, w_ih1, w_hh1, w1)
        a.w_u_pk = P(_pack1(w_u, 8, 7, 1 << 30, NW, 4, H, 1, 7))        # (w_u is assembled per call: not cached)
        a.w_hh0_pk = P(pk_hh0)
        a.w_ih1_pk = P(pk_ih1)
        a.w_hh1_pk = P(pk_hh1)
        a.bias1 = P(bias1)
        a.w1_pk = P(pk_w1)
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
        # executed matrix FLOPs per row-step: W_u (7 wide) + W_hh_l0 + layer 1 + head; the time-invariant
        # condition . W_ih_l0 term (8 H^2 of SURVEY.md's 548 224 at H = 128) is hoisted out of the loop, not counted
        g_exec = 8 * H * 7 + 8 * H * H + 16 * H * H + H * H + 3 * H
        ctx.g_exec = g_exec
        L_.call('gmt_gen_rollout_fwd', ctypes_byref(a), L_.stream(),
                label='gen_fwd_train' if need_grad else 'gen_fwd_nograd', flops=float(T) * B * g_exec)
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
        w1T, wl1T, whh0T = gen_backward_packs(w_hh0, w_ih1, w_hh1, w1)
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
        d_g0 = None
        if GEN_BWD_SUMS_G0 and H <= 128:               # the kernel accumulates d g0 on chip (32 x 4H floats of LDS)
            d_g0 = f(B, 4 * H)
            a.d_g0 = d_g0.data_ptr()
        a.T, a.B, a.H = T, B, H
        a.dropout_p, a.seed = ctx.drop
        hook = AFTER_GEN_BWD_LAUNCH[0] if AFTER_GEN_BWD_LAUNCH else None
        before = None
        if hook is not None:
            before = torch.cuda.Event()
            before.record()
        L_.call('gmt_gen_rollout_bwd', ctypes_byref(a), L_.stream(), label='gen_bwd',
                flops=float(T) * B * (ctx.g_exec - 8 * H * 7))      # the data-gradient half of the backward (dW: gmt_wgrad)
        if hook is not None:
            # the BPTT kernel is B / 32 workgroups (half the CUs at B = 4096): whoever registered the hook may now
            # enqueue independent work on ANOTHER stream that waits for `before` only (Trainer: the next D step's
            # real-sample critic forward), so that it runs beside this launch instead of after it
            del AFTER_GEN_BWD_LAUNCH[:]
            hook(before)
        da0 = gates0.view(T * B, 4 * H)
        da1 = gates1.view(T * B, 4 * H)
        h0v, h1v = h0.view(T * B, H), h1.view(T * B, H)
        if d_g0 is None:
            d_g0 = gates0.sum(0)                                        # time-invariant input: sum over t (unit-major, like g0)
        d_w_u = wgrad_gates(da0, u.view(T * B, 8))[0][:, :7]
        d_w_hh0 = wgrad_gates(da0[B:], h0v[:-B])[0] if T > 1 else torch.zeros_like(w_hh0)
        d_w_ih1, d_bias1 = wgrad_gates(da1, h0d.view(T * B, H), want_db=True)
        d_w_hh1 = wgrad_gates(da1[B:], h1v[:-B])[0] if T > 1 else torch.zeros_like(w_hh1)
        dyp = d_ypre.view(T * B, HH)
        d_w1, d_b1 = wgrad(dyp, h1v, want_db=True)
        dov = d_o.view(T * B, 3)
        d_w2 = wgrad(y.view(T * B, HH), dov)[0].t()                    # [3,HH] via its transpose: N = HH, skinny X = d_o
        d_b2 = d_o.sum(0).sum(0)                                        # (two well-shaped reductions: [T, B*3] then [B, 3]; 0.18 -> 0.02 ms)
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
