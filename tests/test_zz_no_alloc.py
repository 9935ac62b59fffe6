"""gmt_wgrad_noalloc (include/gmt_hip.h): the weight-gradient contraction for callers that cannot allocate.
Through the C ABI, against gmt_wgrad on the same inputs - same kernels, so bit for bit - on two calls in a
row, the second with a smaller input (the static scratch keeps the first call's contents), plus the
size limit.  Collected last (file name): the round's end runs `pytest -x`."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _both(R, N, I, seed, want_db=True, p_drop=0.0):
    from generative_mouse_trajectories_amd import _lib as L_
    g = torch.Generator().manual_seed(seed)
    dA = torch.randn(R, N, generator=g).to(DEV)
    X = torch.randn(R, I, generator=g).to(DEV)
    ks, need = ctypes.c_int(0), ctypes.c_longlong(0)
    L_.call('gmt_wgrad_workspace_floats', N, I, R, ctypes.addressof(ks), ctypes.addressof(need))
    ws = torch.empty(need.value, device=DEV)
    out = []
    for name in ('gmt_wgrad', 'gmt_wgrad_noalloc'):
        dW = torch.full((N, I), float('nan'), device=DEV)
        db = torch.full((N,), float('nan'), device=DEV) if want_db else None
        head = (dA.data_ptr(), N, X.data_ptr(), I, R, N, I, dW.data_ptr(), None if db is None else db.data_ptr(), 0)
        if name == 'gmt_wgrad':
            L_.call(name, *head, ws.data_ptr(), need.value, p_drop, 99, L_.stream())
        else:
            before = torch.cuda.memory_stats()['num_alloc_retries'], torch.cuda.memory_allocated()
            L_.call(name, *head, p_drop, 99, L_.stream())
            assert (torch.cuda.memory_stats()['num_alloc_retries'], torch.cuda.memory_allocated()) == before
        torch.cuda.synchronize()
        out.append((dW, db))
    (w0, b0), (w1, b1) = out
    assert torch.isfinite(w0).all() and torch.equal(w0, w1), float((w0 - w1).abs().max())
    if want_db:
        assert torch.isfinite(b0).all() and torch.equal(b0, b1)
    # and the contraction itself (fp64 reference of the same op)
    Xr = X.double()
    if p_drop > 0:                                   # host replica of the kernels' counter-based mask (element r*ldx + i)
        from generative_mouse_trajectories_amd import ops
        Xr = Xr * ops.dropout_keep_mask(R, 1, I, p_drop, 99, pair16=True).reshape(R, I).double().to(DEV)
    ref = dA.double().t() @ Xr
    assert float((w1.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    return need.value


def test_wgrad_noalloc_matches_wgrad_on_two_calls_in_a_row():
    from generative_mouse_trajectories_amd import _lib as L_
    L_.call('gmt_wgrad_noalloc_init')
    n1 = _both(129 * 256, 512, 256, 1)               # the recurrent-weight shape, split-K over 128 slabs
    n2 = _both(16 * 40, 128, 128, 2)                 # smaller: fewer slabs, the scratch still holds call 1's partial sums
    assert n2 < n1
    _both(16 * 64, 64, 9, 3, want_db=False)          # skinny X, scalar slab reduction
    _both(129 * 64, 512, 128, 4, p_drop=0.1)         # X read through the fused dropout mask


def test_wgrad_noalloc_rejects_sizes_beyond_its_scratch():
    from generative_mouse_trajectories_amd import _lib as L_
    R, N, I = 129 * 4096, 5120, 2048                 # 320 tiles x 2 slabs x 32 768 floats > 65 MiB
    ks, need = ctypes.c_int(0), ctypes.c_longlong(0)
    L_.call('gmt_wgrad_workspace_floats', N, I, R, ctypes.addressof(ks), ctypes.addressof(need))
    assert need.value > 17039360
    dW = torch.zeros(N, I, device=DEV)
    with pytest.raises(L_.GmtError, match='workspace'):
        L_.call('gmt_wgrad_noalloc', dW.data_ptr(), N, dW.data_ptr(), I, R, N, I, dW.data_ptr(), None, 0, 0.0, 0, L_.stream())
