"""The critic's two-layer BiLSTM as one autograd node with the top layer's input gradient contracted inside its backward
sweep (ops.LSTMStack2Fn, gmt_lstm_layer_bwd_fx, H = 128) against the same layers as separate nodes with the separate
input-gradient contraction (ops.LSTMLayerFn + gmt_dx_gemm) - the path the oracle / golden-vector tests pinned in rounds
1-2.  Same forward kernels, so scores are bit-identical; gradients differ by fp32 summation order only."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(fused, B, Lq, ragged, p_drop, seed):
    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd import ops
    cfg = gmt.Config()
    cfg.discriminator_hidden_dim = 128
    cfg.lstm_dropout, cfg.head_dropout = p_drop, 0.0
    torch.manual_seed(seed)
    D = gmt.Discriminator(cfg).to(DEV).train()
    gen = torch.Generator().manual_seed(seed + 1)
    traj = (torch.rand(B, 1, 2, generator=gen) + torch.cumsum(0.01 * torch.randn(B, Lq, 2, generator=gen), 1)).to(DEV).requires_grad_(True)
    dts = (0.5 + 0.5 * torch.rand(B, Lq, generator=gen)).to(DEV)
    lengths = torch.full((B,), Lq)
    if ragged:
        lengths = torch.randint(2, Lq + 1, (B,), generator=gen)
        lengths[0] = Lq
    old = ops.LSTM_STACK_FUSED_DX
    ops.LSTM_STACK_FUSED_DX = fused
    try:
        score = D(traj, dts, lengths, dropout_seed=77)
        w = torch.randn(B, 1, generator=gen).to(DEV)
        (score * w).sum().backward()
    finally:
        ops.LSTM_STACK_FUSED_DX = old
    return score.detach(), traj.grad.detach(), {k: v.grad.detach() for k, v in D.named_parameters()}


@pytest.mark.parametrize('B,Lq,ragged,p_drop', [(64, 9, False, 0.0), (50, 17, True, 0.0), (96, 12, True, 0.1), (32, 1, False, 0.0)])
def test_fused_stack_matches_the_layerwise_path(B, Lq, ragged, p_drop):
    s1, gx1, g1 = _run(True, B, Lq, ragged, p_drop, 5)
    s0, gx0, g0 = _run(False, B, Lq, ragged, p_drop, 5)
    assert torch.equal(s1, s0), 'same forward kernels: scores must be bit-identical'

    def close(a, b, what):
        scale = max(float(b.abs().max()), 1e-12)
        err = float((a - b).abs().max()) / scale
        assert err <= 2e-5, f'{what}: {err:.3e}'
    close(gx1, gx0, 'd trajectories')
    for k in g0:
        close(g1[k], g0[k], k)


def test_fused_stack_is_the_path_taken_at_h128():
    from generative_mouse_trajectories_amd import _lib as L_
    L_.profile_start()
    _run(True, 32, 5, False, 0.0, 3)
    torch.cuda.synchronize()
    prof = L_.profile_stop()
    assert 'lstm_bwd_fx' in prof and 'dx_gemm_K512_N256' not in prof, sorted(prof)
