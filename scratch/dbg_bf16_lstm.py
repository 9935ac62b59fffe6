"""bf16 per-timestep LSTM layer (csrc/lstm_bf16.hip) against a plain torch fp32 loop on the same bf16-rounded operands."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_mouse_trajectories_amd import _lib as L_

dev = torch.device('cuda:0')
torch.manual_seed(0)

def slab_perm(H):
    c = torch.arange(4 * H)
    s, g, u = c // 128, (c % 128) // 32, c % 32
    return (g * H + s * 32 + u).to(dev)

def ref_layer(pre, whh, lengths, L, B, H):
    """pre [L][B][2][4H] (original gate order) fp32; whh [2][4H][H]; returns h_all [L][B][2H], hfin [2][B][H]"""
    outs = [[None] * L, [None] * L]
    fin = []
    for d in range(2):
        h = torch.zeros(B, H, device=dev); c = torch.zeros(B, H, device=dev)
        ts = range(L) if d == 0 else range(L - 1, -1, -1)
        for t in ts:
            a = pre[t, :, d] + h @ whh[d].t()
            i, f, g, o = a.split(H, dim=1)
            cn = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            hn = torch.sigmoid(o) * torch.tanh(cn)
            m = (t < lengths).float().unsqueeze(1)
            c = m * cn + (1 - m) * c
            h = m * hn + (1 - m) * h
            outs[d][t] = h
        fin.append(h)
    return torch.stack([torch.cat([outs[0][t], outs[1][t]], 1) for t in range(L)]), torch.stack(fin)

def run(L, B, H, ragged):
    perm = slab_perm(H)
    pre = (torch.randn(L, B, 2, 4 * H, device=dev)).bfloat16().float().requires_grad_(True)
    whh = (torch.randn(2, 4 * H, H, device=dev) / H ** 0.5).bfloat16().float().requires_grad_(True)
    lengths = torch.randint(1, L + 1, (B,), device=dev, dtype=torch.int32) if ragged else torch.full((B,), L, device=dev, dtype=torch.int32)
    h_ref, fin_ref = ref_layer(pre, whh, lengths, L, B, H)
    gfin = torch.randn(2, B, H, device=dev)
    gh = torch.randn(L, B, 2 * H, device=dev).bfloat16().float() * 0.1
    # padded positions of the reference output carry the frozen state; only valid positions get an output gradient
    mask = (torch.arange(L, device=dev).view(L, 1) < lengths.view(1, B)).float().unsqueeze(-1)
    loss = (fin_ref * gfin).sum() + (h_ref * gh * mask).sum()
    gpre, gwhh = torch.autograd.grad(loss, [pre, whh])

    G = pre.detach()[:, :, :, perm].reshape(L, B, 8 * H).bfloat16().contiguous()
    c_all = torch.empty(L, B, 2 * H, device=dev); h_all = torch.empty(L, B, 2 * H, device=dev, dtype=torch.bfloat16)
    wpk = whh.detach()[:, perm, :].bfloat16().contiguous()
    L_.call('gmt_bf16_lstm_layer_fwd', L_.ptr(G), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(wpk), L_.ptr(lengths), int(ragged), L, B, H, 2, L_.stream())
    torch.cuda.synchronize()
    hh = h_all.float()
    # compare valid positions (direction 1 keeps zeros where the reference keeps nothing meaningful)
    e_h = ((hh - h_ref.detach()) * mask).abs().max().item()
    fin = torch.stack([hh[L - 1, :, :H], hh[0, :, H:]])
    e_f = (fin - fin_ref.detach()).abs().max().item()
    print(f'L={L} B={B} H={H} ragged={ragged}: fwd max abs err h {e_h:.3e} final {e_f:.3e}')
    wT = whh.detach()[:, perm, :].transpose(1, 2).bfloat16().contiguous()      # [2][H][4H]
    dcc = torch.empty(2, B, H, device=dev); dhc = torch.empty(2, B, H, device=dev)
    gh_in = (gh * mask).bfloat16().contiguous()
    L_.call('gmt_bf16_lstm_layer_bwd', L_.ptr(G), L_.ptr(c_all), L_.ptr(gh_in), L_.ptr(gfin.contiguous()), L_.ptr(dcc), L_.ptr(dhc),
            L_.ptr(wT), L_.ptr(lengths), int(ragged), L, B, H, 2, L_.stream())
    torch.cuda.synchronize()
    dA = G.float().view(L, B, 2, 4 * H)
    ref_dA = gpre[:, :, :, perm]
    e_a = (dA - ref_dA).abs().max().item() / ref_dA.abs().max().item()
    print(f'      bwd dA max rel err {e_a:.3e}   (|dA|max {ref_dA.abs().max().item():.3f})')
    assert e_h < 3e-2 and e_f < 3e-2 and e_a < 3e-2

run(5, 256, 128, False)
run(7, 512, 256, True)
run(4, 256, 512, True)
# rate at the config-3 shape (B = 8192 rows: real + fake), a slice of 33 steps
L, B, H = 33, 8192, 512
G = torch.randn(L, B, 8 * H, device=dev).bfloat16(); c_all = torch.empty(L, B, 2 * H, device=dev); h_all = torch.empty(L, B, 2 * H, device=dev, dtype=torch.bfloat16)
wpk = (torch.randn(2, 4 * H, H, device=dev) / H ** 0.5).bfloat16(); wT = wpk.transpose(1, 2).contiguous()
lengths = torch.full((B,), L, device=dev, dtype=torch.int32)
gfin = torch.randn(2, B, H, device=dev); dcc = torch.empty(2, B, H, device=dev)
for name, fn in [('fwd', lambda: L_.call('gmt_bf16_lstm_layer_fwd', L_.ptr(G), L_.ptr(c_all), L_.ptr(h_all), L_.ptr(wpk), None, 0, L, B, H, 2, L_.stream())),
                 ('bwd', lambda: L_.call('gmt_bf16_lstm_layer_bwd', L_.ptr(G), L_.ptr(c_all), None, L_.ptr(gfin), L_.ptr(dcc), None, L_.ptr(wT), None, 0, L, B, H, 2, L_.stream()))]:
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    fl = 2.0 * 2 * B * 4 * H * H * (L - 1)
    print(f'{name}: {t / L * 1e6:.1f} us per step, {fl / t / 1e12:.0f} TFLOP/s (recurrent contraction only)')
