import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch, numpy as np
import helpers as H
import test_gpu_parity as TP
from generative_mouse_trajectories_amd import ops
import generative_mouse_trajectories_amd.ops as OPS
gmt, cfg, G, D = TP.build(H.TINY, 31, 32)
cfg.lstm_dropout = 0.25
B, T = 6, 9
b = TP.dev_batch(H.batch_t(B, T, 33, [9, 8, 7, 5, 3, 2]))
z = torch.randn(B, 8).cuda()
saved = {}
orig = OPS.GenRolloutFn.forward
seq, _ = G(b['starts'], b['ends'], z, target_sequences=b['sequences'], target_lengths=b['lengths'], tf_mask=[False]*T, dropout_seed=12345)
fn = seq.grad_fn
# walk to GenRolloutFnBackward
def find(f, depth=0):
    if f is None or depth > 6: return None
    if 'GenRollout' in type(f).__name__: return f
    for nf, _ in f.next_functions:
        r = find(nf, depth + 1)
        if r is not None: return r
node = find(fn)
st = node.saved_tensors
h0, h0d = st[9], st[-1]
print(h0.shape, h0d.shape)
dm = ops.dropout_keep_mask(T, 32, 32, 0.25, 12345).cuda()
kern_keep = (h0d != 0)
py_keep = (dm > 0)
print('keep frac kernel', kern_keep.float().mean().item(), 'python', py_keep.float().mean().item())
print('mismatch', (kern_keep != py_keep).float().mean().item())
print('value err', ((h0 * dm) - h0d).abs().max().item())
from oracle import gan_oracle as O
bc = H.batch_t(B, T, 33, [9, 8, 7, 5, 3, 2])
Pg = {k: v.detach().cpu().double() for k, v in G.state_dict().items()}
for name, mask_tf in (('none', [False]*T), ('mix', [False, True]*4 + [False]), ('all', [True]*T)):
    for p in (0.0, 0.25):
        cfg.lstm_dropout = p
        with torch.no_grad():
            seq, _ = G(b['starts'], b['ends'], z, target_sequences=b['sequences'], target_lengths=b['lengths'], tf_mask=mask_tf, dropout_seed=12345)
        dmm = ops.dropout_keep_mask(T, 32, 32, p, 12345)[:, :B].double() if p > 0 else None
        ref = O.generator_forward(Pg, bc['starts'].double(), bc['ends'].double(), z.cpu().double(), bc['sequences'].double(), mask_tf, dropout_mask=dmm)
        print(name, p, H.relerr(seq.cpu().numpy(), ref.numpy()))
