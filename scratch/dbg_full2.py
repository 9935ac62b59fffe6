import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import torch, numpy as np
import helpers as H
from oracle import gan_oracle as O
from test_gpu_parity import build, DEV
from test_gpu_fullsize import _batch, B, T
gmt, cfg, G, D = build(H.MID, 5, 6)
G.eval(); D.eval()
starts, ends, seq, lengths, z = _batch(1)
with torch.no_grad():
    full, _ = G(starts.to(DEV), ends.to(DEV), z.to(DEV), target_sequences=seq.to(DEV), target_lengths=lengths.to(DEV), tf_mask=[False] * T)
    traj = gmt.trajectory_to_absolute(starts.to(DEV), full[:, :, :2])
    dpad = torch.nn.functional.pad(full[:, :, 2], (1, 0), value=0.001)
    s_full = D(traj, dpad, (lengths + 1).to(DEV)).cpu()
r = slice(0, 32)
oc, Pg, Pd, Bd = H.oracle_models(H.MID, 5, 6)
g_seq = full[r].cpu()
o_traj = O.to_absolute(starts[r], g_seq[:, :, :2]); o_dpad = torch.nn.functional.pad(g_seq[:, :, 2], (1, 0), value=0.001)
o_s = O.discriminator_forward(Pd, Bd, o_traj, o_dpad, lengths[r] + 1, training=False)
print('traj equal', float((traj[r].cpu() - o_traj).abs().max()), 'dpad equal', float((dpad[r].cpu() - o_dpad).abs().max()))
err = (s_full[r] - o_s).abs().squeeze() / o_s.abs().max()
for i in range(32): print(i, int(lengths[i]), float(s_full[i]), float(o_s[i]), f'{float(err[i]):.2e}')
