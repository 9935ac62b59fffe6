import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import torch, numpy as np
import helpers as H
from oracle import gan_oracle as O
from test_gpu_parity import build, DEV
from generative_mouse_trajectories_amd import ops
gmt, cfg, G, D = build(H.MID, 5, 6)
D.eval()
B, T = 64, 128
g = torch.Generator().manual_seed(1)
starts = torch.rand(B, 2, generator=g); deltas = torch.randn(B, T, 2, generator=g) * 0.01
dts = torch.rand(B, T, generator=g) * 0.02 + 0.004
lengths = torch.randint(T // 2, T + 1, (B,), generator=g); lengths[:8] = T
if len(sys.argv) > 1:
    G.eval()
    ends = starts + deltas.sum(1); z = torch.randn(B, 64, generator=g)
    with torch.no_grad():
        full, _ = G(starts.to(DEV), ends.to(DEV), z.to(DEV), target_sequences=torch.cat([deltas, dts.unsqueeze(-1)], -1).to(DEV), target_lengths=lengths.to(DEV), tf_mask=[False] * T)
    full = full.cpu(); deltas = full[:, :, :2].contiguous(); dts = full[:, :, 2].contiguous()
    print('G out: |dx| max', float(deltas.abs().max()), 'dt range', float(dts.min()), float(dts.max()))
traj = gmt.trajectory_to_absolute(starts, deltas); dpad = torch.nn.functional.pad(dts, (1, 0), value=0.001)
with torch.no_grad():
    s = D(traj.to(DEV), dpad.to(DEV), (lengths + 1).to(DEV)).cpu()
    feat = ops.kinematics_tm(traj.to(DEV).transpose(0, 1).contiguous(), dpad.to(DEV).t().contiguous()).transpose(0, 1).cpu()
for dt in (torch.float32, torch.float64):
    oc, Pg, Pd, Bd = H.oracle_models(H.MID, 5, 6, dt)
    o = O.discriminator_forward(Pd, Bd, traj.to(dt), dpad.to(dt), lengths + 1, training=False)
    f = O.kinematics(traj.to(dt), dpad.to(dt))
    print('feat abs max per column', [float(f[..., c].abs().max()) for c in range(9)])
    print(dt, 'score relerr', H.relerr(s, o), 'max|score|', float(o.abs().max()), 'feat relerr per column',
          [round(H.relerr(feat[..., c], f[..., c]), 9) for c in range(9)])
o32 = O.discriminator_forward(*H.oracle_models(H.MID, 5, 6, torch.float32)[2:], traj, dpad, lengths + 1, training=False)
o64 = O.discriminator_forward(*H.oracle_models(H.MID, 5, 6, torch.float64)[2:], traj.double(), dpad.double(), lengths + 1, training=False)
print('oracle32 vs oracle64', H.relerr(o32, o64))
