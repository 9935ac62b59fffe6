import sys, torch
sys.path.insert(0, '.')
from generative_mouse_trajectories_amd import ops
R, N, I = 4096 * 128, 512, 128
dA = torch.randn(R, N, device='cuda'); X = torch.randn(R, 256, device='cuda')[:, :I]
for _ in range(3):
    dW, db = ops.wgrad(dA, X, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(5): ops.wgrad(dA, X, True)
e1.record(); torch.cuda.synchronize()
print('ms', e0.elapsed_time(e1) / 5)
