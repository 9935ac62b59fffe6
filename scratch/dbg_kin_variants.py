"""kinematics kernels of a library variant (tuning): python scratch/dbg_kin_variants.py scratch/libgmt_kch16.so"""
import sys, torch
sys.path.insert(0, '.')
from generative_mouse_trajectories_amd import _lib as L_
L_.LIB_PATH = sys.argv[1]
from generative_mouse_trajectories_amd import ops
dev = 'cuda:0'; L, Bb = 129, 1 << 18
posb = torch.rand(L, Bb, 2, device=dev); dtsb = torch.rand(L, Bb, device=dev) * 0.02 + 0.004
gfeat = torch.randn(L, Bb, 9, device=dev); gpos = torch.empty(L, Bb, 2, device=dev); gdts = torch.empty(L, Bb, device=dev)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: ops.kinematics_tm(posb, dtsb)); print(sys.argv[1], f'fwd {t*1e6:.0f} us {48*Bb*L/t/1e12:.2f} TB/s', end='  ')
t = timed(lambda: L_.call('gmt_kinematics_bwd', L_.ptr(posb), L_.ptr(dtsb), L_.ptr(gfeat), L_.ptr(gpos), L_.ptr(gdts), L, Bb, 1e-8, L_.stream()))
print(f'bwd {t*1e6:.0f} us {60*Bb*L/t/1e12:.2f} TB/s')
