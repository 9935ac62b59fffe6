"""bf16 GEMM (csrc/gemm_bf16.hip) against torch, and its rate on the config-3 shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_mouse_trajectories_amd import _lib as L_

dev = torch.device('cuda:0')
torch.manual_seed(0)

def gemm(A, W, bias=None, act=None, epi=0, slope=0.2):
    M, K = A.shape; N = W.shape[0]
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    L_.call('gmt_bf16_gemm_nt', L_.ptr(A), A.stride(0), L_.ptr(W), W.stride(0), L_.ptr(C), N,
            L_.ptr(bias) if bias is not None else None, L_.ptr(act) if act is not None else None,
            act.stride(0) if act is not None else 0, epi, slope, M, N, K, L_.stream())
    return C

for (M, N, K) in [(256, 256, 64), (512, 768, 192), (4096, 2048, 512)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev)
    ref = A.float() @ W.float().t() + b
    got = gemm(A, W, b).float()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f'M={M} N={N} K={K} bias: max rel err {err:.2e}')
    assert err < 1e-2
    got = gemm(A, W, b, epi=1).float()
    r1 = torch.nn.functional.leaky_relu(ref, 0.2)
    print('   lrelu', ((got - r1).abs().max() / r1.abs().max()).item())
    act = torch.randn(M, N, device=dev).bfloat16()
    got = gemm(A, W, None, act, epi=2).float()
    r2 = (A.float() @ W.float().t()) * torch.where(act.float() > 0, 1.0, 0.2)
    print('   lrelu-deriv', ((got - r2).abs().max() / r2.abs().max()).item())

for (M, N, K) in [(4096 * 64, 4096, 512), (4096 * 64, 4096, 1024), (4096 * 64, 512, 4096), (4096 * 64, 1024, 4096), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    for _ in range(2): gemm(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gemm(A, W)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    e0.record()
    for _ in range(5): torch.matmul(A, W.t())
    e1.record(); torch.cuda.synchronize()
    t2 = e0.elapsed_time(e1) / 5 * 1e-3
    print(f'M={M} N={N} K={K}: {t*1e3:.3f} ms = {2*M*N*K/t/1e12:.0f} TFLOP/s   (torch/hipBLASLt {2*M*N*K/t2/1e12:.0f})')

def wgrad(dA, X, S):
    M, N = dA.shape; I = X.shape[1]
    slabs = torch.empty(S, N, I, device=dev)
    L_.call('gmt_bf16_wgrad_tn', L_.ptr(dA), dA.stride(0), L_.ptr(X), X.stride(0), L_.ptr(slabs), M, N, I, S, L_.stream())
    return slabs.sum(0)

for (M, N, I, S) in [(64, 256, 256, 1), (640, 512, 256, 3), (8192, 256, 512, 4)]:
    dA = torch.randn(M, N, device=dev).bfloat16(); X = torch.randn(M, I, device=dev).bfloat16()
    ref = dA.float().t() @ X.float()
    got = wgrad(dA, X, S)
    print(f'wgrad M={M} N={N} I={I} S={S}: max rel err {((got - ref).abs().max() / ref.abs().max()).item():.2e}')
for (M, N, I, S) in [(4096 * 64, 4096, 512, 8), (4096 * 64, 4096, 1024, 4), (4096 * 64, 2048, 512, 16)]:
    dA = torch.randn(M, N, device=dev).bfloat16(); X = torch.randn(M, I, device=dev).bfloat16()
    for _ in range(2): wgrad(dA, X, S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): wgrad(dA, X, S)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 3 * 1e-3
    print(f'wgrad M={M} N={N} I={I} S={S}: {t*1e3:.3f} ms = {2*M*N*I/t/1e12:.0f} TFLOP/s')
