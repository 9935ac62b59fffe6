import torch, time
dev = 'cuda:0'
R = 4096 * 129
for lib in ('cublaslt', 'cublas'):
    torch.backends.cuda.preferred_blas_library(lib)
    for N, K in ((256, 512), (128, 512)):
        a = torch.randn(R, K, device=dev); w = torch.randn(K, N, device=dev)
        out = torch.empty(R, N, device=dev)
        for _ in range(3): torch.mm(a, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10): torch.mm(a, w, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(lib, N, K, f'{ms:.3f} ms', f'{2 * R * N * K / ms / 1e9:.1f} TF/s')
        # addmm_ accumulate form used for the second direction
        for _ in range(3): out.addmm_(a, w)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10): out.addmm_(a, w)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(lib, N, K, 'addmm_', f'{ms:.3f} ms', f'{2 * R * N * K / ms / 1e9:.1f} TF/s')
