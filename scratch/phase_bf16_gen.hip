// Phase timing of the generator's step kernels (see phase_bf16.hip).  ./scratch/phase_bf16_gen [B] [groups]
#include "../generative_mouse_trajectories_amd/csrc/gen_bf16.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static void* dz(size_t n) { void* p = nullptr; if (hipMalloc(&p, n) != hipSuccess || hipMemset(p, 0, n) != hipSuccess) { printf("alloc\n"); exit(1); } return p; }
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096, H = 512, T = 10;
    if (argc > 2) row_groups_setting().store(atoi(argv[2]));
    gmt_bf16_gen_args a = {};
    a.cond_pre = dz((size_t)B * 4 * H * 2); a.w0 = dz((size_t)4 * H * (H + 64) * 2); a.w1 = dz((size_t)4 * H * 2 * H * 2);
    a.bias1 = (float*)dz(4 * H * 4); a.wh1 = dz((size_t)(H / 2) * H * 2); a.bh1 = (float*)dz(H / 2 * 4); a.wh2 = (float*)dz(3 * H / 2 * 4);
    a.bh2 = (float*)dz(16); a.ends = (float*)dz((size_t)B * 8); a.out_seq = (float*)dz((size_t)T * B * 12);
    a.x0 = dz((size_t)2 * B * (H + 64) * 2); a.x1 = dz((size_t)2 * B * 2 * H * 2); a.c0 = (float*)dz((size_t)2 * B * H * 4);
    a.c1 = (float*)dz((size_t)2 * B * H * 4); a.pos = (float*)dz((size_t)B * 16);
    a.T = T; a.B = B; a.H = H; a.dropout_p = 0.1f; a.seed = 3;
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 3; ++rep) { if (gmt_bf16_gen_rollout_fwd(&a, st)) { printf("rc\n"); return 1; } CK(hipStreamSynchronize(st)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    gmt_bf16_gen_rollout_fwd(&a, st);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B=%d: %.1f us per step\n", B, ms * 1e3 / T);
    std::vector<long long> buf(4096 * 16);
    CK(hipMemcpyFromSymbol(buf.data(), HIP_SYMBOL(gmt_phase_buf), buf.size() * 8));
    const int nb = B / 128 * 8;
    long long t0 = buf[0]; for (int b = 0; b < nb; ++b) if (buf[b * 16] && buf[b * 16] < t0) t0 = buf[b * 16];
    const char* names[14] = {"cell1 start", "mainloop done", "half0 in", "half0 computed", "half0 stores issued", "half0 stores done",
                             "half1 in", "half1 computed", "half1 stores issued", "half1 stores done",
                             "head start", "head mainloop done", "head reduced", "head done"};
    for (int p = 0; p < 14; ++p) {
        double sum = 0, mx = 0, mn = 1e18; int n = 0;
        for (int b = 0; b < nb; ++b) { if (!buf[b * 16 + p]) continue; const double v = (buf[b * 16 + p] - t0) * 0.01; sum += v; mx = std::max(mx, v); mn = std::min(mn, v); ++n; }
        printf("%-22s mean %7.2f us  min %7.2f  max %7.2f  (%d blocks)\n", names[p], sum / n, mn, mx, n);
    }
    // experiment: every workgroup contracts the SAME tile (all operands L2-hot, no fabric traffic): CU-internal limit of the loop
    {
        hipFuncSetAttribute((const void*)gen_cell_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i)
                hipLaunchKernelGGL(gen_cell_kernel, dim3(256), dim3(G_CELL_THREADS), G_LDS, st, (const __bf16*)a.w1, (const __bf16*)a.x1, 1024LL, 32,
                                   (const __bf16*)nullptr, a.bias1, a.c1, a.c1 + (size_t)B * H, (__bf16*)a.x1 + (size_t)B * 1024 + 512, 1024LL,
                                   (__bf16*)nullptr, 0LL, (__bf16*)nullptr, H, 0, 1, 1, 0.f, 0u, 0LL);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("same-tile cell launch (K=1024): %.1f us per launch\n", ms * 1e3 / 20);
        CK(hipMemcpyFromSymbol(buf.data(), HIP_SYMBOL(gmt_phase_buf), buf.size() * 8));
        double m0 = 0, m1 = 0; for (int b = 0; b < 256; ++b) { m0 += buf[b * 16] * 0.01; m1 += buf[b * 16 + 1] * 0.01; }
        printf("  mainloop %.2f us (16 tiles of 64)\n", (m1 - m0) / 256);
        std::vector<long long> cyc(4096 * 16);
        CK(hipMemcpyFromSymbol(cyc.data(), HIP_SYMBOL(gmt_phase_cyc), cyc.size() * 8));
        double c = 0; for (int b = 0; b < 256; ++b) c += double(cyc[b * 16 + 1] - cyc[b * 16]);
        printf("  mainloop %.0f shader cycles -> %.0f MHz\n", c / 256, c / 256 / ((m1 - m0) / 256));
    }
    return 0;
}
