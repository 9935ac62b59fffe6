// Where does a forward-step workgroup spend its time?  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGMT_PHASE_TIMING
//   -I generative_mouse_trajectories_amd/csrc scratch/phase_bf16.hip -o scratch/phase_bf16 ; ./scratch/phase_bf16 [B] [groups]
#include "../generative_mouse_trajectories_amd/csrc/lstm_bf16.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 4096, H = 512, L = 12, ndir = 2;
    if (argc > 2) gmt_bf16_set_row_groups(atoi(argv[2]));
    void *gates, *hall, *hdrop, *whh; float* call;
    CK(hipMalloc(&gates, (size_t)L * B * ndir * 4 * H * 2)); CK(hipMemset(gates, 0, (size_t)L * B * ndir * 4 * H * 2));
    CK(hipMalloc(&hall, (size_t)L * B * ndir * H * 2)); CK(hipMalloc(&hdrop, (size_t)L * B * ndir * H * 2));
    CK(hipMalloc(&call, (size_t)L * B * ndir * H * 4));
    CK(hipMalloc(&whh, (size_t)ndir * 4 * H * H * 2)); CK(hipMemset(whh, 0, (size_t)ndir * 4 * H * H * 2));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 3; ++rep) {
        int rc = gmt_bf16_lstm_layer_fwd(gates, call, hall, hdrop, whh, nullptr, 0, L, B, H, ndir, 0.1f, 5u, st);
        if (rc) { printf("rc %d\n", rc); return 1; }
        CK(hipStreamSynchronize(st));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    gmt_bf16_lstm_layer_fwd(gates, call, hall, hdrop, whh, nullptr, 0, L, B, H, ndir, 0.1f, 5u, st);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B=%d: %.1f us per step\n", B, ms * 1e3 / L);
    {   // the backward sweep over the same buffers (gates are overwritten with d(pre-activation)): timing, and for the PMC passes
        void* dhout; float *dhfin, *dcc, *dhc; void* whhT;
        CK(hipMalloc(&dhout, (size_t)L * B * ndir * H * 2)); CK(hipMemset(dhout, 0, (size_t)L * B * ndir * H * 2));
        CK(hipMalloc(&dhfin, (size_t)ndir * B * H * 4)); CK(hipMemset(dhfin, 0, (size_t)ndir * B * H * 4));
        CK(hipMalloc(&dcc, (size_t)ndir * B * H * 4)); CK(hipMalloc(&dhc, (size_t)ndir * B * H * 4));
        CK(hipMalloc(&whhT, (size_t)ndir * 4 * H * H * 2)); CK(hipMemset(whhT, 0, (size_t)ndir * 4 * H * H * 2));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            int rc = gmt_bf16_lstm_layer_bwd(gates, call, dhout, dhfin, dcc, dhc, whhT, nullptr, 0, L, B, H, ndir, 0.1f, 5u, st);
            if (rc) { printf("bwd rc %d\n", rc); return 1; }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("backward: %.1f us per step\n", ms * 1e3 / L);
    }
    std::vector<long long> buf(4096 * 16);
    CK(hipMemcpyFromSymbol(buf.data(), HIP_SYMBOL(gmt_phase_buf), buf.size() * 8));
    const int nb = std::min(4096, ndir * (B / 128) * 8);
    // last step's stamps (group 0's blocks overwrite group 1's: same block indices); wall_clock64 ticks at 100 MHz
    long long t0 = buf[0]; for (int b = 0; b < nb; ++b) if (buf[b * 16] && buf[b * 16] < t0) t0 = buf[b * 16];
    const char* names[10] = {"start", "mainloop done", "half0 in", "half0 computed", "half0 stores issued", "half0 stores done",
                             "half1 in", "half1 computed", "half1 stores issued", "half1 stores done"};
    for (int p = 0; p < 10; ++p) {
        double sum = 0, mx = 0, mn = 1e18; int n = 0;
        for (int b = 0; b < nb; ++b) { if (!buf[b * 16]) continue; const double v = (buf[b * 16 + p] - t0) * 0.01; sum += v; mx = std::max(mx, v); mn = std::min(mn, v); ++n; }
        printf("%-22s mean %7.2f us  min %7.2f  max %7.2f  (%d blocks)\n", names[p], sum / n, mn, mx, n);
    }
    return 0;
}
