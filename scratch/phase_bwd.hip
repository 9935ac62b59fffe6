// phase timing of lstm_bwd_kernel<4,true> (tuning aid, not part of the library)
#define GMT_PHASE_TIMING
#include "../generative_mouse_trajectories_amd/csrc/lstm_seq.hip"
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main()
{
    const int L = 129, B = 4096, H = 128, ndir = 2;
    float *gates, *c, *w, *dh, *dhf;
    CK(hipMalloc(&gates, (size_t)ndir * L * B * 4 * H * 4)); CK(hipMemset(gates, 0, (size_t)ndir * L * B * 4 * H * 4));
    CK(hipMalloc(&c, (size_t)ndir * L * B * H * 4)); CK(hipMemset(c, 0, (size_t)ndir * L * B * H * 4));
    CK(hipMalloc(&w, (size_t)ndir * 4 * H * H * 4)); CK(hipMemset(w, 0, (size_t)ndir * 4 * H * H * 4));
    CK(hipMalloc(&dh, (size_t)L * B * ndir * H * 4)); CK(hipMemset(dh, 0, (size_t)L * B * ndir * H * 4));
    CK(hipMalloc(&dhf, (size_t)B * ndir * H * 4)); CK(hipMemset(dhf, 0, (size_t)B * ndir * H * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        int rc = gmt_lstm_layer_bwd(gates, c, w, nullptr, dh, dhf, nullptr, nullptr, nullptr, nullptr, nullptr, L, B, H, ndir, 0.f, 0u, 0);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        if (rc) { printf("rc %d\n", rc); return 1; }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[8]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(gmt_phase_ticks), sizeof(t)));
        double tot = 0; for (int i = 0; i < 6; ++i) tot += t[i];
        printf("%.3f ms  ticks/step: elementwise %.0f  barrier1 %.0f  prefetch-issue %.0f  mfma %.0f  barrier2 %.0f  top %.0f  total %.0f (%.2f ticks/ns)\n",
               ms, t[0] / (double)L, t[1] / (double)L, t[2] / (double)L, t[3] / (double)L, t[4] / (double)L, t[5] / (double)L, tot / L, tot / (ms * 1e6));
    }
    return 0;
}
