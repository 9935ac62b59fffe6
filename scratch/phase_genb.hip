// phase timing of gen_bwd_kernel<4,*> (tuning aid, not part of the library)
#define GMT_PHASE_TIMING
#include "../generative_mouse_trajectories_amd/csrc/gen_rollout.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static float* dz(size_t n, float v = 0.f) { float* p; CK(hipMalloc(&p, n * 4)); if (v == 0.f) CK(hipMemset(p, 0, n * 4)); else { std::vector<float> h(n, v); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); } return p; }
int main()
{
    const int T = 128, B = 4096, H = 128, HH = 64;
    gmt_gen_bwd_args a; memset(&a, 0, sizeof(a));
    a.gates0 = dz((size_t)T * B * 4 * H); a.c0_all = dz((size_t)T * B * H); a.gates1 = dz((size_t)T * B * 4 * H); a.c1_all = dz((size_t)T * B * H);
    a.y_all = dz((size_t)T * B * HH); a.oraw_all = dz((size_t)T * B); a.u_all = dz((size_t)T * B * 8, 1.f); a.d_out = dz((size_t)T * B * 3);
    a.w1T_pk = dz((size_t)H * H); a.w_l1T_pk = dz((size_t)4 * H * 2 * H); a.w_hh0T_pk = dz((size_t)4 * H * H); a.w_uP = dz(4 * H * 8); a.w2 = dz(3 * HH);
    a.d_ypre = dz((size_t)T * B * HH); a.d_o = dz((size_t)T * B * 3); a.d_in0 = dz((size_t)B * 3);
    a.T = T; a.B = B; a.H = H; a.dropout_p = 0.1f; a.seed = 1234;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* nm[8] = {"B1", "B2", "B3", "B4-issue", "B4", "B5", "B6", "B7+top"};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        int rc = gmt_gen_rollout_bwd(&a, 0);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        if (rc) { printf("rc %d\n", rc); return 1; }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[8]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(gmt_phase_ticks), sizeof(t)));
        double tot = 0; for (int i = 0; i < 8; ++i) tot += t[i];
        printf("%.3f ms  ticks/step:", ms);
        for (int i = 0; i < 8; ++i) printf("  %s %.0f", nm[i], t[i] / (double)T);
        printf("  total %.0f (%.2f ticks/ns)\n", tot / T, tot / (ms * 1e6));
    }
    return 0;
}
