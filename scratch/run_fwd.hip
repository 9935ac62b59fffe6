// plain launch of the dominant kernel at the bench shape, for rocprofv3 --pmc passes (no timing hooks)
#include "../generative_mouse_trajectories_amd/csrc/lstm_seq.hip"
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main()
{
    const int L = 129, B = 4096, H = 128, ndir = 2, I = 256;
    float *x, *wih, *whh, *bias, *gates, *c, *h, *hf;
    CK(hipMalloc(&x, (size_t)L * B * I * 4)); CK(hipMemset(x, 0, (size_t)L * B * I * 4));
    CK(hipMalloc(&wih, (size_t)ndir * 4 * H * I * 4)); CK(hipMemset(wih, 0, (size_t)ndir * 4 * H * I * 4));
    CK(hipMalloc(&whh, (size_t)ndir * 4 * H * H * 4)); CK(hipMemset(whh, 0, (size_t)ndir * 4 * H * H * 4));
    CK(hipMalloc(&bias, (size_t)ndir * 4 * H * 4)); CK(hipMemset(bias, 0, (size_t)ndir * 4 * H * 4));
    CK(hipMalloc(&gates, (size_t)ndir * L * B * 4 * H * 4));
    CK(hipMalloc(&c, (size_t)ndir * L * B * H * 4));
    CK(hipMalloc(&h, (size_t)L * B * ndir * H * 4));
    CK(hipMalloc(&hf, (size_t)B * ndir * H * 4));
    for (int rep = 0; rep < 2; ++rep) {
        int rc = gmt_lstm_layer_fwd(x, nullptr, wih, whh, bias, nullptr, gates, c, h, hf, L, B, I, H, ndir, 0.1f, 7u, 0);   // inter-layer dropout on, as in the bench
        if (rc) { printf("rc %d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
    }
    printf("done\n");
    return 0;
}
