// phase timing of lstm_fwd_kernel<4,true,IX> (tuning aid, not part of the library)
#define GMT_PHASE_TIMING
#include "../generative_mouse_trajectories_amd/csrc/lstm_seq.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void fill_rand(float* p, size_t n, float scale, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = scale * ((float)(gmt_hash32((unsigned)i + seed) >> 8) * (1.0f / 8388608.0f) - 1.0f);
}

int main(int argc, char** argv)
{
    const bool rnd = argc > 1;   // any argument: random data instead of zeros (clock under real switching activity)

    const int L = 129, B = 4096, H = 128, ndir = 2;
    for (int IX = 1; IX <= 2; ++IX) {
        const int I = IX * H;
        float *x, *wih, *whh, *bias, *gates, *c, *h, *hf;
        CK(hipMalloc(&x, (size_t)L * B * I * 4)); CK(hipMemset(x, 0, (size_t)L * B * I * 4));
        CK(hipMalloc(&wih, (size_t)ndir * 4 * H * I * 4)); CK(hipMemset(wih, 0, (size_t)ndir * 4 * H * I * 4));
        CK(hipMalloc(&whh, (size_t)ndir * 4 * H * H * 4)); CK(hipMemset(whh, 0, (size_t)ndir * 4 * H * H * 4));
        CK(hipMalloc(&bias, (size_t)ndir * 4 * H * 4)); CK(hipMemset(bias, 0, (size_t)ndir * 4 * H * 4));
        CK(hipMalloc(&gates, (size_t)ndir * L * B * 4 * H * 4));
        CK(hipMalloc(&c, (size_t)ndir * L * B * H * 4));
        CK(hipMalloc(&h, (size_t)L * B * ndir * H * 4));
        CK(hipMalloc(&hf, (size_t)B * ndir * H * 4));
        if (rnd) {
            fill_rand<<<2048, 256>>>(x, (size_t)L * B * I, 0.5f, 1u);
            fill_rand<<<256, 256>>>(wih, (size_t)ndir * 4 * H * I, 0.09f, 2u);
            fill_rand<<<256, 256>>>(whh, (size_t)ndir * 4 * H * H, 0.09f, 3u);
            fill_rand<<<8, 256>>>(bias, (size_t)ndir * 4 * H, 0.09f, 4u);
            CK(hipDeviceSynchronize());
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            int rc = gmt_lstm_layer_fwd(x, nullptr, wih, whh, bias, nullptr, gates, c, h, hf, L, B, I, H, ndir, 0.f, 0u, 0);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            if (rc) { printf("rc %d\n", rc); return 1; }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long t[8]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(gmt_phase_ticks), sizeof(t)));
            double tot = 0; for (int i = 0; i < 6; ++i) tot += t[i];
            printf("IX=%d  %.3f ms  ticks/step: x-part %.0f  h-part %.0f  elementwise %.0f  x_store %.0f  barrier %.0f  top %.0f  total %.0f (%.2f ticks/ns)\n",
                   IX, ms, t[0] / (double)L, t[1] / (double)L, t[2] / (double)L, t[3] / (double)L, t[4] / (double)L, t[5] / (double)L,
                   tot / L, tot / (ms * 1e6));
        }
        hipFree(x); hipFree(wih); hipFree(whh); hipFree(bias); hipFree(gates); hipFree(c); hipFree(h); hipFree(hf);
    }
    return 0;
}
