// Building blocks of the persistent LSTM kernels (gfx950, wave64, fp32-input MFMA).
//
// Work decomposition shared by every recurrent kernel in this library:
//   * a workgroup owns a tile of M = 32 trajectories for ALL timesteps (trajectories are
//     independent, so there is no inter-workgroup synchronisation at all);
//   * wave w of the workgroup owns hidden units [32w, 32w+32) and, for those units, all
//     four gates: its 32x32 accumulator tiles g = 0..3 (i, f, g, o) hold the same
//     (row, unit) element in the same (lane, register), so the gate non-linearities and
//     the cell update happen in registers with no data movement;
//   * the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32: the result is a
//     k-ordered fmaf chain, which keeps the 1e-4 parity budget);  operand maps
//     (cdna_hip_programming.md section 3):
//        A: lane l holds A[m = l & 31][k = l >> 5]          (one VGPR per 2 k)
//        B: lane l holds B[k = l >> 5][n = l & 31]
//        C/D: lane l, reg r holds C[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31]
//   * the A operand (x_t or h_{t-1} of the 32 rows) lives in LDS TRANSPOSED, [k][m] with a
//     row pitch of 33 floats: the 32 lanes of a half-wave read 32 consecutive floats
//     (conflict-free ds_read_b32) and the transposing writes hit 32 distinct banks;
//   * the B operand (weights) is pre-packed once per optimizer step into the exact
//     per-lane fragment order ("fragment-major"), so a wave's weight load is a run of
//     contiguous 1 KiB float4 loads.  When 4H x H fp32 fits the CU's register file
//     (H <= 128: 256 VGPRs per lane over 4 waves) the recurrent weights stay RESIDENT in
//     registers for the whole sequence; otherwise they stream from L2 every step.
#pragma once
#include "gmt_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GMT_TM 32          // trajectories per workgroup tile
#define GMT_LDA 33         // LDS pitch (floats) of a transposed operand row

// Launders a pointer through an empty asm so that loads through it cannot be hoisted out of
// the time loop (LICM would otherwise pin dozens of loop-invariant values in VGPRs that the
// resident weights need).
template <class T> __device__ __forceinline__ T* gmt_opaque_v(T* p) { asm volatile("" : "+v"(p)); return p; }
template <class T> __device__ __forceinline__ T* gmt_opaque_s(T* p) { asm volatile("" : "+s"(p)); return p; }
// the same for a per-lane offset: keeps `offset + constant` from being hoisted into one live register per constant
__device__ __forceinline__ unsigned gmt_opaque_u(unsigned x) { asm volatile("" : "+v"(x)); return x; }

// Counter-based dropout mask (inter-layer LSTM dropout, RecurrentGAN.py:456): the keep decision of
// element `idx` is a pure function of (seed, idx), so forward and backward regenerate the same
// mask and nothing is stored.  PCG output hash on a Weyl-offset counter.
__device__ __host__ __forceinline__ uint32_t gmt_hash32(uint32_t x) {
    x = x * 747796405u + 2891336453u;
    const uint32_t w = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __host__ __forceinline__ bool gmt_keep(uint32_t idx, uint32_t seed, uint32_t thresh24) {
    return (gmt_hash32(idx + seed * 0x9E3779B1u) >> 8) >= thresh24;      // P(drop) = thresh24 / 2^24
}
// The critic's inter-layer mask: one hash decides two neighbouring elements (16 bits each), so a
// float4 of the staged operand costs two hashes.  Element e is kept iff the (e & 1)-th half of
// hash(e >> 1) is >= thresh16; P(drop) = thresh16 / 2^16.
__device__ __host__ __forceinline__ uint32_t gmt_keep_pair_hash(uint32_t pair, uint32_t seed) {
    return gmt_hash32(pair + seed * 0x9E3779B1u);
}
__device__ __host__ __forceinline__ bool gmt_keep16(uint32_t idx, uint32_t seed, uint32_t thresh16) {
    const uint32_t h = gmt_keep_pair_hash(idx >> 1, seed);
    return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= thresh16;
}
// x4 = four consecutive elements starting at idx (idx % 4 == 0), scaled or zeroed in place
__device__ __forceinline__ void gmt_drop4(float4& v, uint32_t idx, uint32_t seed, uint32_t thresh16, float scale) {
    const uint32_t h0 = gmt_keep_pair_hash(idx >> 1, seed), h1 = gmt_keep_pair_hash((idx >> 1) + 1u, seed);
    v.x = (h0 & 0xffffu) >= thresh16 ? v.x * scale : 0.f;
    v.y = (h0 >> 16) >= thresh16 ? v.y * scale : 0.f;
    v.z = (h1 & 0xffffu) >= thresh16 ? v.z * scale : 0.f;
    v.w = (h1 >> 16) >= thresh16 ? v.w * scale : 0.f;
}

// row of accumulator register r for this lane
__device__ __forceinline__ int gmt_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Fragment-major packing of a [N][K] row-major weight block for the forward contraction
// out[m][n] += A[m][k] * W[n][k]:   float index of element e (0..3) of lane `lane`,
// k-step group s4 (k-steps 4*s4 .. 4*s4+3), gate tile g, wave w:
//   ((((w * (K/8) + s4) * 4 + g) * 64 + lane) * 4 + e)   holds  W[g*H + 32w + (lane&31)][2*(4*s4+e) + (lane>>5)]
__device__ __host__ __forceinline__ size_t gmt_pk_index(int w, int s4, int g, int lane, int e, int K) {
    return ((((size_t)w * (K / 8) + s4) * 4 + g) * 64 + lane) * 4 + e;
}

// acc[g] += A(lds, [K][GMT_LDA]) * Wpk   for one wave; weights streamed from global/L2.
// wpk points at this wave's packed block (K/8 groups of 4 gates x 64 lanes x float4).
template <int NG>
__device__ __forceinline__ void gmt_mma_group(f32x16 (&acc)[NG], const float (&a)[4], const float4 (&w)[NG])
{
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], w[g].x, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], w[g].y, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], w[g].z, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], w[g].w, acc[g], 0, 0, 0);
    }
}

template <int NG>
__device__ __forceinline__ void gmt_stream_fetch(float4 (&w)[NG], float (&a)[4], const float4* __restrict__ wpk,
                                                 const float* __restrict__ ap, int s4, int lane)
{
#pragma unroll
    for (int g = 0; g < NG; ++g) w[g] = wpk[(s4 * NG + g) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ap[(8 * s4 + 2 * j) * GMT_LDA];
}

// Two register buffers used in turn (no copies): the operands of k-group s4+1 are requested before
// the 4*NG MFMAs of group s4 are issued, so an L2 round trip hides under a whole group.  The
// scheduling fences keep the compiler from sinking the loads back down to their first use
// (measured on MI355X: 34 -> 29 ns per MFMA, 28.5 being the register-only rate).
// Saved gate buffers (activated i,f,g,o; d(pre-activation); their adjoints) are row-major with UNIT-MAJOR
// columns: element (row, unit, gate) at row*4H + 4*unit + gate.  The four gate values a lane owns for one
// (row, unit) are one aligned float4, so a wave moves them with one 16-byte access per accumulator row
// (32 units x 16 B contiguous per row half) instead of four dword accesses 4H bytes apart.  The host
// permutes the 4H axis of the weights that meet these buffers in a GEMM (ops.gate_perm / gate_unperm).
__device__ __forceinline__ float4 gmt_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void gmt_st4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }

// k-groups per stage of the 4-gate streamed loop (tuning knob)
#ifndef GMT_STREAM_G4
#define GMT_STREAM_G4 1
#endif

// One pipeline stage of the streamed contraction = G k-groups (8 k each) = 4*NG*G >= 16 MFMAs:
// the stage is what one L2 round trip has to hide under.
template <int NG, int G>
__device__ __forceinline__ void gmt_stage_fetch(float4 (&w)[G][NG], float (&a)[G][4], const float4* __restrict__ wpk,
                                                const float* __restrict__ ap, int st, int lane)
{
#pragma unroll
    for (int q = 0; q < G; ++q) gmt_stream_fetch<NG>(w[q], a[q], wpk, ap, st * G + q, lane);
}

template <int NG, int G>
__device__ __forceinline__ void gmt_stage_mma(f32x16 (&acc)[NG], f32x16& alt, const float (&a)[G][4], const float4 (&w)[G][NG])
{
#pragma unroll
    for (int q = 0; q < G; ++q) {
        if constexpr (NG == 1) {
            // one-gate form: a second accumulator takes every other k-group, so the chains are half as long
            if (q & 1) {
                alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][0], w[q][0].x, alt, 0, 0, 0);
                alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][1], w[q][0].y, alt, 0, 0, 0);
                alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][2], w[q][0].z, alt, 0, 0, 0);
                alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][3], w[q][0].w, alt, 0, 0, 0);
                continue;
            }
        }
        gmt_mma_group<NG>(acc, a[q], w[q]);
    }
}

// Two register stages used in turn (no copies): the operands of stage st+1 are requested before
// the MFMAs of stage st are issued, so an L2 round trip hides under a whole stage.  The scheduling
// fences keep the compiler from sinking the loads back down to their first use (measured on
// MI355X: 34 -> 29 ns per MFMA, 28.5 being the register-only rate).  No branch sits between a
// fetch and its use: a conditional fetch makes every wait a full drain.
template <int NG>
__device__ __forceinline__ void gmt_mma_stream(f32x16 (&acc)[NG], const float* __restrict__ lds_a,
                                               const float4* __restrict__ wpk, int K, int lane)
{
    constexpr int G = (NG >= 4) ? GMT_STREAM_G4 : (NG == 2 ? 2 : 4);
    const int hi = lane >> 5, lo = lane & 31;
    const int ns4 = K >> 3;
    const int nst = ns4 / G;
    const float* ap = lds_a + hi * GMT_LDA + lo;
    f32x16 alt;
    if constexpr (NG == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) alt[r] = 0.f;
    }
    if (nst > 0) {
        float4 w0[G][NG], w1[G][NG];
        float a0[G][4], a1[G][4];
        gmt_stage_fetch<NG, G>(w0, a0, wpk, ap, 0, lane);
        int st = 0;
#pragma unroll 1
        for (; st + 1 < nst; st += 2) {
            gmt_stage_fetch<NG, G>(w1, a1, wpk, ap, st + 1, lane);
            __builtin_amdgcn_sched_barrier(0);
            gmt_stage_mma<NG, G>(acc, alt, a0, w0);
            __builtin_amdgcn_sched_barrier(0);
            const int sn = (st + 2 < nst) ? st + 2 : nst - 1;       // last trip: an in-bounds reload, unused when nst is even
            gmt_stage_fetch<NG, G>(w0, a0, wpk, ap, sn, lane);
            __builtin_amdgcn_sched_barrier(0);
            gmt_stage_mma<NG, G>(acc, alt, a1, w1);
        }
        if (nst & 1) gmt_stage_mma<NG, G>(acc, alt, a0, w0);
    }
    if constexpr (G > 1) {
        for (int s4 = nst * G; s4 < ns4; ++s4) {                    // K not a multiple of 8 G: the odd groups
            float4 w[NG]; float a[4];
            gmt_stream_fetch<NG>(w, a, wpk, ap, s4, lane);
            gmt_mma_group<NG>(acc, a, w);
        }
    }
    if constexpr (NG == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += alt[r];
    }
}

// (A fully unrolled variant of gmt_mma_stream with a per-stage hook was tried for interleaving prefetch loads
// with the streamed MFMAs: the register allocator then keeps every stage's operands apart and spills > 1 KB a lane.)

// Same contraction with the weights already on chip: wr[g][s], s = k-step, holds the first KR
// k-steps in registers; the remaining KS - KR k-steps (NG == 4 only) sit in LDS in the packed fragment
// order (wl = this wave's float4 block: [(KS-KR)/4][NG][64 lanes]), which takes registers
// off the H = 128 kernels that would otherwise spill.  The A fragments (and the LDS weights) of the
// next group of k-steps (>= 8 MFMAs) are read before the current group is issued; fences as in
// gmt_mma_stream.
struct gmt_no_side { __device__ __forceinline__ void operator()(int) const {} };

// `side(grp)` is called once per group, ahead of that group's MFMAs, grp a compile-time constant
// after unrolling: the caller uses it to issue independent global loads (e.g. the next step's
// operands, one row per group) so that their issue slots hide under the matrix pipe.
template <int NG, int KS, int KR = KS, class Side = gmt_no_side>
__device__ __forceinline__ void gmt_mma_resident(f32x16 (&acc)[NG], const float* __restrict__ lds_a,
                                                 const float (&wr)[NG][KR], int lane,
                                                 const float4* __restrict__ wl = nullptr, Side side = Side())
{
    constexpr int GS = (NG >= 4) ? 4 : 8;                       // k-steps per group (>= 8 MFMAs)
    constexpr int GQ = GS / 4;                                  // float4 of packed weights per gate and group
    static_assert(KS % GS == 0 && KR % GS == 0, "K must be a multiple of the group size");
    constexpr int NGRP = KS / GS, RGRP = KR / GS;
    const int hi = lane >> 5, lo = lane & 31;
    const float* ap = lds_a + hi * GMT_LDA + lo;
    float a[2][GS];
    float4 wq[2][GQ][NG];
    // one-gate form: a second accumulator takes the odd k-steps, so consecutive MFMAs are independent
    f32x16 alt;
    if constexpr (NG == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) alt[r] = 0.f;
    }
    auto tail_fetch = [&](int grp) {                            // grp >= RGRP
#pragma unroll
        for (int q = 0; q < GQ; ++q)
#pragma unroll
            for (int g = 0; g < NG; ++g) wq[grp & 1][q][g] = wl[(((grp - RGRP) * GQ + q) * NG + g) * 64 + lane];
    };
#pragma unroll
    for (int j = 0; j < GS; ++j) a[0][j] = ap[2 * j * GMT_LDA];
    if constexpr (RGRP == 0) tail_fetch(0);
#pragma unroll
    for (int grp = 0; grp < NGRP; ++grp) {
        if (grp + 1 < NGRP) {
#pragma unroll
            for (int j = 0; j < GS; ++j) a[(grp + 1) & 1][j] = ap[2 * ((grp + 1) * GS + j) * GMT_LDA];
            if (grp + 1 >= RGRP) tail_fetch(grp + 1);
        }
        side(grp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            float w[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (grp < RGRP) w[g] = wr[g][grp * GS + j];
                else {
                    const float4 v = wq[grp & 1][j / 4][g];
                    w[g] = (j & 3) == 0 ? v.x : (j & 3) == 1 ? v.y : (j & 3) == 2 ? v.z : v.w;
                }
            }
            if constexpr (NG == 1) {
                if (j & 1) alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[grp & 1][j], w[0], alt, 0, 0, 0);
                else acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[grp & 1][j], w[0], acc[0], 0, 0, 0);
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[grp & 1][j], w[g], acc[g], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NG == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += alt[r];
    }
}

// load this wave's packed block into registers (KS = K/2 k-steps)
template <int NG, int KS>
__device__ __forceinline__ void gmt_load_resident(float (&wr)[NG][KS], const float4* __restrict__ wpk, int lane)
{
    // (with an LDS-held tail, KS here is the register part KR; the tail is copied by the caller)
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 v = wpk[(s4 * NG + g) * 64 + lane];
            wr[g][4 * s4 + 0] = v.x; wr[g][4 * s4 + 1] = v.y; wr[g][4 * s4 + 2] = v.z; wr[g][4 * s4 + 3] = v.w;
        }
}

// Phase timing for tuning (the scratch/phase_*.hip drivers define GMT_PHASE_TIMING and include a kernel file);
// compiled out of the library.
#ifdef GMT_PHASE_TIMING
__device__ unsigned long long gmt_phase_ticks[8];
#define GMT_PH_DECL unsigned long long ph_t = __builtin_readcyclecounter(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GMT_PH(i) do { unsigned long long n_ = __builtin_readcyclecounter(); ph_acc[i] += n_ - ph_t; ph_t = n_; } while (0)
#define GMT_PH_DUMP do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) gmt_phase_ticks[i_] = ph_acc[i_]; } while (0)
#else
#define GMT_PH_DECL
#define GMT_PH(i)
#define GMT_PH_DUMP
#endif

