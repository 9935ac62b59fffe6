// Building blocks of the bf16-operand / fp32-accumulate kernels (BASELINE config 3: H = 512, where the recurrent
// weights no longer fit a CU and the gate contractions are real dense GEMMs).
//
// One contraction form serves every kernel of this family: a workgroup of 4 or 8 waves computes a tile of
//     D[p][q] = sum_k P[p][k] * Q[q][k]          (64 NPB) x (32 NQB NWQ), K in steps of 32 or 64
// with both operands K-CONTIGUOUS in memory (weights [N][K], activations [M][K]) - "NT" in BLAS terms.  P feeds the
// MFMA's A side and Q its B side, so in the 32x32 accumulator tile (cdna_hip_programming.md section 3)
//     lane l, register r  holds  D[p = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][q = l & 31]:
// the LANE is a Q row and a register quad is FOUR CONSECUTIVE P rows.  Every kernel here puts the weights on P and
// the activation rows (trajectory-steps) on Q, so an epilogue thread owns one trajectory row and 4 consecutive
// output columns at a time: 8-byte bf16 / 16-byte fp32 accesses, per-row masks are per-lane scalars.
//
// Staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers).  Two loops share the image format:
//   mainloop       (the GEMMs): two LDS buffers of (TP + TQ) rows x 128 B, one barrier per 64-deep K-tile, the next tile's DMA
//                  issued in one burst after the barrier ("glds, 2 LDS buffers, BK = 64" of the programming guide);
//   mainloop_ring  (the per-timestep kernels): three slots of 32- or 64-deep K-tiles, counted vmcnt waits, the DMA pieces
//                  spread behind the MFMA clusters.
// The LDS image of a tile is [row][chunks of 16 B]; chunk c of row r is stored at position c ^ ((r >> 1) & 7) (64-deep) or
// c ^ ((r >> 2) & 3) (32-deep) - the DMA destination is lane-linear, so the XOR is applied to the per-lane SOURCE address -
// which makes the ds_read_b128 fragment reads (16-lane groups {0-3,12-15,20-27}, ...) conflict-free.
// Wave (wp, wq) owns P rows [TP/2 wp, +TP/2) x a Q row block: NPB x NQB accumulator tiles of 32 x 32.
#pragma once
#include "gmt_common.h"
#include "bf16_maps.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Phase timing for the scratch/phase_bf16.hip experiment (compiled out of the library): thread 0 of every workgroup stamps
// the wall clock at numbered points of a kernel.
#ifdef GMT_PHASE_TIMING
__device__ long long gmt_phase_buf[4096 * 16], gmt_phase_cyc[4096 * 16];
#define GMT_PHASE(n) do { if (threadIdx.x == 0 && blockIdx.x < 4096) { gmt_phase_buf[blockIdx.x * 16 + (n)] = wall_clock64(); \
                                                                       gmt_phase_cyc[blockIdx.x * 16 + (n)] = clock64(); } } while (0)
#else
#define GMT_PHASE(n) do { } while (0)
#endif

namespace gmt_bf16 {

constexpr int BK = 64;                              // k per LDS stage: one 128-byte row per operand row
constexpr int NTHREADS = 256;
// NPB x NQB accumulator blocks (32 x 32) per wave; the workgroup tile is (64 NPB) x (64 NQB)
template <int NPB, int NQB> struct Tile {
    static constexpr int TP = 64 * NPB, TQ = 64 * NQB;
    static constexpr int STAGE_BYTES = (TP + TQ) * BK * 2;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;           // 4 x 4: 128 KiB (one workgroup per CU); 2 x 2: 64 KiB
};
constexpr int TP = Tile<4, 4>::TP, TQ = Tile<4, 4>::TQ, LDS_BYTES = Tile<4, 4>::LDS_BYTES;

// ROWS x 64 k of a row-major bf16 operand -> LDS image at lds_tile.  g points at (tile row 0, k0); every row and
// k-range must exist (callers check M % tile == 0, N % tile == 0, K % 64 == 0).
template <int ROWS, int NWAVES = 4>
__device__ __forceinline__ void stage(const __bf16* __restrict__ g, long long ld, char* lds_tile, int wave, int lane)
{
#pragma unroll
    for (int j = 0; j < ROWS / (8 * NWAVES); ++j) {
        const int row = j * 8 * NWAVES + wave * 8 + (lane >> 3);
        const int chunk = gmt_bf16_maps::stage_src_chunk(row, lane & 7);
        const __bf16* src = g + (long long)row * ld + chunk * 8;
        char* dst = lds_tile + (j * 8 * NWAVES + wave * 8) * 128;     // wave-uniform; the DMA adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

// acc[i][j] += P-block i (32 rows) x Q-block j (32 rows) of this wave over nk K-tiles.
// lds: Tile::LDS_BYTES, 16-byte aligned.  No barrier after the last tile: callers that reuse the LDS put one.
// NWQ waves along Q (2 or 4; always 2 along P): the workgroup tile is (64 NPB) x (32 NQB NWQ), 128 NWQ threads.
template <int NPB, int NQB, int NWQ = 2>
__device__ __forceinline__ void mainloop(const __bf16* __restrict__ Pg, long long ldp, const __bf16* __restrict__ Qg,
                                         long long ldq, int nk, char* lds, f32x16 (&acc)[NPB][NQB], int wave, int lane)
{
    using T = Tile<NPB, NQB * NWQ / 2>;
    constexpr int NWV = 2 * NWQ;
    const int wp = wave / NWQ, wq = wave % NWQ;
    const int lo = lane & 31, hi = lane >> 5;
    if (nk > 0) {
        stage<T::TP, NWV>(Pg, ldp, lds, wave, lane);
        stage<T::TQ, NWV>(Qg, ldq, lds + T::TP * 128, wave, lane);
    }
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // tile t has landed; everybody is done reading the other buffer
        char* cur = lds + (t & 1) * T::STAGE_BYTES;
        if (t + 1 < nk) {
            char* nxt = lds + ((t + 1) & 1) * T::STAGE_BYTES;
            stage<T::TP, NWV>(Pg + (long long)(t + 1) * BK, ldp, nxt, wave, lane);
            stage<T::TQ, NWV>(Qg + (long long)(t + 1) * BK, ldq, nxt + T::TP * 128, wave, lane);
        }
        const char* pt = cur + (wp * (T::TP / 2) + lo) * 128;
        const char* qt = cur + T::TP * 128 + (wq * (32 * NQB) + lo) * 128;
        // fragments of k-step s+1 are requested before the MFMAs of k-step s (two register sets): with one or two waves per
        // SIMD nothing else hides the LDS latency, and the compiler on its own reads, waits, computes, reads, ...
        bf16x8 pf[2][NPB], qf[2][NQB];
        auto read_frags = [&](int s, int set) {
            const int off = gmt_bf16_maps::stage_addr(lo, 2 * s + hi) - lo * 128;      // (row blocks are multiples of 32 rows: same XOR)
#pragma unroll
            for (int i = 0; i < NPB; ++i) pf[set][i] = *reinterpret_cast<const bf16x8*>(pt + i * 32 * 128 + off);
#pragma unroll
            for (int j = 0; j < NQB; ++j) qf[set][j] = *reinterpret_cast<const bf16x8*>(qt + j * 32 * 128 + off);
        };
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) read_frags(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NPB; ++i)
#pragma unroll
                for (int j = 0; j < NQB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[s & 1][i], qf[s & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The contraction for the per-timestep kernels: a ring of three LDS slots.  While K-tile t is consumed, tile t+1 is landing
// and the DMA pieces of tile t+2 are issued - a few after each k-step's MFMAs, not in one burst: a CU's vector-memory path
// takes an LDS-DMA piece (1 KiB) every ~25 cycles at best (scratch/dma_rate.hip: 48 KiB in 0.45-0.5 us with 8 waves issuing,
// 0.7 with 4), about as long as the tile's MFMAs, and a wave that issues into a full queue is held until its piece is
// taken; a burst after the barrier therefore parks every wave while the matrix pipe idles, then the pipe runs while the
// memory path idles (measured: tile time = DMA time + MFMA time).  Spread, the two overlap.  The wait for a tile is a
// COUNTED s_waitcnt (the pieces of the tile after it stay in flight) and the barrier the raw s_barrier (__syncthreads()
// would drain vmcnt to 0).  Fragment reads run one k-step ahead of their MFMAs, across the tile boundary too: the barrier
// for tile t+1 sits before the LAST k-step's MFMAs of tile t, when every wave holds all its fragments of tile t.
// RBK = 32 or 64 (k per slot); 2 x 2 waves or 2 x 4 (NWQ = 4: two waves per SIMD); tile (64 NPB) x (32 NQB NWQ);
// lds: 3 * (64 NPB + 32 NQB NWQ) * RBK * 2 bytes.
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NPB, int NQB, int RBK, int NST, int NWQ = 2>
__device__ __forceinline__ void mainloop_ring(const __bf16* __restrict__ Pg, long long ldp, const __bf16* __restrict__ Qg,
                                              long long ldq, int nk, char* lds, f32x16 (&acc)[NPB][NQB], int wave, int lane)
{
    static_assert(RBK == 32 || RBK == 64, "k per slot");
    static_assert(NST == 3, "three slots: consumed, landing, being issued");
    static_assert(NWQ == 2 || (NWQ == 4 && RBK == 64), "2 x 2 waves, or 2 x 4 (two per SIMD) with 64-deep slots");
    constexpr int NWV = 2 * NWQ;
    constexpr int TPr = 64 * NPB, TQr = 32 * NQB * NWQ, ROWB = RBK * 2, STG = (TPr + TQr) * ROWB, KS = RBK / 16;
    constexpr int RPP = 1024 / ROWB * NWV;                       // rows per round of pieces (one piece per wave)
    constexpr int PP = TPr / RPP, LPT = (TPr + TQr) / RPP;       // pieces per thread: of P, in all
    static_assert(TPr % RPP == 0 && TQr % RPP == 0 && LPT < 32, "piece counts");
    if (nk <= 0) return;
    const int wp = wave / NWQ, wq = wave % NWQ;
    const int lo = lane & 31, hi = lane >> 5;
    // piece p of K-tile t -> slot: rows [p RPP + wave (RPP / NWV), + RPP / NWV) of P (p < PP) or Q
    auto piece = [&](int t, int slot, int p) {
        const bool isq = p >= PP;
        const int j = isq ? p - PP : p;
        const int lrow = lane / (ROWB / 16), pos = lane % (ROWB / 16);
        const int row = j * RPP + wave * (RPP / NWV) + lrow;
        const int chunk = RBK == 32 ? gmt_bf16_maps::stage32_src_chunk(row, pos) : gmt_bf16_maps::stage_src_chunk(row, pos);
        const __bf16* src = (isq ? Qg + (long long)row * ldq : Pg + (long long)row * ldp) + (long long)t * RBK + chunk * 8;
        char* dst = lds + slot * STG + (isq ? TPr * ROWB : 0) + (j * RPP + wave * (RPP / NWV)) * ROWB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    bf16x8 pf[2][NPB], qf[2][NQB];
    auto read_frags = [&](int slot, int ks, int set) {
        const char* cur = lds + slot * STG;
        const char* pt = cur + (wp * (TPr / 2) + lo) * ROWB;
        const char* qt = cur + TPr * ROWB + (wq * (32 * NQB) + lo) * ROWB;
        int off;
        if constexpr (RBK == 32) off = gmt_bf16_maps::stage32_addr(lo, 2 * ks + hi) - lo * 64;
        else off = gmt_bf16_maps::stage_addr(lo, 2 * ks + hi) - lo * 128;
#pragma unroll
        for (int i = 0; i < NPB; ++i) pf[set][i] = *reinterpret_cast<const bf16x8*>(pt + i * 32 * ROWB + off);
#pragma unroll
        for (int j = 0; j < NQB; ++j) qf[set][j] = *reinterpret_cast<const bf16x8*>(qt + j * 32 * ROWB + off);
    };
#pragma unroll
    for (int p = 0; p < LPT; ++p) piece(0, 0, p);
    if (nk > 1) {
#pragma unroll
        for (int p = 0; p < LPT; ++p) piece(1, 1, p);
        wait_vmcnt<LPT>();                                       // tile 0 has landed (this wave's pieces) ...
    } else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                                // ... everybody's
    read_frags(0, 0, 0);
    int slot = 0;
    for (int t = 0; t < nk; ++t) {
        const int nslot = slot == 2 ? 0 : slot + 1, fslot = nslot == 2 ? 0 : nslot + 1;   // slots of tiles t+1, t+2
        const bool fill = t + 2 < nk;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) read_frags(slot, ks + 1, (ks + 1) & 1);
            else if (t + 1 < nk) {
                // tile t+1 has landed: only the pieces of tile t+2 issued so far may still be in flight
                if (fill) wait_vmcnt<(KS - 1) * LPT / KS>(); else wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave holds all its fragments of tile t
                __builtin_amdgcn_s_barrier();
                read_frags(nslot, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NPB; ++i)
#pragma unroll
                for (int j = 0; j < NQB; ++j)
                {
#ifndef GMT_RING_NO_MFMA
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[ks & 1][i], qf[ks & 1][j], acc[i][j], 0, 0, 0);
#else
                    asm volatile("" :: "v"(pf[ks & 1][i]), "v"(qf[ks & 1][j]));
#endif
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (fill) {                                          // this k-step's share of tile t+2 (slot of tile t-1: free since the last barrier)
#pragma unroll
                for (int p = ks * LPT / KS; p < (ks + 1) * LPT / KS; ++p) {
#ifndef GMT_RING_NO_DMA                                       // (scratch experiments: the loop without its DMA / its MFMAs / its reads)
                    piece(t + 2, fslot, p);
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        slot = nslot;
    }
}

template <int NPB, int NQB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NPB][NQB])
{
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
        for (int j = 0; j < NQB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// fp32 -> bf16, round to nearest even (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d)
{
    bf16x4 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
    return v;
}

// ---- epilogue tiles through LDS.  In the accumulator layout a lane owns ONE row (trajectory) and 4 consecutive columns,
// so a direct global access is 8-16 bytes per lane, each lane on a different row: a 128-byte line is fetched or written
// in 8-16 separate requests, and the L2 takes ~16 requests per clock per XCD whatever their size (measured: the GEMM's
// tile store sped up 5.5 us when its 8 192 sixteen-byte requests became 1 024 full lines).  The kernels therefore move
// their epilogue operands as whole rows - LDS-DMA in, 16 bytes per lane out - and pick their elements out of the LDS.
// Image of a [rows][ROWB bytes] block: lane-linear (the DMA's rule), the 16-byte chunk c of row r stored at position
// c ^ (r & (ROWB/16 - 1)), so that lanes on consecutive rows reading the same column hit different banks.
template <int ROWB> struct RowImage {
    static constexpr int CH = ROWB / 16;                 // chunks per row (8, 16 or 32)
    static constexpr int RPI = 64 / CH;                  // rows per wave-instruction
    static __device__ __forceinline__ int addr(int row, int byte_off) { return gmt_bf16_maps::image_addr(ROWB, row, byte_off); }
    // rows [0, nrows) <- g + rowmap(row) * ld_bytes + colmap(chunk): colmap gives the global byte offset of the image's
    // 16-byte chunk `chunk` (identity * 16 for a contiguous row); nrows % (4 RPI) == 0
    template <class RowMap, class ColMap>
    static __device__ __forceinline__ void dma_in(const char* __restrict__ g, long long ld_bytes, char* lds, int nrows,
                                                  RowMap rowmap, ColMap colmap, int wave, int lane, int nwaves = 4) {
        for (int inst = wave; inst < nrows / RPI; inst += nwaves) {
            const int row = inst * RPI + lane / CH;
            const int chunk = gmt_bf16_maps::image_src_chunk(ROWB, row, lane);
            const char* src = g + (long long)rowmap(row) * ld_bytes + colmap(chunk);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + inst * 1024), 16, 0, 0);
        }
    }
    template <class RowMap, class ColMap>
    static __device__ __forceinline__ void copy_out(char* __restrict__ g, long long ld_bytes, const char* lds, int nrows,
                                                    RowMap rowmap, ColMap colmap, int wave, int lane, int nwaves = 4) {
        for (int inst = wave; inst < nrows / RPI; inst += nwaves) {
            const int row = inst * RPI + lane / CH;
            const int chunk = gmt_bf16_maps::image_src_chunk(ROWB, row, lane);
            const f32x4 v = *reinterpret_cast<const f32x4*>(lds + inst * 1024 + lane * 16);
            *reinterpret_cast<f32x4*>(g + (long long)rowmap(row) * ld_bytes + colmap(chunk)) = v;
        }
    }
    template <class RowMap>
    static __device__ __forceinline__ void dma_in(const char* __restrict__ g, long long ld_bytes, char* lds, int nrows,
                                                  RowMap rowmap, int wave, int lane, int nwaves = 4) {
        dma_in(g, ld_bytes, lds, nrows, rowmap, [](int c) { return c * 16; }, wave, lane, nwaves);
    }
    template <class RowMap>
    static __device__ __forceinline__ void copy_out(char* __restrict__ g, long long ld_bytes, const char* lds, int nrows,
                                                    RowMap rowmap, int wave, int lane, int nwaves = 4) {
        copy_out(g, ld_bytes, lds, nrows, rowmap, [](int c) { return c * 16; }, wave, lane, nwaves);
    }
};

using gmt_bf16_maps::tile_of_block;      // GEMM tile order and the recurrence steps' XCD-compact patches: bf16_maps.h
using gmt_bf16_maps::patch_of_block;

}  // namespace gmt_bf16
