// Index maps of the bf16 kernels (bf16_tile.h, gemm_bf16.hip, lstm_bf16.hip): where a chunk of a global tile lands in LDS,
// where a lane reads it back, which tile a workgroup owns.  Pure integer functions, compiled for the device by hipcc and for
// the host by g++ (tests/hostsim), where the CPU test-suite replays the LDS-DMA's placement rule and the LDS bank rules of
// MI355X_MICROARCH.md on them: every image is read back consistently and conflict-free, every block map is a bijection.
#pragma once
#ifndef GMT_HD
#  ifdef __HIPCC__
#    define GMT_HD __host__ __device__ __forceinline__
#  else
#    define GMT_HD static inline
#  endif
#endif

namespace gmt_bf16_maps {

// ---- K-contiguous operand stage: [rows][64 k] bf16 = 128-byte rows, 8 chunks of 16 B.
// The DMA writes lane-linear: the wave-instruction that covers rows r0 .. r0+7 puts lane l at byte (r0*128 + l*16), i.e.
// row r0 + l/8, position l%8.  Position p of row r holds global chunk p ^ ((r >> 1) & 7).
GMT_HD int stage_src_chunk(int row, int pos) { return pos ^ ((row >> 1) & 7); }
// byte address (within the stage) of global chunk c of row r
GMT_HD int stage_addr(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- the same with 32 k per stage (64-byte rows, 4 chunks; a wave-instruction covers 16 rows): position p of row r holds
// global chunk p ^ ((r >> 2) & 3) - the 16 rows of a ds_read_b128 lane group then fill one 256-byte bank row exactly.
GMT_HD int stage32_src_chunk(int row, int pos) { return pos ^ ((row >> 2) & 3); }
GMT_HD int stage32_addr(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// ---- epilogue row images: [rows][ROWB bytes], chunk c of row r at position c ^ (r & (ROWB/16 - 1))
GMT_HD int image_src_chunk(int rowb, int row, int lane) { return (lane ^ row) & (rowb / 16 - 1); }
GMT_HD int image_addr(int rowb, int row, int byte_off)
{
    return row * rowb + ((((byte_off >> 4) ^ row) & (rowb / 16 - 1)) << 4) + (byte_off & 15);
}

// ---- weight-gradient stage: [64 r][256 columns] bf16 = 512-byte rows; chunk c of row r at position c ^ ((r & 3) << 2)
GMT_HD int wg_src_chunk(int row, int pos) { return pos ^ ((row & 3) << 2); }
GMT_HD int wg_addr(int row, int col) { return row * 512 + ((((col >> 3)) ^ ((row & 3) << 2)) << 4) + (col & 7) * 2; }
// ds_read_b64_tr_b16: lane l of a wave addresses 4 consecutive columns of one row and receives 4 consecutive rows of ONE
// column.  For the fragment of column block col0 (32 columns), k-step s (16 rows), half h (rows +0..3 or +4..7):
GMT_HD void wg_tr_lane(int lane, int col0, int s, int half, int& row, int& col)
{
    const int grp = lane >> 4, i = lane & 15;
    row = 16 * s + 8 * (grp >> 1) + 4 * half + (i >> 2);
    col = col0 + 16 * (grp & 1) + 4 * (i & 3);
}

// ---- gate columns ("slab" order): (direction d, unit u, gate g) of a [.., ndir*4H] plane
GMT_HD int slab_col(int H, int d, int u, int g) { return d * 4 * H + (u / 32) * 128 + g * 32 + (u % 32); }

// ---- which tile a workgroup owns.  GEMM: blocks on one XCD (bid % 8) take consecutive tiles of 4 (rows) x n_p patches.
GMT_HD void tile_of_block(long long bid, long long nblocks, int n_p, long long n_q, int& tp, long long& tq)
{
    long long lin = bid;
    if ((nblocks & 7) == 0) lin = (bid & 7) * (nblocks >> 3) + (bid >> 3);
    if ((n_q & 3) == 0) {
        const long long per_group = 4LL * n_p;
        const long long grp = lin / per_group, idx = lin % per_group;
        tp = (int)(idx >> 2);
        tq = grp * 4 + (idx & 3);
    } else {
        tp = (int)(lin % n_p);
        tq = lin / n_p;
    }
}
// recurrence steps: the workgroups that share an XCD take a (n_mt / 4) x (n_pt / 2) patch of each direction's tile grid
GMT_HD void patch_of_block(int bid, int nblocks, int n_mt, int n_pt, int& d, int& pt, int& mt)
{
    const int per_dir = n_mt * n_pt;
    if ((n_mt & 3) == 0 && (n_pt & 1) == 0 && (nblocks & 7) == 0) {
        const int x = bid & 7, i = bid >> 3;
        const int A = n_mt >> 2, Bp = n_pt >> 1;
        d = i / (A * Bp);
        const int r = i % (A * Bp);
        pt = (x & 1) * Bp + r % Bp;
        mt = (x >> 1) * A + r / Bp;
    } else {
        d = bid / per_dir;
        const int rem = bid % per_dir;
        pt = rem % n_pt;
        mt = rem / n_pt;
    }
}

}  // namespace gmt_bf16_maps
