"""CPU checks of the bf16 kernels' index maps (csrc/bf16_maps.h, compiled for the host by tests/hostsim): the LDS-DMA's
placement rule (a wave-instruction writes lane l's 16 bytes at base + 16 l) and the LDS bank rules of
MI355X_MICROARCH.md (section LDS) are replayed on the very functions the kernels use.
  * every LDS image is read back where the DMA put it;
  * the MFMA fragment reads (ds_read_b128), the transposing reads (ds_read_b64_tr_b16) and the cell-state reads are
    conflict-free, the 8-byte gate / h accesses at most 2-way;
  * every block -> tile map is a bijection, and the recurrence's patches keep an XCD on few tiles.
No GPU."""
import ctypes
import os
import subprocess

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'hostsim', 'libhostsim.so')
# ds_read_b128 is serviced in four 16-lane groups (MI355X_MICROARCH.md, LDS table)
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


@pytest.fixture(scope='module')
def sim():
    src = os.path.join(HERE, 'hostsim', 'hostsim.cpp')
    subprocess.check_call(['g++', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-std=c++17',
                           '-Dunroll=', '-Wno-unknown-pragmas', '-o', SO, src])
    return ctypes.CDLL(SO)


def _worst_way(addrs_bytes, width, nbanks):
    """max number of DISTINCT addresses that land on one bank (1 = conflict-free); each access covers width bytes"""
    per_bank = {}
    for a in addrs_bytes:
        for b in range(a // 4, (a + width) // 4):
            per_bank.setdefault(b % nbanks, set()).add(b)
    return max(len(v) for v in per_bank.values())


def test_operand_stage_image(sim):
    rows = 256
    where = {}
    for r0 in range(0, rows, 8):                      # one wave-instruction: 8 rows x 128 B
        for lane in range(64):
            row, pos = r0 + lane // 8, lane % 8
            where[r0 * 128 + lane * 16] = (row, sim.sim_stage_src_chunk(row, pos))
    assert len(where) == rows * 8
    for row in range(rows):
        for c in range(8):
            assert where[sim.sim_stage_addr(row, c)] == (row, c)
    # fragment reads: lane (lo, hi) reads chunk 2 s + hi of row base + lo, base a multiple of 32
    for base in (0, 32, 160):
        for s in range(4):
            for grp in B128_GROUPS:
                addrs = [sim.sim_stage_addr(base + (l & 31), 2 * s + (l >> 5)) for l in grp]
                assert _worst_way(addrs, 16, 64) == 1


def test_half_depth_operand_stage_image(sim):
    rows = 256
    where = {}
    for r0 in range(0, rows, 16):                     # one wave-instruction: 16 rows x 64 B
        for lane in range(64):
            row, pos = r0 + lane // 4, lane % 4
            where[r0 * 64 + lane * 16] = (row, sim.sim_stage32_src_chunk(row, pos))
    assert len(where) == rows * 4
    for row in range(rows):
        for c in range(4):
            assert where[sim.sim_stage32_addr(row, c)] == (row, c)
    for base in (0, 96):
        for s in range(2):
            for grp in B128_GROUPS:
                addrs = [sim.sim_stage32_addr(base + (l & 31), 2 * s + (l >> 5)) for l in grp]
                assert _worst_way(addrs, 16, 64) == 1


@pytest.mark.parametrize('rowb', [128, 256, 512])
def test_epilogue_row_images(sim, rowb):
    ch, rows = rowb // 16, 128
    rpi = 64 // ch
    where = {}
    for inst in range(rows // rpi):
        for lane in range(64):
            row = inst * rpi + lane // ch
            where[inst * 1024 + lane * 16] = (row, sim.sim_image_src_chunk(rowb, row, lane))
    assert len(where) == rows * ch
    for row in range(rows):
        for c in range(ch):
            assert where[sim.sim_image_addr(rowb, row, c * 16)] == (row, c)
            assert sim.sim_image_addr(rowb, row, c * 16 + 8) == sim.sim_image_addr(rowb, row, c * 16) + 8
    lanes_lo = list(range(32))
    if rowb == 512:       # gates: ds_read_b64 / ds_write_b64, lane = row, all lanes the same column
        for off in (0, 8, 72, 504):
            addrs = [sim.sim_image_addr(rowb, 32 + lo, off) for lo in lanes_lo]
            assert _worst_way(addrs, 8, 64) <= 2                        # read: one 32-lane half, 64 banks
            for g in range(0, 32, 16):
                assert _worst_way(addrs[g:g + 16], 8, 32) <= 2         # write: 16 contiguous lanes, 32 banks
    if rowb == 256:       # cell state: ds_read_b128 / ds_write_b128
        for off in (0, 16, 240):
            for grp in B128_GROUPS[:2]:
                assert _worst_way([sim.sim_image_addr(rowb, 64 + l, off) for l in grp], 16, 64) == 1
    if rowb == 128:       # h: ds_write_b64
        for off in (0, 8, 120):
            addrs = [sim.sim_image_addr(rowb, lo, off) for lo in lanes_lo]
            for g in range(0, 32, 16):
                assert _worst_way(addrs[g:g + 16], 8, 32) <= 2


def test_weight_gradient_image_and_transposing_reads(sim):
    where = {}
    for r0 in range(0, 64, 2):                        # one wave-instruction: 2 rows x 512 B
        for lane in range(64):
            row, pos = r0 + lane // 32, lane % 32
            where[r0 * 512 + lane * 16] = (row, sim.sim_wg_src_chunk(row, pos))
    for row in range(64):
        for col in range(0, 256, 8):
            assert where[sim.sim_wg_addr(row, col)] == (row, col // 8)
    r, c = ctypes.c_int(), ctypes.c_int()
    for col0 in (0, 32, 224):
        for s in range(4):
            for half in range(2):
                per_lane = []
                for lane in range(64):
                    sim.sim_wg_tr_lane(lane, col0, s, half, ctypes.byref(r), ctypes.byref(c))
                    per_lane.append((r.value, c.value))
                for grp in range(4):                  # a 16-lane group addresses 4 rows x 16 columns, lane 4 q + p: row q, columns 4 p ..
                    rows = sorted({per_lane[16 * grp + i][0] for i in range(16)})
                    k0 = 16 * s + 8 * (grp >> 1) + 4 * half
                    assert rows == [k0, k0 + 1, k0 + 2, k0 + 3]      # = k of the MFMA fragment: 8 hi + 4 half + q
                    for i in range(16):
                        assert per_lane[16 * grp + i] == (k0 + i // 4, col0 + 16 * (grp & 1) + 4 * (i % 4))
                for h32 in range(2):                  # bank conflicts count per 32-lane half (64 banks, 8 B per lane)
                    addrs = [sim.sim_wg_addr(*per_lane[l]) for l in range(32 * h32, 32 * h32 + 32)]
                    assert _worst_way(addrs, 8, 64) == 1


def test_block_to_tile_maps_are_bijections(sim):
    tp, tq = ctypes.c_int(), ctypes.c_longlong()
    for n_p, n_q in [(16, 1024), (2, 8), (3, 5), (8, 12), (1, 7)]:
        nb = n_p * n_q
        seen = set()
        for bid in range(nb):
            sim.sim_tile_of_block(ctypes.c_longlong(bid), ctypes.c_longlong(nb), n_p, ctypes.c_longlong(n_q), ctypes.byref(tp), ctypes.byref(tq))
            assert 0 <= tp.value < n_p and 0 <= tq.value < n_q
            seen.add((tp.value, tq.value))
        assert len(seen) == nb
    d, pt, mt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    for ndir, n_mt, n_pt in [(2, 32, 8), (2, 16, 8), (2, 64, 4), (1, 4, 2), (2, 3, 8), (2, 1, 2)]:
        nb = ndir * n_mt * n_pt
        seen, per_xcd = set(), {}
        for bid in range(nb):
            sim.sim_patch_of_block(bid, nb, n_mt, n_pt, ctypes.byref(d), ctypes.byref(pt), ctypes.byref(mt))
            assert 0 <= d.value < ndir and 0 <= pt.value < n_pt and 0 <= mt.value < n_mt
            seen.add((d.value, pt.value, mt.value))
            per_xcd.setdefault(bid % 8, []).append((d.value, pt.value, mt.value))
        assert len(seen) == nb
        if n_mt % 4 == 0 and n_pt % 2 == 0 and nb % 8 == 0:
            for tiles in per_xcd.values():            # an XCD's workgroups share few weight tiles and few row tiles
                assert len({t[1] for t in tiles}) <= n_pt // 2 and len({t[2] for t in tiles}) <= n_mt // 4


def test_slab_order_matches_the_host_permutation(sim):
    from generative_mouse_trajectories_amd import ops_bf16
    for H in (256, 512):
        perm = ops_bf16.slab_perm(H, 'cpu')
        for u in (0, 31, 32, 100, H - 1):
            for g in range(4):
                assert int(perm[sim.sim_slab_col(H, 0, u, g)]) == g * H + u
                assert sim.sim_slab_col(H, 1, u, g) == 4 * H + sim.sim_slab_col(H, 0, u, g)
        w = torch.arange(4 * H * 3, dtype=torch.float32).view(4 * H, 3)
        assert torch.equal(ops_bf16.slab_unperm_rows(w[perm], H), w)
