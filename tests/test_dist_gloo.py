"""Data-parallel path on CPU: two gloo ranks with different gradient shards all-reduce the flat
gradient buffer once and apply the fused-Adam arithmetic with grad_scale = 1/world; the result
must equal single-process Adam on the averaged gradient, and both replicas must stay identical."""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..'))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from generative_mouse_trajectories_amd.trainer import FlatParams, allreduce_flat_grad
    torch.manual_seed(0)                                   # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    flat = FlatParams(model)
    flat.zero_grad()
    torch.manual_seed(100 + rank)                          # different shard per rank
    x = torch.randn(16, 7)
    model(x).pow(2).sum().backward()
    flat.gather_grads()                                    # per-parameter gradients -> the flat buffer
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(flat.params, flat.views_g))
    local = flat.grad.clone()
    allreduce_flat_grad(flat, world, None)
    sim = ctypes.CDLL(os.path.join(HERE, 'hostsim', 'libhostsim.so'))
    p = flat.flat.numpy(); g = (flat.grad * (1.0 / world)).numpy().copy()
    m = np.zeros_like(p); v = np.zeros_like(p)
    fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    sim.sim_adam(fp(p), fp(g), fp(m), fp(v), ctypes.c_longlong(p.size), ctypes.c_float(0.5), ctypes.c_float(0.9),
                 ctypes.c_float(1e-8), ctypes.c_float(1e-4 / (1 - 0.5)), ctypes.c_float(math.sqrt(1 - 0.9)))
    np.savez(os.path.join(out_dir, f'r{rank}.npz'), local=local.numpy(), summed=flat.grad.numpy(), p=p,
             w0=model[0].weight.detach().numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_and_update(tmp_path):
    subprocess.check_call(['g++', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-std=c++17', '-Dunroll=',
                           '-Wno-unknown-pragmas', '-o', os.path.join(HERE, 'hostsim', 'libhostsim.so'),
                           os.path.join(HERE, 'hostsim', 'hostsim.cpp')])
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'r0.npz'); r1 = np.load(tmp_path / 'r1.npz')
    np.testing.assert_allclose(r0['summed'], r0['local'] + r1['local'], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(r0['summed'], r1['summed'])
    np.testing.assert_array_equal(r0['p'], r1['p'])               # replicas stay bit-identical
    np.testing.assert_array_equal(r0['w0'], r1['w0'])             # nn.Parameter is a view of the flat buffer
    # single-process reference: torch Adam on the averaged gradient
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ps = list(model.parameters())
    avg = 0.5 * (r0['local'] + r1['local'])
    off = 0
    for q in ps:
        q.grad = torch.from_numpy(avg[off:off + q.numel()].reshape(q.shape).copy())
        off += (q.numel() + 3) // 4 * 4
    torch.optim.Adam(ps, lr=1e-4, betas=(0.5, 0.9)).step()
    off = 0
    for q in ps:
        np.testing.assert_allclose(r0['p'][off:off + q.numel()], q.detach().numpy().reshape(-1), rtol=2e-7, atol=2e-9)
        off += (q.numel() + 3) // 4 * 4


def _worker_overlap(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from generative_mouse_trajectories_amd.trainer import FlatParams, OverlappedGradSync
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 6), torch.nn.Tanh(), torch.nn.Linear(6, 5), torch.nn.Tanh(),
                                torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    for p in model[5].parameters():
        p.requires_grad_(False)                           # a frozen tail: not part of the flat buffer
    flat = FlatParams(model)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    sync = OverlappedGradSync(flat, names, lambda n: n.split('.')[0], world, None)
    assert [len(b['idx']) for b in sync.buckets] == [2, 2, 2]
    assert sync.buckets[0]['lo'] == 0 and sync.buckets[-1]['hi'] == flat.total
    assert all(a['hi'] == b['lo'] for a, b in zip(sync.buckets, sync.buckets[1:]))
    torch.manual_seed(100 + rank)
    x = torch.randn(16, 7)
    res = {}
    for it in range(2):                                   # twice: the bookkeeping resets between steps
        flat.zero_grad()
        loss = model(x).pow(2).sum() * (it + 1)
        local = torch.autograd.grad(loss, flat.params, retain_graph=True)
        sync.begin()
        loss.backward()
        launched_in_backward = list(sync.launched)        # started from the autograd hooks, before finish()
        sync.finish()
        assert all(launched_in_backward), launched_in_backward
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(flat.params, flat.views_g))
        lf = torch.zeros_like(flat.grad)
        for g, o in zip(local, flat.offsets):
            lf[o:o + g.numel()] = g.reshape(-1)
        res[f'local{it}'] = lf.numpy().copy()
        res[f'summed{it}'] = flat.grad.numpy().copy()
    # a step in which one bucket gets no gradient at all: finish() still exchanges it (zeros)
    flat.zero_grad()
    sync.begin()
    model[4](torch.randn(4, 5)).sum().backward()
    assert sync.launched == [False, False, True]
    sync.finish()
    assert torch.all(flat.grad[:sync.buckets[2]['lo']] == 0)
    # a bucket that completes out of turn waits: only the FIRST module gets gradients -> nothing starts in backward
    flat.zero_grad()
    sync.begin()
    model[0](torch.randn(4, 7)).sum().backward()
    assert sync.launched == [False, False, False] and sync.pending == [0, 2, 2]
    sync.finish()
    assert all(sync.launched)
    assert torch.all(flat.grad[sync.buckets[1]['lo']:] == 0)
    np.savez(os.path.join(out_dir, f'o{rank}.npz'), **res)
    dist.destroy_process_group()


def test_two_rank_gloo_bucketed_allreduce_overlapped_with_backward(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_overlap, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'o0.npz'); r1 = np.load(tmp_path / 'o1.npz')
    for it in range(2):
        np.testing.assert_allclose(r0[f'summed{it}'], r0[f'local{it}'] + r1[f'local{it}'], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(r0[f'summed{it}'], r1[f'summed{it}'])
