"""CPU checks of the host side: the C-ABI library loads and exports every symbol that
include/gmt_hip.h declares (no compute calls), the drop-in modules expose the reference's
state-dict keys, dataset / collate follow the reference contract, and the package refuses
CPU tensors instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import helpers as H
from oracle import gan_oracle as O

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def test_library_exports_every_declared_symbol():
    from generative_mouse_trajectories_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'gmt_hip.h')).read()
    declared = sorted(set(re.findall(r'^int (gmt_\w+)\(', hdr, flags=re.M)))
    assert len(declared) >= 12
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_lib.exported_symbols()) == declared
    L.gmt_version.restype = ctypes.c_int
    assert L.gmt_version() == _lib.ABI_VERSION


def test_state_dict_keys_match_reference_layout():
    import generative_mouse_trajectories_amd as gmt
    cfg = gmt.Config()
    G, D = gmt.Generator(cfg), gmt.Discriminator(cfg)
    oc = O.OracleConfig()
    assert list(G.state_dict().keys()) == list(O.generator_shapes(oc).keys())
    ps, bs = O.discriminator_shapes(oc)
    sd = D.state_dict()
    assert set(sd.keys()) == set(ps) | set(bs)
    for k, shp in {**O.generator_shapes(oc)}.items():
        assert tuple(G.state_dict()[k].shape) == shp, k
    for k, shp in {**ps, **bs}.items():
        assert tuple(sd[k].shape) == shp, k
    assert sum(p.numel() for p in G.parameters()) == 1178118
    assert sum(p.numel() for p in D.parameters()) == 718465


def test_no_cpu_fallback():
    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd._lib import GmtError
    cfg = gmt.Config()
    with pytest.raises(GmtError):
        gmt.compute_kinematics(torch.zeros(2, 5, 2), torch.ones(2, 5))
    D = gmt.Discriminator(cfg)
    with pytest.raises(GmtError):
        D(torch.zeros(2, 5, 2), torch.ones(2, 5), torch.tensor([5, 4]))
    with pytest.raises(GmtError):
        gmt.Trainer(gmt.Generator(cfg), D, cfg, torch.device('cpu'))


def test_dataset_and_collate_contract(tmp_path):
    import generative_mouse_trajectories_amd as gmt
    g = H.load('telemetry.npz')
    cols = list(g['columns']); d = g['data'][:400]
    p = tmp_path / 'mini.csv'
    with open(p, 'w') as fh:
        fh.write('timestamp,' + ','.join(cols) + ',click_events\n')
        for r in d:
            fh.write('0,' + ','.join(repr(float(v)) for v in r) + ',None\n')
    cfg = gmt.Config()
    ds = gmt.MouseTrajectoryDataset(str(p), cfg)
    td = d[:, cols.index('total_duration')]
    n_seg = int((td == 0).sum())
    assert len(ds) % 4 == 0 and 0 < len(ds) // 4 <= n_seg
    b = gmt.collate_trajectories([ds[i] for i in range(8)])
    assert b['sequences'].shape[0] == 8 and b['sequences'].shape[2] == 3
    L = b['lengths']
    assert torch.all(L[:-1] >= L[1:]) and b['sequences'].shape[1] == int(L[0])
    for i in range(8):
        assert torch.all(b['sequences'][i, int(L[i]):] == 0)
        assert torch.all(b['sequences'][i, :int(L[i]), 2] >= 1e-6)
    # first base trajectory: end = start + sum(deltas) up to float rounding, coordinates normalised
    t0 = ds[0]
    assert torch.allclose(t0['start'] + t0['deltas'].sum(0), t0['end'], atol=1e-4)
    # mirrored copies (RG.py:205-272)
    n = len(ds) // 4
    assert torch.allclose(ds[n]['deltas'][:, 0], -t0['deltas'][:, 0])
    assert torch.allclose(ds[n]['start'][0], 1.0 - t0['start'][0])


def test_length_bucketed_batches_cut_padding(tmp_path):
    """batches of similar length, every trajectory once per epoch, usable as a
    DataLoader batch_sampler with the reference's collate function."""
    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd.data import padding_fraction
    from torch.utils.data import DataLoader
    sample = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'telemetry_sample.csv')
    cfg = gmt.Config()
    cfg.min_trajectory_length = 3
    ds = gmt.MouseTrajectoryDataset(sample, cfg)
    assert len(ds) == 6 * 4
    bs = gmt.LengthBucketedBatchSampler.from_dataset(ds, 4, seed=1)
    seen = []
    for b in DataLoader(ds, batch_sampler=bs, collate_fn=gmt.collate_trajectories):
        L = b['lengths']
        assert int(L[0]) == b['sequences'].shape[1] and torch.all(L[:-1] >= L[1:])
        seen.append(L.tolist())
    assert len(seen) == len(bs) == 6
    # the four mirrored copies of a trajectory have its length: bucketed batches of 4 are padding-free here
    assert all(len(set(l)) == 1 for l in seen)

    lengths = np.random.default_rng(0).integers(5, 200, 1000)
    s = gmt.LengthBucketedBatchSampler(lengths, 64, jitter=0.05, seed=3)
    e0, e1 = list(s), list(s)
    assert sorted(i for b in e0 for i in b) == list(range(1000)) and len(e0) == len(s) == 16
    assert sorted(i for b in e1 for i in b) == list(range(1000)) and e0 != e1
    rnd = np.random.default_rng(1).permutation(1000)
    random_batches = [rnd[i:i + 64] for i in range(0, 1000, 64)]
    assert padding_fraction(lengths, e0) < 0.12 < 0.3 < padding_fraction(lengths, random_batches)
    assert len(gmt.LengthBucketedBatchSampler(lengths, 64, drop_last=True)) == 15
    s2 = gmt.LengthBucketedBatchSampler(lengths, 64, jitter=0.05, seed=3)
    assert list(s2) == e0                                # same seed, same epoch: same batches


REF_STYLE_CKPT = '''
# what RG.py:1211-1227 writes when RecurrentGAN.py runs as a script: the Config INSTANCE is pickled as
# __main__.Config, optimizers are torch.optim.Adam over module.parameters()
import sys, torch
sys.path.insert(0, sys.argv[2])
import generative_mouse_trajectories_amd as gmt
class Config(gmt.Config):
    pass
cfg = Config(); cfg.batch_size = 32
torch.manual_seed(5)
G, D = gmt.Generator(cfg), gmt.Discriminator(cfg)
opts = []
for m in (G, D):
    o = torch.optim.Adam(m.parameters(), lr=cfg.learning_rate, betas=cfg.betas)
    for _ in range(3):
        for p in m.parameters():
            p.grad = torch.randn_like(p)
        o.step()
    opts.append(o)
torch.save({'epoch': 7, 'generator_state_dict': G.state_dict(), 'discriminator_state_dict': D.state_dict(),
            'g_optimizer_state_dict': opts[0].state_dict(), 'd_optimizer_state_dict': opts[1].state_dict(),
            'config': cfg, 'best_loss': 0.25}, sys.argv[1])
'''


def make_ref_style_checkpoint(tmp_path):
    import subprocess
    import sys
    script, ckpt = tmp_path / 'mk.py', tmp_path / 'ref_style.pt'
    script.write_text(REF_STYLE_CKPT)
    subprocess.check_call([sys.executable, str(script), str(ckpt), ROOT])
    return str(ckpt)


def test_reference_style_checkpoint_loads_without_unpickling_code(tmp_path):
    """the reference pickles its Config instance into the checkpoint and loads
    with weights_only=False; here the file is read with the restricted unpickler only."""
    import pickle
    from generative_mouse_trajectories_amd.trainer import load_checkpoint_file
    path = make_ref_style_checkpoint(tmp_path)
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, weights_only=True)              # refused as is: __main__.Config
    c = load_checkpoint_file(path, 'cpu')
    assert c['epoch'] == 7 and c['best_loss'] == 0.25 and c['config'] == {'batch_size': 32}
    assert 'lstm.weight_ih_l1_reverse' in c['discriminator_state_dict']
    st = c['d_optimizer_state_dict']['state']
    assert len(st) == 26 and float(st[0]['step']) == 3.0
    # anything else in the file stays refused
    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))
    bad = tmp_path / 'bad.pt'
    torch.save({'epoch': 0, 'config': Evil()}, str(bad))
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint_file(str(bad), 'cpu')


def test_teacher_forcing_schedule():
    oc = O.OracleConfig()
    assert O.teacher_forcing_ratio(0, oc) == 1.0
    assert abs(O.teacher_forcing_ratio(50, oc) - 0.5) < 1e-12
    assert O.teacher_forcing_ratio(100, oc) == 0.0


def test_step_metrics_resolve_on_every_kind_of_read():
    """trainer.StepMetrics: the dict a step function returns fills itself in at the first read of a value, whatever
    the way of reading; keys / len / `in` never wait."""
    import json
    import pickle
    from generative_mouse_trajectories_amd.trainer import StepMetrics

    class Ev:
        def __init__(self): self.n = 0
        def synchronize(self): self.n += 1
        def query(self): return True

    def make():
        ev = Ev()
        return StepMetrics(('a', 'b'), torch.tensor([1.5, 2.5]), ev, {'c': 3.0}), ev

    want = {'a': 1.5, 'b': 2.5, 'c': 3.0}
    m, ev = make()
    assert len(m) == 3 and 'a' in m and ev.n == 0
    assert m['a'] == 1.5 and ev.n == 1 and m['b'] == 2.5 and ev.n == 1
    for read in (lambda m: dict(m), lambda m: {**m}, lambda m: json.loads(json.dumps(m)), lambda m: dict(m.items()),
                 lambda m: m.copy(), lambda m: {k: m[k] for k in m}, lambda m: pickle.loads(pickle.dumps(m)),
                 lambda m: dict(zip(m.keys(), m.values())), lambda m: {k: m.get(k) for k in ('a', 'b', 'c')},
                 lambda m: eval(repr(m))):
        m, ev = make()
        assert read(m) == want and ev.n == 1
    m1, _ = make(); m2, _ = make()
    assert m1 == m2 and (m1, m2) == (want, want) and not (m1 != m2)
