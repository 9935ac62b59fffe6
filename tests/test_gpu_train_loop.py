"""Config 1 plumbing on the GPU: the reference's run loop on a real recording
(the committed excerpt of its CSVs) - dataset -> DataLoader -> Trainer.train -> checkpoints - and the
checkpoint formats: this Trainer's own files resume bit-identically, a reference-style file (pickled
Config instance, torch.optim.Adam state) loads through the restricted unpickler."""
import copy
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

import test_abi_and_host as TH

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
SAMPLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'telemetry_sample.csv')


def _cfg(gmt):
    cfg = gmt.Config()
    cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 16, 64, 64
    cfg.min_trajectory_length, cfg.batch_size, cfg.gp_every_n = 3, 8, 2
    return cfg


def test_run_loop_on_a_recording_and_bit_identical_resume(tmp_path, capsys):
    import generative_mouse_trajectories_amd as gmt
    cfg = _cfg(gmt)
    ds = gmt.MouseTrajectoryDataset(SAMPLE, cfg)
    assert len(ds) == 24
    loader = DataLoader(ds, batch_sampler=gmt.LengthBucketedBatchSampler.from_dataset(ds, cfg.batch_size, seed=2),
                        collate_fn=gmt.collate_trajectories)
    torch.manual_seed(0)
    tr = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, DEV, log_dir=str(tmp_path / 'runs'))
    ck = tmp_path / 'ck'
    tr.train(loader, num_epochs=2, checkpoint_dir=str(ck))
    out = capsys.readouterr().out
    assert 'Epoch 0/2' in out and 'Epoch 1/2' in out and 'Training complete!' in out
    assert (ck / 'checkpoint_epoch_0.pt').exists() and (ck / 'checkpoint_epoch_0_best.pt').exists()
    assert tr.d_step == 2 * 3 * cfg.n_critic and tr.g_optimizer.step_count == 6
    for p in list(tr.generator.parameters()) + list(tr.discriminator.parameters()):
        assert torch.isfinite(p).all()

    # resume: a fresh trainer that loads the file continues exactly like the one that wrote it
    path = str(tmp_path / 'now.pt')
    tr.save_checkpoint(path)
    torch.manual_seed(1)
    tr2 = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, DEV, log_dir=str(tmp_path / 'runs2'))
    tr2.load_checkpoint(path)
    tr2.d_step = tr.d_step
    assert tr2.epoch == tr.epoch and tr2.best_loss == tr.best_loss
    assert tr2.d_optimizer.step_count == tr.d_optimizer.step_count
    batch = gmt.collate_trajectories([ds[i] for i in range(8)])
    B, T = batch['sequences'].shape[:2]
    g = torch.Generator().manual_seed(9)
    noise_d = {'z': torch.randn(B, cfg.latent_dim, generator=g), 'gp_idx': torch.randperm(B, generator=g),
               'alpha': torch.rand(2, 1, 1, generator=g)}
    noise_g = {'z': torch.randn(B, cfg.latent_dim, generator=g), 'tf_mask': [True, False] * (T // 2) + [False] * (T % 2)}
    # the remaining RNG draws are the injected ones (the determinism recipe of SURVEY.md section 8c:
    # the head's nn.Dropout keeps the p it was built with)
    cfg.lstm_dropout = 0.0
    for t_ in (tr, tr2):
        t_.generator.train(); t_.discriminator.train()
        t_.discriminator.output_layers[2].p = 0.0
    m1 = (tr.train_discriminator_step(batch, noise_d), tr.train_generator_step(batch, noise_g))
    m2 = (tr2.train_discriminator_step(batch, noise_d), tr2.train_generator_step(batch, noise_g))
    assert m1 == m2, (m1, m2)
    assert torch.equal(tr.g_flat.flat, tr2.g_flat.flat) and torch.equal(tr.d_flat.flat, tr2.d_flat.flat)
    assert torch.equal(tr.d_optimizer.exp_avg_sq, tr2.d_optimizer.exp_avg_sq)


def test_reference_style_checkpoint_into_trainer(tmp_path):
    import generative_mouse_trajectories_amd as gmt
    from generative_mouse_trajectories_amd.trainer import load_checkpoint_file
    path = TH.make_ref_style_checkpoint(tmp_path)
    raw = load_checkpoint_file(path, 'cpu')
    cfg = gmt.Config()
    torch.manual_seed(77)
    tr = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, DEV, log_dir=str(tmp_path / 'runs'))
    tr.load_checkpoint(path)
    assert tr.epoch == 7 and tr.best_loss == 0.25
    for mod, key, opt, flat in ((tr.generator, 'generator_state_dict', tr.g_optimizer, tr.g_flat),
                                (tr.discriminator, 'discriminator_state_dict', tr.d_optimizer, tr.d_flat)):
        sd = mod.state_dict()
        for k, v in raw[key].items():
            assert torch.equal(sd[k].cpu(), v), k
        ost = raw[key.replace('_state_dict', '')[0] + '_optimizer_state_dict']['state']
        assert opt.step_count == 3
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            # parameters still live in the flat buffer, moments were re-homed into the flat moment buffers
            assert p.data_ptr() == flat.flat.data_ptr() + 4 * o
            assert torch.equal(opt.exp_avg[o:o + p.numel()].view_as(p).cpu(), ost[i]['exp_avg'])
            assert torch.equal(opt.exp_avg_sq[o:o + p.numel()].view_as(p).cpu(), ost[i]['exp_avg_sq'])
    # and the loaded state trains
    ds = gmt.MouseTrajectoryDataset(SAMPLE, _cfg(gmt))
    batch = gmt.collate_trajectories([ds[i] for i in range(8)])
    m = tr.train_discriminator_step(batch)
    assert all(np.isfinite(v) for v in m.values())


def test_run_loop_on_a_recording_at_config3_width_bf16(tmp_path, capsys):
    """The same run loop with compute_dtype = 'bf16' and both networks at H = 512 (BASELINE configs[2]'s width) on the recording
    (ragged batches of 8 trajectories, padded to the kernels' 256 rows; gradient penalty every 2nd D step; both dropouts on):
    two epochs, finite parameters, a checkpoint that resumes to the same step - the step functions' numbers are
    tests/test_gpu_bf16.py's."""
    import generative_mouse_trajectories_amd as gmt
    cfg = _cfg(gmt)
    cfg.generator_hidden_dim = cfg.discriminator_hidden_dim = 512
    cfg.compute_dtype = 'bf16'
    ds = gmt.MouseTrajectoryDataset(SAMPLE, cfg)
    loader = DataLoader(ds, batch_sampler=gmt.LengthBucketedBatchSampler.from_dataset(ds, cfg.batch_size, seed=2),
                        collate_fn=gmt.collate_trajectories)
    torch.manual_seed(0)
    tr = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, DEV, log_dir=str(tmp_path / 'runs'))
    ck = tmp_path / 'ck'
    tr.train(loader, num_epochs=2, checkpoint_dir=str(ck))
    out = capsys.readouterr().out
    assert 'Epoch 1/2' in out and 'Training complete!' in out
    assert (ck / 'checkpoint_epoch_0.pt').exists()
    assert tr.d_step == 2 * 3 * cfg.n_critic
    for p in list(tr.generator.parameters()) + list(tr.discriminator.parameters()):
        assert torch.isfinite(p).all()
    path = str(tmp_path / 'now.pt')
    tr.save_checkpoint(path)
    torch.manual_seed(1)
    tr2 = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, DEV, log_dir=str(tmp_path / 'runs2'))
    tr2.load_checkpoint(path)
    tr2.d_step = tr.d_step
    batch = gmt.collate_trajectories([ds[i] for i in range(8)])
    B, T = batch['sequences'].shape[:2]
    g = torch.Generator().manual_seed(9)
    noise_d = {'z': torch.randn(B, cfg.latent_dim, generator=g), 'gp_idx': torch.randperm(B, generator=g),
               'alpha': torch.rand(2, 1, 1, generator=g)}
    noise_g = {'z': torch.randn(B, cfg.latent_dim, generator=g), 'tf_mask': [True, False] * (T // 2) + [False] * (T % 2)}
    cfg.lstm_dropout = 0.0
    for t_ in (tr, tr2):
        t_.generator.train(); t_.discriminator.train()
        t_.discriminator.output_layers[2].p = 0.0
    m1 = (tr.train_discriminator_step(batch, noise_d), tr.train_generator_step(batch, noise_g))
    m2 = (tr2.train_discriminator_step(batch, noise_d), tr2.train_generator_step(batch, noise_g))
    assert m1 == m2, (m1, m2)
    assert torch.equal(tr.g_flat.flat, tr2.g_flat.flat) and torch.equal(tr.d_flat.flat, tr2.d_flat.flat)
    # sampling through the bf16 rollout (RG.py:606-690)
    starts, ends = batch['starts'].to(DEV), batch['ends'].to(DEV)
    traj, lens = tr.generator.generate(starts, ends)
    assert traj.shape[0] == B and bool(torch.isfinite(traj).all()) and int(lens.min()) >= 1
