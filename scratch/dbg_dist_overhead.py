"""Where does the ~4 ms/step of the forced one-rank RCCL run go?  mode: none | init | sync"""
import os, sys, time, torch
sys.path.insert(0, '.')
import torch.distributed as dist
import generative_mouse_trajectories_amd as gmt
from bench import synthetic_batch
mode = sys.argv[1]
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
pg = None
print('affinity before', len(os.sched_getaffinity(0)), 'threads', torch.get_num_threads())
if mode != 'none':
    kw = {} if mode == 'lazy' else {'device_id': dev}
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29544', rank=0, world_size=1, **kw)
    pg = dist.group.WORLD
    if mode != 'lazy':
        dist.all_reduce(torch.ones(4, device=dev))
    print('affinity after', len(os.sched_getaffinity(0)), 'threads', torch.get_num_threads())
cfg = gmt.Config(); cfg.latent_dim, cfg.generator_hidden_dim, cfg.discriminator_hidden_dim = 64, 128, 128
torch.manual_seed(0)
tr = gmt.Trainer(gmt.Generator(cfg), gmt.Discriminator(cfg), cfg, dev, log_dir='/tmp/gmt_runs_x', world_size=1, process_group=pg)
if mode == 'sync':
    tr.world_size = 2; tr.g_optimizer.grad_scale = tr.d_optimizer.grad_scale = 1.0; tr._enable_overlapped_sync()
tr.epoch = cfg.teacher_forcing_decay_epochs; tr.overlap_g_forward = True
batch = synthetic_batch(4096, 128, 1234, dev); tf = [False] * 128
def step():
    tr.train_discriminator_step(batch); tr.train_generator_step(batch, {'tf_mask': tf})
for _ in range(8): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step()
torch.cuda.synchronize(); print(mode, f'{(time.perf_counter() - t0) / 8 * 1e3:.2f} ms/step')
if pg is not None: dist.destroy_process_group()
