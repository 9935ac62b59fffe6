"""One rank of the two-process data-parallel rehearsal on ONE GPU (tests/test_zz_gpu_dist.py): the real
Trainer(world_size=2) over a gloo group, both ranks on cuda:0.  Started by tests/conftest.py at session start,
before the pytest process itself has touched the GPU.  Writes its flat parameters / gradients / metrics to
<out>/r<rank>.npz and a time-stamped trace of where it is to <out>/trace<rank>.log (read when a run stalls)."""
import faulthandler
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'dist')]


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    tf = open(os.path.join(out, f'trace{rank}.log'), 'w')
    t0 = time.time()

    def trace(msg):
        tf.write(f'{time.time() - t0:8.2f}s {msg}\n')
        tf.flush()

    faulthandler.dump_traceback_later(240, exit=True, file=open(os.path.join(out, f'stall{rank}.log'), 'w'))
    trace('start')
    import numpy as np
    import torch
    import torch.distributed as dist
    import two_rank_common as C
    trace('torch imported')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    trace('process group up')
    import generative_mouse_trajectories_amd as gmt
    dev = torch.device('cuda:0')
    cfg = C.config(gmt)
    torch.manual_seed(100 + rank)            # replicas start DIFFERENT on purpose: Trainer takes rank 0's state
    G, D = gmt.Generator(cfg), gmt.Discriminator(cfg)
    tr = gmt.Trainer(G, D, cfg, dev, log_dir=os.path.join(out, f'runs{rank}'), world_size=world, process_group=dist.group.WORLD)
    tr.epoch = C.EPOCH
    G.train(); D.train()
    torch.cuda.synchronize()
    trace('trainer built, replicas synchronised')
    full = C.full_batch()
    lo, hi = rank * C.B_LOCAL, (rank + 1) * C.B_LOCAL
    batch = {k: v[lo:hi].contiguous() for k, v in full.items()}
    metrics, grads = [], {}
    for it in range(C.ITERS):
        nd, ng = C.noise(it, rank)
        m_d = tr.train_discriminator_step(batch, nd)
        torch.cuda.synchronize()
        trace(f'iteration {it}: D step done {m_d}')
        grads[f'd_grad{it}'] = tr.d_flat.grad.cpu().numpy().copy()
        m_g = tr.train_generator_step(batch, ng)            # teacher-forcing coins: Trainer._tf_coins (same on every rank)
        torch.cuda.synchronize()
        trace(f'iteration {it}: G step done {m_g}')
        grads[f'g_grad{it}'] = tr.g_flat.grad.cpu().numpy().copy()
        metrics.append([m_d[k] for k in sorted(m_d)] + [m_g[k] for k in sorted(m_g)])
    res = dict(g=tr.g_flat.flat.cpu().numpy(), d=tr.d_flat.flat.cpu().numpy(),
               bufs=np.concatenate([b.detach().cpu().numpy().reshape(-1) for b in D.buffers()]),
               metrics=np.asarray(metrics, dtype=np.float64), **grads)
    # the run loop (RG.py:1242-1303) on per-rank shards: the epoch metrics that drive ReduceLROnPlateau, best-loss
    # and early stopping are averaged over the ranks, checkpoints are written by rank 0 only
    tr.g_scheduler.patience = tr.d_scheduler.patience = 0          # make the plateau logic act within three epochs
    halves = [{k: v[:C.B_LOCAL // 2].contiguous() for k, v in batch.items()}, {k: v[C.B_LOCAL // 2:].contiguous() for k, v in batch.items()}]
    ck = os.path.join(out, 'ckpt')
    tr.epoch = 0
    tr.train(halves, 3, checkpoint_dir=ck)
    torch.cuda.synchronize()
    trace('train() returned')
    res.update(g_after=tr.g_flat.flat.cpu().numpy(), d_after=tr.d_flat.flat.cpu().numpy(),
               run_state=np.asarray([tr.g_optimizer.param_groups[0]['lr'], tr.d_optimizer.param_groups[0]['lr'], tr.best_loss,
                                     tr.patience_counter, tr.epoch], dtype=np.float64))
    np.savez(os.path.join(out, f'r{rank}.npz'), **res)
    trace('results written')
    dist.barrier()
    dist.destroy_process_group()
    trace('done')
    faulthandler.cancel_dump_traceback_later()


if __name__ == '__main__':
    main()
