import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


# ------------------------------------------------------------------------------------------------------------
# Two-rank data-parallel rehearsal on the one GPU (tests/test_zz_gpu_dist.py): the two child processes are
# started HERE, at session start, before this pytest process has made any GPU call (a process that has
# initialised the GPU must not start programs), and run to completion; the test then reads what they wrote.
# ------------------------------------------------------------------------------------------------------------
def _launch_two_rank_job(config):
    import subprocess
    import tempfile
    import time
    out = tempfile.mkdtemp(prefix='gmt_two_rank_')
    config._gmt_two_rank = {'dir': out, 'status': 'not run'}
    port = 24000 + os.getpid() % 4000
    child = os.path.join(ROOT, 'tests', 'dist', 'two_rank_child.py')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, '-u', child, str(r), '2', str(port), out], env=env,
                              stdout=open(os.path.join(out, f'out{r}.log'), 'w'), stderr=subprocess.STDOUT)
             for r in range(2)]
    t0 = time.time()
    limit = 420.0                       # the first `import torch` on a fresh box alone can take two minutes
    while any(p.poll() is None for p in procs) and time.time() - t0 < limit:
        time.sleep(2.0)
        if int(time.time() - t0) % 20 < 2:
            sys.stderr.write(f'[two-rank rehearsal] {time.time() - t0:.0f}s ...\n')
            sys.stderr.flush()
    stalled = [p for p in procs if p.poll() is None]
    for p in stalled:                   # exactly the processes started above
        p.kill()
        p.wait()
    config._gmt_two_rank['status'] = 'stalled' if stalled else ('ok' if all(p.returncode == 0 for p in procs) else 'failed')
    config._gmt_two_rank['seconds'] = time.time() - t0


def pytest_sessionstart(session):
    config = session.config
    expr = config.getoption('markexpr', default='') or ''
    if 'not gpu' in expr or os.environ.get('GMT_SKIP_TWO_RANK'):
        return
    args = [str(a) for a in config.args]
    if any(a.endswith('.py') or '::' in a for a in args) and not any('test_zz_gpu_dist' in a for a in args):
        return                          # a run of other, named test files
    import torch
    if torch.cuda.device_count() < 1:   # (counting devices does not initialise the GPU)
        return
    _launch_two_rank_job(config)
