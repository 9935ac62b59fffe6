"""In-tree build of libgmt_hip.so (gfx950 only).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(PKG, 'libgmt_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']
# per-file extra flags: the f64 telemetry kernel must not fuse a*b+c (bit parity with MT.rs)
EXTRA = {'telemetry.hip': ['-ffp-contract=off'],
         # the per-trajectory loss arithmetic follows the CPU reference operation by operation (no fused multiply-add)
         'loss.hip': ['-ffp-contract=off']}


def _deps_mtime() -> float:
    hs = glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(PKG, '..', 'include', 'gmt_hip.h')]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, hm: float) -> str:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= hm):
        return obj
    cmd = [HIPCC] + BASE_FLAGS + EXTRA.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int = 8) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hm = _deps_mtime()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hm), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
