#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS summary of one .hip source (hipcc -save-temps, gfx950):
    python tools/kernel_resources.py generative_mouse_trajectories_amd/csrc/dx_gemm.hip [extra hipcc flags]
Prints one line per kernel: VGPRs (arch + acc), SGPRs, scratch bytes per lane, static LDS."""
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
tmp = tempfile.mkdtemp()
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-function', '-c', src,
       '-o', os.path.join(tmp, 'x.o'), '-save-temps=obj'] + sys.argv[2:]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr)
asm = [f for f in os.listdir(tmp) if f.endswith('.s') and 'amdgcn' in f][0]
text = open(os.path.join(tmp, asm)).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', text, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body) or [None, '?'])[1]
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(.*', '', dem.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''))
    print(f'{dem:60s} next_free_vgpr {g("next_free_vgpr"):>4} accum_offset {g("accum_offset"):>4} sgpr {g("next_free_sgpr"):>3} '
          f'scratch {g("private_segment_fixed_size"):>5} B  lds {g("group_segment_fixed_size"):>6}')
