import json, sys
for f in sys.argv[1:]:
    line = [l for l in open(f) if l.startswith('{')][-1]
    d = json.loads(line); k = d['roofline']['all_kernels_ms_per_step']
    print(f, round(d['ms_per_step'], 2), 'frac', round(d['roofline']['frac'], 4), 'avg_ms', round(d['roofline']['avg_launch_ms'], 3),
          {n: round(v, 2) for n, v in k.items() if n.startswith('lstm_fwd') or n == 'lstm_bwd'})
    print('   host', {n: round(v, 2) for n, v in d.get('host_ms_per_step', {}).items()}, 'traj/s', round(d['value']))
