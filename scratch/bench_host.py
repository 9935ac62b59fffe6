import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f, round(d['ms_per_step'], 2), {k: round(v, 1) for k, v in d['host_ms_per_step'].items()})
