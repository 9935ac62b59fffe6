import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "step_tflops", "hbm_peak_allocated_gb") if k in d}, {k: v for k, v in d.items() if k.startswith('frac_of')})
r = d["roofline"]
print('dominant:', r["kernel"], round(r["achieved"], 1), r["unit"], round(r["frac"], 3), 'avg ms', round(r["avg_launch_ms"], 3))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f'{k:28s} {v["launches_per_step"]:6.2f} {v["ms_per_step"]:8.2f} ms', round(v.get("achieved", 0)), v.get("unit", ""))
