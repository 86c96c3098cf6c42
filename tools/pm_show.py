#!/usr/bin/env python3
"""prints the per-particle-map bench reports under gpurun_out/pm/ (tools/pm_bench.sh)"""
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/pm/*_report.json")):
    if len(sys.argv) > 1 and not any(a in f for a in sys.argv[1:]):
        continue
    try:
        d = json.load(open(f))["per_particle_maps"]
    except Exception as e:
        print(f, "ERR", e); continue
    k = d["kernels"]
    g = lambda n: ("%.1f" % k[n]["avg_launch_us"]) if n in k else "-"
    print(f"{f.split('/')[-1][:-12]:28s} update {d['update_ms']*1e3:8.1f} us  resample {d['resample_ms']*1e3:7.1f} us | lik {g('likelihood'):>7} particle {g('score'):>7} reduce {g('reduce'):>5} "
          f"resample {g('resample'):>5} copy {g('mapcopy'):>7} | lik frac {k.get('likelihood',{}).get('hbm_frac',0):.3f}")
