#!/usr/bin/env python3
"""Builds and runs the microbenchmarks on the GPU box and writes their raw output plus a parsed summary:
   gpurun_out/microbench/*.txt and gpurun_out/microbench/microbench.json  (copy the json to profiles/<round>/: bench.py reads the
   look-up ceilings of the scoring kernel from the newest profiles/*/microbench.json).
   usage (on the GPU box):  python3 tools/microbench/run_all.py"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "microbench")
os.makedirs(OUT, exist_ok=True)
res = {"source": "tools/microbench/*.hip on MI355X (gfx950), hipcc -O3"}
for name, args in (("gather_coalesce", []), ("gather8", []), ("gather_addr", []), ("dda_chain", []), ("persistent_step", ["-1"]), ("fp64_rate", [])):
    src = os.path.join(ROOT, "tools", "microbench", name + ".hip")
    exe = os.path.join("/tmp", "mb_" + name)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", src, "-o", exe])
    try:
        txt = subprocess.run([exe] + args, capture_output=True, text=True, timeout=180).stdout
    except subprocess.TimeoutExpired as e:
        txt = (e.stdout or b"").decode() + "\nTIMEOUT\n"
    open(os.path.join(OUT, name + ".txt"), "w").write(txt)
    res[name + "_raw"] = txt.strip().splitlines()

def grab(lines, key, pat):
    for l in lines:
        if key in l:
            m = re.search(pat, l)
            if m:
                return float(m.group(1))
    return None

gc = res["gather_coalesce_raw"]
res["gather_independent_lanes_per_s"] = (grab(gc, "mode 0", r"([\d.]+) G lanes/s") or 0) * 1e9 or None
res["gather_neighbour_quads_lanes_per_s"] = (grab(gc, "quads of neighbour lanes", r"([\d.]+) G lanes/s") or 0) * 1e9 or None
res["gather_one_line_lanes_per_s"] = (grab(gc, "random lanes in 16x1 cells", r"([\d.]+) G lanes/s") or 0) * 1e9 or None
dd = res["dda_chain_raw"]
res["dda_clocks_per_step_statement_per_step"] = grab(dd, "(library)", r"([\d.]+) s_memtime")
res["dda_clocks_per_step_one_statement_per_word"] = grab(dd, "library order, 32 steps", r"([\d.]+) s_memtime")
fr = res["fp64_rate_raw"]
for key, tag in (("fp64_mul_ticks_per_instr_one_wavefront", "v_mul_f64           256"), ("fp64_mul_ticks_per_instr_four_wavefronts", "v_mul_f64          1024")):
    res[key] = grab(fr, tag, r"-> ([\d.]+) ticks")
ps = res["persistent_step_raw"]
for key, tag in (("four_launches_us", "L  four launches"), ("one_launch_us", "P  one launch"), ("two_launches_us", "P2 two launches")):
    vals = [float(m.group(1)) for l in ps if tag in l for m in [re.search(r"([\d.]+) us per step", l)] if m]
    res["persistent_step_" + key] = min(vals) if vals else None
json.dump(res, open(os.path.join(OUT, "microbench.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.endswith("_raw")}, indent=1))
