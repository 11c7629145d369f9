#!/usr/bin/env python3
"""Pretty-print the JSON line of a bench.py log: headline + per-kernel table (sorted by time per step)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms/step", d.get("frac_of_fp32_mfma_peak", d.get("frac_of_bf16_mfma_peak")))
tot = 0.0
for k, v in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["total_ms_per_step"]):
    tot += v["total_ms_per_step"]
    print(f"{k:44s} {v['tflops']:7.2f} TF  {v['total_ms_per_step']:7.3f} ms/step  {v['avg_launch_ms']:.4f} ms x {v['launches_per_step']}")
print(f"matrix kernels {tot:.2f} ms/step, everything else {d['ms_per_step'] - tot:.2f}")
for k in ("roofline", "inference", "cpu_baseline"):
    if k in d:
        print(k, d[k])
