#!/usr/bin/env python3
"""Print the interesting parts of a bench.py JSON line: python tools/show_bench.py <file>"""
import json, sys
d = json.load(open(sys.argv[1]))
print(d["ms_per_step"], "ms/step", d["value"], d["unit"], "mfma_utilisation_step", d["mfma_utilisation_step"], "rccl", d.get("rccl"))
r = d["roofline"]
print("roofline:", r["kernel"], r["kinds"], r["achieved"], r["unit"], "frac", r["frac"], "traffic", r["traffic"], "operands", r["operand_bytes_per_launch"],
      "avg us", r["avg_launch_us"], "launches", r["launches_per_step"])
for k, v in r["all_gemm_kernels"].items():
    print("   ", k, v)
print("hbm_kernel:", d.get("hbm_kernel"))
for k, v in d.get("extra", {}).items():
    print("extra", k, {kk: vv for kk, vv in v.items() if kk not in ("workload", "roofline", "what", "includes")}, (v.get("roofline") or {}).get("frac"))
print("cpu_baseline:", d.get("cpu_baseline"))
print("parity_check:", d.get("parity_check"))
