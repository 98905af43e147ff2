#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --kernel-trace --stats CSV: python tools/prof_summary.py <kernel_stats.csv> <n_steps>"""
import csv, sys
f, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total GPU time per step: {tot/steps/1e6:.2f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('unsigned short', 'u16')[:72]
    print(f"{float(r['TotalDurationNs'])/steps/1e6:8.3f} ms/step  calls/step {int(r['Calls'])/steps:6.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {n}")
