#!/usr/bin/env python3
"""The kernel sequence of the last train step of a rocprofv3 --kernel-trace CSV, in launch order: start offset, duration, the gap in
front of it, name.  python tools/probes/step_sequence.py <kernel_trace.csv> [adamw launches per step = 4] [only kernels shorter than N us]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[0])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
short = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
adam = [i for i, r in enumerate(rows) if "adamw_" in r[2]]
lo, hi = adam[-per - 1] + 1, adam[-1]
t0 = rows[lo][0]
tot_small = 0.0
n_small = 0
for i in range(lo, hi + 1):
    s, e, n = rows[i]
    gap = (s - rows[i - 1][1]) / 1e3
    d = (e - s) / 1e3
    if d < 10: tot_small += d + max(gap, 0); n_small += 1
    if d < short:
        print(f"{(s - t0) / 1e3:10.1f} us  {d:8.1f} us  gap {gap:6.1f}  {n.replace('void ', '').replace('(anonymous namespace)::', '')[:120]}")
print(f"kernels shorter than 10 us: {n_small}, their time + the gap in front of each: {tot_small / 1e3:.3f} ms")
