#!/usr/bin/env python3
"""Idle time between kernels of the last train step in a rocprofv3 --kernel-trace CSV:
python tools/trace_gaps.py <kernel_trace.csv> [adamw launches per step = 4]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[0])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
adam = [i for i, r in enumerate(rows) if "adamw_" in r[2]]
assert len(adam) >= 2 * per, "need at least two steps in the trace"
lo, hi = adam[-per - 1] + 1, adam[-1]          # from just after the previous step's last AdamW launch to this step's last one
step = rows[lo:hi + 1]
busy = sum(e - s for s, e, _ in step)
span = step[-1][1] - step[0][0]
gaps = [(step[i + 1][0] - step[i][1], step[i][2][:50], step[i + 1][2][:50]) for i in range(len(step) - 1)]
pos = [g for g in gaps if g[0] > 0]
print(f"kernels {len(step)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms  "
      f"({len(pos)} gaps, mean {sum(g[0] for g in pos) / max(1, len(pos)) / 1e3:.2f} us)")
for g in sorted(gaps, key=lambda g: -g[0])[:8]:
    print(f"  {g[0] / 1e3:8.1f} us  after {g[1]}  before {g[2]}")
agg = {}
for s_, e_, n_ in step:
    k = n_.replace("void ", "").replace("(anonymous namespace)::", "")[:70]
    a = agg.setdefault(k, [0, 0])
    a[0] += e_ - s_; a[1] += 1
print("kernel time inside that step (no setup / data-generation kernels):")
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"  {t / 1e6:8.3f} ms  x{n:4d}  avg {t / n / 1e3:8.1f} us  {k}")
