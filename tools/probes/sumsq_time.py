#!/usr/bin/env python3
"""sum of squares over a CSM-1B-sized bf16 gradient range, isolated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
n = 1_550_000_128
g = torch.full((n,), 0.01, dtype=torch.bfloat16, device="cuda")
parts = torch.empty(ops.sumsq_blocks(), device="cuda")
ops.sumsq_bf16(g, parts); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.sumsq_bf16(g, parts)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"sumsq {ms * 1e3:.1f} us  {2 * n / ms / 1e6:.0f} GB/s  sum {float(parts.sum()):.6g} (n * bf16(0.01)^2 = {n * float(torch.tensor(0.01).bfloat16()) ** 2:.6g})")
