#!/usr/bin/env python3
"""AdamW (split master) pass over a CSM-1B-sized arena for several grid sizes (csm_set_adamw_blocks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
n = 1_550_000_128 // 8 * 8
lo = torch.zeros(n, dtype=torch.int16, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
p = torch.zeros(n, dtype=torch.bfloat16, device=dev); g = torch.full((n,), 0.01, dtype=torch.bfloat16, device=dev)
coef = torch.tensor([1.0, 1.0], device=dev)
def t(k=5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(k): ops.adamw_step_split(lo, m, v, p, g, 1e-4, 0.9, 0.999, 1e-8, 0.01, i + 1, coef)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for blocks in (4096, 16384, 65536, 262144, 1048576):
    ops.lib.csm_set_adamw_blocks(blocks)
    t(2)
    ms = t()
    print(f"blocks {blocks:7d}: {ms:.3f} ms  {26 * n / ms / 1e6:.0f} GB/s", flush=True)
