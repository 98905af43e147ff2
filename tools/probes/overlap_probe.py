#!/usr/bin/env python3
"""Can the HBM-bound AdamW pass hide under MFMA-bound GEMMs?  Times 0.6 G-parameter AdamW (side stream) and a chain of
forward-shaped GEMMs (main stream) serially and concurrently, for a few AdamW grid sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops

dev = "cuda"
n = 600_000_000
master = torch.zeros(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
p = torch.zeros(n, dtype=torch.bfloat16, device=dev); g = torch.randn(n, device=dev).to(torch.bfloat16)
x = (torch.randn(8192, 2048, device=dev) * 0.5).to(torch.bfloat16)
w13 = (torch.randn(16384, 2048, device=dev) * 0.02).to(torch.bfloat16)
w2 = (torch.randn(2048, 8192, device=dev) * 0.02).to(torch.bfloat16)
gu = torch.empty(8192, 16384, dtype=torch.bfloat16, device=dev); act = torch.empty(8192, 8192, dtype=torch.bfloat16, device=dev)
out = torch.empty(8192, 2048, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream()

def gemms(k=8):
    for _ in range(k):
        ops.linear_swiglu_fwd(x, w13, gu, act)
        ops.linear_fwd(act, w2, out)

def adam():
    ops.adamw_step(master, m, v, p, g, 1e-5, 0.9, 0.999, 1e-8, 0.01, 1)

def timeit(fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts[1:])[len(ts[1:]) // 2]

def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        adam()
    gemms()
    torch.cuda.current_stream().wait_stream(side)

for blocks in (16384, 2048, 512, 256):
    ops.lib.csm_set_adamw_blocks(blocks)
    tg, ta = timeit(gemms), timeit(adam)
    tb = timeit(both)
    print(f"adamw blocks {blocks:6d}: gemms {tg:.2f} ms, adamw {ta:.2f} ms ({30.0 * n / ta / 1e9:.2f} TB/s), serial {tg + ta:.2f}, concurrent {tb:.2f} ms "
          f"-> hidden {(tg + ta - tb) / ta * 100:.0f}% of adamw")
