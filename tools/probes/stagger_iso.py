#!/usr/bin/env python3
"""Isolated view of the four-wave GEMM's start offsets (csm_set_gemm_tuning keys 2..5) on the two fused-SwiGLU products."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
M, d, F = 8192, 2048, 8192
dy, w2, gu, dgu = rnd(M, d), rnd(d, F), rnd(M, 2 * F), torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev)
x, w13, act = rnd(M, d), rnd(2 * F, d), torch.empty(M, F, dtype=torch.bfloat16, device=dev)
gu2 = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev)
cases = {"w2_dx+swiglu_bwd": lambda: ops.linear_dx_swiglu_bwd(dy, w2, gu, dgu), "w13_fwd+swiglu": lambda: ops.linear_swiglu_fwd(x, w13, gu2, act)}
def t(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for G, fwd, bwd in [(1, 0, 0), (2, 20000, 20000), (2, 400, 900), (2, 800, 1800), (4, 250, 600), (4, 500, 1200), (8, 150, 300), (1, 0, 0)]:
    for k, v in ((2, G), (3, fwd), (4, bwd)):
        ops.lib.csm_set_gemm_tuning(k, v)
    print(f"groups {G} ticks fwd {fwd} bwd {bwd}: " + "  ".join(f"{n} {t(fn):7.1f} us" for n, fn in cases.items()), flush=True)
