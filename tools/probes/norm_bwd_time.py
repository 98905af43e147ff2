#!/usr/bin/env python3
"""rmsnorm_bwd + its column sum at the backbone shape (8192 x 2048), isolated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
M, D = 8192, 2048
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
x, dy, dres, w = rnd(M, D), rnd(M, D), rnd(M, D), rnd(D)
rstd = torch.rand(M, device=dev) + 0.5
dx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
nb = ops.lib.csm_rmsnorm_bwd_blocks()
parts = torch.empty(nb, D, device=dev); gw = torch.zeros(D, dtype=torch.bfloat16, device=dev)
def t(fn, k=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
print(f"blocks {nb}: rmsnorm_bwd {t(lambda: ops.rmsnorm_bwd(x, w, rstd, dy, dx, dres, parts)):.1f} us   colsum {t(lambda: ops.colsum_bf16(parts, gw)):.1f} us")
