#!/usr/bin/env python3
"""RMSNorm backward / column-sum timing at the backbone shape (M = 8192, D = 2048)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
M, D = 8192, 2048
dev = "cuda"
x = torch.randn(M, D, device=dev).to(torch.bfloat16); dy = torch.randn(M, D, device=dev).to(torch.bfloat16)
dres = torch.randn(M, D, device=dev).to(torch.bfloat16); w = torch.ones(D, device=dev, dtype=torch.bfloat16)
rstd = torch.rand(M, device=dev) + 0.5; dx = torch.empty_like(x)
nb = ops.lib.csm_rmsnorm_bwd_blocks()
parts = torch.empty(nb, D, device=dev); gw = torch.zeros(D, device=dev, dtype=torch.bfloat16)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = t(lambda: ops.rmsnorm_bwd(x, w, rstd, dy, dx, dres, parts))
b = t(lambda: ops.colsum_bf16(parts, gw, accumulate=True))
print(f"rmsnorm_bwd {a:.1f} us ({4 * M * D * 2 / a / 1e6:.2f} TB/s)   colsum[{nb}x{D}] {b:.1f} us")
