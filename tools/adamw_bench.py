#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
n = 973146112
dev = "cuda"
master = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
p = master.to(torch.bfloat16); g = (torch.randn(n, device=dev) * 0.01).to(torch.bfloat16)
nc = torch.ones(2, device=dev)
for z in (0, 1, 0, 1):
    for _ in range(2): ops.adamw_step(master, m, v, p, g, 1e-5, 0.9, 0.999, 1e-8, 0.01, 1, nc, zero_grad=bool(z))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5): ops.adamw_step(master, m, v, p, g, 1e-5, 0.9, 0.999, 1e-8, 0.01, 2 + i, nc, zero_grad=bool(z))
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    print(f"zero_grad={z}: {t*1e3:.3f} ms  {(28 + 2 * z) * n / t / 1e12:.2f} TB/s")
import ctypes
for b in (2048, 8192, 16384, 65536, 4096):
    ops.lib._handle if False else None
    fn = getattr(ops.lib, "csm_set_adamw_blocks"); fn.argtypes = [ctypes.c_int]; fn(b)
    for _ in range(2): ops.adamw_step(master, m, v, p, g, 1e-5, 0.9, 0.999, 1e-8, 0.01, 1, nc, zero_grad=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5): ops.adamw_step(master, m, v, p, g, 1e-5, 0.9, 0.999, 1e-8, 0.01, 2 + i, nc, zero_grad=True)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    print(f"blocks={b}: {t*1e3:.3f} ms  {30 * n / t / 1e12:.2f} TB/s")
