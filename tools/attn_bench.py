#!/usr/bin/env python3
"""Attention micro-benchmark at the CSM-1B backbone shape (random bf16 data)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
B, S, H, KV, hd = int(os.environ.get("AB_B", 4)), 2048, 32, 8, 64
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * S, (H + 2 * KV) * hd, device=dev, generator=g)).to(torch.bfloat16)
dout = (torch.randn(B * S, H * hd, device=dev, generator=g)).to(torch.bfloat16)
out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(2, *lse.shape, dtype=lse.dtype, device=lse.device)
fl = 4.0 * B * H * S * S / 2 * hd
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for v in [int(x) for x in os.environ.get('AB_VARIANTS', '768,0,768,0').split(',')]:
    ops.lib.csm_set_attn_variant(v)
    tf = timeit(lambda: ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd))
    tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd))
    print(f"variant={v} (768 = first-generation kernels, 0 = default): fwd {tf*1e6:7.1f} us {fl/tf/1e12:6.1f} TF/s | bwd {tb*1e6:7.1f} us {2.5*fl/tb/1e12:6.1f} TF/s (2.5x fwd flop)")
