#!/usr/bin/env python3
"""GEMM kernels on the depth-decoder shapes of the train step (M = 512 frames x 32 positions, d = 1024, F = 8192), where the
contraction is short (K = 1024): 128x128 kernel (variant 1, 2-3 workgroups per CU) vs 256x256 (variant 3, one per CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
M, d, F = int(os.environ.get("GB_M", 16384)), 1024, 8192
g = torch.Generator(device=dev).manual_seed(0)
def r(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
x, w13, w2 = r(M, d), r(2 * F, d), r(d, F)
gu, act, dy = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev), torch.empty(M, F, dtype=torch.bfloat16, device=dev), r(M, d)
dgu, y, dx = torch.empty_like(gu), torch.empty(M, d, dtype=torch.bfloat16, device=dev), torch.empty(M, d, dtype=torch.bfloat16, device=dev)
wq, qkv, wo = r(1536, d), torch.empty(M, 1536, dtype=torch.bfloat16, device=dev), r(d, d)
gw13, gw2 = torch.empty_like(w13), torch.empty_like(w2)
ops.linear_swiglu_fwd(x, w13, gu, act)
cases = [
    ("w13 fwd + swiglu", 2.0 * M * 2 * F * d, lambda: ops.linear_swiglu_fwd(x, w13, gu, act)),
    ("w13 fwd plain", 2.0 * M * 2 * F * d, lambda: ops.linear_fwd(x, w13, gu)),
    ("w2 fwd", 2.0 * M * F * d, lambda: ops.linear_fwd(act, w2, y)),
    ("w2 dx + swiglu bwd", 2.0 * M * F * d, lambda: ops.linear_dx_swiglu_bwd(dy, w2, gu, dgu)),
    ("w2 dx plain", 2.0 * M * F * d, lambda: ops.linear_dx(dy, w2, act)),
    ("w13 dx", 2.0 * M * 2 * F * d, lambda: ops.linear_dx(dgu, w13, dx)),
    ("w13 dW", 2.0 * M * 2 * F * d, lambda: ops.gemm(dgu, x, gw13, None, True, True)),
    ("w2 dW", 2.0 * M * F * d, lambda: ops.gemm(dy, act, gw2, None, True, True)),
    ("qkv fwd", 2.0 * M * 1536 * d, lambda: ops.linear_fwd(x, wq, qkv)),
    ("o fwd", 2.0 * M * d * d, lambda: ops.linear_fwd(x, wo, y)),
]
def t(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for name, fl, fn in cases:
    out = []
    for v in (1, 3, 2):
        ops.lib.csm_set_gemm_variant(v)
        best = min(t(fn) for _ in range(3))
        out.append(f"v{v} {best * 1e6:7.1f} us {fl / best / 1e12:7.1f} TF/s")
    ops.lib.csm_set_gemm_variant(2)
    print(f"{name:20s} " + " | ".join(out))
