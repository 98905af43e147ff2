#!/usr/bin/env python3
"""A/B of the 256x256 kernel's epilogue-read prefetch (csm_set_gemm_tuning(0, v)) on the two fused products that read a
tile-sized operand in their epilogue: w2 dgrad + SwiGLU backward, and o_proj / w2 forward + bf16 residual."""
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
x, wo, res, y = rnd(M, d), rnd(d, d), rnd(M, d), torch.empty(M, d, dtype=torch.bfloat16, device=dev)
act, w2f = rnd(M, F), rnd(d, F)
cases = {
    "w2_dx+swiglu_bwd": (lambda: ops.linear_dx_swiglu_bwd(dy, w2, gu, dgu), 2.0 * M * F * d),
    "o_fwd+residual": (lambda: ops.gemm(x, wo, y, res, False, False), 2.0 * M * d * d),
    "w2_fwd+residual": (lambda: ops.gemm(act, w2f, y, res, False, False), 2.0 * M * d * F),
}
outs = {}
for r in range(6):
    for name, (fn, fl) in cases.items():
        for v in (0, 1):
            ops.lib.csm_set_gemm_tuning(0, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            if r: outs.setdefault((name, v), []).append(e0.elapsed_time(e1) / 5 * 1e-3)
            if r == 0: outs[(name, v, "bits")] = (dgu if "swiglu" in name else y).clone()
for name, (fn, fl) in cases.items():
    assert os.environ.get("TOUCH_NOCHECK") or torch.equal(outs[(name, 0, "bits")], outs[(name, 1, "bits")]), name
    for v in (0, 1):
        t = sorted(outs[(name, v)])[len(outs[(name, v)]) // 2]
        print(f"{name:20s} touch={v}  {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s")
