#!/usr/bin/env python3
"""GEMM micro-benchmark on the CSM-1B train-step shapes (random bf16 data, interleaved rounds in one process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops

M = int(os.environ.get("GB_M", 8192))
SHAPES = [  # (name, mode, M, N, K)
    ("qkv_fwd", "nt", M, 3072, 2048), ("o_fwd", "nt", M, 2048, 2048), ("w13_fwd", "nt", M, 16384, 2048), ("w2_fwd", "nt", M, 2048, 8192),
    ("qkv_dx", "nn", M, 2048, 3072), ("w13_dx", "nn", M, 2048, 16384), ("w2_dx", "nn", M, 8192, 2048),
    ("qkv_dw", "tn", 3072, 2048, M), ("w13_dw", "tn", 16384, 2048, M), ("w2_dw", "tn", 2048, 8192, M),
]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
ZERO = bool(os.environ.get("GB_ZERO"))     # all-zero operands: the clock the chip holds depends on the data (DVFS)
def rnd(*s): return torch.zeros(*s, device=dev, dtype=torch.bfloat16) if ZERO else (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
cases = []
for name, mode, m, n, k in SHAPES:
    if mode == "nt": A, B, tA, tB = rnd(m, k), rnd(n, k), False, False
    elif mode == "nn": A, B, tA, tB = rnd(m, k), rnd(k, n), False, True
    else: A, B, tA, tB = rnd(k, m), rnd(k, n), True, True
    C = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    cases.append((name, mode, m, n, k, A, B, C, tA, tB))
# correctness spot check on the first shape of each mode
for name, mode, m, n, k, A, B, C, tA, tB in cases:
    if name in ("o_fwd", "qkv_dx", "qkv_dw") and not os.environ.get("GB_NOCHECK"):
        ops.lib.csm_set_gemm_variant(1)
        ops.gemm(A, B, C, None, tA, tB)
        a = A.float().t() if tA else A.float(); b = B.float() if tB else B.float().t()
        ref = a[:256] @ b
        err = (C[:256].float() - ref).abs().max().item() / ref.abs().max().item()
        print(f"check {name}: rel err {err:.2e}")
        assert err < 2e-2
rounds = int(os.environ.get("GB_ROUNDS", 5))
variants = [int(v) for v in os.environ.get("GB_VARIANTS", "1,2").split(",")]
res = {(v, c[0]): [] for c in cases for v in variants}
for r in range(rounds + 1):
    for name, mode, m, n, k, A, B, C, tA, tB in cases:
        for v in variants:
            ops.lib.csm_set_gemm_variant(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm(A, B, C, None, tA, tB)
            e1.record(); torch.cuda.synchronize()
            if r: res[(v, name)].append(e0.elapsed_time(e1) / 5 * 1e-3)
for v in variants:
    tot_f = tot_t = 0
    for name, mode, m, n, k, *_ in cases:
        t = sorted(res[(v, name)])[len(res[(v, name)]) // 2]
        f = 2.0 * m * n * k
        tot_f += f; tot_t += t
        print(f"v{v} {name:8s} {mode} M={m:5d} N={n:5d} K={k:5d}  {t*1e6:8.1f} us  {f/t/1e12:7.1f} TF/s")
    print(f"v{v} TOTAL {tot_f/tot_t/1e12:.1f} TF/s")
