#!/usr/bin/env python3
"""One GEMM shape, a few launches - the workload for rocprofv3 --pmc passes over the 256x256 kernel (tools/probes/gemm_pmc.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
mode, M, N, K = os.environ.get("GP_MODE", "nn"), int(os.environ.get("GP_M", 8192)), int(os.environ.get("GP_N", 2048)), int(os.environ.get("GP_K", 16384))
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
if mode == "nt": A, B, tA, tB = rnd(M, K), rnd(N, K), False, False
elif mode == "nn": A, B, tA, tB = rnd(M, K), rnd(K, N), False, True
else: A, B, tA, tB = rnd(K, M), rnd(K, N), True, True
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(int(os.environ.get("GP_REPS", 6))):
    ops.gemm(A, B, C, None, tA, tB)
torch.cuda.synchronize()
