#!/usr/bin/env python3
"""First contact of the four-wave GEMM kernel (variant 4) with hardware: tiny shapes first, each against variant 3."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
BF = torch.bfloat16
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(BF).to(dev)
shapes = [(256, 256, 64), (256, 256, 128), (256, 256, 192), (256, 256, 448), (512, 768, 256), (304, 520, 128), (1000, 2112, 1024)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for mode in (os.environ.get("W4_MODES", "nn,nt,tn").split(",")):
    for (M, N, K) in shapes:
        if mode == "nt": A, B, tA, tB = rnd(M, K), rnd(N, K), False, False
        elif mode == "nn": A, B, tA, tB = rnd(M, K), rnd(K, N), False, True
        else: A, B, tA, tB = rnd(K, M), rnd(K, N), True, True
        outs = {}
        for v in (3, 4):
            ops.lib.csm_set_gemm_variant(v)
            C = torch.zeros(M, N, dtype=BF, device=dev)
            ops.gemm(A, B, C, None, tA, tB)
            torch.cuda.synchronize()
            outs[v] = C
        ops.lib.csm_set_gemm_variant(2)
        ref = (A.float().t() if tA else A.float()) @ (B.float() if tB else B.float().t())
        e3 = (outs[3].float() - ref).abs().max().item() / ref.abs().max().item()
        e4 = (outs[4].float() - ref).abs().max().item() / ref.abs().max().item()
        print(f"{mode} {M}x{N}x{K}: rel err v3 {e3:.2e}  v4 {e4:.2e}  bit-equal {torch.equal(outs[3], outs[4])}", flush=True)
