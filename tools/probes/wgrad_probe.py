#!/usr/bin/env python3
"""Small-output weight gradients: direct GEMM (auto tile) vs split-K slabs + column sum, per shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K in (("o_proj dW", 8192, 2048, 2048), ("qkv dW", 8192, 3072, 2048), ("dec qkv dW", 16384, 1536, 1024),
                      ("dec o dW", 16384, 1024, 1024), ("dec w2 dW", 16384, 1024, 8192), ("dec w13 dW", 16384, 16384, 1024)):
    dy = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16); x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    res = []
    res.append(("heuristic", t(lambda: ops.linear_dw(dy, x, out, accumulate=True))))
    res.append(("direct", t(lambda: ops.gemm(dy, x, out, out, True, True))))
    for v in (1, 3):  # 128x128 LDS-DMA kernel, forced 256x256 kernel
        ops.lib.csm_set_gemm_variant(v)
        res.append((f"direct v{v}", t(lambda: ops.gemm(dy, x, out, out, True, True))))
    ops.lib.csm_set_gemm_variant(2)
    for splits in (2, 4, 8, 16):
        chunk = M // splits
        ws = torch.empty(splits, N * K, dtype=torch.float32, device=dev)
        def f():
            ops.gemm(dy[:chunk], x[:chunk], ws[0].view(N, K), None, True, True, 1.0, batch=splits, sA=chunk * dy.stride(0), sB=chunk * x.stride(0), sC=N * K)
            ops.colsum_bf16(ws, out.view(-1), accumulate=True)
        res.append((f"split{splits}", t(f)))
    fl = 2.0 * M * N * K
    print(f"{name:11s} M={M} N={N} K={K}: " + "  ".join(f"{k} {v:.0f}us({fl / v / 1e6:.0f}TF)" for k, v in res))
