#!/usr/bin/env python3
"""Do independent GEMMs on a second stream fill the tails (launch gap, prologue, epilogue drain, partial last round) of
the GEMMs on the first?  One backbone layer's backward GEMMs: the dX chain on stream A, the dW GEMMs on stream B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
M, d, F = 8192, 2048, 8192
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
dx, act, gu, hn, dh, o, xn = r(M, d), r(M, F), r(M, 2 * F), r(M, d), r(M, d), r(M, d), r(M, d)
dgu, dhn, do_, dqkv, dxn = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev), r(M, d), r(M, d), r(M, 3072), r(M, d)
w2, w13, wo, wqkv = r(d, F), r(2 * F, d), r(d, d), r(3072, d)
gw2, gw13, gwo, gwqkv = torch.zeros_like(w2), torch.zeros_like(w13), torch.zeros_like(wo), torch.zeros_like(wqkv)
side = torch.cuda.Stream()

def chain():
    ops.linear_dx_swiglu_bwd(dx, w2, gu, dgu)
    ops.linear_dx(dgu, w13, dhn)
    ops.linear_dx(dh, wo, do_)
    ops.linear_dx(dqkv, wqkv, dxn)

def wgrads():
    ops.linear_dw(dx, act, gw2)
    ops.linear_dw(dgu, hn, gw13)
    ops.linear_dw(dh, o, gwo)
    ops.linear_dw(dqkv, xn, gwqkv)

def serial(n=4):
    for _ in range(n):
        chain(); wgrads()

def two(n=4):
    for _ in range(n):
        chain()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            wgrads()
    torch.cuda.current_stream().wait_stream(side)

def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for _ in range(3):
    print(f"one stream {t(serial):.3f} ms   two streams {t(two):.3f} ms")
