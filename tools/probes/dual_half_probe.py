#!/usr/bin/env python3
"""Does running the two halves of a batch as two independent streams raise the chip's utilisation?

A one-workgroup-per-CU GEMM runs the whole chip in lockstep: every CU is in its main loop (HBM idle) and then every CU
is in its epilogue (MFMA idle).  Two half-batch chains on two HIP streams de-phase that: one half's epilogues, norms and
attention tails run beside the other half's main loops.  This probe times the backbone stack (16 layers, forward +
backward with weight gradients, no optimiser) of CSM-1B at S=2048:

    single      B=4 on one stream                      (what the engine does today)
    dual        2 x B=2 on two plain streams           (host enqueues A's whole pass, then B's)
    dual-thr    the same, each stream fed by its own host thread
    dual-mask   2 x B=2 on two streams with complementary CU masks (hipExtStreamCreateWithCUMask), if the call works

The two halves write the same gradient arena without ordering - fine for a timing probe, not for the product.
"""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.engine import _Stack
from csm.models.model import Model
from csm.training.trainer import csm_1b_args

S = 2048
model = Model(csm_1b_args(), device="cuda:0", seed=0)
model.ensure_grads()
dev = model.device
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.randn(4 * S, 2048, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dy = (torch.randn(4 * S, 2048, device=dev, generator=g) * 0.01).to(torch.bfloat16)
full, A, B = _Stack(model, "backbone"), _Stack(model, "backbone"), _Stack(model, "backbone")


def fb(stack, xs, dys, nb):
    stack.forward(xs, nb, S, True)
    stack.backward(dys, nb, S, True, 1.0)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def single():
    fb(full, x, dy, 4)


def make_dual(s1, s2, threads=False):
    def run_a():
        with torch.cuda.stream(s1):
            fb(A, x[:2 * S], dy[:2 * S], 2)

    def run_b():
        with torch.cuda.stream(s2):
            fb(B, x[2 * S:], dy[2 * S:], 2)

    def dual():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        if threads:
            ta, tb = threading.Thread(target=run_a), threading.Thread(target=run_b)
            ta.start(); tb.start(); ta.join(); tb.join()
        else:
            run_a(); run_b()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return dual


print(f"single            {timed(single):8.3f} ms   (backbone fwd+bwd, B=4)")
p1, p2 = torch.cuda.Stream(), torch.cuda.Stream()
print(f"dual              {timed(make_dual(p1, p2)):8.3f} ms   (2 x B=2, two plain streams)")
print(f"dual-thr          {timed(make_dual(p1, p2, True)):8.3f} ms   (2 x B=2, two plain streams, two host threads)")
print(f"single            {timed(single):8.3f} ms")

# ---- CU-masked streams
try:
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    cen = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "libxcc_census.so"))
    cen.census.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]

    def masked_stream(bits):
        words = (C.c_uint32 * 8)()
        for i in bits:
            words[i // 32] |= 1 << (i % 32)
        h = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
        return torch.cuda.ExternalStream(h.value)

    def census(stream, label):
        out = torch.full((2 * 1024,), -1, dtype=torch.int32, device=dev)
        cen.census(stream.cuda_stream, out.data_ptr(), 1024, 20000)
        torch.cuda.synchronize()
        o = out.view(-1, 2).cpu()
        xccs = sorted(set(int(v) for v in o[:, 0]))
        cus = len(set((int(a), int(b) & 0xFF00 | (int(b) >> 13 & 7) << 16 | (int(b) >> 12 & 1) << 20) for a, b in o.tolist()))
        print(f"  census {label}: XCCs {xccs}, distinct (xcc, se, sh, cu) = {cus}")

    for name, (bits_a, bits_b) in {
        "xcd-split (bit i -> xcd i%8; xcds 0-3 | 4-7)": ([i for i in range(256) if i % 8 < 4], [i for i in range(256) if i % 8 >= 4]),
        "contiguous (bits 0-127 | 128-255)": (list(range(128)), list(range(128, 256))),
    }.items():
        m1, m2 = masked_stream(bits_a), masked_stream(bits_b)
        print(name)
        census(m1, "A"); census(m2, "B")
        print(f"dual-mask         {timed(make_dual(m1, m2)):8.3f} ms")
        print(f"dual-mask-thr     {timed(make_dual(m1, m2, True)):8.3f} ms")
except Exception as e:  # noqa: BLE001
    print("CU-mask part failed:", repr(e))
print(f"single            {timed(single):8.3f} ms")
