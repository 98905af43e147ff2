#!/usr/bin/env python3
"""Mimi codec on the GPU: encode / decode time for 10 s of 24 kHz audio (seeded random weights with the HF key names)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from transformers import MimiConfig, MimiModel
from csm.codec import MimiCodec

torch.manual_seed(0)
hf = MimiModel(MimiConfig()).eval()
with torch.no_grad():
    for n, b in hf.named_buffers():
        if n.endswith("embed_sum"):
            b.copy_(torch.randn(b.shape))
codec = MimiCodec(hf.state_dict(), device="cuda")
wav = torch.randn(1, 1, 240000) * 0.1
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r
te, codes = t(lambda: codec.encode(wav))
td, out = t(lambda: codec.decode(codes))
print(f"encode 10 s: {te*1e3:.1f} ms ({10/te:.0f}x real time) -> codes {tuple(codes.shape)};  decode: {td*1e3:.1f} ms ({10/td:.0f}x real time) -> {tuple(out.shape)}")
