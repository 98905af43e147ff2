"""Per-workgroup fixed cost of the asm dK/dV kernel: the same number of loop steps per CU at S=2048 (4 workgroups per CU) and
S=1024 (8 per CU), and at S=512 (16 per CU); rocprofv3-free: event timing of dQ+dK/dV minus the dQ kernel timed alone via variant
(the dQ kernel's time is printed from a separate run with the gen2 dK/dV, so look at the differences)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
H, KV, hd = 32, 8, 64
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, S in ((4, 2048), (16, 1024), (64, 512)):
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B * S, (H + 2 * KV) * hd, device=dev, generator=g).to(torch.bfloat16)
    dout = torch.randn(B * S, H * hd, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
    ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
    nk = S // 64
    steps = B * KV * sum(S // 32 - 2 * k for k in range(nk)) / 256
    wgs = B * KV * nk / 256
    ops.lib.csm_set_attn_variant(0)
    t = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd))
    print(f"B={B} S={S}: {steps:.0f} loop steps and {wgs:.0f} workgroups per CU: dQ + dK/dV {t:7.1f} us", flush=True)
