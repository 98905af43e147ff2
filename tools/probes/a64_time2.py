"""dK/dV kernel alone (event timing of attn_bwd minus nothing: prints the whole backward; the dQ kernel is constant across libs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
B, S, H, KV, hd = int(os.environ.get("AB_B", 4)), 2048, 32, 8, 64
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B * S, (H + 2 * KV) * hd, device=dev, generator=g).to(torch.bfloat16)
dout = torch.randn(B * S, H * hd, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for v, name in ((1270, "gen2 dK/dV"), (0, "asm, heaviest first"), (2048, "asm, pair by pair")):
    ops.lib.csm_set_attn_variant(v if v != 2048 else (246 | 2048))
    t = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd))
    print(f"  {name:22s}: attn_bwd (dQ + dK/dV) {t:7.1f} us", flush=True)
ops.lib.csm_set_attn_variant(0)
