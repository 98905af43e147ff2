"""The depth decoder's attention (512 frames x 32 positions, 8 q / 2 kv heads of 128) forward + backward, grouped mapping against
block-per-head (csm_set_attn_variant bit 14)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "csm-train-pytorch_amd"))
from csm.hip import ops
dev = torch.device("cuda:0")
B, S, H, KV, hd = 512, 32, 8, 2, 128
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, (H + 2 * KV) * hd, generator=g)).to(torch.bfloat16).to(dev)
dout = (torch.randn(B * S, H * hd, generator=g)).to(torch.bfloat16).to(dev)
out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
dqkv = torch.zeros_like(qkv)
delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
def timeit(f, n=100):
    for _ in range(10): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
base = 2 | (1 << 2) | (3 << 4) | (1 << 6) | (1 << 7)
for rep in range(2):
    for name, word in (("grouped", base), ("per_head", base | (1 << 14))):
        ops.lib.csm_set_attn_variant(word)
        tf = timeit(lambda: ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd))
        tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd))
        print(f"{name:9s} forward {tf:6.1f} us   backward (dQ + dK/dV) {tb:6.1f} us")
ops.lib.csm_set_attn_variant(0)
