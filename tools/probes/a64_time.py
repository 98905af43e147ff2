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
for v in (1270, 0, 246 | 4096):
    ops.lib.csm_set_attn_variant(v)
    for _ in range(20):
        ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
        ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd)
    torch.cuda.synchronize()
