"""Cycle stamps of the asm dQ kernel (csm_attn64_set_debug): per workgroup item the cycles of the part before the loop, of the
loop (per 32-key tile) and of the epilogue, and the shader clock (cycle counter against the 100 MHz real-time counter)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
dev = "cuda"
B, S, H, KV, hd = 4, 2048, 32, 8, 64
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B * S, (H + 2 * KV) * hd, device=dev, generator=g).to(torch.bfloat16)
dout = torch.randn(B * S, H * hd, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
for _ in range(5):
    ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd)
dbg = torch.zeros(1024 * 4 * 8 * 8, dtype=torch.int64, device=dev)
ops.lib.csm_attn64_set_debug(dbg.data_ptr())
ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd)
torch.cuda.synchronize()
ops.lib.csm_attn64_set_debug(None)
d = dbg.cpu().view(1024, 4, 8, 8)
rows = d[d[..., 0] > 0].double()          # [items, 8]
n, st0, st1, st2, st3, rt0, rt1 = [rows[:, i] for i in range(7)]
clk = (st3 - st0) / ((rt1 - rt0) / 100e6) / 1e9
print(f"{rows.shape[0]} (wave, item) records; shader clock {clk.mean():.2f} GHz (min {clk.min():.2f}, max {clk.max():.2f})")
print(f"before the loop: {(st1 - st0).mean():8.0f} cycles (max {(st1 - st0).max():.0f})")
print(f"loop           : {((st2 - st1) / n).mean():8.0f} cycles per 32-key tile (24 MFMAs = 768), fixed part by fit below")
print(f"after the loop : {(st3 - st2).mean():8.0f} cycles (max {(st3 - st2).max():.0f})")
x = d[..., 7][d[..., 0] > 0]
print(f"   of which: block 1 (requests only) {(x >> 32).double().mean():.0f} cycles, through the delta computation {(x & 0xffffffff).double().mean():.0f}")
for rd in range(8):
    rr = d[:, :, rd, :]
    rr = rr[rr[..., 0] > 0].double()
    if len(rr):
        print(f"   round {rd}: tiles {rr[:, 0].mean():5.1f}  before {(rr[:, 2] - rr[:, 1]).mean():8.0f}  loop/tile {((rr[:, 3] - rr[:, 2]) / rr[:, 0]).mean():7.0f}  after {(rr[:, 4] - rr[:, 3]).mean():7.0f}")
# least squares loop = a + b n
A = torch.stack([torch.ones_like(n), n], 1)
sol = torch.linalg.lstsq(A, (st2 - st1).unsqueeze(1)).solution.flatten()
print(f"loop cycles ~ {sol[0]:.0f} + {sol[1]:.0f} x tiles")
first = d[:, 0, :, :]                      # wave 0 of each workgroup
tot = []
for w in range(first.shape[0]):
    r = first[w][first[w][:, 0] > 0]
    if len(r):
        tot.append(float(r[:, 4].max() - r[:, 1].min()))
print(f"workgroup lifetime (first stamp to last): mean {sum(tot) / len(tot):.0f} cycles, max {max(tot):.0f} ({len(tot)} workgroups)")
