"""Repeatability stress of the asm dK/dV kernel: N runs per configuration, every result compared with the first and with the
second-generation kernel's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
from csm.models.model import llama3_rope_table
dev = "cuda"
N = int(os.environ.get("N", 40))
for (B, S, H, KV) in [(1, 2048, 4, 1), (4, 2048, 32, 8), (2, 256, 8, 2)]:
    hd = 64
    g = torch.Generator().manual_seed(S + hd)
    qkv = torch.randn(B * S, (H + 2 * KV) * hd, generator=g).to(torch.bfloat16).to(dev)
    dout = torch.randn(B * S, H * hd, generator=g).to(torch.bfloat16).to(dev)
    out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
    delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
    table = llama3_rope_table(S, hd, 500000.0, 32.0).to(dev).contiguous()
    for rope in (None, table):
        ops.lib.csm_set_attn_variant(1270)
        ref = torch.zeros_like(qkv)
        ops.attn_bwd(qkv, out, dout, lse, ref, delta, B, S, H, KV, hd, rope_table=rope)
        ops.lib.csm_set_attn_variant(246 | 4096)
        first, bad = None, 0
        tsum0 = table.double().sum().item()
        tclone = table.clone()
        for i in range(N):
            dqkv = torch.full_like(qkv, float("nan"))
            ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd, rope_table=rope)
            if first is None:
                first = dqkv.clone()
                d = (first.float() - ref.float()).abs()
                print(f"B={B} S={S} H={H} KV={KV} rope={rope is not None}: first vs gen2 max {d.max().item():.3e}", flush=True)
            elif not torch.equal(first, dqkv):
                bad += 1
                d = (first.float() - dqkv.float()).abs()
                rows = torch.nonzero(d.max(1).values > 0).flatten()
                cols = torch.nonzero(d.max(0).values > 0).flatten()
                if bad <= 3:
                    print(f"   run {i}: differs max {d.max().item():.3e} rows {rows[:10].tolist()}..({rows.numel()}) cols {cols[:24].tolist()}..({cols.numel()})", flush=True)
        torch.cuda.synchronize()
        dt = (table - tclone).abs()
        print(f"   {bad} of {N - 1} repeats differ; rope table changed: {bool((dt > 0).any())} (max {dt.max().item():.3e}, cols {torch.nonzero(dt.view(S, -1).max(0).values > 0).flatten()[:8].tolist()})", flush=True)
        if rope is not None:
            ops.lib.csm_set_attn_variant(1270)
            g1 = torch.zeros_like(qkv); g2 = torch.zeros_like(qkv)
            ops.attn_bwd(qkv, out, dout, lse, g1, delta, B, S, H, KV, hd, rope_table=rope)
            ops.attn_bwd(qkv, out, dout, lse, g2, delta, B, S, H, KV, hd, rope_table=rope)
            print(f"   gen2 with rope repeat-identical: {torch.equal(g1, g2)}; gen2 now vs gen2 before the asm runs: {(g1.float() - ref.float()).abs().max().item():.3e}", flush=True)
            ops.lib.csm_set_attn_variant(0)
