#!/usr/bin/env python3
"""Generated-asm dK/dV attention kernel (attention64_asm.hip) against the second-generation kernel (attention64.hip) and the
fp32 oracle on several shapes, run-to-run bit-identity, and timing at the backbone's shape.  variant 5366 = defaults with both asm kernels off (bits 10, 12); the
asm kernel off (csm_set_attn_variant bit 10)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from csm.hip import ops
from oracle import csm_oracle as O
dev = "cuda"
OFF = 1270          # defaults with the asm dK/dV kernel off
ON = 246 | 4096     # defaults + the asm dQ kernel


def run(B, S, H, KV, hd=64, seed=0, rope=False, check_ref=True):
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B * S, (H + 2 * KV) * hd, generator=g)).to(torch.bfloat16)
    dout = (torch.randn(B * S, H * hd, generator=g)).to(torch.bfloat16)
    qd, dd = qkv.to(dev), dout.to(dev)
    out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_fwd(qd, out, lse, B, S, H, KV, hd)
    delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
    table = None
    if rope:
        from csm.models.model import llama3_rope_table
        table = llama3_rope_table(S, hd, 500000.0, 32.0).to(dev).contiguous()
    res = {}
    for v in (OFF, ON, ON):
        ops.lib.csm_set_attn_variant(v)
        dqkv = torch.full_like(qd, float("nan"))
        ops.attn_bwd(qd, out, dd, lse, dqkv, delta, B, S, H, KV, hd, rope_table=table)
        torch.cuda.synchronize()
        res.setdefault(v, []).append(dqkv.float().cpu())
    ops.lib.csm_set_attn_variant(0)
    old, new, new2 = res[OFF][0], res[ON][0], res[ON][1]
    kv = slice(0, None)
    sc = old[:, kv].abs().max().item()
    d = (old[:, kv] - new[:, kv]).abs().max().item()
    msg = f"B={B} S={S} H={H} KV={KV} rope={rope}: |asm - gen2| max {d:.3e} (scale {sc:.3e}) repeat-identical {torch.equal(new, new2)} nan {bool(torch.isnan(new).any())}"
    if check_ref and not rope:
        qr = qkv.float().requires_grad_(True)
        q = qr[:, :H * hd].view(B, S, H, hd); k = qr[:, H * hd:(H + KV) * hd].view(B, S, KV, hd); vv = qr[:, (H + KV) * hd:].view(B, S, KV, hd)
        O.attention(q, k, vv).reshape(B * S, H * hd).backward(dout.float())
        gr = qr.grad[:, kv]
        msg += f" | vs oracle: asm {(new[:, kv] - gr).abs().max().item():.3e} gen2 {(old[:, kv] - gr).abs().max().item():.3e} (ref scale {gr.abs().max().item():.3e})"
    print(msg, flush=True)
    if rope or d > 1e-2 * sc:
        q_, k_, v_ = slice(0, H * hd), slice(H * hd, (H + KV) * hd), slice((H + KV) * hd, None)
        for nm, sl in (("dq", q_), ("dk", k_), ("dv", v_)):
            dd_ = (old[:, sl] - new[:, sl]).abs()
            rr_ = (new[:, sl] - new2[:, sl]).abs()
            rows = torch.nonzero(dd_.max(1).values > 1e-2 * sc).flatten()
            print(f"   {nm}: asm-gen2 max {dd_.max().item():.3e}, asm repeat diff {rr_.max().item():.3e}, bad rows {rows[:12].tolist()} (of {rows.numel()})"
                  f" bad cols {torch.nonzero(dd_.max(0).values > 1e-2 * sc).flatten()[:16].tolist()}", flush=True)


if __name__ == "__main__":
    for sh in [(1, 64, 4, 1), (1, 128, 4, 1), (2, 192, 8, 2), (1, 512, 8, 2), (1, 2048, 4, 1)]:
        run(*sh)
    run(2, 256, 8, 2, rope=True)
    run(2, 256, 8, 2, rope=False)
    run(1, 64, 4, 1, rope=True)
    if os.environ.get("A64_TIME", "1") == "1":
        B, S, H, KV, hd = 4, 2048, 32, 8, 64
        g = torch.Generator(device=dev).manual_seed(0)
        qkv = torch.randn(B * S, (H + 2 * KV) * hd, device=dev, generator=g).to(torch.bfloat16)
        dout = torch.randn(B * S, H * hd, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
        lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
        dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
        ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
        def timeit(fn, n=20):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        for rep in range(2):
            for v in (OFF, 0):
                ops.lib.csm_set_attn_variant(v)
                t = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd))
                print(f"variant {v}: attn_bwd (dQ + dK/dV) {t:7.1f} us", flush=True)
        ops.lib.csm_set_attn_variant(0)
