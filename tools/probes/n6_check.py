"""256 x 192 tile kernel against the 256 x 256 one: where do they differ?  python tools/probes/n6_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "csm-train-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from csm.hip import ops
from oracle import csm_oracle as O
dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator().manual_seed(5)
def rnd(shape, s=1.0): return (torch.randn(shape, generator=g) * s).to(BF).to(dev)
def report(tag, a, b):
    d = (a.float() - b.float()).abs()
    bad = (d > 0).nonzero()
    if bad.numel() == 0:
        print(tag, "equal"); return
    rows, cols = bad[:, 0], bad[:, 1]
    print(tag, "DIFF n=", bad.shape[0], "max", d.max().item(), "rows", rows.min().item(), rows.max().item(), "cols", cols.min().item(), cols.max().item(),
          "cols%192 set", sorted(set((cols % 192 // 16).tolist())), "rows%256//16", sorted(set((rows % 256 // 16).tolist()))[:16])
for (M, N, K, tA, rope) in [(256, 768, 64, False, None), (256, 768, 128, False, None), (256, 768, 192, False, None), (256, 768, 512, False, (128, 640)), (256, 768, 192, True, None),
                            (4096, 768, 512, False, (512, 640)), (8192, 3072, 2048, False, (2048, 2560))]:
    A = rnd((K, M) if tA else (M, K)); B = rnd((N, K), 0.1); R = rnd((M, N))
    out = {}
    for on in (1, 0):
        ops.lib.csm_set_gemm_tuning(8, on); ops.lib.csm_set_gemm_variant(4)
        C = torch.empty(M, N, dtype=BF, device=dev); ops.gemm(A, B, C, None, tA, False)
        name = ops.lib.csm_gemm_last_kernel().decode()
        acc = R.clone(); ops.gemm(A, B, acc, acc, tA, False, alpha=0.5)
        res = [C, acc]
        if rope and not tA:
            S, p0 = rope
            table = O.rope_table(S, 64).to(dev).contiguous()
            q = torch.empty(M, N, dtype=BF, device=dev); ops.linear_rope_fwd(A, B, q, table, S, p0, 64); res.append(q)
        out[on] = (name, res)
    torch.cuda.synchronize()
    print(M, N, K, tA, rope, out[1][0], out[0][0])
    for i, (a, b) in enumerate(zip(out[1][1], out[0][1])): report(f"   out{i}", a, b)

# which one is right?  fp32 accumulators from the fp32-output kernel, the rotation in float64
M, N, K, S, p0 = 4096, 768, 512, 512, 640
g = torch.Generator().manual_seed(77)
A = rnd((M, K)); B = rnd((N, K), 0.1)
table = O.rope_table(S, 64).to(dev).contiguous()
acc32 = torch.empty(M, N, dtype=torch.float32, device=dev)
ops.lib.csm_set_gemm_variant(4)
ops.gemm(A, B, acc32, None, False, False)
outs = {}
for on in (1, 0):
    ops.lib.csm_set_gemm_tuning(8, on)
    q = torch.empty(M, N, dtype=BF, device=dev); ops.linear_rope_fwd(A, B, q, table, S, p0, 64); outs[on] = q
ops.lib.csm_set_gemm_tuning(8, 1); ops.lib.csm_set_gemm_variant(2)
a = acc32.double().view(M, N // 2, 2)
pos = torch.arange(M, device=dev) % S
cs = table[pos].double()                                   # [M, 32, 2]
c = cs[:, :, 0].repeat(1, N // 64); s = cs[:, :, 1].repeat(1, N // 64)
x0, x1 = a[..., 0], a[..., 1]
exact = torch.stack([x0 * c - x1 * s, x1 * c + x0 * s], -1).view(M, N)
exact[:, p0:] = acc32.double()[:, p0:]
bad = (outs[1].float() != outs[0].float()).nonzero()
print("diffs", bad.shape[0])
for (r, cc) in bad.tolist()[:12]:
    e = exact[r, cc].item()
    print(f"  ({r},{cc}) exact {e:.9f}  n6 {outs[1][r, cc].item():.6f}  w4 {outs[0][r, cc].item():.6f}  acc pair {acc32[r, cc & ~1].item():.7f} {acc32[r, cc | 1].item():.7f}")
for on in (1, 0):
    d = (outs[on].double() - exact).abs().max().item()
    print("max |kernel - exact|", "n6" if on else "w4", d)

# kernel time, tiles on / off (bench shapes: backbone q|k|v 8192 x 3072 x 2048 with RoPE, decoder q|k|v 16384 x 1536 x 1024 plain + RoPE)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, N, K, S, p0) in [(8192, 3072, 2048, 2048, 2560), (16384, 1536, 1024, 2048, 1280), (8192, 1536, 2048, 2048, 0), (4096, 3072, 2048, 2048, 2560)]:
    A = rnd((M, K)); B = rnd((N, K), 0.1); C = torch.empty(M, N, dtype=BF, device=dev)
    table = O.rope_table(S, 64).to(dev).contiguous()
    for on in (1, 0, 1, 0):
        ops.lib.csm_set_gemm_tuning(8, on)
        f = (lambda: ops.linear_rope_fwd(A, B, C, table, S, p0, 64)) if p0 else (lambda: ops.gemm(A, B, C, None, False, False))
        t = timeit(f)
        print(f"{M}x{N}x{K} rope={p0} n6={on}: {t:.1f} us  {2 * M * N * K / t / 1e6:.0f} TFLOP/s  {ops.lib.csm_gemm_last_kernel().decode()}")
ops.lib.csm_set_gemm_tuning(8, 1)
