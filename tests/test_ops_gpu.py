"""Per-kernel parity: HIP (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerances: index / integer outputs bit-exact; bf16 kernels are compared with the oracle evaluated in fp32 on the
SAME bf16-rounded inputs, so the only error left is fp32-accumulate ordering + one bf16 output rounding
(relative 2^-8): tolerance = 2e-2 * output scale unless stated.
"""
import math

import pytest
import torch

from oracle import csm_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rnd(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(BF)


def close(name, got, ref, tol):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    assert math.isfinite(err) and err <= tol * scale, f"{name}: max abs err {err:.4g} vs scale {scale:.4g} (tol {tol})"
    return err / scale


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (200, 136, 72), (1000, 2112, 1024), (64, 8, 256), (300, 128, 8)])
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm(dev, M, N, K, mode):
    from csm.hip import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    if mode == "nt":      # C = A[M,K] B[N,K]^T
        A, B = rnd((M, K), g), rnd((N, K), g)
        ref = A.float() @ B.float().t()
        tA, tB = False, False
    elif mode == "nn":    # C = A[M,K] B[K,N]
        if N % 8:
            pytest.skip("N must be a multiple of 8 for [K][N] operands")
        A, B = rnd((M, K), g), rnd((K, N), g)
        ref = A.float() @ B.float()
        tA, tB = False, True
    else:                 # C = A[K,M]^T B[K,N]
        if N % 8 or M % 8:
            pytest.skip("M,N must be multiples of 8 for k-strided operands")
        A, B = rnd((K, M), g), rnd((K, N), g)
        ref = A.float().t() @ B.float()
        tA, tB = True, True
    R = rnd((M, N), g)
    Ad, Bd, Rd = A.to(dev), B.to(dev), R.to(dev)
    C = torch.empty(M, N, dtype=BF, device=dev)
    ops.gemm(Ad, Bd, C, None, tA, tB)
    close(f"gemm {mode}", C, ref, 1e-2)
    C32 = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(Ad, Bd, C32, Rd, tA, tB, alpha=0.5)
    close(f"gemm {mode} f32+R", C32, 0.5 * ref + R.float(), 2e-5 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 256), (304, 520, 128), (1000, 2112, 1024), (264, 8, 192)])
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm_256_tile_kernel(dev, M, N, K, mode, variant=3):
    """The deep-pipelined 256x256 kernel forced on (variant 3), including ragged edges and several K-tile counts
    (1, 2, 3, 4, 16 tiles exercise prologue / steady state / tail of the LDS-DMA pipeline)."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(M + N * 3 + K * 7)
    if mode == "nt":
        A, B, tA, tB = rnd((M, K), g), rnd((N, K), g), False, False
        ref = A.float() @ B.float().t()
    elif mode == "nn":
        A, B, tA, tB = rnd((M, K), g), rnd((K, N), g), False, True
        ref = A.float() @ B.float()
    else:
        A, B, tA, tB = rnd((K, M), g), rnd((K, N), g), True, True
        ref = A.float().t() @ B.float()
    R = rnd((M, N), g)
    ops.lib.csm_set_gemm_variant(variant)
    try:
        C = torch.empty(M, N, dtype=BF, device=dev)
        ops.gemm(A.to(dev), B.to(dev), C, None, tA, tB)
        close(f"gemm256 {mode}", C, ref, 1e-2)
        C32 = torch.empty(M, N, dtype=torch.float32, device=dev)
        for _ in range(3):   # repeated launches: a staging race would show up as run-to-run differences
            ops.gemm(A.to(dev), B.to(dev), C32, R.to(dev), tA, tB, alpha=0.5)
            close(f"gemm256 {mode} f32+R", C32, 0.5 * ref + R.float(), 2e-5 * math.sqrt(K))
    finally:
        ops.lib.csm_set_gemm_variant(2)


def test_gemm_256_many_tiles(dev):
    """More 256x256 tiles than CUs (several dispatch rounds), ragged edges, fp32 and bf16 outputs, residual."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(123)
    ops.lib.csm_set_gemm_variant(3)
    try:
        for (M, N, K, mode) in [(5000, 6000, 128, "nt"), (4360, 5120, 192, "nn"), (4096, 6152, 256, "tn"), (9000, 4000, 2048, "nt")]:
            if mode == "nt":
                A, B, tA, tB = rnd((M, K), g), rnd((N, K), g), False, False
            elif mode == "nn":
                A, B, tA, tB = rnd((M, K), g), rnd((K, N), g), False, True
            else:
                A, B, tA, tB = rnd((K, M), g), rnd((K, N), g), True, True
            Ad, Bd = A.to(dev), B.to(dev)
            a = Ad.float().t() if tA else Ad.float()
            b = Bd.float() if tB else Bd.float().t()
            ref = a @ b
            C = torch.empty(M, N, dtype=BF, device=dev)
            ops.gemm(Ad, Bd, C, None, tA, tB)
            close(f"many tiles {mode} {M}x{N}x{K}", C, ref, 1e-2)
            R = rnd((M, N), g).to(dev)
            C32 = torch.empty(M, N, dtype=torch.float32, device=dev)
            ops.gemm(Ad, Bd, C32, R, tA, tB, alpha=0.5)
            close(f"many tiles {mode} f32+R", C32, 0.5 * ref + R.float(), 2e-5 * math.sqrt(K))
            C2 = torch.empty(M, N, dtype=BF, device=dev)
            ops.gemm(Ad, Bd, C2, None, tA, tB)
            assert torch.equal(C, C2)
    finally:
        ops.lib.csm_set_gemm_variant(2)


@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm_256_persistent_rounds_with_residual(dev, mode):
    """Several rounds of FULL 256x256 tiles per workgroup (the persistent form: the next tile's loads are requested before the
    finished tile is stored and waited for with vmcnt counted past the epilogue's loads and stores) with a bf16 residual
    that aliases the output (gradient accumulation, C += A.B) and without: same bits as one tile per workgroup."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(77)
    M, N, K = 4096, 8192, 320                     # 512 tiles = 2 rounds; K = 5 K-tiles
    if mode == "nt":
        A, B, tA, tB = rnd((M, K), g), rnd((N, K), g), False, False
    elif mode == "nn":
        A, B, tA, tB = rnd((M, K), g), rnd((K, N), g), False, True
    else:
        A, B, tA, tB = rnd((K, M), g), rnd((K, N), g), True, True
    Ad, Bd = A.to(dev), B.to(dev)
    ref = (Ad.float().t() if tA else Ad.float()) @ (Bd.float() if tB else Bd.float().t())
    R = rnd((M, N), g).to(dev)
    out = {}
    try:
        for persistent in (1, 0):
            ops.lib.csm_set_gemm256_persistent(persistent)
            C = torch.empty(M, N, dtype=BF, device=dev)
            ops.gemm(Ad, Bd, C, None, tA, tB)
            acc = R.clone()
            ops.gemm(Ad, Bd, acc, acc, tA, tB, alpha=0.5)             # in place: C = R + alpha * A.B with R == C
            out[persistent] = (C, acc)
    finally:
        ops.lib.csm_set_gemm256_persistent(1)
    close(f"persistent {mode}", out[1][0], ref, 1e-2)
    close(f"persistent {mode} + R", out[1][1], 0.5 * ref + R.float(), 1e-2)
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1])


@pytest.mark.parametrize("mode,M,N,K", [("tn", 16384, 2048, 8192), ("tn", 2048, 8192, 8192), ("nn", 8192, 8192, 2048), ("nt", 8192, 16384, 2048)])
def test_gemm_256_persistent_long_contraction(dev, mode, M, N, K):
    """The persistent form at the train step's own shapes (w13 / w2 weight gradients: 128 K-tiles, operands of 100-300 MB,
    i.e. every K-tile comes from HBM, not from L2, so an LDS-DMA half-tile that is read before it has landed shows): same bits
    as one tile per workgroup, three times over, with and without accumulation into an aliasing bf16 output."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(K + M)
    if mode == "nt":
        A, B, tA, tB = rnd((M, K), g), rnd((N, K), g), False, False
    elif mode == "nn":
        A, B, tA, tB = rnd((M, K), g), rnd((K, N), g), False, True
    else:
        A, B, tA, tB = rnd((K, M), g), rnd((K, N), g), True, True
    Ad, Bd = A.to(dev), B.to(dev)
    R = rnd((M, N), g).to(dev)
    out = {}
    try:
        for persistent in (0, 1, 1, 1):
            ops.lib.csm_set_gemm256_persistent(persistent)
            C = torch.empty(M, N, dtype=BF, device=dev)
            ops.gemm(Ad, Bd, C, None, tA, tB)
            acc = R.clone()
            ops.gemm(Ad, Bd, acc, acc, tA, tB, alpha=0.5)
            if persistent == 0:
                out[0] = (C, acc)
            else:
                assert torch.equal(C, out[0][0]), f"{mode}: persistent != one tile per workgroup"
                assert torch.equal(acc, out[0][1]), f"{mode}: persistent != one tile per workgroup (accumulating)"
    finally:
        ops.lib.csm_set_gemm256_persistent(1)
    ref = (Ad.float().t() if tA else Ad.float()) @ (Bd.float() if tB else Bd.float().t())
    close(f"long-K {mode}", out[0][0], ref, 1e-2)


def test_gemm_256_bitwise_repeatable(dev):
    from csm.hip import ops
    g = torch.Generator().manual_seed(77)
    A, B = rnd((2048, 2048), g).to(dev), rnd((2048, 2048), g).to(dev)
    try:
        outs = []
        for v in (3, 3, 3):
            ops.lib.csm_set_gemm_variant(v)
            C = torch.empty(2048, 2048, dtype=torch.float32, device=dev)
            ops.gemm(A, B, C, None, False, False)
            outs.append(C)
        ops.lib.csm_set_gemm_variant(1)
        C1 = torch.empty(2048, 2048, dtype=torch.float32, device=dev)
        ops.gemm(A, B, C1, None, False, False)
    finally:
        ops.lib.csm_set_gemm_variant(2)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert torch.equal(C1, outs[0]), "same k-order accumulation in both tile kernels"


def test_gemm_asymmetric_identity(dev):
    """A = I with an asymmetric B catches a transposed C write (both operand orders)."""
    from csm.hip import ops
    n = 128
    eye = torch.eye(n, dtype=BF)
    B = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(BF)
    C = torch.empty(n, n, dtype=torch.float32, device=dev)
    ops.gemm(eye.to(dev), B.to(dev), C, None, False, False)
    assert torch.equal(C.cpu(), B.float().t())
    ops.gemm(eye.to(dev), B.to(dev), C, None, False, True)
    assert torch.equal(C.cpu(), B.float())
    ops.gemm(B.to(dev), eye.to(dev), C, None, True, True)
    assert torch.equal(C.cpu(), B.float().t())


def test_gemm_batched(dev):
    from csm.hip import ops
    g = torch.Generator().manual_seed(5)
    nb, M, K, N = 3, 72, 128, 200
    A = rnd((M, nb * K), g)             # batch b reads columns [b*K, (b+1)*K)
    B = rnd((nb, K, N), g)
    C = torch.empty(nb, M, N, dtype=torch.float32, device=dev)
    Ad, Bd = A.to(dev), B.to(dev)
    ops.gemm(Ad[:, :K], Bd[0], C[0], None, False, True, batch=nb, sA=K, sB=K * N, sC=M * N)
    ref = torch.stack([A[:, b * K:(b + 1) * K].float() @ B[b].float() for b in range(nb)])
    close("gemm batched", C, ref, 2e-5 * math.sqrt(K))


@pytest.mark.parametrize("M,D", [(37, 256), (64, 1024), (130, 2048)])
def test_rmsnorm(dev, M, D):
    from csm.hip import ops
    g = torch.Generator().manual_seed(D + M)
    x, w, dy, dres = rnd((M, D), g, 2.0), (1 + 0.1 * torch.randn(D, generator=g)).to(BF), rnd((M, D), g), rnd((M, D), g)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    y_ref = O.rmsnorm(xr, wr)
    y_ref.backward(dy.float())
    xd, wd = x.to(dev), w.to(dev)
    y = torch.empty(M, D, dtype=BF, device=dev)
    rstd = torch.empty(M, dtype=torch.float32, device=dev)
    ops.rmsnorm_fwd(xd, wd, y, rstd)
    close("rmsnorm fwd", y, y_ref, 1e-2)
    close("rstd", rstd, torch.rsqrt(x.float().pow(2).mean(-1) + 1e-5), 1e-5)
    dx = torch.empty(M, D, dtype=BF, device=dev)
    parts = torch.empty(ops.lib.csm_rmsnorm_bwd_blocks(), D, dtype=torch.float32, device=dev)
    ops.rmsnorm_bwd(xd, wd, rstd, dy.to(dev), dx, dres.to(dev), parts)
    close("rmsnorm dx", dx, xr.grad + dres.float(), 1e-2)
    dw = torch.zeros(D, dtype=BF, device=dev)
    ops.colsum_bf16(parts, dw)
    close("rmsnorm dscale", dw, wr.grad, 1e-2)


@pytest.mark.parametrize("acc", [False, True])
def test_two_linear_dw_grouped(dev, acc):
    """csm_gemm_bf16_two_wgrad: two weight gradients of different output shapes in one launch = the two separate launches of
    the same tile kernel, bit for bit."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(77)
    M = 1024
    dy1, x1 = rnd((M, 768), g, 0.5).to(dev), rnd((M, 512), g, 0.5).to(dev)
    dy2, x2 = rnd((M, 512), g, 0.5).to(dev), rnd((M, 320), g, 0.5).to(dev)
    w1, w2 = rnd((768, 512), g, 0.1).to(dev), rnd((512, 320), g, 0.1).to(dev)
    r1, r2 = w1.clone(), w2.clone()
    ops.lib.csm_set_gemm_variant(3)
    try:
        ops.gemm(dy1, x1, r1, r1 if acc else None, True, True, 0.25)
        ops.gemm(dy2, x2, r2, r2 if acc else None, True, True, 0.25)
    finally:
        ops.lib.csm_set_gemm_variant(2)
    assert ops.two_linear_dw(dy1, x1, w1, dy2, x2, w2, accumulate=acc, alpha=0.25)
    assert torch.equal(w1, r1) and torch.equal(w2, r2)
    if not acc:
        close("grouped dW1", w1, 0.25 * dy1.float().t() @ x1.float(), 1e-2)
        close("grouped dW2", w2, 0.25 * dy2.float().t() @ x2.float(), 1e-2)


@pytest.mark.parametrize("acc", [False, True])
def test_colsum_multi_equals_single(dev, acc):
    """csm_colsum_bf16_multi (11 pairs = two launches of <= 8) against csm_colsum_bf16 per pair: the same bits."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(5)
    rows, D = 256, 2048
    pairs, refs = [], []
    for _ in range(11):
        p_ = torch.randn(rows, D, generator=g).to(dev)
        d0 = rnd((D,), g, 0.3).to(dev)
        r = d0.clone()
        ops.colsum_bf16(p_, r, accumulate=acc)
        pairs.append((p_, d0)); refs.append(r)
    ops.colsum_bf16_multi(pairs, accumulate=acc)
    for (p_, d0), r in zip(pairs, refs):
        assert torch.equal(d0, r)
    if not acc:
        close("colsum", pairs[0][1], pairs[0][0].sum(0), 1e-2)


@pytest.mark.parametrize("acc", [False, True])
def test_multi_linear_dw_equals_separate_products(dev, acc):
    """csm_gemm_bf16_multi_wgrad: six weight gradients (three layers' fused q|k|v and output projections, ragged shapes
    included) in one launch = each product launched alone on the four-wave kernel, bit for bit; and with the four-wave kernel
    switched off the entry point falls back to one ordinary launch per product."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(78)
    M = 1024
    shapes = [(768, 512), (512, 512), (776, 520), (264, 512), (768, 256), (512, 328)]
    probs, refs = [], []
    for n_, k_ in shapes:
        dy, x, w = rnd((M, n_), g, 0.5).to(dev), rnd((M, k_), g, 0.5).to(dev), rnd((n_, k_), g, 0.1).to(dev)
        r = w.clone()
        ops.lib.csm_set_gemm_variant(4)
        try:
            ops.gemm(dy, x, r, r if acc else None, True, True, 0.25)
        finally:
            ops.lib.csm_set_gemm_variant(2)
        probs.append((dy, x, w)); refs.append(r)
    assert ops.multi_linear_dw(probs, accumulate=acc, alpha=0.25)
    assert ops.lib.csm_gemm_last_kernel().decode() == "gemm256w4_multi_tn_kernel"
    for (dy, x, w), r in zip(probs, refs):
        assert torch.equal(w, r)
        if not acc:
            close("multi dW", w, 0.25 * dy.float().t() @ x.float(), 1e-2)
    ops.lib.csm_set_gemm_tuning(1, 0)                    # four-wave kernel off: the same call, one product per launch
    try:
        outs = [(dy, x, torch.zeros_like(w)) for dy, x, w in probs]
        assert ops.multi_linear_dw(outs, accumulate=False, alpha=0.25)
        assert ops.lib.csm_gemm_last_kernel().decode() != "gemm256w4_multi_tn_kernel"
        for (dy, x, w) in outs:
            close("multi dW (fallback)", w, 0.25 * dy.float().t() @ x.float(), 1e-2)
    finally:
        ops.lib.csm_set_gemm_tuning(1, 1)
    assert not ops.multi_linear_dw([(p[0][:1000], p[1][:1000], p[2]) for p in probs])     # M not a multiple of 64: caller falls back


def test_adamw_split_master_is_bit_exact(dev):
    """csm_adamw_step_split (master = bf16 working copy + 16-bit lower half, 26 B/param) against csm_adamw_step (plain fp32
    master, 28 B/param): identical master, m and v bits after three steps with clipping; the working copy is the
    half-up rounding of the master."""
    from csm.hip import ops
    from csm.training.optim import join_master, split_master
    g = torch.Generator().manual_seed(31)
    n = 8 * 4096
    master0 = (torch.randn(n, generator=g) * 0.05).to(dev)
    master0[:64] = master0[:64].to(BF).float()            # some exactly representable values (lo == 0)
    grads = [(torch.randn(n, generator=g) * (0.3 if i else 3.0)).to(BF).to(dev) for i in range(3)]
    coef = torch.tensor([2.0, 0.5], device=dev)
    ma, m1, v1 = master0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pa = torch.empty(n, dtype=BF, device=dev)
    pb = torch.empty(n, dtype=BF, device=dev)
    lo, m2, v2 = split_master(master0, pb), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    assert torch.equal(join_master(pb, lo), master0)
    for step, gr in enumerate(grads, 1):
        ga, gb = gr.clone(), gr.clone()
        ops.adamw_step(ma, m1, v1, pa, ga, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, coef, zero_grad=(step == 2))
        ops.adamw_step_split(lo, m2, v2, pb, gb, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, coef, zero_grad=(step == 2))
        assert torch.equal(join_master(pb, lo).view(torch.int32), ma.view(torch.int32)), step
        assert torch.equal(m1, m2) and torch.equal(v1, v2) and torch.equal(ga, gb)
        bits = ma.view(torch.int32)
        assert torch.equal(pb.view(torch.int16), (((bits + 0x8000) >> 16) & 0xFFFF).to(torch.int16))
        assert float((pb.float() - pa.float()).abs().max()) <= float(pa.float().abs().max()) * 2.0 ** -7    # they differ on ties only
        assert float((pb.view(torch.int16) != pa.view(torch.int16)).float().mean()) < 1e-3


@pytest.mark.parametrize("M,Nout,Kin,swiglu,acc", [(512, 256, 320, False, False), (1024, 2048, 512, False, True), (2048, 256, 512, True, False),
                                                  (4096, 2048, 2048, False, True), (192, 128, 64, True, True)])
def test_linear_dx_dw_pair(dev, M, Nout, Kin, swiglu, acc):
    """csm_gemm_bf16_dgrad_wgrad: dgrad (optionally with the SwiGLU-backward epilogue) and wgrad of a Linear layer in one
    launch with interleaved tiles - against the two separate launches (same tile kernel: bit-identical) and the oracle."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(M + Nout + Kin)
    dy = rnd((M, Nout), g, 0.5).to(dev)
    w = rnd((Nout, Kin), g, 0.1).to(dev)
    x = rnd((M, Kin), g, 0.5).to(dev)
    gu = rnd((M, 2 * Kin), g, 1.0).to(dev) if swiglu else None
    dw0 = rnd((Nout, Kin), g, 0.2).to(dev)
    # separate launches, 256x256 kernel forced so that the arithmetic order is the paired kernel's
    ops.lib.csm_set_gemm_variant(3)
    try:
        dx_ref = torch.empty(M, 2 * Kin if swiglu else Kin, dtype=BF, device=dev)
        if swiglu:
            ops.linear_dx_swiglu_bwd(dy, w, gu, dx_ref)
        else:
            ops.linear_dx(dy, w, dx_ref)
        dw_ref = dw0.clone()
        ops.gemm(dy, x, dw_ref, dw_ref if acc else None, True, True, 0.5)
    finally:
        ops.lib.csm_set_gemm_variant(2)
    dx = torch.empty_like(dx_ref)
    dw = dw0.clone()
    assert ops.linear_dx_dw(dy, w, dx, x, dw, accumulate=acc, alpha=0.5, swiglu_gu=gu)
    assert torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref), "paired launch must reproduce the separate launches bit for bit"
    ref_dw = 0.5 * dy.float().t() @ x.float() + (dw0.float() if acc else 0)
    close("paired dW", dw, ref_dw, 1e-2)
    if not swiglu:
        close("paired dX", dx, dy.float() @ w.float(), 1e-2)


@pytest.mark.parametrize("M,S,H,KV,hd,K", [(100, 50, 4, 2, 64, 256), (96, 32, 2, 1, 128, 192), (4096, 2048, 32, 8, 64, 2048)])
def test_linear_rope_fused_epilogue(dev, M, S, H, KV, hd, K):
    """csm_gemm_bf16_rope: the q|k|v projection with RoPE applied to the fp32 accumulators of the q and k columns, against
    the oracle's projection + torchtune-style rotation (v columns untouched); exercises both GEMM tile kernels and the
    narrow (column overhang) epilogue path through the ragged first case."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(M + hd)
    N = (H + 2 * KV) * hd
    x = rnd((M, K), g, 1.0 if K < 1024 else 0.25)
    w = rnd((N, K), g, 0.1)
    table = O.rope_table(max(S, 64), hd)
    y = x.float() @ w.float().t()
    pos = (torch.arange(M) % S).view(1, M)
    ref = y.clone()
    ref[:, :H * hd] = O.rope(y[:, :H * hd].reshape(1, M, H, hd), table, pos).reshape(M, -1)
    ref[:, H * hd:(H + KV) * hd] = O.rope(y[:, H * hd:(H + KV) * hd].reshape(1, M, KV, hd), table, pos).reshape(M, -1)
    out = torch.empty(M, N, dtype=BF, device=dev)
    ops.linear_rope_fwd(x.to(dev), w.to(dev), out, table.to(dev), S, (H + KV) * hd, hd)
    close("fused qkv + rope", out, ref, 1.5e-2)
    plain = torch.empty(M, N, dtype=BF, device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), plain)
    assert torch.equal(out[:, (H + KV) * hd:], plain[:, (H + KV) * hd:]), "v columns must be the plain projection"


@pytest.mark.parametrize("M,N,K,kx,transB,epi", [
    (100, 72, 64, 32, False, "none"),          # ragged edges, 128x128 kernel, narrow epilogue path
    (300, 256, 192, 64, True, "none"),         # nn (input-gradient form), two extension k-steps, residual
    (4096, 3072, 2048, 32, False, "rope"),     # the q|k|v projection of the backbone: 128x128 kernel, RoPE epilogue on the sum
    (8192, 2048, 3072, 32, True, "none"),      # the backbone's q|k|v input gradient: 256x256 kernel, residual
    (2048, 4096, 512, 32, False, "swiglu"),    # SwiGLU epilogue on the sum (256x256 kernel)
])
def test_gemm_k_extension(dev, M, N, K, kx, transB, epi):
    """csm_gemm_bf16_kext: C = A.B + xA.xB^T (+R) with the extension's k-steps taken inside the same product - a frozen
    projection plus its LoRA adapters (reference LoRALinear.__call__, src/csm/mlx/components/lora.py:85-105) - against the
    fp32 reference of the sum; the epilogues (RoPE, SwiGLU) must see the sum."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(M + N + kx)
    a = rnd((M, K), g, 0.5 if K < 1024 else 0.25)
    b = rnd((K, N) if transB else (N, K), g, 0.1)
    xa = torch.zeros(M, kx, dtype=BF)
    xb = torch.zeros(N, kx, dtype=BF)
    r = kx - 8                                   # the padding columns stay zero, as adapter ranks below the pad width do
    xa[:, :r] = rnd((M, r), g, 0.5)
    xb[:, :r] = rnd((N, r), g, 0.2)
    y = a.float() @ (b.float() if transB else b.float().t()) + xa.float() @ xb.float().t()
    out = torch.empty(M, N, dtype=BF, device=dev)
    if epi == "rope":
        H, KV, hd, S = 32, 8, 64, 2048
        table = O.rope_table(S, hd)
        pos = (torch.arange(M) % S).view(1, M)
        ref = y.clone()
        ref[:, :H * hd] = O.rope(y[:, :H * hd].reshape(1, M, H, hd), table, pos).reshape(M, -1)
        ref[:, H * hd:(H + KV) * hd] = O.rope(y[:, H * hd:(H + KV) * hd].reshape(1, M, KV, hd), table, pos).reshape(M, -1)
        ops.gemm_kext(a.to(dev), b.to(dev), out, xa.to(dev), xb.to(dev), rope=(table.to(dev), S, (H + KV) * hd, hd))
        close("k-extension + rope", out, ref, 1.5e-2)
    elif epi == "swiglu":
        act = torch.empty(M, N // 2, dtype=BF, device=dev)
        ops.gemm_kext(a.to(dev), b.to(dev), out, xa.to(dev), xb.to(dev), swiglu_act=act)
        close("k-extension gate/up", out, y, 1e-2)
        yb = out.float().cpu()
        close("k-extension silu(gate)*up", act, torch.nn.functional.silu(yb[:, 0::2]) * yb[:, 1::2], 1e-2)
    else:
        res = rnd((M, N), g, 1.0) if transB else None
        ops.gemm_kext(a.to(dev), b.to(dev), out, xa.to(dev), xb.to(dev), R=None if res is None else res.to(dev), transB=transB)
        close("k-extension", out, y + (res.float() if res is not None else 0), 1e-2)
    # all-zero extension operands reproduce the plain product bit for bit
    plain = torch.empty(M, N, dtype=BF, device=dev)
    if epi == "none" and not transB:
        ops.gemm(a.to(dev), b.to(dev), plain)
        z = torch.empty(M, N, dtype=BF, device=dev)
        ops.gemm_kext(a.to(dev), b.to(dev), z, torch.zeros(M, kx, dtype=BF, device=dev), torch.zeros(N, kx, dtype=BF, device=dev))
        assert torch.equal(z, plain)


@pytest.mark.parametrize("M,N,K,kx,transB", [(1024, 768, 256, 32, False), (520, 1000, 192, 64, True), (2048, 4096, 512, 96, False)])
def test_gemm_k_extension_four_wave_equals_eight_wave(dev, M, N, K, kx, transB):
    """The K-extension on the four-wave 256x256 kernel (the extension's k-steps as one more asm block on the asm-owned
    accumulators) against the eight-wave kernel's: the same MFMA sequence per accumulator, so the same bits - with and
    without a residual, ragged edges included."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(M + 3 * N + kx)
    a = rnd((M, K), g, 0.5).to(dev)
    b = rnd((K, N) if transB else (N, K), g, 0.1).to(dev)
    xa, xb = rnd((M, kx), g, 0.5).to(dev), rnd((N, kx), g, 0.2).to(dev)
    res = rnd((M, N), g, 1.0).to(dev)
    outs = {}
    for variant in (3, 4):
        ops.lib.csm_set_gemm_variant(variant)
        try:
            for with_r in (False, True):
                o = torch.empty(M, N, dtype=BF, device=dev)
                ops.gemm_kext(a, b, o, xa, xb, R=res if with_r else None, transB=transB)
                want = "gemm256w4_kernel" if variant == 4 else "gemm256p_kernel"
                assert ops.lib.csm_gemm_last_kernel().decode().startswith(want), ops.lib.csm_gemm_last_kernel()
                outs[(variant, with_r)] = o
        finally:
            ops.lib.csm_set_gemm_variant(2)
    for with_r in (False, True):
        assert torch.equal(outs[(3, with_r)], outs[(4, with_r)])
    y = a.float() @ (b.float() if transB else b.float().t()) + xa.float() @ xb.float().t()
    close("k-extension (four-wave)", outs[(4, False)], y, 1e-2)


@pytest.mark.parametrize("M,N,K", [(16384, 32, 2048), (4096, 64, 3072), (100, 32, 384), (33, 64, 128), (50, 32, 192)])
def test_skinny_nt(dev, M, N, K):
    """csm_skinny_nt_bf16: out = alpha x wt^T for a LoRA group's ranks, against fp32 (ragged M, K halves that are not a
    multiple of the unrolled step, both widths)."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(M + N)
    x = rnd((M, K), g, 0.5)
    wt = rnd((N, K), g, 0.1)
    out = torch.empty(M, N, dtype=BF, device=dev)
    ops.skinny_nt(x.to(dev), wt.to(dev), out, alpha=2.0)
    close("skinny nt", out, 2.0 * (x.float() @ wt.float().t()), 1e-2)
    xs = x.to(dev)[:, :K // 2 // 128 * 128] if K >= 256 else None        # a strided view as input (ld > K)
    if xs is not None:
        o2 = torch.empty(M, N, dtype=BF, device=dev)
        ops.skinny_nt(xs, wt.to(dev)[:, :xs.shape[1]], o2)
        close("skinny nt strided", o2, x[:, :xs.shape[1]].float() @ wt[:, :xs.shape[1]].float().t(), 1e-2)


@pytest.mark.parametrize("hd,H,KV", [(64, 4, 2), (128, 2, 1)])
def test_rope(dev, hd, H, KV):
    from csm.hip import ops
    g = torch.Generator().manual_seed(hd)
    B, S = 2, 50
    ld = (H + 2 * KV) * hd
    qkv = rnd((B * S, ld), g)
    table = O.rope_table(128, hd)
    pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
    q = qkv[:, :H * hd].view(B, S, H, hd)
    k = qkv[:, H * hd:(H + KV) * hd].view(B, S, KV, hd)
    ref = qkv.clone().float()
    ref[:, :H * hd] = O.rope(q.float(), table, pos).reshape(B * S, -1)
    ref[:, H * hd:(H + KV) * hd] = O.rope(k.float(), table, pos).reshape(B * S, -1)
    d = qkv.to(dev)
    ops.rope(d, table.to(dev), S, H + KV, hd)
    close("rope fwd", d, ref, 1e-2)
    assert torch.equal(d[:, (H + KV) * hd:].cpu(), qkv[:, (H + KV) * hd:]), "v must be untouched"
    # inverse rotation == autograd backward of the rotation
    qf = q.float().requires_grad_(True)
    up = rnd((B, S, H, hd), g).float()
    O.rope(qf, table, pos).backward(up)
    buf = torch.zeros(B * S, ld, dtype=BF)
    buf[:, :H * hd] = up.reshape(B * S, -1).to(BF)
    bd = buf.to(dev)
    ops.rope(bd, table.to(dev), S, H + KV, hd, inverse=True)
    close("rope bwd", bd[:, :H * hd], qf.grad.reshape(B * S, -1), 1e-2)
    # explicit positions
    p2 = torch.randint(0, 128, (B * S,), generator=g, dtype=torch.int32)
    d2 = qkv.to(dev)
    ops.rope(d2, table.to(dev), S, H + KV, hd, pos=p2.to(dev))
    ref2 = O.rope(q.float(), table, p2.long().view(B, S)).reshape(B * S, -1)
    close("rope pos", d2[:, :H * hd], ref2, 1e-2)


def test_swiglu(dev):
    """Stand-alone kernels and the two fused GEMM epilogues; gate/up are interleaved along the feature axis."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(1)
    M, F, D = 200, 512, 256
    x, w1, w3, w2 = rnd((M, D), g), rnd((F, D), g, 0.1), rnd((F, D), g, 0.1), rnd((D, F), g, 0.1)
    dout = rnd((M, D), g)
    xr = x.float()
    gate, up = xr @ w1.float().t(), xr @ w3.float().t()
    gate_q, up_q = gate.to(BF).float().requires_grad_(True), up.to(BF).float().requires_grad_(True)
    act_ref = torch.nn.functional.silu(gate_q) * up_q
    dact_ref = dout.float() @ w2.float()
    act_ref.backward(dact_ref.to(BF).float())
    w13 = torch.stack([w1, w3], dim=1).reshape(2 * F, D).contiguous()        # rows: g0,u0,g1,u1,...
    gu_ref = torch.stack([gate, up], dim=2).reshape(M, 2 * F)
    dgu_ref = torch.stack([gate_q.grad, up_q.grad], dim=2).reshape(M, 2 * F)
    # fused forward
    gu = torch.empty(M, 2 * F, dtype=BF, device=dev)
    act = torch.empty(M, F, dtype=BF, device=dev)
    ops.linear_swiglu_fwd(x.to(dev), w13.to(dev), gu, act)
    close("fused gu", gu, gu_ref, 1e-2)
    close("fused act", act, act_ref, 1.5e-2)
    # stand-alone forward on the same gu
    act2 = torch.empty(M, F, dtype=BF, device=dev)
    ops.swiglu_fwd(gu, act2)
    close("swiglu fwd", act2, act_ref, 1.5e-2)
    # fused backward: dgu = SwiGLU'(gu) . (dout w2)
    dgu = torch.empty(M, 2 * F, dtype=BF, device=dev)
    ops.linear_dx_swiglu_bwd(dout.to(dev), w2.to(dev), gu_ref.to(BF).to(dev), dgu)
    close("fused dgu", dgu, dgu_ref, 2e-2)
    # stand-alone backward
    dgu2 = torch.empty(M, 2 * F, dtype=BF, device=dev)
    ops.swiglu_bwd(gu_ref.to(BF).to(dev), dact_ref.to(BF).to(dev), dgu2)
    close("swiglu bwd", dgu2, dgu_ref, 1e-2)
    # the 256x256 kernel's epilogues as well
    ops.lib.csm_set_gemm_variant(3)
    try:
        ops.linear_swiglu_fwd(x.to(dev), w13.to(dev), gu, act)
        close("fused act (256 kernel)", act, act_ref, 1.5e-2)
        ops.linear_dx_swiglu_bwd(dout.to(dev), w2.to(dev), gu_ref.to(BF).to(dev), dgu)
        close("fused dgu (256 kernel)", dgu, dgu_ref, 2e-2)
    finally:
        ops.lib.csm_set_gemm_variant(2)


def test_swiglu_fused_epilogues_persistent_rounds(dev):
    """The SwiGLU-forward and -backward GEMM epilogues over several rounds of full tiles per persistent workgroup (waits counted
    past 32 / 64 epilogue memory operations): against the stand-alone kernels' arithmetic and bit-equal to one tile per workgroup."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(5)
    M, F, D = 8192, 4096, 320                                            # fwd: 32 x 32 = 1024 tiles; bwd: 32 x 16 = 512 tiles
    x, w13, w2 = rnd((M, D), g).to(dev), rnd((2 * F, D), g, 0.1).to(dev), rnd((D, F), g, 0.1).to(dev)
    dout = rnd((M, D), g).to(dev)
    res = {}
    try:
        for persistent in (1, 0):
            ops.lib.csm_set_gemm256_persistent(persistent)
            gu = torch.empty(M, 2 * F, dtype=BF, device=dev)
            act = torch.empty(M, F, dtype=BF, device=dev)
            ops.linear_swiglu_fwd(x, w13, gu, act)
            dgu = torch.empty(M, 2 * F, dtype=BF, device=dev)
            ops.linear_dx_swiglu_bwd(dout, w2, gu, dgu)
            res[persistent] = (gu, act, dgu)
    finally:
        ops.lib.csm_set_gemm256_persistent(1)
    gu, act, dgu = res[1]
    close("gu", gu, x.float() @ w13.float().t(), 1e-2)
    act2 = torch.empty(M, F, dtype=BF, device=dev)
    ops.swiglu_fwd(gu, act2)
    close("fused act vs stand-alone", act, act2.float(), 1.5e-2)      # (the fused form activates the fp32 sums, before their bf16 rounding)
    dact = torch.empty(M, F, dtype=BF, device=dev)
    ops.linear_dx(dout, w2, dact)
    dgu2 = torch.empty(M, 2 * F, dtype=BF, device=dev)
    ops.swiglu_bwd(gu, dact, dgu2)
    close("fused dgu vs unfused", dgu, dgu2.float(), 2e-2)
    for a_, b_ in zip(res[1], res[0]):
        assert torch.equal(a_, b_)


def _attn_ref(qkv, B, S, H, KV, hd):
    q = qkv[:, :H * hd].view(B, S, H, hd)
    k = qkv[:, H * hd:(H + KV) * hd].view(B, S, KV, hd)
    v = qkv[:, (H + KV) * hd:].view(B, S, KV, hd)
    return O.attention(q, k, v).reshape(B * S, H * hd)


@pytest.mark.parametrize("B,S,H,KV,hd", [(1, 64, 1, 1, 64), (2, 200, 4, 2, 64), (1, 512, 8, 2, 64), (3, 32, 2, 1, 128),
                                        (2, 100, 4, 2, 128), (1, 31, 2, 2, 64), (2, 1000, 4, 1, 64), (1, 2048, 4, 1, 64)])
def test_attention(dev, B, S, H, KV, hd):
    from csm.hip import ops
    g = torch.Generator().manual_seed(S + hd)
    qkv = rnd((B * S, (H + 2 * KV) * hd), g)
    dout = rnd((B * S, H * hd), g)
    qr = qkv.float().requires_grad_(True)
    ref = _attn_ref(qr, B, S, H, KV, hd)
    ref.backward(dout.float())
    qd = qkv.to(dev)
    out = torch.empty(B * S, H * hd, dtype=BF, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_fwd(qd, out, lse, B, S, H, KV, hd)
    close("attn fwd", out, ref, 1.5e-2)
    # lse check
    q = qkv[:, :H * hd].view(B, S, H, hd).float().transpose(1, 2)
    k = qkv[:, H * hd:(H + KV) * hd].view(B, S, KV, hd).float().repeat_interleave(H // KV, 2).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~torch.tril(torch.ones(S, S, dtype=torch.bool)), float("-inf"))
    close("attn lse", lse, torch.logsumexp(s, -1), 1e-3)
    dqkv = torch.zeros_like(qd)
    delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
    ops.attn_bwd(qd, out, dout.to(dev), lse, dqkv, delta, B, S, H, KV, hd)
    gq = qr.grad
    close("attn dq", dqkv[:, :H * hd], gq[:, :H * hd], 2e-2)
    close("attn dk", dqkv[:, H * hd:(H + KV) * hd], gq[:, H * hd:(H + KV) * hd], 2e-2)
    close("attn dv", dqkv[:, (H + KV) * hd:], gq[:, (H + KV) * hd:], 2e-2)
    # RoPE backward fused into the dQ / dK epilogues == the separate inverse pass (one bf16 rounding fewer), dV untouched
    from csm.models.model import llama3_rope_table
    table = llama3_rope_table(S, hd, 500000.0, 32.0).to(dev).contiguous()
    ops.rope(dqkv, table, S, H + KV, hd, inverse=True)
    fused = torch.zeros_like(qd)
    ops.attn_bwd(qd, out, dout.to(dev), lse, fused, delta, B, S, H, KV, hd, rope_table=table)
    close("attn bwd + rope^T (q, k)", fused[:, :(H + KV) * hd], dqkv[:, :(H + KV) * hd].float(), 1e-2)
    assert torch.equal(fused[:, (H + KV) * hd:], dqkv[:, (H + KV) * hd:])


@pytest.mark.parametrize("B,S,H,KV", [(5, 32, 8, 2), (3, 17, 4, 1), (512, 32, 8, 2)])
def test_attention_short_sequences_grouped_mapping(dev, B, S, H, KV):
    """Round 4: head_dim 128, S <= 32 and four q heads per kv head (the depth decoder's 32-position frames) - forward and dQ
    run with one workgroup per (batch, kv head), wave = q head, so the group's K / V are staged once and no wave is left without
    queries.  Against the fp32 oracle, and bit-equal to the block-per-head mapping (csm_set_attn_variant bit 14) in output, lse,
    dQ / dK / dV, with and without the RoPE^T epilogue: every query tile sees the same key tiles in the same order."""
    from csm.hip import ops
    from csm.models.model import llama3_rope_table
    hd = 128
    g = torch.Generator().manual_seed(11 * S + B)
    qkv = rnd((B * S, (H + 2 * KV) * hd), g)
    dout = rnd((B * S, H * hd), g)
    table = llama3_rope_table(S, hd, 500000.0, 32.0).to(dev).contiguous()
    qd, dd = qkv.to(dev), dout.to(dev)
    res = {}
    try:
        for name, word in (("grouped", 0), ("per_head", 1 << 14)):
            ops.lib.csm_set_attn_variant(word | 2 | (1 << 2) | (3 << 4) | (1 << 6) | (1 << 7) if word else 0)
            out = torch.empty(B * S, H * hd, dtype=BF, device=dev)
            lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
            ops.attn_fwd(qd, out, lse, B, S, H, KV, hd)
            dqkv, fused = torch.zeros_like(qd), torch.zeros_like(qd)
            delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
            ops.attn_bwd(qd, out, dd, lse, dqkv, delta, B, S, H, KV, hd)
            ops.attn_bwd(qd, out, dd, lse, fused, delta, B, S, H, KV, hd, rope_table=table)
            res[name] = (out, lse, dqkv, fused)
    finally:
        ops.lib.csm_set_attn_variant(0)
    for a, b, what in zip(res["grouped"], res["per_head"], ("out", "lse", "dqkv", "dqkv + rope^T")):
        assert torch.equal(a, b), what
    if B <= 8:
        qr = qkv.float().requires_grad_(True)
        ref = _attn_ref(qr, B, S, H, KV, hd)
        ref.backward(dout.float())
        close("attn fwd (grouped)", res["grouped"][0], ref, 1.5e-2)
        close("attn dqkv (grouped)", res["grouped"][2], qr.grad, 2e-2)


def test_attention_spike(dev):
    """One key dominates one query row late in the sequence: forces the online-softmax rescale path."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(3)
    B, S, H, KV, hd = 1, 256, 1, 1, 64
    qkv = rnd((S, 3 * hd), g, 0.5)
    qkv[200, :hd] = 4.0
    qkv[150, hd:2 * hd] = 4.0
    ref = _attn_ref(qkv.float(), B, S, H, KV, hd)
    out = torch.empty(S, hd, dtype=BF, device=dev)
    lse = torch.empty(1, 1, S, dtype=torch.float32, device=dev)
    ops.attn_fwd(qkv.to(dev), out, lse, B, S, H, KV, hd)
    close("attn spike", out, ref, 1.5e-2)


def test_embed(dev):
    from csm.hip import ops
    cfg = O.tiny_cfg()
    params = {k: v.to(BF) for k, v in O.init_params(cfg, seed=2).items()}
    tokens, mask, _ = O.synthetic_batch(cfg, 2, 40, seed=4)
    mask[0, 3, :] = True   # a row with every slot live
    ref = O.embed_masked_sum({k: v.float() for k, v in params.items()}, cfg, tokens, mask)
    M = tokens.shape[0] * tokens.shape[1]
    out = torch.empty(M, cfg.backbone.dim, dtype=BF, device=dev)
    te, ae = params["text_embeddings.weight"].to(dev), params["audio_embeddings.weight"].to(dev)
    tk, mk = tokens.view(M, -1).to(dev), mask.view(M, -1).to(torch.uint8).to(dev)
    ops.embed_fwd(tk, mk, te, ae, out, cfg.audio_vocab)
    close("embed fwd", out, ref.view(M, -1), 1e-2)
    dh = rnd((M, cfg.backbone.dim), torch.Generator().manual_seed(9))
    pt = {k: params[k].float().requires_grad_(True) for k in ("text_embeddings.weight", "audio_embeddings.weight")}
    O.embed_masked_sum(pt, cfg, tokens, mask).view(M, -1).backward(dh.float())
    # sorted / deterministic form, with extra occurrences coming from a second source tensor
    K, V, TV = cfg.n_codebooks, cfg.audio_vocab, cfg.text_vocab
    slot = torch.arange(K + 1)
    tk2 = tokens.view(M, -1)
    rows = torch.where(slot < K, TV + tk2 + slot * V, tk2)
    rows = torch.where(mask.view(M, -1), rows, torch.full_like(rows, TV + K * V))
    src = torch.arange(M).unsqueeze(1).expand(M, K + 1)
    g2 = torch.Generator().manual_seed(10)
    extra_rows = TV + torch.randint(0, K * V, (50,), generator=g2)
    dseq = rnd((50, cfg.backbone.dim), g2)
    rows_all = torch.cat([rows.reshape(-1), extra_rows])
    src_all = torch.cat([src.reshape(-1), M + torch.arange(50)])
    order = torch.argsort(rows_all, stable=True)
    gt = torch.ones_like(te)      # accumulate on top of existing gradient values
    ga = torch.ones_like(ae)
    ops.embed_bwd_sorted(rows_all[order].contiguous().to(dev), src_all[order].contiguous().to(dev), dh.to(dev), dseq.to(dev), gt, ga)
    ref_a = pt["audio_embeddings.weight"].grad.clone()
    ref_a.index_add_(0, extra_rows - TV, dseq.float())
    close("embed sorted d_text", gt, 1.0 + pt["text_embeddings.weight"].grad, 1e-2)
    close("embed sorted d_audio", ga, 1.0 + ref_a, 1e-2)
    gt2, ga2 = torch.ones_like(te), torch.ones_like(ae)
    ops.embed_bwd_sorted(rows_all[order].contiguous().to(dev), src_all[order].contiguous().to(dev), dh.to(dev), dseq.to(dev), gt2, ga2)
    assert torch.equal(gt, gt2) and torch.equal(ga, ga2), "sorted embedding backward must be bitwise repeatable"


def test_ce(dev):
    from csm.hip import ops
    g = torch.Generator().manual_seed(6)
    R, V, ld = 77, 2051, 2112
    logits = torch.zeros(R, ld)
    logits[:, :V] = torch.randn(R, V, generator=g) * 3
    tgt = torch.randint(0, V, (R,), generator=g)
    tgt[5] = -1
    lr = logits[:, :V].clone().requires_grad_(True)
    valid = tgt >= 0
    loss_ref = torch.nn.functional.cross_entropy(lr[valid], tgt[valid], reduction="sum")
    loss_ref.backward()
    ld_ = logits.to(dev)
    rows = torch.empty(R, dtype=torch.float32, device=dev)
    dl = torch.full((R, ld), 7.0, dtype=BF, device=dev)
    ops.ce_fwd_bwd(ld_, tgt.to(dev), rows, dl, V, 0.25)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    ops.reduce_sum(rows, out)
    assert abs(out.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    gref = torch.zeros(R, V)
    gref[valid] = lr.grad[valid] * 0.25
    close("ce dlogits", dl[:, :V], gref, 1e-2)
    assert (dl[:, V:] == 0).all() and (dl[5] == 0).all()


def test_optimizer(dev):
    from csm.hip import ops
    g = torch.Generator().manual_seed(8)
    n = 8 * 1000
    p0 = torch.randn(n, generator=g)
    master = p0.clone().to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    pbf = master.to(BF)
    po, mo, vo = p0.clone(), torch.zeros(n), torch.zeros(n)
    nb = ops.sumsq_blocks()
    for step in range(1, 4):
        grad = rnd((n,), g, 0.3)
        parts = torch.empty(nb, dtype=torch.float32, device=dev)
        ops.sumsq_bf16(grad.to(dev), parts)
        nc = torch.empty(2, dtype=torch.float32, device=dev)
        ops.clip_coef(parts, 1.0, nc)
        gl = [grad.float().clone()]
        norm, coef = O.clip_grad_norm(gl, 1.0)
        assert abs(nc[0].item() - norm.item()) <= 1e-4 * norm.item()
        assert abs(nc[1].item() - coef) <= 1e-4 * coef
        ops.adamw_step(master, m, v, pbf, grad.to(dev), 1e-2, 0.9, 0.999, 1e-8, 0.01, step, nc)
        O.adamw_step(po, gl[0], mo, vo, step, 1e-2)
    close("adamw master", master, po, 1e-5)
    close("adamw m", m, mo, 1e-4)
    close("adamw v", v, vo, 1e-4)
    assert torch.equal(pbf.cpu(), master.cpu().to(BF))


def test_sampler_golden(dev):
    import numpy as np, os
    from csm.hip import ops
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))
    logits, q, want = (torch.from_numpy(z[k]) for k in ("sampler_logits", "sampler_q", "sampler_out"))
    out = torch.empty(logits.shape[0], dtype=torch.int32, device=dev)
    ops.sample_topk(logits.to(dev), q.to(dev), out, 50, 0.9)
    assert torch.equal(out.cpu(), want.view(-1)), (out.cpu(), want.view(-1))
    # live oracle comparison on fresh draws, padded row stride
    g = torch.Generator().manual_seed(31)
    lg = torch.randn(64, 2051, generator=g) * 2
    qq = torch.empty(64, 2051).exponential_(1, generator=g)
    pad = torch.zeros(64, 2112)
    pad[:, :2051] = lg
    out = torch.empty(64, dtype=torch.int32, device=dev)
    ops.sample_topk(pad.to(dev), qq.to(dev), out, 50, 0.9, V=2051)
    assert torch.equal(out.cpu(), O.sample_topk(lg, 50, 0.9, qq).view(-1))
    # round 4: the kept values are compacted and finished by one wave when there are at most 64 of them; more (topk > 64, or ties
    # at the k-th value - torch keeps every value >= the k-th largest) take the block-wide form.  Both against the oracle:
    # topk 64 (the last one-wave case), 65 and 200 (block-wide), and rows whose k-th value is shared by many tokens.
    for topk in (1, 2, 64, 65, 200):
        ops.sample_topk(pad.to(dev), qq.to(dev), out, topk, 0.9, V=2051)
        assert torch.equal(out.cpu(), O.sample_topk(lg, topk, 0.9, qq).view(-1)), topk
    tied = lg.clone()
    tied[:, 100:400] = tied[:, 100:101]                        # 300 equal values per row ...
    tied[:32, 100:400] += 3.0                                  # ... in half of the rows inside the top 50 (kept set of ~330), below it in the others
    pad[:, :2051] = tied
    ops.sample_topk(pad.to(dev), qq.to(dev), out, 50, 0.9, V=2051)
    assert torch.equal(out.cpu(), O.sample_topk(tied, 50, 0.9, qq).view(-1)), "ties at the k-th value"
    quant = (lg * 4).round() / 4                               # heavy ties everywhere (values on a 0.25 grid)
    pad[:, :2051] = quant
    ops.sample_topk(pad.to(dev), qq.to(dev), out, 50, 0.9, V=2051)
    assert torch.equal(out.cpu(), O.sample_topk(quant, 50, 0.9, qq).view(-1)), "quantised logits"


def test_rvq(dev):
    import numpy as np, os
    from csm.hip import ops
    g = torch.Generator().manual_seed(21)   # same draws as tests/golden/make_golden.py
    cbs = torch.randn(8, 2048, 256, generator=g)
    x = torch.randn(40, 256, generator=g) * 4
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))
    codes = torch.empty(8, 40, dtype=torch.int64, device=dev)
    ops.rvq_encode(x.to(dev), cbs.to(dev), codes, 1)
    assert torch.equal(codes.cpu(), torch.from_numpy(z["rvq_codes"])), "RVQ indices must be bit-exact"
    out = torch.empty(40, 256, dtype=torch.float32, device=dev)
    ops.rvq_decode(codes, cbs.to(dev), out)
    assert torch.equal(out.cpu(), O.rvq_decode(codes.cpu(), cbs)), "RVQ decode is an ordered fp32 sum: exact"
    # encode -> decode -> encode is a fixed point for the first (semantic) quantiser
    codes2 = torch.empty(1, 40, dtype=torch.int64, device=dev)
    ops.rvq_encode(cbs[0][codes[0].cpu()].to(dev).contiguous(), cbs[:1].to(dev).contiguous(), codes2, 1)
    assert torch.equal(codes2[0], codes[0])


def test_gemm_epilogue_read_prefetch_changes_no_bit(dev):
    """csm_set_gemm_tuning(0, v): the 256x256 kernel's touches of a fused epilogue's read operand (gate/up of the SwiGLU
    backward, a bf16 residual - extra loads inside the K loop whose results are discarded, waited for with vmcnt counted one
    higher) must leave every output bit as it is, for every operand layout, with several tiles per persistent workgroup."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(321)
    M, d, F = 4096, 2048, 4096
    dy, w2, gu = rnd((M, d), g).to(dev), rnd((d, F), g, 0.05).to(dev), rnd((M, 2 * F), g).to(dev)
    cases = [("nt", rnd((M, 1024), g).to(dev), rnd((4096, 1024), g, 0.05).to(dev), False, False),
             ("nn", rnd((M, 1024), g).to(dev), rnd((1024, 4096), g, 0.05).to(dev), False, True),
             ("tn", rnd((2048, M), g).to(dev), rnd((2048, 4096), g, 0.05).to(dev), True, True)]
    R = rnd((M, 4096), g).to(dev)
    out = {}
    try:
        for v in (0, 1):
            ops.lib.csm_set_gemm_tuning(0, v)
            dgu = torch.empty(M, 2 * F, dtype=BF, device=dev)
            ops.linear_dx_swiglu_bwd(dy, w2, gu, dgu)
            res = [dgu]
            for name, A, B, tA, tB in cases:
                acc = R.clone()
                ops.gemm(A, B, acc, acc, tA, tB, alpha=0.5)          # accumulate into an aliasing bf16 residual
                res.append(acc)
            out[v] = res
    finally:
        ops.lib.csm_set_gemm_tuning(0, 1)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    ref = torch.empty(M, 2 * F, dtype=BF, device=dev)                # and the fused result is still the unfused one's
    dact = torch.empty(M, F, dtype=BF, device=dev)
    ops.gemm(dy, w2, dact, None, False, True)
    ops.swiglu_bwd(gu, dact, ref)
    close("swiglu bwd fused vs unfused", out[1][0], ref, 2e-2)


def test_adamw_skipped_step_on_device(dev):
    """A negative clip coefficient on the device = "this optimiser step does not happen" (how a data-parallel step with
    incomplete gradients is dropped on every rank without a host sync): parameters, master halves and moments keep their
    bits, zero_grad is still honoured; a non-negative coefficient gives the ordinary update."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(5)
    n = 8 * 1000
    for split in (False, True):
        p0 = (torch.randn(n, generator=g) * 0.1).to(BF).to(dev)
        grad0 = (torch.randn(n, generator=g) * 0.01).to(BF).to(dev)
        m0, v0 = torch.rand(n, generator=g).to(dev) * 0.01, torch.rand(n, generator=g).to(dev) * 1e-4
        for coef, moved in ((-1.0, False), (0.5, True)):
            p, grad, m, v = p0.clone(), grad0.clone(), m0.clone(), v0.clone()
            nc = torch.tensor([3.0, coef], dtype=torch.float32, device=dev)
            if split:
                lo = torch.randint(0, 2 ** 15, (n,), generator=g).to(torch.int16).to(dev)
                lo0 = lo.clone()
                ops.adamw_step_split(lo, m, v, p, grad, 1e-3, 0.9, 0.999, 1e-8, 0.01, 3, nc, zero_grad=True)
                assert torch.equal(lo, lo0) != moved
            else:
                master = p.float()
                ops.adamw_step(master, m, v, p, grad, 1e-3, 0.9, 0.999, 1e-8, 0.01, 3, nc, zero_grad=True)
                assert torch.equal(master, p0.float()) != moved
            assert torch.equal(p, p0) != moved and torch.equal(m, m0) != moved and torch.equal(v, v0) != moved
            assert float(grad.float().abs().max()) == 0.0, "zero_grad is honoured either way"


@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm_four_wave_kernel_equals_eight_wave(dev, mode):
    """The four-wave 256x256 kernel (hand-scheduled inline-asm K loop, AGPR accumulators, 128x128 wave tiles: variant 4)
    against the eight-wave one (variant 3): the same products in the same order, so the same bits - one to 16 K-tiles,
    ragged edges, several tiles per persistent workgroup, bf16 and fp32 outputs, a residual that aliases the output - and
    against the oracle's arithmetic."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(808)
    for (M, N, K) in [(256, 256, 64), (256, 256, 128), (256, 512, 192), (304, 520, 448), (1000, 2112, 1024), (4096, 8192, 320), (8192, 2048, 4096)]:
        if mode == "nt":
            A, B, tA, tB = rnd((M, K), g), rnd((N, K), g), False, False
        elif mode == "nn":
            A, B, tA, tB = rnd((M, K), g), rnd((K, N), g), False, True
        else:
            A, B, tA, tB = rnd((K, M), g), rnd((K, N), g), True, True
        Ad, Bd = A.to(dev), B.to(dev)
        R = rnd((M, N), g).to(dev)
        out = {}
        try:
            for v in (3, 4):
                ops.lib.csm_set_gemm_variant(v)
                C = torch.empty(M, N, dtype=BF, device=dev)
                ops.gemm(Ad, Bd, C, None, tA, tB)
                acc = R.clone()
                ops.gemm(Ad, Bd, acc, acc, tA, tB, alpha=0.5)
                C32 = torch.empty(M, N, dtype=torch.float32, device=dev)
                ops.gemm(Ad, Bd, C32, R, tA, tB, alpha=0.25)
                out[v] = (C, acc, C32)
        finally:
            ops.lib.csm_set_gemm_variant(2)
        for a, b in zip(out[3], out[4]):
            assert torch.equal(a, b), f"{mode} {M}x{N}x{K}"
        ref = (Ad.float().t() if tA else Ad.float()) @ (Bd.float() if tB else Bd.float().t())
        close(f"w4 {mode} {M}x{N}x{K}", out[4][0], ref, 1e-2)
        close(f"w4 {mode} f32+R", out[4][2], 0.25 * ref + R.float(), 2e-5 * math.sqrt(K))


def test_gemm_four_wave_kernel_fused_epilogues(dev):
    """The fused epilogues through the four-wave kernel - SwiGLU forward (gate/up + activation), SwiGLU backward (reads
    gate/up, writes d gate/up), RoPE on the q|k columns of the fused projection - bit-equal to the eight-wave kernel's, over
    several tiles per persistent workgroup and with one tile per workgroup."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(909)
    M, F, D = 4096, 4096, 512
    x, w13, w2 = rnd((M, D), g).to(dev), rnd((2 * F, D), g, 0.1).to(dev), rnd((D, F), g, 0.1).to(dev)
    dout = rnd((M, D), g).to(dev)
    S, H, KV, hd = 512, 8, 2, 64
    wq = rnd(((H + 2 * KV) * hd, D), g, 0.1).to(dev)
    table = O.rope_table(S, hd).to(dev).contiguous()
    res = {}
    try:
        for key, (variant, persistent) in {"w8": (3, 1), "w4": (4, 1), "w4_one": (4, 0)}.items():
            ops.lib.csm_set_gemm_variant(variant)
            ops.lib.csm_set_gemm256_persistent(persistent)
            gu, act = torch.empty(M, 2 * F, dtype=BF, device=dev), torch.empty(M, F, dtype=BF, device=dev)
            ops.linear_swiglu_fwd(x, w13, gu, act)
            dgu = torch.empty(M, 2 * F, dtype=BF, device=dev)
            ops.linear_dx_swiglu_bwd(dout, w2, gu, dgu)
            qkv = torch.empty(M, (H + 2 * KV) * hd, dtype=BF, device=dev)
            ops.linear_rope_fwd(x, wq, qkv, table, S, (H + KV) * hd, hd)
            res[key] = (gu, act, dgu, qkv)
    finally:
        ops.lib.csm_set_gemm_variant(2)
        ops.lib.csm_set_gemm256_persistent(1)
    for key in ("w4", "w4_one"):
        for a, b in zip(res["w8"], res[key]):
            assert torch.equal(a, b), key


def test_gemm_four_wave_256x192_tiles_equal_256x256(dev):
    """Round 4: where 256 x 192 output tiles fill the rounds of the 256 CUs better than 256 x 256 ones (the fused q|k|v projection of
    the bench batch: 384 tiles = 1.5 rounds become 512 tiles of 3/4 the work = 2 rounds of 3/4), the four-wave kernel takes them
    (gemm256w4n6_kernel; csm_set_gemm_tuning(8, 0) switches it off).  Each output element is the same k-ordered sum through the
    same MFMA, so the results are bit-equal: plain, with a residual that aliases the output, with the RoPE epilogue whose
    rotated/unrotated boundary falls inside a 192-column tile, with A transposed, one round and several rounds."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(1906)
    hd = 64
    cases = [(256, 768, 64, False, None), (256, 768, 192, True, None), (512, 1536, 1024, False, (128, 1280)),
             (8192, 1536, 2048, False, None), (8192, 3072, 2048, False, (2048, 2560)), (16384, 1536, 1024, False, (2048, 1280)),
             (2048, 3072, 320, True, None)]
    for (M, N, K, tA, rope) in cases:
        A = rnd((K, M) if tA else (M, K), g).to(dev)
        B = rnd((N, K), g, 0.1).to(dev)
        R = rnd((M, N), g).to(dev)
        out = {}
        try:
            for on in (1, 0):
                ops.lib.csm_set_gemm_tuning(8, on)
                ops.lib.csm_set_gemm_variant(4)
                C = torch.empty(M, N, dtype=BF, device=dev)
                ops.gemm(A, B, C, None, tA, False)
                name = ops.lib.csm_gemm_last_kernel().decode()
                assert name.startswith("gemm256w4n6_kernel" if on else "gemm256w4_kernel"), (name, M, N, K)
                acc = R.clone()
                ops.gemm(A, B, acc, acc, tA, False, alpha=0.5)
                res = [C, acc]
                if rope is not None and not tA:
                    S, p0 = rope
                    table = O.rope_table(S, hd).to(dev).contiguous()
                    qkv = torch.empty(M, N, dtype=BF, device=dev)
                    ops.linear_rope_fwd(A, B, qkv, table, S, p0, hd)
                    assert ops.lib.csm_gemm_last_kernel().decode().startswith("gemm256w4n6_kernel" if on else "gemm256w4_kernel")
                    res.append(qkv)
                out[on] = res
        finally:
            ops.lib.csm_set_gemm_variant(2)
            ops.lib.csm_set_gemm_tuning(8, 1)
        for i, (a, b) in enumerate(zip(out[0], out[1])):
            assert torch.equal(a, b), f"{M}x{N}x{K} tA={tA} output {i}: {(a.float() - b.float()).abs().max().item()}"
        ref = (A.float().t() if tA else A.float()) @ B.float().t()
        close(f"w4n6 {M}x{N}x{K}", out[1][0], ref, 1e-2)
    # shapes the 192-column tiling does not help (or cannot take) keep the 256 x 256 kernel
    for (M, N, K) in [(4096, 2048, 512), (16384, 3072, 128), (300, 768, 64)]:
        A, B = rnd((M, K), g).to(dev), rnd((N, K), g).to(dev)
        ops.lib.csm_set_gemm_variant(4)
        try:
            ops.gemm(A, B, torch.empty(M, N, dtype=BF, device=dev), None, False, False)
            assert ops.lib.csm_gemm_last_kernel().decode().startswith("gemm256w4_kernel"), (M, N, K)
        finally:
            ops.lib.csm_set_gemm_variant(2)


@pytest.mark.parametrize("B,S,H,KV", [(1, 64, 4, 1), (1, 128, 4, 1), (2, 192, 8, 2), (1, 512, 8, 2), (2, 320, 4, 1), (3, 384, 4, 1), (1, 2048, 8, 2)])
def test_attention_backward_generated_asm_kernels(dev, B, S, H, KV):
    """attention64_asm.hip (round 4): the dK/dV pass as one wave per (64 keys, query head) and the dQ pass as one wave per (64
    queries, query head; it takes 128-query blocks, other lengths run the second-generation dQ kernel), each with a generated,
    hand-allocated asm loop.  Against the fp32 oracle (same tolerance as test_attention), against the second-generation kernel it replaces
    (csm_set_attn_variant bit 10 switches it off: the two differ only in the order the four heads / two parities are summed),
    with and without the RoPE^T epilogue, both work orders (bit 11), and bit-identical from run to run."""
    from csm.hip import ops
    from csm.models.model import llama3_rope_table
    hd = 64
    g = torch.Generator().manual_seed(7 * S + H)
    qkv = rnd((B * S, (H + 2 * KV) * hd), g)
    dout = rnd((B * S, H * hd), g)
    qr = qkv.float().requires_grad_(True)
    _attn_ref(qr, B, S, H, KV, hd).backward(dout.float())
    gq = qr.grad
    qd, dd = qkv.to(dev), dout.to(dev)
    out = torch.empty(B * S, H * hd, dtype=BF, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_fwd(qd, out, lse, B, S, H, KV, hd)
    delta = torch.empty(2, B, H, S, dtype=torch.float32, device=dev)
    table = llama3_rope_table(S, hd, 500000.0, 32.0).to(dev).contiguous()
    kv = slice(H * hd, None)
    DEFAULTS = 2 | (1 << 2) | (3 << 4) | (1 << 6) | (1 << 7)
    res = {}
    try:
        DQ = 1 << 12                              # the asm dQ kernel is off by default (slower inside the step): switched on here
        for name, v in (("gen2", DEFAULTS | (1 << 10)), ("asm", DEFAULTS | DQ), ("asm_pairs", DEFAULTS | DQ | (1 << 11) | (1 << 13)),
                        ("asm_again", DEFAULTS | DQ), ("default", 0)):
            ops.lib.csm_set_attn_variant(v)
            for rope in (None, table):
                dqkv = torch.full_like(qd, float("nan"))
                ops.attn_bwd(qd, out, dd, lse, dqkv, delta, B, S, H, KV, hd, rope_table=rope)
                res[name, rope is not None] = dqkv.clone()
                if name == "asm_again":   # (the asm dQ kernel takes 128-query blocks: S % 128 == 0; the dK/dV one 64-key blocks)
                    assert ops.lib.csm_attn_last_dkv_kernel() == (3 if S % 128 == 0 else 1), "the asm kernels must have taken this shape"
                if name == "default":
                    assert ops.lib.csm_attn_last_dkv_kernel() == 1, "default: asm dK/dV, compiler-scheduled dQ"
    finally:
        ops.lib.csm_set_attn_variant(0)
    close("asm dq", res["asm", False][:, :H * hd], gq[:, :H * hd], 2e-2)
    close("asm dk", res["asm", False][:, H * hd:(H + KV) * hd], gq[:, H * hd:(H + KV) * hd], 2e-2)
    close("asm dv", res["asm", False][:, (H + KV) * hd:], gq[:, (H + KV) * hd:], 2e-2)
    for rope in (False, True):
        assert not torch.isnan(res["asm", rope].float()).any()
        close("asm vs second generation", res["asm", rope], res["gen2", rope].float(), 2e-3)
        assert torch.equal(res["asm", rope], res["asm_again", rope]), "run-to-run bit-identical"
        assert torch.equal(res["asm", rope], res["asm_pairs", rope]), "the work order must not change a bit"
        assert torch.equal(res["default", rope][:, kv], res["asm", rope][:, kv]) and torch.equal(res["default", rope][:, :H * hd], res["gen2", rope][:, :H * hd])
