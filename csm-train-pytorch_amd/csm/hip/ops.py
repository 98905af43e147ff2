"""Torch-tensor front end of the C ABI: shape/dtype checks on the host, raw pointers + the current stream below.

PyTorch is plumbing here (device memory, streams); every arithmetic op is a hand-written gfx950 kernel.
"""
from typing import Optional

import torch

from . import check, lib

BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _mat(t: torch.Tensor, dtype=BF16):
    """2-D row-major view with unit inner stride -> (ptr, rows, cols, ld)."""
    assert t.is_cuda and t.dtype == dtype and t.dim() == 2 and t.stride(1) == 1, (t.dtype, t.shape, t.stride())
    return t.data_ptr(), t.shape[0], t.shape[1], t.stride(0)


def gemm(A, B, C, R=None, transA=False, transB=False, alpha=1.0, batch=1, sA=0, sB=0, sC=0, sR=0):
    """C[M,N] = alpha * opA(A) . opB(B)^T (+ R).  A: [M,K] (or [K,M] if transA); B: [N,K] (or [K,N] if transB)."""
    pa, a0, a1, lda = _mat(A)
    pb, b0, b1, ldb = _mat(B)
    M, K = (a1, a0) if transA else (a0, a1)
    N, Kb = (b1, b0) if transB else (b0, b1)
    assert K == Kb, (A.shape, B.shape, transA, transB)
    out_f32 = C.dtype == torch.float32
    pc, c0, c1, ldc = _mat(C, C.dtype)
    assert (c0, c1) == (M, N), (C.shape, M, N)
    pr, ldr = None, 0
    if R is not None:
        pr, r0, r1, ldr = _mat(R)
        assert (r0, r1) == (M, N)
    check(lib.csm_gemm_bf16(pa, pb, pc, pr, M, N, K, lda, ldb, ldc, ldr, int(transA), int(transB), int(out_f32),
                            float(alpha), batch, sA, sB, sC, sR, _stream()), "csm_gemm_bf16")
    return C


def linear_swiglu_fwd(x, w13, gu, act, residual=None):
    """gu[M,2F] = x w13^T (+ residual) with gate/up interleaved, act[M,F] = silu(gate)*up, one GEMM launch.
    ``residual`` [M,2F] (may be ``gu`` itself) is how LoRA adapters on w1 / w3 join: their product is written first."""
    pa, M, K, lda = _mat(x)
    pb, N, Kb, ldb = _mat(w13)
    assert K == Kb and gu.shape == (M, N) and act.shape == (M, N // 2) and gu.is_contiguous() and act.is_contiguous()
    pr, ldr = (None, 0) if residual is None else (residual.data_ptr(), residual.stride(0))
    assert residual is None or (residual.shape == (M, N) and residual.stride(1) == 1 and residual.dtype == BF16)
    check(lib.csm_gemm_bf16_ex(pa, pb, gu.data_ptr(), pr, M, N, K, lda, ldb, N, ldr, 0, 0, 0, 1.0, 1, 0, 0, 0, 0, 1, None,
                               act.data_ptr(), N // 2, _stream()), "csm_gemm_bf16_ex(swiglu fwd)")


def linear_dx_swiglu_bwd(dy, w2, gu, dgu):
    """dgu[M,2F] = SwiGLU'(gu) applied to (dy[M,d] w2[d,F]); the intermediate d(act) is never stored."""
    pa, M, K, lda = _mat(dy)
    pb, Kb, F, ldb = _mat(w2)
    assert K == Kb and gu.shape == (M, 2 * F) and dgu.shape == (M, 2 * F) and gu.is_contiguous() and dgu.is_contiguous()
    check(lib.csm_gemm_bf16_ex(pa, pb, dgu.data_ptr(), None, M, F, K, lda, ldb, 2 * F, 0, 0, 1, 0, 1.0, 1, 0, 0, 0, 0, 2,
                               gu.data_ptr(), None, 2 * F, _stream()), "csm_gemm_bf16_ex(swiglu bwd)")


def linear_rope_fwd(x, w, out, table, S, n_rope_cols, head_dim):
    """out[M,N] = x[M,K] w[N,K]^T with RoPE (positions = row % S) applied to columns [0, n_rope_cols) in the GEMM epilogue."""
    pa, M, K, lda = _mat(x)
    pb, N, Kb, ldb = _mat(w)
    assert K == Kb and out.shape == (M, N) and out.dtype == BF16 and out.stride(1) == 1
    assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[0] >= S and table.shape[1] * 2 == head_dim
    check(lib.csm_gemm_bf16_rope(pa, pb, out.data_ptr(), M, N, K, lda, ldb, out.stride(0), table.data_ptr(), S, n_rope_cols, head_dim,
                                 _stream()), "csm_gemm_bf16_rope")
    return out


def gemm_kext(A, B, C, xA, xB, R=None, transB=False, rope=None, swiglu_act=None, swiglu_bwd_gu=None):
    """C[M,N] = A . opB(B)^T + xA[M,kx] . xB[N,kx]^T (+ R): a frozen projection with its LoRA adapters as extra k-steps of the
    same product.  ``rope`` = (table, S, n_rope_cols, head_dim), ``swiglu_act`` = act[M,N/2] or ``swiglu_bwd_gu`` = gate/up [M,2N]
    (then C = d(gate/up) [M,2N]) select the fused epilogues, which see the sum."""
    pa, M, K, lda = _mat(A)
    pb, b0, b1, ldb = _mat(B)
    N, Kb = (b1, b0) if transB else (b0, b1)
    assert K == Kb, (A.shape, B.shape, transB)
    kx = xA.shape[1]
    assert xA.shape == (M, kx) and xB.shape == (N, kx) and xA.is_contiguous() and xB.is_contiguous() and xA.dtype == BF16 and xB.dtype == BF16
    assert C.shape == (M, 2 * N if swiglu_bwd_gu is not None else N) and C.dtype == BF16 and C.stride(1) == 1
    pr, ldr = (None, 0) if R is None else (R.data_ptr(), R.stride(0))
    assert R is None or (R.shape == (M, N) and R.stride(1) == 1 and R.dtype == BF16)
    epi, aux_in, aux_out, ld_aux, rc, hd = 0, None, None, 0, 0, 0
    if rope is not None:
        table, S, rc, hd = rope
        assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[0] >= S and table.shape[1] * 2 == hd
        epi, aux_in, ld_aux = 3, table.data_ptr(), S
    elif swiglu_act is not None:
        assert swiglu_act.shape == (M, N // 2) and swiglu_act.is_contiguous() and C.is_contiguous()
        epi, aux_out, ld_aux = 1, swiglu_act.data_ptr(), N // 2
    elif swiglu_bwd_gu is not None:      # the product is d(act) [M,N]; C = d(gate/up) [M,2N] from gate/up [M,2N]
        assert swiglu_bwd_gu.shape == (M, 2 * N) and swiglu_bwd_gu.is_contiguous() and C.is_contiguous() and R is None
        epi, aux_in, ld_aux = 2, swiglu_bwd_gu.data_ptr(), 2 * N
    check(lib.csm_gemm_bf16_kext(pa, pb, C.data_ptr(), pr, M, N, K, lda, ldb, C.stride(0), ldr, 0, int(transB), xA.data_ptr(),
                                 xB.data_ptr(), kx, epi, aux_in, aux_out, ld_aux, rc, hd, _stream()), "csm_gemm_bf16_kext")
    return C


def skinny_nt(x, wt, out, alpha=1.0):
    """out[M,N] = alpha * x[M,K] wt[N,K]^T for N in (32, 64): the bandwidth-shaped product of a LoRA group (falls back to the
    GEMM for other widths)."""
    pa, M, K, lda = _mat(x)
    pb, N, Kb, ldb = _mat(wt)
    assert K == Kb and out.shape == (M, N) and out.dtype == BF16 and out.stride(1) == 1
    if N not in (32, 64) or K % 128:
        return gemm(x, wt, out, None, False, False, alpha)
    check(lib.csm_skinny_nt_bf16(pa, pb, out.data_ptr(), M, N, K, lda, ldb, out.stride(0), float(alpha), _stream()), "csm_skinny_nt_bf16")
    return out


def linear_fwd(x, w, out, residual=None, alpha=1.0):
    """out[M,N] = x[M,K] w[N,K]^T (+ residual)."""
    return gemm(x, w, out, residual, False, False, alpha)


def linear_dx(dy, w, out, residual=None, alpha=1.0):
    """out[M,K] = dy[M,N] w[N,K] (+ residual)."""
    return gemm(dy, w, out, residual, False, True, alpha)


def linear_dx_dw(dy, w, dx, x, dw, accumulate=False, alpha=1.0, swiglu_gu=None):
    """dx[M,Kin] = dy[M,Nout] w[Nout,Kin] and dw[Nout,Kin] (+)= alpha * dy^T x[M,Kin] in ONE launch (tiles of the two
    products interleaved).  With ``swiglu_gu`` [M, 2 Kin] the first product carries the SwiGLU-backward epilogue and dx is
    d(gate/up) [M, 2 Kin] (w = w2).  Returns False when the shapes do not suit the paired kernel (caller falls back)."""
    M, Nout = dy.shape
    Kin = w.shape[1]
    if M % 64 or Nout % 64 or Kin % 8 or not (dy.stride(1) == w.stride(1) == x.stride(1) == dw.stride(1) == dx.stride(1) == 1):
        return False
    assert w.shape == (Nout, Kin) and x.shape == (M, Kin) and dw.shape == (Nout, Kin), (dy.shape, w.shape, x.shape, dw.shape)
    assert dx.shape == (M, 2 * Kin if swiglu_gu is not None else Kin)
    epi, aux, ld_aux = (0, None, 0) if swiglu_gu is None else (2, swiglu_gu.data_ptr(), swiglu_gu.stride(0))
    check(lib.csm_gemm_bf16_dgrad_wgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), x.data_ptr(), dw.data_ptr(), M, Nout, Kin,
                                        dy.stride(0), w.stride(0), dx.stride(0), x.stride(0), dw.stride(0), epi, aux, ld_aux,
                                        int(accumulate), float(alpha), _stream()), "csm_gemm_bf16_dgrad_wgrad")
    return True


def two_linear_dw(dy1, x1, dw1, dy2, x2, dw2, accumulate=False, alpha=1.0):
    """dw1[N1,K1] (+)= alpha dy1^T x1 and dw2[N2,K2] (+)= alpha dy2^T x2 (same row count M) in ONE launch of 256x256 tiles.
    Returns False when the shapes do not suit it (caller falls back to two linear_dw calls)."""
    M = dy1.shape[0]
    ts = (dy1, x1, dw1, dy2, x2, dw2)
    if M % 64 or dy2.shape[0] != M or any(t.stride(1) != 1 for t in ts) or any(d % 8 for t in ts for d in t.shape[1:]):
        return False
    assert x1.shape[0] == M and x2.shape[0] == M and dw1.shape == (dy1.shape[1], x1.shape[1]) and dw2.shape == (dy2.shape[1], x2.shape[1])
    check(lib.csm_gemm_bf16_two_wgrad(dy1.data_ptr(), x1.data_ptr(), dw1.data_ptr(), dy1.shape[1], x1.shape[1], dy1.stride(0), x1.stride(0),
                                      dw1.stride(0), dy2.data_ptr(), x2.data_ptr(), dw2.data_ptr(), dy2.shape[1], x2.shape[1], dy2.stride(0),
                                      x2.stride(0), dw2.stride(0), M, int(accumulate), float(alpha), _stream()), "csm_gemm_bf16_two_wgrad")
    return True


def multi_linear_dw(problems, accumulate=False, alpha=1.0):
    """dw_i[N_i,K_i] (+)= alpha dy_i^T x_i for a list of (dy_i, x_i, dw_i) sharing the row count M, in ONE launch of 256x256 tiles
    (csm_gemm_bf16_multi_wgrad).  Returns False when the shapes do not suit it (caller falls back to its per-layer launches)."""
    import ctypes as C
    n = len(problems)
    if n < 1 or n > 12:
        return False
    M = problems[0][0].shape[0]
    for dy, x, dw in problems:
        ts = (dy, x, dw)
        if M % 64 or dy.shape[0] != M or x.shape[0] != M or any(t.stride(1) != 1 for t in ts) or any(d % 8 for t in ts for d in t.shape[1:]):
            return False
        assert dw.shape == (dy.shape[1], x.shape[1])
    vp, ip = C.c_void_p * n, C.c_int * n
    check(lib.csm_gemm_bf16_multi_wgrad(n, vp(*[p[0].data_ptr() for p in problems]), vp(*[p[1].data_ptr() for p in problems]),
                                        vp(*[p[2].data_ptr() for p in problems]), ip(*[p[0].shape[1] for p in problems]),
                                        ip(*[p[1].shape[1] for p in problems]), ip(*[p[0].stride(0) for p in problems]),
                                        ip(*[p[1].stride(0) for p in problems]), ip(*[p[2].stride(0) for p in problems]),
                                        M, int(accumulate), float(alpha), _stream()), "csm_gemm_bf16_multi_wgrad")
    return True


_splitk_ws = {}


def linear_dw(dy, x, out, accumulate=False, alpha=1.0):
    """out[N,K] (+)= dy[M,N]^T x[M,K].

    Weight gradients with a small output and a long contraction (attention output projection, every decoder matrix)
    would put a handful of 256x256 tiles on 256 CUs; those are split along the contraction over the GEMM's batch
    dimension into fp32 partial slabs (so that tiles x splits fills the chip) and summed into the bf16 gradient by the
    column-sum kernel."""
    Mrows, N = dy.shape
    Kout = x.shape[1]
    tiles = ((N + 255) // 256) * ((Kout + 255) // 256)
    tiles128 = ((N + 127) // 128) * ((Kout + 127) // 128)
    # measured (tools/probes/wgrad_probe.py): once the 128x128 kernel has >= ~320 tiles for its 512 workgroup slots the
    # direct GEMM beats slabs + column sum (fused-qkv dW: 126 vs 160 us); below that the split wins (o_proj dW: 108 vs 132)
    # (round 3, four-wave kernel: half a round of 256x256 tiles - the depth decoder's w2 gradient, 128 tiles - is better split in
    # two than given to the 128x128 kernel: 261 vs 295 us)
    if (tiles <= 96 and tiles128 < 320 or 96 < tiles <= 128) and Mrows >= 4096 and out.is_contiguous() and Mrows % 64 == 0:
        splits = 1
        while tiles * splits * 2 <= 256 and (Mrows // (splits * 2)) % 64 == 0 and Mrows // (splits * 2) >= 512:
            splits *= 2
        if splits > 1:
            chunk = Mrows // splits
            key = (splits, N, Kout, dy.device)
            ws = _splitk_ws.get(key)
            if ws is None:
                ws = _splitk_ws[key] = torch.empty(splits, N * Kout, dtype=torch.float32, device=dy.device)
            gemm(dy[:chunk], x[:chunk], ws[0].view(N, Kout), None, True, True, alpha, batch=splits,
                 sA=chunk * dy.stride(0), sB=chunk * x.stride(0), sC=N * Kout)
            colsum_bf16(ws, out.view(-1), accumulate=accumulate)
            return out
    return gemm(dy, x, out, out if accumulate else None, True, True, alpha)


def rmsnorm_fwd(x, scale, y, rstd, eps=1e-5):
    M, D = x.shape
    check(lib.csm_rmsnorm_fwd(x.data_ptr(), scale.data_ptr(), y.data_ptr(), _ptr(rstd), M, D, eps, _stream()), "csm_rmsnorm_fwd")
    return y


def rmsnorm_bwd(x, scale, rstd, dy, dx, dres=None, dscale_partials=None):
    M, D = x.shape
    check(lib.csm_rmsnorm_bwd(x.data_ptr(), scale.data_ptr(), rstd.data_ptr(), dy.data_ptr(), _ptr(dres), dx.data_ptr(),
                              _ptr(dscale_partials), M, D, _stream()), "csm_rmsnorm_bwd")
    return dx


def colsum_bf16(partials, dst, accumulate=False):
    rows, D = partials.shape
    check(lib.csm_colsum_bf16(partials.data_ptr(), rows, D, dst.data_ptr(), int(accumulate), _stream()), "csm_colsum_bf16")


def colsum_bf16_multi(pairs, accumulate=False):
    """dst_i (+)= column sums of partials_i for a list of (partials_i [rows, D] fp32, dst_i [D] bf16) of one shape: one launch per
    8 pairs, the arithmetic of colsum_bf16 per pair."""
    import ctypes as C
    rows, D = pairs[0][0].shape
    assert all(p.shape == (rows, D) and p.is_contiguous() and d.numel() == D for p, d in pairs)
    for i in range(0, len(pairs), 8):
        chunk = pairs[i:i + 8]
        n = len(chunk)
        vp = C.c_void_p * n
        check(lib.csm_colsum_bf16_multi(n, vp(*[p.data_ptr() for p, _ in chunk]), vp(*[d.data_ptr() for _, d in chunk]), rows, D,
                                        int(accumulate), _stream()), "csm_colsum_bf16_multi")


def dropout_bf16(x, out, p, seed, accumulate=False):
    """out (+)= dropout(x, p) on [M, D] bf16 row-strided matrices; the mask depends only on (seed, row*D + col)."""
    M, D = x.shape
    assert out.shape == x.shape and x.stride(1) == 1 and out.stride(1) == 1
    check(lib.csm_dropout_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), M, D, float(p), int(seed) & (2 ** 64 - 1),
                               int(accumulate), _stream()), "csm_dropout_bf16")
    return out


def bias_add_bf16(y, bias):
    M, D = y.shape
    assert y.stride(1) == 1 and bias.numel() == D and bias.is_contiguous()
    check(lib.csm_bias_add_bf16(y.data_ptr(), y.stride(0), bias.data_ptr(), M, D, _stream()), "csm_bias_add_bf16")


def bias_grad_bf16(dy, gbias, accumulate=True, slices=64):
    """gbias (+)= sum over rows of dy [M, D] (bf16, row-strided), fp32 accumulation in two stages."""
    M, D = dy.shape
    assert dy.stride(1) == 1
    slices = max(1, min(slices, (M + 3) // 4))
    part = torch.empty(slices, D, dtype=torch.float32, device=dy.device)
    check(lib.csm_colsum_rows_bf16(dy.data_ptr(), dy.stride(0), M, D, part.data_ptr(), slices, _stream()), "csm_colsum_rows_bf16")
    colsum_bf16(part, gbias, accumulate=accumulate)


def rope(qkv, table, S, n_heads_qk, head_dim, pos=None, inverse=False):
    M, ld = qkv.shape
    assert table.dtype == torch.float32 and table.is_contiguous()
    check(lib.csm_rope(qkv.data_ptr(), table.data_ptr(), _ptr(pos), M, S, n_heads_qk, head_dim, qkv.stride(0), int(inverse),
                       _stream()), "csm_rope")
    return qkv


def attn_fwd(qkv, out, lse, B, S, H, KV, HD):
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.shape == (B * S, (H + 2 * KV) * HD)
    check(lib.csm_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, KV, HD, _stream()), "csm_attn_fwd")
    return out


def attn_bwd(qkv, out, dout, lse, dqkv, delta_ws, B, S, H, KV, HD, rope_table=None):
    """dqkv = gradient of the (rotated) q | k | v rows; with ``rope_table`` the RoPE backward is applied in the dQ / dK
    epilogues and dqkv is the gradient of the un-rotated projection output (positions = row index in the sequence)."""
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dqkv.is_contiguous()
    need = lib.csm_attn_bwd_workspace_bytes(B, S, H)
    assert delta_ws.dtype == torch.float32 and delta_ws.is_contiguous() and delta_ws.numel() * 4 >= need, f"delta_ws: {need} bytes of scratch"
    if rope_table is None:
        check(lib.csm_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                               delta_ws.data_ptr(), B, S, H, KV, HD, _stream()), "csm_attn_bwd")
    else:
        assert rope_table.dtype == torch.float32 and rope_table.is_contiguous() and rope_table.shape[0] >= S
        check(lib.csm_attn_bwd_rope(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                    delta_ws.data_ptr(), rope_table.data_ptr(), B, S, H, KV, HD, _stream()), "csm_attn_bwd_rope")
    return dqkv


def swiglu_fwd(gu, out):
    M, F2 = gu.shape
    check(lib.csm_swiglu_fwd(gu.data_ptr(), out.data_ptr(), M, F2 // 2, _stream()), "csm_swiglu_fwd")
    return out


def swiglu_bwd(gu, dout, dgu):
    M, F2 = gu.shape
    check(lib.csm_swiglu_bwd(gu.data_ptr(), dout.data_ptr(), dgu.data_ptr(), M, F2 // 2, _stream()), "csm_swiglu_bwd")
    return dgu


def embed_fwd(tokens, mask_u8, text_emb, audio_emb, out, audio_vocab):
    M, K1 = tokens.shape
    assert tokens.dtype == torch.int64 and mask_u8.dtype == torch.uint8 and tokens.is_contiguous() and mask_u8.is_contiguous()
    check(lib.csm_embed_fwd(tokens.data_ptr(), mask_u8.data_ptr(), text_emb.data_ptr(), audio_emb.data_ptr(), out.data_ptr(),
                            M, K1 - 1, out.shape[1], audio_vocab, _stream()), "csm_embed_fwd")
    return out


def embed_bwd_sorted(sorted_rows, src_index, dh, dseq, g_text, g_audio):
    assert sorted_rows.dtype == torch.int64 and src_index.dtype == torch.int64 and sorted_rows.is_contiguous() and src_index.is_contiguous()
    check(lib.csm_embed_bwd_sorted(sorted_rows.data_ptr(), src_index.data_ptr(), sorted_rows.numel(), dh.data_ptr(), _ptr(dseq),
                                   dh.shape[0], g_text.data_ptr(), g_audio.data_ptr(), g_text.shape[0],
                                   g_text.shape[0] + g_audio.shape[0], dh.shape[1], _stream()), "csm_embed_bwd_sorted")


def rows_add_bf16(dst, rows_i32, src, src_stride_rows):
    check(lib.csm_rows_add_bf16(dst.data_ptr(), rows_i32.data_ptr(), src.data_ptr(), rows_i32.numel(), int(src_stride_rows),
                                dst.shape[1], _stream()), "csm_rows_add_bf16")


def rows_take_bf16(table, rows_i32, out):
    """out[n] = table[rows[n]]; table[rows[n]] = 0 (unique rows; negative = padding -> zero row)."""
    assert rows_i32.dtype == torch.int32 and rows_i32.is_contiguous() and out.is_contiguous() and out.shape == (rows_i32.numel(), table.shape[1])
    check(lib.csm_rows_take_bf16(table.data_ptr(), rows_i32.data_ptr(), out.data_ptr(), rows_i32.numel(), table.shape[1], _stream()),
          "csm_rows_take_bf16")
    return out


def decoder_input_fwd(hidden, rows_i32, codes, audio_emb, out, audio_vocab):
    N, K = codes.shape
    assert rows_i32.dtype == torch.int32 and codes.dtype == torch.int64 and codes.is_contiguous()
    check(lib.csm_decoder_input_fwd(hidden.data_ptr(), rows_i32.data_ptr(), codes.data_ptr(), audio_emb.data_ptr(),
                                    out.data_ptr(), N, K, hidden.shape[1], audio_vocab, _stream()), "csm_decoder_input_fwd")
    return out


def ce_fwd_bwd(logits_f32, targets, loss_rows, dlogits, V, grad_scale):
    R, ldl = logits_f32.shape[0], logits_f32.stride(0)
    assert logits_f32.dtype == torch.float32 and targets.dtype == torch.int64 and targets.is_contiguous()
    ldd = dlogits.stride(0) if dlogits is not None else 0
    check(lib.csm_ce_fwd_bwd(logits_f32.data_ptr(), targets.data_ptr(), loss_rows.data_ptr(), _ptr(dlogits), R, V, ldl, ldd,
                             float(grad_scale), _stream()), "csm_ce_fwd_bwd")


def reduce_sum(x_f32, out_f32, scale=1.0):
    check(lib.csm_reduce_sum_f32(x_f32.data_ptr(), x_f32.numel(), float(scale), out_f32.data_ptr(), _stream()), "csm_reduce_sum_f32")


def sumsq_blocks() -> int:
    return lib.csm_sumsq_blocks()


def sumsq_bf16(g, partials):
    check(lib.csm_sumsq_bf16(g.data_ptr(), g.numel(), partials.data_ptr(), _stream()), "csm_sumsq_bf16")


def clip_coef(partials, max_norm, norm_and_coef):
    check(lib.csm_clip_coef(partials.data_ptr(), partials.numel(), float(max_norm), norm_and_coef.data_ptr(), _stream()), "csm_clip_coef")


def adamw_step(master, m, v, param, grad, lr, beta1, beta2, eps, wd, step, norm_and_coef=None, grad_mul=1.0, zero_grad=False):
    n = master.numel()
    assert master.dtype == torch.float32 and param.dtype == BF16 and grad.dtype == BF16 and param.numel() == n == grad.numel()
    check(lib.csm_adamw_step(master.data_ptr(), m.data_ptr(), v.data_ptr(), param.data_ptr(), grad.data_ptr(), n, lr, beta1,
                             beta2, eps, wd, int(step), _ptr(norm_and_coef), float(grad_mul), int(zero_grad), _stream()), "csm_adamw_step")


def adamw_step_split(master_lo, m, v, param, grad, lr, beta1, beta2, eps, wd, step, norm_and_coef=None, grad_mul=1.0, zero_grad=False):
    n = master_lo.numel()
    assert master_lo.dtype == torch.int16 and param.dtype == BF16 and grad.dtype == BF16 and param.numel() == n == grad.numel()
    check(lib.csm_adamw_step_split(master_lo.data_ptr(), m.data_ptr(), v.data_ptr(), param.data_ptr(), grad.data_ptr(), n, lr, beta1,
                                   beta2, eps, wd, int(step), _ptr(norm_and_coef), float(grad_mul), int(zero_grad), _stream()),
          "csm_adamw_step_split")


def gemv(x, W, y, residual=None):
    """y[B,N] = x[B,K] W[N,K]^T (+ residual), B <= 4."""
    B, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K and y.shape == (B, N) and x.stride(1) == 1 and W.stride(1) == 1 and y.stride(1) == 1
    check(lib.csm_gemv_bf16(x.data_ptr(), W.data_ptr(), y.data_ptr(), _ptr(residual), B, N, K, W.stride(0), x.stride(0),
                            y.stride(0), int(y.dtype == torch.float32), _stream()), "csm_gemv_bf16")
    return y


def gemv_ex(x, W, y, residual=None, norm_scale=None, eps=1e-5, swiglu=False, row_index=None, row_offset=0):
    """gemv with an RMSNorm prologue on x (norm_scale), the SwiGLU pairing of interleaved gate/up rows (y is [B, N/2]) and /
    or x gathered from a table: batch row b = x[row_index[b] + row_offset] (row_index int32 [B] on the device)."""
    B = y.shape[0]
    K = x.shape[1]
    N = W.shape[0]
    assert W.shape[1] == K and y.shape == (B, N // 2 if swiglu else N) and x.stride(1) == 1 and W.stride(1) == 1 and y.stride(1) == 1
    assert row_index is not None or x.shape[0] == B
    assert row_index is None or (row_index.dtype == torch.int32 and row_index.numel() == B and row_index.is_contiguous())
    check(lib.csm_gemv_bf16_ex(x.data_ptr(), W.data_ptr(), y.data_ptr(), _ptr(residual), B, N, K, W.stride(0), x.stride(0),
                               y.stride(0), int(y.dtype == torch.float32), _ptr(norm_scale), float(eps), int(swiglu),
                               _ptr(row_index), int(row_offset), _stream()), "csm_gemv_bf16_ex")
    return y


def attn_decode_rope(qkv, kcache, vcache, out, pos_i32, table, H, KV, HD, pos_host=None):
    """rope(q, new k) + append(new k, v) + one-position attention against the caches, one launch.  ``pos_host``: the position all
    rows share, as a host integer (the depth decoder's step; HD = 128, H = 4 KV, S_max <= 32): the launch that issues every load
    of its prologue at once."""
    B, _, S_max, _ = kcache.shape
    assert table.dtype == torch.float32 and table.is_contiguous()
    if pos_host is not None and HD == 128 and H == 4 * KV and S_max <= 32 and out.is_contiguous():
        check(lib.csm_attn_decode_rope_at(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), int(pos_host),
                                          table.data_ptr(), B, H, KV, HD, S_max, qkv.stride(0), _stream()), "csm_attn_decode_rope_at")
        return out
    check(lib.csm_attn_decode_rope(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), pos_i32.data_ptr(),
                                   table.data_ptr(), B, H, KV, HD, S_max, qkv.stride(0), _stream()), "csm_attn_decode_rope")
    return out


def gemv_attn(qkv, kcache, vcache, pos_i32, table, W, y, residual, H, KV, HD, pos_host=None):
    """A depth-decoder layer's rope + append + attention + output projection (+ residual) in one launch (S_max <= 64,
    HD = 128); bit-identical to ``attn_decode_rope`` followed by ``gemv``.  ``pos_host``: the position as a host integer (B = 1,
    S_max <= 32): the launch that issues every load of its prologue at once."""
    B, _, S_max, _ = kcache.shape
    N = W.shape[0]
    assert table.dtype == torch.float32 and table.is_contiguous() and W.shape[1] == H * HD and W.stride(1) == 1 and y.shape == (B, N)
    if (pos_host is not None and B == 1 and S_max <= 32 and HD == 128 and H * HD == 1024 and H <= 8 and KV <= 2
            and H % 2 == 0 and (H // KV) % 2 == 0):
        assert qkv.is_contiguous() and y.is_contiguous() and (residual is None or residual.is_contiguous())
        check(lib.csm_gemv_attn_at_bf16(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), int(pos_host), table.data_ptr(), W.data_ptr(),
                                        y.data_ptr(), _ptr(residual), N, H, KV, HD, S_max, W.stride(0), _stream()), "csm_gemv_attn_at_bf16")
        return y
    check(lib.csm_gemv_attn_bf16(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), pos_i32.data_ptr(), table.data_ptr(),
                                 W.data_ptr(), y.data_ptr(), _ptr(residual), B, N, H, KV, HD, S_max, qkv.stride(0), W.stride(0),
                                 y.stride(0), _stream()), "csm_gemv_attn_bf16")
    return y


def gemv_t(x, W, y):
    """y[B,N] = x[B,K] W[K,N], B <= 4."""
    B, K = x.shape
    N = W.shape[1]
    assert W.shape[0] == K and y.shape == (B, N) and W.stride(1) == 1
    check(lib.csm_gemv_t_bf16(x.data_ptr(), W.data_ptr(), y.data_ptr(), B, N, K, W.stride(0), x.stride(0), y.stride(0),
                              int(y.dtype == torch.float32), _stream()), "csm_gemv_t_bf16")
    return y


def kv_append(qkv, kcache, vcache, pos_i32, H, KV, HD):
    B, _, S_max, _ = kcache.shape
    check(lib.csm_kv_append(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), pos_i32.data_ptr(), B, H, KV, HD, S_max,
                            qkv.stride(0), _stream()), "csm_kv_append")


def attn_decode(qkv, kcache, vcache, out, pos_i32, H, KV, HD):
    B, _, S_max, _ = kcache.shape
    check(lib.csm_attn_decode(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), pos_i32.data_ptr(), B, H, KV,
                              HD, S_max, qkv.stride(0), _stream()), "csm_attn_decode")
    return out


def sample_topk(logits_f32, q_f32, out_i32, topk, temperature, V=None):
    rows = logits_f32.shape[0]
    V = V or logits_f32.shape[1]
    assert q_f32.is_contiguous() and q_f32.shape == (rows, V)
    check(lib.csm_sample_topk(logits_f32.data_ptr(), q_f32.data_ptr(), out_i32.data_ptr(), rows, V, logits_f32.stride(0),
                              int(topk), float(temperature), _stream()), "csm_sample_topk")
    return out_i32


def rvq_encode(x_f32, codebooks_f32, codes_i64, n_semantic=1):
    T, D = x_f32.shape
    K, Cn, D2 = codebooks_f32.shape
    assert D == D2 and codes_i64.shape == (K, T) and x_f32.is_contiguous() and codebooks_f32.is_contiguous()
    check(lib.csm_rvq_encode(x_f32.data_ptr(), codebooks_f32.data_ptr(), codes_i64.data_ptr(), T, K, Cn, D, n_semantic,
                             _stream()), "csm_rvq_encode")
    return codes_i64


def rvq_decode(codes_i64, codebooks_f32, out_f32):
    K, T = codes_i64.shape
    _, Cn, D = codebooks_f32.shape
    check(lib.csm_rvq_decode(codes_i64.data_ptr(), codebooks_f32.data_ptr(), out_f32.data_ptr(), T, K, Cn, D, _stream()),
          "csm_rvq_decode")
    return out_f32
