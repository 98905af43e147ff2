"""CPU oracle for the CSM training / generation hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-torch (fp32/fp64, CPU) restatement of the arithmetic on the
reference's PyTorch path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the shipped package under
``csm-train-pytorch_amd/`` never does and fails loudly without its HIP library.

What each function follows (paths relative to the reference checkout):

* ``embed_masked_sum``      - ``src/csm/models/model.py:202-217`` (``_embed_audio``/``_embed_tokens``)
                              + mask-mul-sum ``src/csm/training/utils.py:85-87``.
* ``rmsnorm``/``rope``/``attention``/``transformer`` - torchtune 0.4.0 ``llama3_2`` as configured at
                              ``src/csm/models/model.py:11-42`` (third-party, pinned in ``pyproject.toml:18``,
                              NOT vendored; restated from its published behaviour).
* ``semantic_loss``         - ``src/csm/training/utils.py:96-106``.
* ``acoustic_loss``         - teacher-forced restatement of ``Model.generate_frame``
                              ``src/csm/models/model.py:171-193`` (the reference leaves a placeholder
                              at ``src/csm/training/utils.py:109-117``).
* ``lora_delta``            - ``src/csm/mlx/components/lora.py:71-105,140-153``.
* ``clip_grad_norm``/``adamw_step`` - ``torch.nn.utils.clip_grad_norm_`` / ``torch.optim.AdamW`` as driven by
                              ``src/csm/training/trainer.py:166-173,271-277``.
* ``sample_topk``           - ``src/csm/models/model.py:79-96`` with the Exp(1) noise injected.
* ``rvq_encode``/``rvq_decode`` - moshi 0.2.2 Mimi split residual VQ (third-party, ``pyproject.toml:17``,
                              call sites ``src/csm/generator.py:117,209``): 1 semantic + (K-1) acoustic
                              residual chain, nearest codeword by squared L2, first index wins ties.

Pinning status: the embedding, loss-slicing/CE/weighting and sampler parts are pinned against the
reference's own functions executed in the build container (``tests/golden/make_golden.py``); the
transformer block is cross-checked against the installed HF ``CsmForConditionalGeneration``.
The reference holds no golden vectors of its own.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- configs
@dataclass
class StackCfg:
    dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    ffn: int
    max_seq_len: int = 2048
    norm_eps: float = 1e-5
    rope_base: float = 500_000.0
    rope_scale: float = 32.0

    @property
    def head_dim(self) -> int:
        return self.dim // self.n_heads


@dataclass
class CsmCfg:
    backbone: StackCfg
    decoder: StackCfg
    text_vocab: int = 128256
    audio_vocab: int = 2051
    n_codebooks: int = 32


def csm_1b_cfg() -> CsmCfg:
    """``src/csm/models/model.py:11-42`` + ``src/csm/generator.py:232-238``."""
    return CsmCfg(
        backbone=StackCfg(2048, 16, 32, 8, 8192),
        decoder=StackCfg(1024, 4, 8, 2, 8192),
    )


def tiny_cfg() -> CsmCfg:
    """Small shape used by golden fixtures (head dims kept at the real 64 / 128)."""
    return CsmCfg(
        backbone=StackCfg(256, 2, 4, 2, 512, max_seq_len=128),
        decoder=StackCfg(256, 2, 2, 1, 512, max_seq_len=128),
        text_vocab=300, audio_vocab=67, n_codebooks=4,
    )


# --------------------------------------------------------------------------- parameters
def stack_param_shapes(prefix: str, c: StackCfg) -> Dict[str, Tuple[int, ...]]:
    hd = c.head_dim
    s: Dict[str, Tuple[int, ...]] = {}
    for i in range(c.n_layers):
        p = f"{prefix}.layers.{i}"
        s[f"{p}.sa_norm.scale"] = (c.dim,)
        s[f"{p}.attn.q_proj.weight"] = (c.n_heads * hd, c.dim)
        s[f"{p}.attn.k_proj.weight"] = (c.n_kv_heads * hd, c.dim)
        s[f"{p}.attn.v_proj.weight"] = (c.n_kv_heads * hd, c.dim)
        s[f"{p}.attn.output_proj.weight"] = (c.dim, c.dim)
        s[f"{p}.mlp_norm.scale"] = (c.dim,)
        s[f"{p}.mlp.w1.weight"] = (c.ffn, c.dim)
        s[f"{p}.mlp.w3.weight"] = (c.ffn, c.dim)
        s[f"{p}.mlp.w2.weight"] = (c.dim, c.ffn)
    s[f"{prefix}.norm.scale"] = (c.dim,)
    return s


def param_shapes(cfg: CsmCfg) -> Dict[str, Tuple[int, ...]]:
    """State-dict names and shapes of ``Model`` (``src/csm/models/model.py:113-126``)."""
    s = {}
    s.update(stack_param_shapes("backbone", cfg.backbone))
    s.update(stack_param_shapes("decoder", cfg.decoder))
    s["text_embeddings.weight"] = (cfg.text_vocab, cfg.backbone.dim)
    s["audio_embeddings.weight"] = (cfg.audio_vocab * cfg.n_codebooks, cfg.backbone.dim)
    s["projection.weight"] = (cfg.decoder.dim, cfg.backbone.dim)
    s["codebook0_head.weight"] = (cfg.audio_vocab, cfg.backbone.dim)
    s["audio_head"] = (cfg.n_codebooks - 1, cfg.decoder.dim, cfg.audio_vocab)
    return s


def init_params(cfg: CsmCfg, seed: int = 0, std: float = 0.02, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded random init: N(0, std) matrices, norm scales = 1."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".scale"):
            out[name] = torch.ones(shape, dtype=dtype)
        else:
            out[name] = (torch.randn(shape, generator=g, dtype=torch.float32) * std).to(dtype)
    return out


# --------------------------------------------------------------------------- blocks
def llama3_inv_freq(head_dim: int, base: float = 500_000.0, scale: float = 32.0,
                    low_freq_factor: float = 1.0, high_freq_factor: float = 4.0,
                    old_context_len: int = 8192) -> torch.Tensor:
    """torchtune 0.4.0 ``Llama3ScaledRoPE.rope_init`` + ``apply_scaling`` (appendix A), in the arithmetic torchtune
    uses: ``freqs = 1 / base ** (arange(0, dim, 2)[:dim // 2].float() / dim)`` is an fp32 tensor, and the scaling loop
    iterates over its 0-d fp32 elements, so the wavelength, the smoothing factor and the scaled frequency are all
    rounded to fp32 at every operation (a float64 table differs from it in the last bit of some frequencies, which
    is ~1e-4 rad at position 2047)."""
    freqs = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    low_wl = old_context_len / low_freq_factor
    high_wl = old_context_len / high_freq_factor
    out = []
    for f in freqs:                                    # 0-d fp32 tensors: python-scalar operands do not widen them
        wl = 2 * math.pi / f
        if wl < high_wl:
            out.append(f)
        elif wl > low_wl:
            out.append(f / scale)
        else:
            smooth = (old_context_len / wl - low_freq_factor) / (high_freq_factor - low_freq_factor)
            out.append((1 - smooth) * f / scale + smooth * f)
    return torch.tensor([float(x) for x in out], dtype=freqs.dtype)


def rope_table(max_seq_len: int, head_dim: int, base: float = 500_000.0, scale: float = 32.0) -> torch.Tensor:
    """[max_seq_len, head_dim/2, 2] = (cos, sin)(pos * theta'), fp32: torchtune's ``build_rope_cache``
    (``seq_idx = arange(max_seq_len, dtype=theta.dtype)``, ``einsum('i, j -> ij', seq_idx, theta).float()``)."""
    theta = llama3_inv_freq(head_dim, base, scale)
    pos = torch.arange(max_seq_len, dtype=theta.dtype)
    ang = torch.einsum("i,j->ij", pos, theta).float()
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1)


def rope(x: torch.Tensor, table: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """Interleaved-pair rotation.  x [B,S,H,hd]; table [P,hd/2,2]; pos [B,S] int."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    t = table[pos].unsqueeze(2)  # [B,S,1,hd/2,2]
    out = torch.stack(
        [xs[..., 0] * t[..., 0] - xs[..., 1] * t[..., 1],
         xs[..., 1] * t[..., 0] + xs[..., 0] * t[..., 1]], dim=-1)
    return out.flatten(-2).type_as(x)


def rmsnorm(x: torch.Tensor, scale: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    x32 = x.float()
    y = x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)
    return y.type_as(x) * scale


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = True) -> torch.Tensor:
    """q [B,S,H,hd], k/v [B,S,KV,hd]; q-head j uses kv-head j // (H/KV)."""
    B, S, H, hd = q.shape
    KV = k.shape[2]
    rep = H // KV
    k = k.repeat_interleave(rep, dim=2)
    v = v.repeat_interleave(rep, dim=2)
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if causal:
        m = torch.tril(torch.ones(S, S, dtype=torch.bool))
        s = s.masked_fill(~m, float("-inf"))
    p = torch.softmax(s.float(), dim=-1).type_as(q)
    return (p @ v).transpose(1, 2)


def lora_delta(x: torch.Tensor, lora: Optional[Dict[str, torch.Tensor]], name: str, scaling: float) -> torch.Tensor:
    """(alpha/r) * (drop(x) A^T) B^T (+ lora_bias), or 0 when the module has no adapter (lora.py:71-105).
    Dropout is expressed through an optional ``{name}.lora_dropout_scale`` entry: the inverted-dropout multiplier
    (0 or 1/(1-p)) per element of x, supplied by the test so that both sides use the same mask."""
    if lora is None or f"{name}.lora_A" not in lora:
        return 0.0
    A, Bm = lora[f"{name}.lora_A"], lora[f"{name}.lora_B"]
    ds = lora.get(f"{name}.lora_dropout_scale")
    xl = x if ds is None else x * ds.reshape(x.shape).to(x.dtype)
    out = scaling * ((xl @ A.t()) @ Bm.t())
    if f"{name}.lora_bias" in lora:
        out = out + lora[f"{name}.lora_bias"]
    return out


def transformer(params: Dict[str, torch.Tensor], prefix: str, c: StackCfg, h: torch.Tensor,
                pos: torch.Tensor, lora: Optional[Dict[str, torch.Tensor]] = None,
                lora_scaling: float = 2.0, table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Pre-norm Llama block stack + final RMSNorm; output upcast to fp32 like torchtune."""
    B, S, _ = h.shape
    hd = c.head_dim
    if table is None:
        table = rope_table(c.max_seq_len, hd, c.rope_base, c.rope_scale)
    for i in range(c.n_layers):
        p = f"{prefix}.layers.{i}"
        xn = rmsnorm(h, params[f"{p}.sa_norm.scale"], c.norm_eps)

        def lin(name, inp):
            return inp @ params[f"{p}.{name}.weight"].t() + lora_delta(inp, lora, f"{p}.{name}", lora_scaling)

        q = lin("attn.q_proj", xn).view(B, S, c.n_heads, hd)
        k = lin("attn.k_proj", xn).view(B, S, c.n_kv_heads, hd)
        v = lin("attn.v_proj", xn).view(B, S, c.n_kv_heads, hd)
        q, k = rope(q, table, pos), rope(k, table, pos)
        o = attention(q, k, v).reshape(B, S, c.dim)
        h = h + lin("attn.output_proj", o)
        hn = rmsnorm(h, params[f"{p}.mlp_norm.scale"], c.norm_eps)
        h = h + lin("mlp.w2", F.silu(lin("mlp.w1", hn)) * lin("mlp.w3", hn))
    return rmsnorm(h, params[f"{prefix}.norm.scale"], c.norm_eps).float()


# --------------------------------------------------------------------------- model-level pieces
def embed_tokens(params, cfg: CsmCfg, tokens: torch.Tensor) -> torch.Tensor:
    """[B,S,K+1] ids -> [B,S,K+1,D]: audio slot c uses row tok + c*V_a, text from the last column."""
    K, V = cfg.n_codebooks, cfg.audio_vocab
    text = params["text_embeddings.weight"][tokens[:, :, -1]].unsqueeze(-2)
    aidx = tokens[:, :, :-1] + V * torch.arange(K)
    audio = params["audio_embeddings.weight"][aidx]
    return torch.cat([audio, text], dim=-2)


def embed_masked_sum(params, cfg: CsmCfg, tokens: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    e = embed_tokens(params, cfg, tokens)
    return (e * mask.unsqueeze(-1)).sum(dim=2)


def backbone_hidden(params, cfg, tokens, mask, lora=None, lora_scaling=2.0):
    B, S, _ = tokens.shape
    pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
    h = embed_masked_sum(params, cfg, tokens, mask)
    return transformer(params, "backbone", cfg.backbone, h, pos, lora, lora_scaling)


def semantic_loss(params, hidden: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """CE of codebook-0 logits at positions [0, S-1) vs targets[:, :S-1, 0]; mean over all rows."""
    logits = hidden[:, :-1] @ params["codebook0_head.weight"].t().float()
    tgt = targets[:, : logits.size(1), 0]
    return F.cross_entropy(logits.reshape(-1, logits.size(-1)), tgt.reshape(-1))


def acoustic_loss(params, cfg: CsmCfg, hidden: torch.Tensor, targets: torch.Tensor,
                  rows: Optional[torch.Tensor] = None, lora=None, lora_scaling=2.0,
                  return_logits: bool = False):
    """Teacher-forced depth-decoder CE.

    For flattened row r=(b,p), p < S-1, with backbone state h and frame codes c_0..c_{K-1} =
    targets[b,p]: decoder sequence [h, E_a(0,c_0), ..., E_a(K-2,c_{K-2})] -> projection -> decoder ->
    logits_i = dec[i] @ audio_head[i-1] for i=1..K-1, CE vs c_i; mean over rows and codebooks.
    ``rows`` selects a subset of the B*(S-1) rows (compute amortisation); default all.
    """
    B, S, D = hidden.shape
    K, V = cfg.n_codebooks, cfg.audio_vocab
    h = hidden[:, :-1].reshape(-1, D)
    codes = targets[:, : S - 1].reshape(-1, K)
    if rows is not None:
        h, codes = h[rows], codes[rows]
    N = h.shape[0]
    aidx = codes[:, : K - 1] + V * torch.arange(K - 1)
    seq = torch.cat([h.unsqueeze(1).to(params["audio_embeddings.weight"].dtype),
                     params["audio_embeddings.weight"][aidx]], dim=1)  # [N,K,D]
    x = seq @ params["projection.weight"].t()
    pos = torch.arange(K).unsqueeze(0).repeat(N, 1)
    dec = transformer(params, "decoder", cfg.decoder, x, pos, lora, lora_scaling)  # [N,K,d'] fp32
    logits = torch.einsum("nkd,kdv->nkv", dec[:, 1:], params["audio_head"].float())
    loss = F.cross_entropy(logits.reshape(-1, V), codes[:, 1:].reshape(-1))
    return (loss, logits) if return_logits else loss


def compute_loss(params, cfg: CsmCfg, tokens, mask, targets, semantic_weight=100.0, acoustic_weight=1.0,
                 acoustic_rows="off", lora=None, lora_scaling=2.0):
    """Restatement of ``compute_loss`` (``src/csm/training/utils.py:56-119``).

    ``acoustic_rows="off"`` reproduces the reference exactly (acoustic term is the literal 0
    placeholder).  ``None`` trains the decoder on every row, a LongTensor on that subset.
    """
    hidden = backbone_hidden(params, cfg, tokens, mask, lora, lora_scaling)
    sem = semantic_loss(params, hidden, targets)
    if isinstance(acoustic_rows, str) and acoustic_rows == "off":
        ac = torch.tensor(0.0)
    else:
        ac = acoustic_loss(params, cfg, hidden, targets, acoustic_rows, lora, lora_scaling)
    total = semantic_weight * sem + acoustic_weight * ac
    return total, {"semantic_loss": sem, "acoustic_loss": ac}


# --------------------------------------------------------------------------- optimiser pieces
def clip_grad_norm(grads: List[torch.Tensor], max_norm: float, eps: float = 1e-6) -> Tuple[torch.Tensor, float]:
    """torch ``clip_grad_norm_``: coef = min(1, max_norm / (||g||_2 + 1e-6)); returns (norm, coef)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = min(1.0, float(max_norm / (total + eps)))
    for g in grads:
        g.mul_(coef)
    return total, coef


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
               beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01) -> None:
    """One torch.optim.AdamW update (decoupled decay, bias-corrected, eps outside the sqrt ratio)."""
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------- sampler
def sample_topk(logits: torch.Tensor, topk: int, temperature: float, q: torch.Tensor) -> torch.Tensor:
    """``sample_topk`` with the Exp(1) draw ``q`` supplied by the caller -> int32 [.., 1]."""
    logits = logits / temperature
    kth = torch.topk(logits, topk)[0][..., -1, None]
    scores = logits.masked_fill(logits < kth, -float("inf"))
    probs = F.softmax(F.log_softmax(scores, dim=-1), dim=-1)
    return torch.argmax(probs / q, dim=-1, keepdim=True).to(torch.int32)


# --------------------------------------------------------------------------- RVQ (Mimi split residual VQ)
def rvq_encode(x: torch.Tensor, codebooks: torch.Tensor, n_semantic: int = 1) -> torch.Tensor:
    """x [T,D] latent frames (already input-projected), codebooks [K,C,D] -> codes [K,T] int64.

    Semantic quantiser(s) see x; the acoustic chain starts again from x (split RVQ) and each layer
    quantises the running residual (fp32).  Distance = squared L2; argmin keeps the first minimum.
    """
    K = codebooks.shape[0]
    codes = torch.empty(K, x.shape[0], dtype=torch.int64)

    def chain(lo, hi):
        r = x.float().clone()
        for kk in range(lo, hi):
            cb = codebooks[kk].float()
            # distances in float64 so that the winner does not depend on a summation order
            d = ((r.double()[:, None, :] - cb.double()[None, :, :]) ** 2).sum(-1)
            idx = torch.argmin(d, dim=-1)
            codes[kk] = idx
            r = r - cb[idx]

    chain(0, n_semantic)
    chain(n_semantic, K)
    return codes


def rvq_decode(codes: torch.Tensor, codebooks: torch.Tensor) -> torch.Tensor:
    """codes [K,T] -> sum_k codebooks[k, codes[k]] : [T,D] fp32 (sum in ascending k)."""
    out = torch.zeros(codes.shape[1], codebooks.shape[2], dtype=torch.float32)
    for kk in range(codes.shape[0]):
        out += codebooks[kk].float()[codes[kk]]
    return out


# --------------------------------------------------------------------------- synthetic data
def synthetic_batch(cfg: CsmCfg, batch: int, seq: int, seed: int, n_segments: int = 2):
    """Interleaved [text | audio] segments; returns (tokens [B,S,K+1] i64, mask bool, targets [B,S,K] i64)."""
    g = torch.Generator().manual_seed(seed)
    K = cfg.n_codebooks
    tokens = torch.zeros(batch, seq, K + 1, dtype=torch.int64)
    mask = torch.zeros(batch, seq, K + 1, dtype=torch.bool)
    hi_txt = max(2, min(48, seq // (2 * n_segments)))
    lo_txt = max(1, min(16, hi_txt - 1))
    for b in range(batch):
        p = 0
        for sgi in range(n_segments):
            end = seq if sgi == n_segments - 1 else (seq * (sgi + 1)) // n_segments
            nt = int(torch.randint(lo_txt, hi_txt + 1, (1,), generator=g))
            nt = min(nt, end - p)
            tokens[b, p:p + nt, K] = torch.randint(0, cfg.text_vocab, (nt,), generator=g)
            mask[b, p:p + nt, K] = True
            p += nt
            na = end - p
            if na > 0:
                codes = torch.randint(0, cfg.audio_vocab - 3, (na, K), generator=g)
                codes[-1] = 0  # EOS frame
                tokens[b, p:end, :K] = codes
                mask[b, p:end, :K] = True
            p = end
    targets = torch.randint(0, cfg.audio_vocab - 3, (batch, seq, K), generator=g)
    return tokens, mask, targets


# --------------------------------------------------------------------------- generation (no KV cache: recompute)
def generate_frame(params, cfg: CsmCfg, tokens: torch.Tensor, mask: torch.Tensor, temperature: float,
                   topk: int, qs: List[torch.Tensor]) -> torch.Tensor:
    """Restatement of ``Model.generate_frame`` (``src/csm/models/model.py:140-195``) without caches.

    tokens/mask hold the WHOLE sequence so far ([B,S,K+1]); qs = K Exp(1) noise tensors [B,V_a].
    Returns sampled codes [B,K] int32.  A KV cache changes no arithmetic, only what is recomputed.
    """
    K = cfg.n_codebooks
    hidden = backbone_hidden(params, cfg, tokens, mask)
    last_h = hidden[:, -1, :]
    c0_logits = last_h @ params["codebook0_head.weight"].t().float()
    c0 = sample_topk(c0_logits, topk, temperature, qs[0])
    samples = [c0]
    seq = [last_h.unsqueeze(1), params["audio_embeddings.weight"][c0.long() + 0 * cfg.audio_vocab].float()]
    for i in range(1, K):
        x = torch.cat(seq, dim=1) @ params["projection.weight"].t().float()
        pos = torch.arange(x.shape[1]).unsqueeze(0).repeat(x.shape[0], 1)
        dec = transformer({k: v.float() for k, v in params.items() if k.startswith("decoder.")},
                          "decoder", cfg.decoder, x, pos)
        logits = dec[:, -1, :] @ params["audio_head"][i - 1].float()
        ci = sample_topk(logits, topk, temperature, qs[i])
        samples.append(ci)
        seq.append(params["audio_embeddings.weight"][ci.long() + i * cfg.audio_vocab].float())
    return torch.cat(samples, dim=1)
