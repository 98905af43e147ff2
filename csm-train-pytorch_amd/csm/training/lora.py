"""LoRA adapters for the HIP path.  Semantics of reference ``src/csm/mlx/components/lora.py``:

  y = x W0^T + (alpha / r) * ((drop(x) A^T) B^T) (+ lora_bias),  A ~ N(0, 1/sqrt(in)) [r, in],  B = 0 [out, r],
  lora_bias = 0 [out] when ``use_bias`` (``LoRALinear`` lora.py:14-105; dropout is inverted dropout on the adapter
  input only, lora.py:85-90), merge W0 + (alpha/r) B A (lora.py:140-153), applied to both stacks, default
  target modules ["q_proj", "v_proj"], all layers unless ``target_layers`` (``apply_lora_to_model`` lora.py:741-860).

Names are ``{backbone|decoder}.layers.{i}.attn.{q_proj,k_proj,v_proj,output_proj}.lora_{A,B}`` and
``...mlp.{w1,w2,w3}.lora_{A,B}``; the stack prefix fixes the reference's key collision between backbone and decoder
adapters.  The four skinny products per adapter go through the same MFMA GEMM as everything
else (N or K = r = 8), accumulating into the frozen projection's output / input gradient in the GEMM epilogue.
"""
import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from ..hip import ops

BF16 = torch.bfloat16
ATTN = ("q_proj", "k_proj", "v_proj", "output_proj")
MLP = ("w1", "w2", "w3")


class LoRAAdapter:
    def __init__(self, name, A, B, gA, gB, scaling, bias=None, gbias=None, dropout=0.0, seed=0, state=None):
        self.name, self.A, self.B, self.gA, self.gB, self.scaling = name, A, B, gA, gB, scaling
        self.bias, self.gbias, self.dropout, self.seed, self.state = bias, gbias, float(dropout), int(seed), state
        self.r = A.shape[0]
        self._draw = None          # dropout seed of the forward whose backward is pending

    def _dropped(self, x):
        """drop(x): the mask is regenerated from ``self._draw`` (set in forward), never stored."""
        if self._draw is None:
            return x
        return ops.dropout_bf16(x, torch.empty(x.shape, dtype=BF16, device=x.device), self.dropout, self._draw)

    def project(self, x: torch.Tensor, t: Optional[torch.Tensor] = None) -> torch.Tensor:
        """t = drop(x) A^T  [M, r] (draws this forward's dropout mask); ``t`` may be a column block of a wider buffer."""
        self._draw = None
        if self.dropout > 0 and self.state.training:
            self.state.draws += 1
            self._draw = (self.seed * 0x9E3779B1 + self.state.draws * 0x85EBCA77) & (2 ** 63 - 1)
        if t is None:
            t = torch.empty(x.shape[0], self.r, dtype=BF16, device=x.device)
        ops.gemm(self._dropped(x), self.A, t)
        return t

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """y += scaling * (drop(x) A^T) B^T (+ bias); returns t = drop(x) A^T (kept for the backward).  The second
        product accumulates into the frozen projection's output in its GEMM epilogue (C = R + alpha * acc, R = C = y)."""
        t = self.project(x)
        ops.gemm(t, self.B, y, y, alpha=self.scaling)
        if self.bias is not None:
            ops.bias_add_bf16(y, self.bias)
        return t

    @staticmethod
    def _skinny_wgrad(a, b, out, alpha):
        """out[Ma, Nb] += alpha * a[M, Ma]^T b[M, Nb] where one of Ma / Nb is the LoRA rank: the output has a handful of
        tiles and a very long contraction, so the contraction is split over the batch dimension of the GEMM (fp32
        partial slabs) and summed by the column-sum kernel - otherwise 16 workgroups would walk M = 16k rows serially."""
        M = a.shape[0]
        chunk = 256
        nsplit = M // chunk
        if nsplit < 8 or M % chunk:
            ops.gemm(a, b, out, out, True, True, alpha=alpha)
            return
        Ma, Nb = a.shape[1], b.shape[1]
        part = torch.empty(nsplit, Ma * Nb, dtype=torch.float32, device=a.device)
        ops.gemm(a[:chunk], b[:chunk], part[0].view(Ma, Nb), None, True, True, alpha=alpha, batch=nsplit,
                 sA=chunk * a.stride(0), sB=chunk * b.stride(0), sC=Ma * Nb)
        ops.colsum_bf16(part, out.view(-1), accumulate=True)

    def backward(self, x, dy, t, dx):
        """dB += s dy^T t ; dt = s dy B ; dA += dt^T drop(x) ; dx += mask * (dt A) / (1-p) ; dbias += colsum(dy)."""
        dt = torch.empty_like(t)
        ops.gemm(dy, self.B, dt, None, False, True, alpha=self.scaling)
        self._skinny_wgrad(dy, t, self.gB, self.scaling)
        self._skinny_wgrad(dt, self._dropped(x), self.gA, 1.0)
        if self._draw is None:
            ops.gemm(dt, self.A, dx, dx, False, True)
        else:
            dxl = torch.empty(x.shape, dtype=BF16, device=x.device)
            ops.gemm(dt, self.A, dxl, None, False, True)
            ops.dropout_bf16(dxl, dx, self.dropout, self._draw, accumulate=True)
        if self.gbias is not None:
            ops.bias_grad_bf16(dy, self.gbias, accumulate=True)


class LoRAState:
    """All adapters of a model in one bf16 arena (+ gradient arena)."""

    def __init__(self, model, r: int, alpha: float, dropout: float, target_modules: List[str],
                 target_layers: Optional[List[int]], use_bias: bool, seed: int = 0):
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError("lora_dropout must be in [0, 1)")
        if r < 1:
            raise ValueError("lora_r must be positive")
        # any rank (the reference documents r = 4 and its CLI accepts any --lora-r): A / B are stored padded to the next
        # multiple of 8 (16-byte rows, MFMA k-step) with the padding rows of A and columns of B at zero.  The padding
        # stays exactly zero under training - its gradients are products with those zeros - so the padded adapter IS the
        # rank-r adapter; the reference-shaped [r, in] / [out, r] tensors are views (named_tensors).
        self.r, self.r_pad, self.alpha, self.dropout, self.scaling = r, (r + 7) // 8 * 8, alpha, dropout, alpha / r
        self.target_modules, self.target_layers, self.use_bias = list(target_modules), target_layers, use_bias
        self.training, self.draws = True, 0        # dropout is live only while training; draws counts masks drawn
        plan = []
        for prefix, c in (("backbone", model.bb), ("decoder", model.dc)):
            hq, hk = c.num_heads * c.head_dim, c.num_kv_heads * c.head_dim
            dims = {"q_proj": (hq, c.embed_dim), "k_proj": (hk, c.embed_dim), "v_proj": (hk, c.embed_dim),
                    "output_proj": (c.embed_dim, c.embed_dim), "w1": (c.intermediate_dim, c.embed_dim),
                    "w3": (c.intermediate_dim, c.embed_dim), "w2": (c.embed_dim, c.intermediate_dim)}
            for i in range(c.num_layers):
                if target_layers is not None and i not in target_layers:
                    continue
                for mod in target_modules:
                    if mod not in dims:
                        raise ValueError(f"unknown LoRA target module {mod!r}")
                    sub = "attn" if mod in ATTN else "mlp"
                    out_f, in_f = dims[mod]
                    plan.append((prefix, i, mod, f"{prefix}.layers.{i}.{sub}.{mod}", out_f, in_f))
        rp = self.r_pad
        total = sum(rp * in_f + out_f * rp + (out_f if use_bias else 0) for *_, out_f, in_f in plan)
        total = (total + 7) // 8 * 8
        dev = model.device
        self.arena = torch.zeros(total, dtype=BF16, device=dev)
        self.grad_arena = torch.zeros(total, dtype=BF16, device=dev)
        self.adapters: Dict[tuple, LoRAAdapter] = OrderedDict()
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        off = 0
        for prefix, i, mod, name, out_f, in_f in plan:
            A = self.arena[off:off + rp * in_f].view(rp, in_f)
            gA = self.grad_arena[off:off + rp * in_f].view(rp, in_f)
            off += rp * in_f
            B = self.arena[off:off + out_f * rp].view(out_f, rp)
            gB = self.grad_arena[off:off + out_f * rp].view(out_f, rp)
            off += out_f * rp
            bias = gbias = None
            if use_bias:
                bias, gbias = self.arena[off:off + out_f], self.grad_arena[off:off + out_f]
                off += out_f
            A[:r].copy_((torch.randn(r, in_f, generator=g, device=dev) / math.sqrt(in_f)).to(BF16))
            self.adapters[(prefix, i, mod)] = LoRAAdapter(name, A, B, gA, gB, self.scaling, bias, gbias, dropout,
                                                          seed * 1000003 + len(self.adapters), self)

    def get(self, prefix, layer, module):
        return self.adapters.get((prefix, layer, module))

    def named_tensors(self):
        for ad in self.adapters.values():
            yield f"{ad.name}.lora_A", ad.A[:self.r]            # reference shapes: A [r, in], B [out, r] (lora.py:62-66)
            yield f"{ad.name}.lora_B", ad.B[:, :self.r]
            if ad.bias is not None:
                yield f"{ad.name}.lora_bias", ad.bias

    def num_params(self) -> int:
        return sum(t.numel() for _, t in self.named_tensors())


def apply_lora_to_model(model, r: int = 8, alpha: float = 16.0, dropout: float = 0.0,
                        target_modules: Optional[List[str]] = None, target_layers: Optional[List[int]] = None,
                        use_bias: bool = False, seed: int = 0):
    """Reference ``apply_lora_to_model`` (lora.py:741-860): freezes the base model, attaches adapters, and gives the
    model ``get_lora_params()`` / ``merge_lora_weights()``."""
    if target_modules is None:
        target_modules = ["q_proj", "v_proj"]     # reference default, lora.py:801-803
    model.lora = LoRAState(model, r, alpha, dropout, target_modules, target_layers, use_bias, seed)
    for k in model.trainable:
        model.trainable[k] = False
    model.get_lora_params = lambda: OrderedDict(model.lora.named_tensors())
    model.merge_lora_weights = lambda: merge_lora_weights(model)
    return model


@torch.no_grad()
def merge_lora_weights(model):
    """W0 += (alpha/r) * B A for every adapter (reference ``merge_with_base`` lora.py:140-153), on the GPU via the
    same GEMM (C = B[out,r] . (A[r,in] read as [K=r][N=in]) + C).  Returns the model."""
    lo = model.lora
    views = model._views(model.arena)
    for (prefix, i, mod), ad in lo.adapters.items():
        sub = "attn" if mod in ATTN else "mlp"
        W = views[f"{prefix}.layers.{i}.{sub}.{mod}.weight"]
        ops.gemm(ad.B, ad.A, W, W, False, True, alpha=lo.scaling)
    return model
