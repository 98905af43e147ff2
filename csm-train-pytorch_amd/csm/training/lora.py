"""LoRA adapters for the HIP path.  Semantics of reference ``src/csm/mlx/components/lora.py``:

  y = x W0^T + (alpha / r) * ((drop(x) A^T) B^T) (+ lora_bias),  A ~ N(0, 1/sqrt(in)) [r, in],  B = 0 [out, r],
  lora_bias = 0 [out] when ``use_bias`` (``LoRALinear`` lora.py:14-105; dropout is inverted dropout on the adapter
  input only, lora.py:85-90), merge W0 + (alpha/r) B A (lora.py:140-153), applied to both stacks, default
  target modules ["q_proj", "v_proj"], all layers unless ``target_layers`` (``apply_lora_to_model`` lora.py:741-860).

Names are ``{backbone|decoder}.layers.{i}.attn.{q_proj,k_proj,v_proj,output_proj}.lora_{A,B}`` and
``...mlp.{w1,w2,w3}.lora_{A,B}``; the stack prefix fixes the reference's key collision between backbone and decoder
adapters.

Storage follows the GEMMs the adapters ride on, not the module list.  The adapters of one layer that share an input and
feed one fused frozen projection form a GROUP - q|k|v (input: the normed hidden state, output: the fused qkv buffer),
output_proj, w1|w3 (output: the interleaved gate/up buffer), w2 - and a group owns two matrices:

  At [in, KX]   column block c0..c0+r_pad of adapter a = A_a^T          (KX = the group's ranks padded to a multiple of 32)
  Bx [N,  KX]   rows of the fused projection that belong to adapter a, same column block = B_a; every other entry 0

so that the whole group is ONE extra operand pair of the frozen projection's GEMM (``csm_gemm_bf16_kext``: y = x W0^T +
(s x At) Bx^T, the extension's k-steps taken inside the same product, before the RoPE / SwiGLU epilogue) and of its
input-gradient GEMM (dx = dy W0 + (s dy Bx) At^T), plus one skinny product each for s x At, s dy Bx and the two weight
gradients.  The reference-shaped tensors (A [r, in], B [out, r]) are strided views of those matrices.  Zero entries of
Bx outside the adapters' blocks get no gradient (masked) and never move.  Adapters with dropout (while training) or a
bias take the per-adapter path below instead, on the same storage.
"""
import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from ..hip import ops

BF16 = torch.bfloat16
ATTN = ("q_proj", "k_proj", "v_proj", "output_proj")
MLP = ("w1", "w2", "w3")
GROUPS = OrderedDict([("attn_in", ("q_proj", "k_proj", "v_proj")), ("attn_out", ("output_proj",)),
                      ("mlp_in", ("w1", "w3")), ("mlp_out", ("w2",))])
GROUP_OF = {mod: g for g, mods in GROUPS.items() for mod in mods}


def skinny_wgrad(a, b, out, alpha=1.0):
    """out[Ma, Nb] += alpha * a[M, Ma]^T b[M, Nb] where Ma or Nb is a handful of LoRA ranks: the output has few tiles and a
    very long contraction, so the contraction is split over the batch dimension of the GEMM (fp32 partial slabs) and
    summed by the column-sum kernel - otherwise a few workgroups would walk M = 16k rows serially."""
    M = a.shape[0]
    chunk = 1024 if (M % 1024 == 0 and M >= 8192) else 256
    nsplit = M // chunk
    if nsplit < 8 or M % chunk:
        ops.gemm(a, b, out, out, True, True, alpha=alpha)
        return
    Ma, Nb = a.shape[1], b.shape[1]
    part = torch.empty(nsplit, Ma * Nb, dtype=torch.float32, device=a.device)
    ops.gemm(a[:chunk], b[:chunk], part[0].view(Ma, Nb), None, True, True, alpha=alpha, batch=nsplit,
             sA=chunk * a.stride(0), sB=chunk * b.stride(0), sC=Ma * Nb)
    if out.is_contiguous():
        ops.colsum_bf16(part, out.view(-1), accumulate=True)
    else:                                          # a strided view of a group matrix (per-adapter path)
        tmp = torch.empty(Ma, Nb, dtype=BF16, device=a.device)
        ops.colsum_bf16(part, tmp.view(-1), accumulate=False)
        out.add_(tmp)


class LoRAAdapter:
    """One module's adapter: ``At`` [in, r_pad] and ``B`` [out, r_pad] are row-major strided views of its group's
    matrices (``A`` / ``gA`` give the reference orientation [r_pad, in])."""

    def __init__(self, name, At, B, gAt, gB, scaling, bias=None, gbias=None, dropout=0.0, seed=0, state=None):
        self.name, self.At, self.B, self.gAt, self.gB, self.scaling = name, At, B, gAt, gB, scaling
        self.bias, self.gbias, self.dropout, self.seed, self.state = bias, gbias, float(dropout), int(seed), state
        self.r = At.shape[1]
        self._draw = None          # dropout seed of the forward whose backward is pending

    @property
    def A(self):
        return self.At.t()

    @property
    def gA(self):
        return self.gAt.t()

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
        ops.gemm(self._dropped(x), self.At, t, None, False, True)
        return t

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """y += scaling * (drop(x) A^T) B^T (+ bias); returns t = drop(x) A^T (kept for the backward).  The second
        product accumulates into the frozen projection's output in its GEMM epilogue (C = R + alpha * acc, R = C = y)."""
        t = self.project(x)
        ops.gemm(t, self.B, y, y, alpha=self.scaling)
        if self.bias is not None:
            ops.bias_add_bf16(y, self.bias)
        return t

    def backward(self, x, dy, t, dx):
        """dB += s dy^T t ; dt = s dy B ; dA += dt^T drop(x) ; dx += mask * (dt A) / (1-p) ; dbias += colsum(dy)."""
        dt = torch.empty_like(t)
        ops.gemm(dy, self.B, dt, None, False, True, alpha=self.scaling)
        skinny_wgrad(dy, t, self.gB, self.scaling)
        skinny_wgrad(self._dropped(x), dt, self.gAt, 1.0)
        if self._draw is None:
            ops.gemm(dt, self.At, dx, dx)
        else:
            dxl = torch.empty(x.shape, dtype=BF16, device=x.device)
            ops.gemm(dt, self.At, dxl)
            ops.dropout_bf16(dxl, dx, self.dropout, self._draw, accumulate=True)
        if self.gbias is not None:
            ops.bias_grad_bf16(dy, self.gbias, accumulate=True)


class LoRAGroup:
    """The adapters of one layer that share an input and a fused frozen projection (see the module docstring)."""

    def __init__(self, state, name, At, Bx, gAt, gBx):
        self.state, self.name, self.At, self.Bx, self.gAt, self.gBx = state, name, At, Bx, gAt, gBx
        self.kx = At.shape[1]
        self.adapters: Dict[str, LoRAAdapter] = OrderedDict()
        self.mask = None           # 0/1 [N, KX] over Bx when entries outside the adapters' blocks could receive a gradient
        self.AtT = self.BxT = None  # [KX, in] / [KX, N] copies for the bandwidth-shaped skinny products (LoRAState.refresh)

    def fusable(self) -> bool:
        """One operand pair for the whole group: no bias, and no dropout mask to draw in this pass."""
        return all(ad.bias is None and not (ad.dropout > 0 and self.state.training) for ad in self.adapters.values())

    # ---- the fused path (engine): forward / backward of "frozen projection + group" -----------------------------
    def project(self, x):
        """tx = s x At  [M, KX]: the extension operand of the forward GEMM (kept for the backward)."""
        tx = torch.empty(x.shape[0], self.kx, dtype=BF16, device=x.device)
        if self.AtT is not None:
            ops.skinny_nt(x, self.AtT, tx, alpha=self.state.scaling)
        else:
            ops.gemm(x, self.At, tx, None, False, True, alpha=self.state.scaling)
        return tx

    def backward(self, x, dy, tx):
        """Weight gradients of the group and the extension operand of the input-gradient GEMM: returns dts = s dy Bx."""
        dts = torch.empty(dy.shape[0], self.kx, dtype=BF16, device=dy.device)
        if self.BxT is not None:
            ops.skinny_nt(dy, self.BxT, dts, alpha=self.state.scaling)
        else:
            ops.gemm(dy, self.Bx, dts, None, False, True, alpha=self.state.scaling)
        skinny_wgrad(dy, tx, self.gBx)              # d/dBx = dy^T (s x At)
        if self.mask is not None:
            self.gBx.mul_(self.mask)
        skinny_wgrad(x, dts, self.gAt)              # d/dAt = x^T (s dy Bx)
        return dts


class LoRAState:
    """All adapters of a model in one bf16 arena (+ gradient arena)."""

    def __init__(self, model, r: int, alpha: float, dropout: float, target_modules: List[str],
                 target_layers: Optional[List[int]], use_bias: bool, seed: int = 0):
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError("lora_dropout must be in [0, 1)")
        if r < 1:
            raise ValueError("lora_r must be positive")
        # any rank (the reference documents r = 4 and its CLI accepts any --lora-r): an adapter's column block is padded to
        # the next multiple of 8 (MFMA k granularity of the per-adapter path) with the padding at zero.  It stays exactly
        # zero under training - its gradients are products with those zeros - so the padded adapter IS the rank-r adapter;
        # the reference-shaped [r, in] / [out, r] tensors are views (named_tensors).
        self.r, self.r_pad, self.alpha, self.dropout, self.scaling = r, (r + 7) // 8 * 8, alpha, dropout, alpha / r
        self.target_modules, self.target_layers, self.use_bias = list(target_modules), target_layers, use_bias
        self.training, self.draws = True, 0        # dropout is live only while training; draws counts masks drawn
        for mod in target_modules:
            if mod not in GROUP_OF:
                raise ValueError(f"unknown LoRA target module {mod!r}")
        rp = self.r_pad
        plan = []                                  # (prefix, layer, group, in_f, rows, [(mod, out_f, row slice)])
        for prefix, c in (("backbone", model.bb), ("decoder", model.dc)):
            hq, hk, d, F = c.num_heads * c.head_dim, c.num_kv_heads * c.head_dim, c.embed_dim, c.intermediate_dim
            members = {"attn_in": (d, hq + 2 * hk, [("q_proj", hq, slice(0, hq)), ("k_proj", hk, slice(hq, hq + hk)),
                                                    ("v_proj", hk, slice(hq + hk, hq + 2 * hk))]),
                       "attn_out": (hq, d, [("output_proj", d, slice(0, d))]),
                       # w13 keeps gate / up rows interleaved (g0, u0, g1, u1, ...): w1 = even rows, w3 = odd rows
                       "mlp_in": (d, 2 * F, [("w1", F, slice(0, 2 * F, 2)), ("w3", F, slice(1, 2 * F, 2))]),
                       "mlp_out": (F, d, [("w2", d, slice(0, d))])}
            for i in range(c.num_layers):
                if target_layers is not None and i not in target_layers:
                    continue
                for gname, (in_f, rows, mods) in members.items():
                    present = [(mod, out_f, sl) for mod, out_f, sl in mods if mod in target_modules]
                    if present:
                        plan.append((prefix, i, gname, in_f, rows, present))
        kx_of = lambda n: (n * rp + 31) // 32 * 32   # noqa: E731
        total = sum((in_f + rows) * kx_of(len(p)) + (sum(o for _, o, _ in p) if use_bias else 0) for *_, in_f, rows, p in plan)
        total = (total + 7) // 8 * 8
        dev = model.device
        self.arena = torch.zeros(total, dtype=BF16, device=dev)
        self.grad_arena = torch.zeros(total, dtype=BF16, device=dev)
        self.adapters: Dict[tuple, LoRAAdapter] = OrderedDict()
        self.groups: Dict[tuple, LoRAGroup] = OrderedDict()
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        off = 0

        def take(n, shape):
            nonlocal off
            w, gr = self.arena[off:off + n].view(shape), self.grad_arena[off:off + n].view(shape)
            off += n
            return w, gr

        # the groups of one kind in one stack lie side by side ([layers, in, KX] and [layers, N, KX]): refresh() transposes
        # a whole stack with one copy
        stacks = OrderedDict()
        for entry in plan:
            stacks.setdefault((entry[0], entry[2]), []).append(entry)
        self.stacks = []
        built = {}
        for (prefix, gname), entries in stacks.items():
            _, _, _, in_f, rows, present = entries[0]
            kx, L = kx_of(len(present)), len(entries)
            At_all, gAt_all = take(L * in_f * kx, (L, in_f, kx))
            Bx_all, gBx_all = take(L * rows * kx, (L, rows, kx))
            grps = []
            for li, (_, i, _, _, _, _) in enumerate(entries):
                grp = LoRAGroup(self, f"{prefix}.layers.{i}.{gname}", At_all[li], Bx_all[li], gAt_all[li], gBx_all[li])
                built[(prefix, i, gname)] = (grp, present, in_f, rows)
                grps.append(grp)
            self.stacks.append((At_all, Bx_all, grps))
        for prefix, i, gname, in_f, rows, present in plan:          # adapters in plan order (layer-major), as named_tensors lists them
            grp, _, _, _ = built[(prefix, i, gname)]
            At, Bx, gAt, gBx, kx = grp.At, grp.Bx, grp.gAt, grp.gBx, grp.kx
            covered = torch.zeros(rows, kx, dtype=BF16, device=dev)
            for j, (mod, out_f, sl) in enumerate(present):
                c0 = j * rp
                bias = gbias = None
                if use_bias:
                    bias, gbias = take(out_f, (out_f,))
                sub = "attn" if mod in ATTN else "mlp"
                At[:, c0:c0 + r].copy_((torch.randn(r, in_f, generator=g, device=dev) / math.sqrt(in_f)).to(BF16).t())
                ad = LoRAAdapter(f"{prefix}.layers.{i}.{sub}.{mod}", At[:, c0:c0 + rp], Bx[sl, c0:c0 + rp], gAt[:, c0:c0 + rp],
                                 gBx[sl, c0:c0 + rp], self.scaling, bias, gbias, dropout, seed * 1000003 + len(self.adapters), self)
                covered[sl, c0:c0 + rp] = 1
                self.adapters[(prefix, i, mod)] = ad
                grp.adapters[mod] = ad
            if bool((covered[:, :len(present) * rp] == 0).any()):
                grp.mask = covered
            self.groups[(prefix, i, gname)] = grp

    @torch.no_grad()
    def refresh(self):
        """[KX, in] / [KX, N] transposes of every group's matrices, one copy per stack and kind: what the bandwidth-shaped
        skinny products (csm_skinny_nt_bf16) read.  Called at the start of every forward pass (the parameters may have
        moved since the last one)."""
        for At_all, Bx_all, grps in self.stacks:
            AtT, BxT = At_all.transpose(1, 2).contiguous(), Bx_all.transpose(1, 2).contiguous()
            for li, grp in enumerate(grps):
                grp.AtT, grp.BxT = AtT[li], BxT[li]

    def get(self, prefix, layer, module):
        return self.adapters.get((prefix, layer, module))

    def group(self, prefix, layer, gname):
        return self.groups.get((prefix, layer, gname))

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
    same GEMM (C = B[out,r] . At[in,r]^T + C).  Returns the model."""
    lo = model.lora
    views = model._views(model.arena)
    for (prefix, i, mod), ad in lo.adapters.items():
        sub = "attn" if mod in ATTN else "mlp"
        W = views[f"{prefix}.layers.{i}.{sub}.{mod}.weight"]
        ops.gemm(ad.B, ad.At, W, W, False, False, alpha=lo.scaling)
    model.params_rewritten(lora=False)        # an optimiser over the base weights must not keep the old lower halves
    return model
