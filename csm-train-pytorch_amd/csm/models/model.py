"""CSM model on MI355X - same public surface as reference ``src/csm/models/model.py``.

``ModelArgs``, ``Model`` (``setup_caches``, ``generate_frame``, ``reset_caches``, ``_embed_audio``,
``_embed_tokens``), ``sample_topk``, ``_create_causal_mask`` and ``_index_causal_mask`` keep the reference's names,
arguments and state-dict keys (torchtune naming: ``backbone.layers.{i}.attn.q_proj.weight`` ...), but nothing
underneath is torchtune: the parameters are views into flat bf16 arenas laid out for the HIP kernels
(fused q|k|v block, w1/w3 rows interleaved in one block, vocabulary padded to a multiple of 64 for the head GEMMs) and every forward /
backward op is a launch into ``libcsm_hip.so``.

HBM layout (one contiguous bf16 arena, one matching bf16 gradient arena; optimiser keeps fp32 master/m/v):
    [ backbone: per layer sa_norm | qkv | output_proj | mlp_norm | w1w3 | w2 ; final norm ]
    [ decoder : same ]
    [ embeddings: text_embeddings | audio_embeddings ]
    [ other: projection | codebook0_head (rows padded) | audio_head (last dim padded) ]
The four groups are exactly the learning-rate groups of reference ``src/csm/training/trainer.py:143-159``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ..hip import ops, require_device

BF16 = torch.bfloat16


@dataclass
class StackConfig:
    embed_dim: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    intermediate_dim: int
    max_seq_len: int = 2048
    norm_eps: float = 1e-5
    rope_base: float = 500_000.0
    scale_factor: float = 32.0

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def qkv_dim(self) -> int:
        return (self.num_heads + 2 * self.num_kv_heads) * self.head_dim


def llama3_2_1B() -> StackConfig:
    """Same hyper-parameters as reference ``llama3_2_1B`` (src/csm/models/model.py:11-25)."""
    return StackConfig(2048, 16, 32, 8, 8192)


def llama3_2_100M() -> StackConfig:
    """Same hyper-parameters as reference ``llama3_2_100M`` (src/csm/models/model.py:28-42)."""
    return StackConfig(1024, 4, 8, 2, 8192)


FLAVORS = {
    "llama-1B": llama3_2_1B,
    "llama-100M": llama3_2_100M,
    # small shapes for tests (head_dim kept at the real 64 / 128)
    "llama-tiny-backbone": lambda: StackConfig(256, 2, 4, 2, 512, max_seq_len=128),
    "llama-tiny-decoder": lambda: StackConfig(256, 2, 2, 1, 512, max_seq_len=128),
    # ONE layer at the full CSM-1B / CSM-100M widths: the full-shape parity fixtures (tests/golden/make_golden_full.py)
    "llama-1B-L1": lambda: StackConfig(2048, 1, 32, 8, 8192),
    "llama-100M-L1": lambda: StackConfig(1024, 1, 8, 2, 8192),
}


def _create_causal_mask(seq_len: int, device: torch.device):
    """API parity with reference model.py:59-61.  The HIP attention kernels apply causality in-kernel."""
    return torch.tril(torch.ones(seq_len, seq_len, dtype=torch.bool, device=device))


def _index_causal_mask(mask: torch.Tensor, input_pos: torch.Tensor):
    """API parity with reference model.py:64-76."""
    return mask[input_pos, :]


def llama3_rope_table(max_seq_len: int, head_dim: int, base: float, scale: float, low: float = 1.0, high: float = 4.0,
                      old_ctx: int = 8192) -> torch.Tensor:
    """[P, hd/2, 2] (cos, sin) of the Llama-3 scaled frequencies, fp32, built on the host the way torchtune 0.4.0's
    ``Llama3ScaledRoPE`` builds its cache (the stacks of reference models/model.py:11-42): every step of the frequency
    scaling is an fp32 tensor operation, not double arithmetic rounded at the end."""
    one = torch.ones((), dtype=torch.float32)
    freqs = one / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    theta = torch.empty_like(freqs)
    for i in range(freqs.numel()):
        f = freqs[i]
        wl = 2 * math.pi / f
        if wl < old_ctx / high:
            theta[i] = f
        elif wl > old_ctx / low:
            theta[i] = f / scale
        else:
            smooth = (old_ctx / wl - low) / (high - low)
            theta[i] = (1 - smooth) * f / scale + smooth * f
    ang = torch.einsum("i,j->ij", torch.arange(max_seq_len, dtype=torch.float32), theta).float()
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous()


def sample_topk(logits: torch.Tensor, topk: int, temperature: float, q: Optional[torch.Tensor] = None):
    """Reference ``sample_topk`` (model.py:85-96) on the GPU: returns int32 [..., 1].

    The Exp(1) draw of ``_multinomial_sample_one_no_sync`` comes from torch's generator (or ``q`` when given, which is
    how parity tests pin the result); threshold / softmax / argmax run in one HIP kernel.
    """
    lg = logits.float()
    lead = lg.shape[:-1]
    lg2 = lg.reshape(-1, lg.shape[-1])
    if lg2.stride(1) != 1:
        lg2 = lg2.contiguous()
    if q is None:
        q = torch.empty(lg2.shape, dtype=torch.float32, device=lg2.device).exponential_(1)
    q = q.reshape(lg2.shape).float().contiguous()
    out = torch.empty(lg2.shape[0], dtype=torch.int32, device=lg2.device)
    ops.sample_topk(lg2, q, out, topk, temperature, V=lg2.shape[1])
    return out.reshape(*lead, 1)


@dataclass
class ModelArgs:
    """Arguments for the CSM model (reference model.py:99-107)."""

    backbone_flavor: str
    decoder_flavor: str
    text_vocab_size: int
    audio_vocab_size: int
    audio_num_codebooks: int


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


class _Slot:
    __slots__ = ("name", "shape", "offset", "numel", "group")

    def __init__(self, name, shape, offset, group):
        self.name, self.shape, self.offset, self.group = name, tuple(shape), offset, group
        self.numel = 1
        for s in shape:
            self.numel *= s


class Model(nn.Module):
    """Conversational Speech Model (reference model.py:110-217) over flat HIP-friendly arenas."""

    def __init__(self, args: ModelArgs, device: Optional[str] = None, seed: Optional[int] = None):
        super().__init__()
        self.args = args
        self.bb: StackConfig = FLAVORS[args.backbone_flavor]()
        self.dc: StackConfig = FLAVORS[args.decoder_flavor]()
        self.vocab_pad = _pad64(args.audio_vocab_size)
        self._slots: Dict[str, _Slot] = OrderedDict()       # internal (fused / padded) blocks
        self._groups: Dict[str, Tuple[int, int]] = {}        # group -> (offset, numel)
        self._plan()
        self._device = torch.device("cpu")
        self.arena: Optional[torch.Tensor] = None
        self.grad_arena: Optional[torch.Tensor] = None
        self.lora = None                                      # set by csm.training.lora.apply_lora_to_model
        self.acoustic_mode = "off"                            # "off" (reference placeholder) | "all" | "amortized"
        self.acoustic_fraction = 1.0 / 16.0
        # state of each group's slice of the gradient arena: "zero" (all zeros), "live" (being accumulated into), "stale"
        # (consumed by a lazy optimizer step: the next backward overwrites it) - see Engine.backward / FusedAdamW.step
        self.grad_state = {"backbone": "zero", "decoder": "zero", "embeddings": "zero", "other": "zero"}
        self.target_ignore_index = None                       # e.g. -100: padded target frames leave the loss (data.IGNORE_INDEX)
        self._tables: Dict[str, torch.Tensor] = {}
        self._engine = None
        self._init_seed = seed
        self._fp32_source = None
        self.use_kv_cache = True                              # generate_frame: KV caches (False = prefix recompute)
        self.use_hip_graph = True                             # replay decode frames as one captured HIP graph
        # which parameter groups receive weight gradients (freeze flags of trainer.prepare_optimizer / LoRA)
        self.trainable = {"backbone": True, "decoder": True, "embeddings": True, "other": True}
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ layout
    def _plan(self):
        a = self.args
        off = 0

        def add(name, shape, group):
            nonlocal off
            off = (off + 63) // 64 * 64
            s = _Slot(name, shape, off, group)
            self._slots[name] = s
            off += s.numel

        def stack(prefix, c: StackConfig):
            for i in range(c.num_layers):
                p = f"{prefix}.layers.{i}"
                add(f"{p}.sa_norm.scale", (c.embed_dim,), prefix)
                add(f"{p}.attn.qkv", (c.qkv_dim, c.embed_dim), prefix)
                add(f"{p}.attn.output_proj.weight", (c.embed_dim, c.embed_dim), prefix)
                add(f"{p}.mlp_norm.scale", (c.embed_dim,), prefix)
                add(f"{p}.mlp.w13", (2 * c.intermediate_dim, c.embed_dim), prefix)
                add(f"{p}.mlp.w2.weight", (c.embed_dim, c.intermediate_dim), prefix)
            add(f"{prefix}.norm.scale", (c.embed_dim,), prefix)

        for grp, fn in (("backbone", lambda: stack("backbone", self.bb)), ("decoder", lambda: stack("decoder", self.dc)),
                        ("embeddings", lambda: (add("text_embeddings.weight", (a.text_vocab_size, self.bb.embed_dim), "embeddings"),
                                                add("audio_embeddings.weight", (a.audio_vocab_size * a.audio_num_codebooks, self.bb.embed_dim), "embeddings"))),
                        ("other", lambda: (add("projection.weight", (self.dc.embed_dim, self.bb.embed_dim), "other"),
                                           add("codebook0_head.padded", (self.vocab_pad, self.bb.embed_dim), "other"),
                                           add("audio_head.padded", (a.audio_num_codebooks - 1, self.dc.embed_dim, self.vocab_pad), "other")))):
            off = (off + 63) // 64 * 64
            start = off
            fn()
            off = (off + 63) // 64 * 64
            self._groups[grp] = (start, off - start)
        self._numel = off

    def _views(self, arena: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        """Reference-named tensors as views of ``arena`` (same key order as the reference module tree)."""
        a = self.args
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()

        def blk(name):
            s = self._slots[name]
            return arena[s.offset:s.offset + s.numel].view(s.shape)

        def stack(prefix, c: StackConfig):
            hq, hk = c.num_heads * c.head_dim, c.num_kv_heads * c.head_dim
            for i in range(c.num_layers):
                p = f"{prefix}.layers.{i}"
                qkv, w13 = blk(f"{p}.attn.qkv"), blk(f"{p}.mlp.w13")
                out[f"{p}.attn.q_proj.weight"] = qkv[:hq]
                out[f"{p}.attn.k_proj.weight"] = qkv[hq:hq + hk]
                out[f"{p}.attn.v_proj.weight"] = qkv[hq + hk:]
                out[f"{p}.attn.output_proj.weight"] = blk(f"{p}.attn.output_proj.weight")
                out[f"{p}.mlp.w1.weight"] = w13[0::2]       # gate rows and up rows are interleaved (SwiGLU GEMM epilogue)
                out[f"{p}.mlp.w2.weight"] = blk(f"{p}.mlp.w2.weight")
                out[f"{p}.mlp.w3.weight"] = w13[1::2]
                out[f"{p}.sa_norm.scale"] = blk(f"{p}.sa_norm.scale")
                out[f"{p}.mlp_norm.scale"] = blk(f"{p}.mlp_norm.scale")
            out[f"{prefix}.norm.scale"] = blk(f"{prefix}.norm.scale")

        stack("backbone", self.bb)
        stack("decoder", self.dc)
        out["text_embeddings.weight"] = blk("text_embeddings.weight")
        out["audio_embeddings.weight"] = blk("audio_embeddings.weight")
        out["projection.weight"] = blk("projection.weight")
        out["codebook0_head.weight"] = blk("codebook0_head.padded")[:a.audio_vocab_size]
        out["audio_head"] = blk("audio_head.padded")[:, :, :a.audio_vocab_size]
        return out

    def block(self, name: str, grad: bool = False) -> torch.Tensor:
        """Internal fused / padded block by slot name (what the kernels consume)."""
        s = self._slots[name]
        src = self.grad_arena if grad else self.arena
        return src[s.offset:s.offset + s.numel].view(s.shape)

    def group_range(self, group: str) -> Tuple[int, int]:
        return self._groups[group]

    # ------------------------------------------------------------------ device / init
    def to(self, *args, **kwargs):  # noqa: D401 - nn.Module.to signature
        """Move to a device.  The working dtype on the GPU is always bf16 (fp32 master copies live in the optimiser);
        a ``dtype=`` argument is accepted for call-site compatibility with the reference and otherwise ignored."""
        device = kwargs.get("device", None)
        for a in args:
            if isinstance(a, (str, torch.device)):
                device = a
        if device is None:
            return self
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("csm (MI355X build): the model lives on a gfx950 GPU; there is no CPU execution path")
        require_device(device.index or 0)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        old = self.arena
        self.arena = torch.zeros(self._numel, dtype=BF16, device=device)
        if old is not None:
            self.arena.copy_(old)
        self.grad_arena = None
        self._device = device
        self._bind()
        if old is None:
            self.reset_parameters(self._init_seed)
        self._tables = {}
        self._engine = None
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    def _bind(self):
        for name, view in self._views(self.arena).items():
            mod, _, leaf = name.rpartition(".")
            holder = self
            if mod:
                for part in mod.split("."):
                    if not hasattr(holder, part):
                        holder.add_module(part, nn.Module())
                    holder = getattr(holder, part)
            if leaf in holder._parameters:
                del holder._parameters[leaf]
            holder.register_parameter(leaf, nn.Parameter(view, requires_grad=True))

    def ensure_grads(self) -> torch.Tensor:
        """Allocate the bf16 gradient arena and expose it as ``param.grad`` views."""
        if self.grad_arena is None:
            self.grad_arena = torch.zeros_like(self.arena)
            self.grad_state.update({k: "zero" for k in self.grad_state})
            gviews = self._views(self.grad_arena)
            for name, p in self.named_parameters():
                if name in gviews:
                    p.grad = gviews[name]
        return self.grad_arena

    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None, std: float = 0.02):
        """N(0, 0.02) matrices, unit norm scales, zero padding (the reference leaves ``audio_head`` uninitialised:
        model.py:126)."""
        g = torch.Generator(device=self._device)
        g.manual_seed(0 if seed is None else seed)
        self.arena.zero_()
        for name, v in self._views(self.arena).items():
            if name.endswith(".scale"):
                v.fill_(1.0)
            else:
                v.copy_((torch.randn(v.shape, generator=g, device=self._device, dtype=torch.float32) * std).to(BF16))

    @property
    def device(self) -> torch.device:
        return self._device

    # ------------------------------------------------------------------ state dict (reference keys)
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else OrderedDict()
        if self._engine is not None:
            self._engine._need()         # (ZeRO-1: updated parameter shards may still be on their way, training/dp.py)
        for name, v in self._views(self.arena).items():
            out[prefix + name] = v if keep_vars else v.detach().clone()
        if self.lora is not None:
            for name, v in self.lora.named_tensors():
                out[prefix + name] = v if keep_vars else v.detach().clone()
        return out

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        views = self._views(self.arena)
        lora_names = dict(self.lora.named_tensors()) if self.lora is not None else {}
        missing = [k for k in views if k not in state_dict]
        unexpected = [k for k in state_dict if k not in views and k not in lora_names
                      and not k.endswith("causal_mask") and "kv_cache" not in k]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing keys {missing[:5]}..., unexpected keys {unexpected[:5]}...")
        for k, v in state_dict.items():
            dst = views.get(k, lora_names.get(k))
            if dst is None:
                continue
            if tuple(v.shape) != tuple(dst.shape):
                raise RuntimeError(f"load_state_dict: shape mismatch for {k}: {tuple(v.shape)} vs {tuple(dst.shape)}")
            dst.copy_(v.to(device=dst.device, dtype=dst.dtype))
        self._fp32_source = {k: v for k, v in state_dict.items() if torch.is_tensor(v) and v.dtype == torch.float32} or None
        self.params_rewritten(base=any(k in views for k in state_dict), lora=any(k in lora_names for k in state_dict),
                              from_fp32_source=True)
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def params_rewritten(self, base: bool = True, lora: bool = True, from_fp32_source: bool = False):
        """Tell every live optimiser built on this model that the working weights changed underneath it (its fp32 master
        is partly stored IN the bf16 arena: training/optim.py).  Any rewrite that is not ``load_state_dict`` itself also
        retires the fp32 state dict kept from the last load - it no longer describes the weights."""
        if not from_fp32_source and base:
            self._fp32_source = None
        alive = []
        for ref in getattr(self, "_optimizers", []):
            opt = ref()
            if opt is not None:
                opt.params_rewritten(base, lora, from_fp32_source)
                alive.append(ref)
        if hasattr(self, "_optimizers"):
            self._optimizers = alive

    # ------------------------------------------------------------------ helpers shared with the engine
    def rope_table(self, which: str) -> torch.Tensor:
        if which not in self._tables:
            c = self.bb if which == "backbone" else self.dc
            self._tables[which] = llama3_rope_table(c.max_seq_len, c.head_dim, c.rope_base, c.scale_factor).to(self._device)
        return self._tables[which]

    @property
    def engine(self):
        if self._engine is None:
            from ..engine import Engine
            self._engine = Engine(self)
        return self._engine

    # ------------------------------------------------------------------ reference API
    def setup_caches(self, max_batch_size: int) -> None:
        """Reference model.py:128-138.  Registers the two causal-mask buffers for API parity; the KV state of
        ``generate_frame`` (KV caches of both stacks) is kept by the engine's ``DecodeState``."""
        dev = self._device
        self._max_batch = max_batch_size
        self.register_buffer("backbone_causal_mask", _create_causal_mask(self.bb.max_seq_len, dev), persistent=False)
        self.register_buffer("decoder_causal_mask", _create_causal_mask(self.args.audio_num_codebooks, dev), persistent=False)
        self._gen_hist = None
        self._decode_state = None

    def caches_are_enabled(self) -> bool:
        return hasattr(self, "backbone_causal_mask")

    def reset_caches(self):
        self._gen_hist = None
        self._decode_state = None

    def _embed_audio(self, codebook: int, tokens: torch.Tensor) -> torch.Tensor:
        """Reference model.py:202-204 (a plain row gather: indexing, no arithmetic)."""
        return self.audio_embeddings.weight[tokens + codebook * self.args.audio_vocab_size]

    def _embed_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        """Reference model.py:206-217: [B,S,K+1] ids -> [B,S,K+1,D] rows (gather only; the masked SUM used by the
        train / generate paths is the fused ``csm_embed_fwd`` kernel)."""
        text = self.text_embeddings.weight[tokens[:, :, -1]].unsqueeze(-2)
        idx = tokens[:, :, :-1] + self.args.audio_vocab_size * torch.arange(self.args.audio_num_codebooks, device=tokens.device)
        audio = self.audio_embeddings.weight[idx]
        return torch.cat([audio, text], dim=-2)

    @torch.no_grad()
    def generate_frame(self, tokens: torch.Tensor, tokens_mask: torch.Tensor, input_pos: torch.Tensor, temperature: float,
                       topk: int, noise: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """Reference model.py:140-195: one frame of K codes, [B, K] int32.  ``noise`` (K tensors [B, V_a] of Exp(1)
        draws) pins the sampler for parity tests."""
        assert self.caches_are_enabled(), "backbone caches are not enabled"
        return self.engine.generate_frame(tokens, tokens_mask, input_pos, temperature, topk, noise)
