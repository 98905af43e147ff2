"""Fused AdamW over the model's flat arenas (replaces ``optim.AdamW(param_groups, weight_decay)`` +
``clip_grad_norm_`` of reference src/csm/training/trainer.py:166-173,271-277).

One launch per learning-rate group (backbone / decoder / embeddings / other - contiguous ranges of the arena, the
same name-substring routing as the reference) plus one for the LoRA arena; fp32 master weights and moments, bf16
working copy and gradients: 28 bytes of HBM traffic per parameter and step.  The clip coefficient stays on the
device and is applied inside the AdamW kernel, so clipping costs one read of the gradients and no host sync.
"""
from typing import Dict, List, Optional

import torch

from ..hip import ops

F32 = torch.float32


class FusedAdamW:
    def __init__(self, model, group_lrs: Dict[str, float], weight_decay: float = 0.01, betas=(0.9, 0.999),
                 eps: float = 1e-8, lora_lr: Optional[float] = None, lora_weight_decay: float = 0.0):
        self.model = model
        self.betas, self.eps = betas, eps
        self.param_groups: List[dict] = []
        self.state: Dict[str, Dict[str, torch.Tensor]] = {}
        self.step_count = 0
        self._coef: Optional[torch.Tensor] = None
        model.ensure_grads()
        for name, lr in group_lrs.items():
            if lr is None or not model.trainable.get(name, False):
                continue
            off, n = model.group_range(name)
            self.param_groups.append(dict(name=name, lr=float(lr), weight_decay=weight_decay, offset=off, numel=n,
                                          param=model.arena[off:off + n], grad=model.grad_arena[off:off + n]))
        if model.lora is not None and lora_lr is not None:
            lo = model.lora
            self.param_groups.append(dict(name="lora", lr=float(lora_lr), weight_decay=lora_weight_decay, offset=0,
                                          numel=lo.arena.numel(), param=lo.arena, grad=lo.grad_arena))
        for g in self.param_groups:
            p = g["param"]
            master = p.float()
            src = getattr(model, "_fp32_source", None)
            if src and g["name"] != "lora":
                # a checkpoint loaded in fp32 seeds the master copy at full precision
                views, base = model._views(model.arena), model.arena.storage_offset()
                for k, t in src.items():
                    v = views.get(k)
                    if v is None:
                        continue
                    so = v.storage_offset() - base
                    if g["offset"] <= so < g["offset"] + g["numel"]:
                        torch.as_strided(master, v.size(), v.stride(), so - g["offset"]).copy_(t.to(master.device))
            self.state[g["name"]] = dict(master=master, m=torch.zeros_like(master), v=torch.zeros_like(master))
        self._partials = torch.empty(max(1, len(self.param_groups)) * ops.sumsq_blocks(), dtype=F32, device=model.device)
        self._norm_coef = torch.ones(2, dtype=F32, device=model.device)

    def named_master(self):
        """(reference parameter name, fp32 master view) for every trainable base parameter."""
        m = self.model
        views, base = m._views(m.arena), m.arena.storage_offset()
        for g in self.param_groups:
            if g["name"] == "lora":
                continue
            master = self.state[g["name"]]["master"]
            for k, v in views.items():
                so = v.storage_offset() - base
                if g["offset"] <= so < g["offset"] + g["numel"]:
                    yield k, torch.as_strided(master, v.size(), v.stride(), so - g["offset"])

    def num_trainable(self) -> int:
        return sum(g["numel"] for g in self.param_groups)

    def zero_grad(self, set_to_none: bool = False):
        for g in self.param_groups:
            g["grad"].zero_()
            if g["name"] in self.model.grad_state:
                self.model.grad_state[g["name"]] = "zero"
        self._coef = None

    def _settle(self):
        """A group whose gradients a lazy step consumed and that no backward has rewritten since holds stale values:
        zero it before it is read again."""
        gs = self.model.grad_state
        for g in self.param_groups:
            if gs.get(g["name"]) == "stale":
                g["grad"].zero_()
                gs[g["name"]] = "zero"

    def clip_grad_norm(self, max_norm: float) -> torch.Tensor:
        """Global L2 norm over every trainable gradient; the coefficient is consumed by the next ``step``.
        Returns the norm as a 0-d GPU tensor (no host sync)."""
        self._settle()
        nb = ops.sumsq_blocks()
        for i, g in enumerate(self.param_groups):
            ops.sumsq_bf16(g["grad"], self._partials[i * nb:(i + 1) * nb])
        ops.clip_coef(self._partials[:len(self.param_groups) * nb], max_norm, self._norm_coef)
        self._coef = self._norm_coef
        return self._norm_coef[0]

    def step(self, zero_grad=False):
        """One AdamW update.  ``zero_grad=True`` clears the gradients in the same pass over memory.  ``zero_grad="lazy"``
        (what the trainers use) clears only the embedding tables, whose backward is a scatter-add that needs zeros
        underneath; the dense groups are marked stale instead and the next backward OVERWRITES them - their weight-gradient
        GEMMs then skip the read of the old value and nobody writes 2 bytes per parameter of zeros."""
        self._settle()
        self.step_count += 1
        b1, b2 = self.betas
        gs = self.model.grad_state
        for g in self.param_groups:
            st = self.state[g["name"]]
            name = g["name"]
            zero = bool(zero_grad) if zero_grad != "lazy" else (name == "embeddings" or name not in gs)
            ops.adamw_step(st["master"], st["m"], st["v"], g["param"], g["grad"], g["lr"], b1, b2, self.eps,
                           g["weight_decay"], self.step_count, self._coef, zero_grad=zero)
            if name in gs:
                if zero:
                    gs[name] = "zero"
                elif zero_grad == "lazy" and gs[name] == "live":
                    gs[name] = "stale"
        self._coef = None

    def state_dict(self):
        return {"step": self.step_count,
                "groups": [{k: g[k] for k in ("name", "lr", "weight_decay", "offset", "numel")} for g in self.param_groups],
                "state": {n: {k: t.cpu() for k, t in st.items()} for n, st in self.state.items()}}

    def load_state_dict(self, sd):
        """Groups are matched by NAME (backbone / decoder / embeddings / other / lora), never by position: resuming with
        different freeze flags must fail loudly instead of giving a group another group's learning rate."""
        saved_by_name = {g["name"]: g for g in sd["groups"]}
        mine = [g["name"] for g in self.param_groups]
        if sorted(saved_by_name) != sorted(mine) or sorted(sd["state"]) != sorted(mine):
            raise ValueError(f"optimizer state was saved for parameter groups {sorted(saved_by_name)} but this optimizer has "
                             f"{sorted(mine)} (different freeze flags / LoRA setting?)")
        for g in self.param_groups:
            saved = saved_by_name[g["name"]]
            if (saved["offset"], saved["numel"]) != (g["offset"], g["numel"]):
                raise ValueError(f"optimizer group {g['name']!r}: saved range {saved['offset']}+{saved['numel']} != {g['offset']}+{g['numel']}")
        self.step_count = sd["step"]
        for n, st in sd["state"].items():
            for k, t in st.items():
                self.state[n][k].copy_(t)
        for g in self.param_groups:
            saved = saved_by_name[g["name"]]
            g["lr"], g["weight_decay"] = saved["lr"], saved["weight_decay"]
