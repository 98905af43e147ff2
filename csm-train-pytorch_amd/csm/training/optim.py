"""Fused AdamW over the model's flat arenas (replaces ``optim.AdamW(param_groups, weight_decay)`` +
``clip_grad_norm_`` of reference src/csm/training/trainer.py:166-173,271-277).

One launch per learning-rate group (backbone / decoder / embeddings / other - contiguous ranges of the arena, the
same name-substring routing as the reference) plus one for the LoRA arena; fp32 master weights and moments, bf16
working copy and gradients.  The clip coefficient stays on the device and is applied inside the AdamW kernel, so
clipping costs one read of the gradients and no host sync.

Master weights are fp32 but not stored as such: the bf16 working copy IS the upper half of the master (rounded
half-up) and a 16-bit ``lo`` tensor holds the lower half, master = ((hi - (lo >> 15)) << 16) | lo exactly - 26 instead of
28 bytes of HBM traffic per parameter and step and 2 bytes less state (``CSM_ADAMW_SPLIT=0`` restores the plain fp32
master for A/B).  ``state_dict`` / ``named_master`` still speak fp32 masters, so checkpoints keep their format.
"""
import os
from typing import Dict, List, Optional

import torch

from ..hip import ops

F32 = torch.float32
SPLIT_MASTER = os.environ.get("CSM_ADAMW_SPLIT", "1") == "1"


def split_master(master: torch.Tensor, param_out: torch.Tensor) -> torch.Tensor:
    """fp32 master -> bf16 working copy (upper half, rounded half-up, written into ``param_out``) + int16 lower half."""
    bits = master.contiguous().view(torch.int32)
    param_out.view(torch.int16).copy_(((bits + 0x8000) >> 16).to(torch.int16))
    return (bits & 0xFFFF).to(torch.int16)          # values >= 0x8000 wrap to negative int16: same 16 bits


def join_master(param: torch.Tensor, lo: torch.Tensor) -> torch.Tensor:
    """The exact fp32 master from the two halves."""
    hi = param.view(torch.int16).to(torch.int32) & 0xFFFF
    l = lo.to(torch.int32) & 0xFFFF
    return (((hi - (l >> 15)) << 16) | l).view(torch.float32)


def seed_master(model, offset: int, numel: int, param: torch.Tensor, from_fp32_source: bool = False) -> torch.Tensor:
    """fp32 master of the arena range [offset, offset + numel) from the current bf16 working weights ``param``.
    ``from_fp32_source``: the weights are the ones ``Model.load_state_dict`` just wrote (or wrote before this optimiser
    existed and nobody has touched since), so a state dict loaded in fp32 (``Model._fp32_source``) seeds its parameters at full
    precision.  Every other rewrite (merge, broadcast, restore) must NOT look at that dict: it holds the checkpoint as loaded,
    not the current weights.  (A tensor that only partly lies inside the range - a ZeRO-1 shard boundary - contributes the
    part that does.)"""
    master = param.float()
    src = getattr(model, "_fp32_source", None) if from_fp32_source else None
    if src:
        views, base = model._views(model.arena), model.arena.storage_offset()
        for k, t in src.items():
            v = views.get(k)
            if v is None:
                continue
            so = v.storage_offset() - base
            end = _span_end(v, so)
            if so >= offset and end <= offset + numel:
                torch.as_strided(master, v.size(), v.stride(), so - offset).copy_(t.to(master.device))
            elif so < offset + numel and end > offset:
                # straddles the range (a sharded optimiser's piece): through a scratch image of the tensor's whole span
                # (the image starts from what ``master`` already holds: interleaved views - w1 / w3 rows of w13 - share a span)
                a_, b_ = max(so, offset), min(end, offset + numel)
                scratch = torch.zeros(end - so, dtype=master.dtype, device=master.device)
                scratch[a_ - so:b_ - so] = master[a_ - offset:b_ - offset]
                torch.as_strided(scratch, v.size(), v.stride(), 0).copy_(t.to(master.device))
                master[a_ - offset:b_ - offset] = scratch[a_ - so:b_ - so]
    return master


def _span_end(v: torch.Tensor, so: int) -> int:
    """One past the last arena element a (possibly strided) view touches."""
    return so + sum((sz - 1) * st for sz, st in zip(v.size(), v.stride())) + 1


class FusedAdamW:
    sharded = False

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
            master = self._seed_master(g, from_fp32_source=True)
            if SPLIT_MASTER:
                lo = split_master(master, g["param"])  # the working copy becomes the master's (half-up rounded) upper half
                self.state[g["name"]] = dict(lo=lo, m=torch.zeros_like(master), v=torch.zeros_like(master))
                del master
            else:
                self.state[g["name"]] = dict(master=master, m=torch.zeros_like(master), v=torch.zeros_like(master))
        self._partials = torch.empty(max(1, len(self.param_groups)) * ops.sumsq_blocks(), dtype=F32, device=model.device)
        self._norm_coef = torch.ones(2, dtype=F32, device=model.device)
        # the model tells its optimisers when somebody rewrites the working weights behind their back (params_rewritten)
        import weakref
        if not hasattr(model, "_optimizers"):
            model._optimizers = []
        model._optimizers.append(weakref.ref(self))

    def _seed_master(self, g, from_fp32_source: bool = False) -> torch.Tensor:
        return seed_master(self.model, g["offset"], g["numel"], g["param"], from_fp32_source and g["name"] != "lora")

    def named_master(self):
        """(reference parameter name, fp32 master view) for every trainable base parameter."""
        m = self.model
        views, base = m._views(m.arena), m.arena.storage_offset()
        for g in self.param_groups:
            if g["name"] == "lora":
                continue
            master = self.master(g["name"])
            for k, v in views.items():
                so = v.storage_offset() - base
                if g["offset"] <= so < g["offset"] + g["numel"]:
                    yield k, torch.as_strided(master, v.size(), v.stride(), so - g["offset"])

    def _group(self, name):
        return next(g for g in self.param_groups if g["name"] == name)

    def master(self, name: str) -> torch.Tensor:
        """fp32 master weights of a group (a fresh tensor when the master is stored split)."""
        st = self.state[name]
        return st["master"] if "master" in st else join_master(self._group(name)["param"], st["lo"])

    def set_master(self, name: str, master: torch.Tensor):
        """Overwrite a group's master weights (and with them the bf16 working copy)."""
        st, g = self.state[name], self._group(name)
        if "master" in st:
            st["master"].copy_(master)
            g["param"].copy_(st["master"])
        else:
            st["lo"].copy_(split_master(master.to(device=g["param"].device, dtype=F32), g["param"]))

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

    def step(self, zero_grad=False, skip: Optional[torch.Tensor] = None):
        """One AdamW update.  ``zero_grad=True`` clears the gradients in the same pass over memory.  ``zero_grad="lazy"``
        (what the trainers use) clears only the embedding tables, whose backward is a scatter-add that needs zeros
        underneath; the dense groups are marked stale instead and the next backward OVERWRITES them - their weight-gradient
        GEMMs then skip the read of the old value and nobody writes 2 bytes per parameter of zeros.

        ``skip``: a device int32 tensor; non-zero means "drop this step" (weights and moments untouched, gradients cleared
        as ``zero_grad`` says).  The decision stays on the device - it travels to the kernels as a negative clip coefficient -
        so the host launches the same work either way (training/dp.py: text-row exchange over capacity).  The caller
        un-counts a dropped step with ``uncount_step`` once the host has seen the flag.  (Known deviation, ADVICE r03: the host
        sees the flag of step k at ``arm()`` of step k + 2, so step k + 1 runs its bias correction with t one too high - a
        factor (1 - beta^t) / (1 - beta^(t+1)) on one step.  Steps are only ever dropped while the text-row exchange capacity
        grows, i.e. in the first steps of a run; the counter stays on the host so that the kernels' bias corrections remain the
        host-computed doubles torch's AdamW uses.)"""
        self._settle()
        self.step_count += 1
        self.model._fp32_source = None          # the weights move on: the loaded fp32 dict no longer describes them
        if skip is not None:
            coef = self._coef[1] if self._coef is not None else torch.ones((), dtype=F32, device=skip.device)
            self._norm_coef[1] = torch.where(skip.reshape(()) != 0, torch.full_like(coef, -1.0), coef)
            self._coef = self._norm_coef
        b1, b2 = self.betas
        gs = self.model.grad_state
        for g in self.param_groups:
            st = self.state[g["name"]]
            name = g["name"]
            zero = bool(zero_grad) if zero_grad != "lazy" else (name == "embeddings" or name not in gs)
            if "lo" in st:
                ops.adamw_step_split(st["lo"], st["m"], st["v"], g["param"], g["grad"], g["lr"], b1, b2, self.eps,
                                     g["weight_decay"], self.step_count, self._coef, zero_grad=zero)
            else:
                ops.adamw_step(st["master"], st["m"], st["v"], g["param"], g["grad"], g["lr"], b1, b2, self.eps,
                               g["weight_decay"], self.step_count, self._coef, zero_grad=zero)
            if name in gs:
                if zero:
                    gs[name] = "zero"
                elif zero_grad == "lazy" and gs[name] == "live":
                    gs[name] = "stale"
        self._coef = None

    def uncount_step(self):
        """A step that was dropped on the device (``step(skip=...)``) does not count towards the bias correction."""
        self.step_count = max(0, self.step_count - 1)

    def params_rewritten(self, base: bool = True, lora: bool = True, from_fp32_source: bool = False):
        """The bf16 working weights were overwritten from outside (``Model.load_state_dict``, ``broadcast_parameters``,
        ``merge_lora_weights``): make them the master again.  With the split master the working copy IS the master's upper half,
        so a stale ``lo`` would silently shift every weight by up to one bf16 ulp; re-seeding makes master == the new weights.
        Only ``Model.load_state_dict`` passes ``from_fp32_source`` (master at fp32 where the loaded state dict was fp32); for a
        merge / broadcast / restore the new bf16 weights themselves are the master.  Moments are kept."""
        for g in self.param_groups:
            if lora if g["name"] == "lora" else base:
                self.set_master(g["name"], self._seed_master(g, from_fp32_source))

    def state_dict(self):
        return {"step": self.step_count,
                "groups": [{k: g[k] for k in ("name", "lr", "weight_decay", "offset", "numel")} for g in self.param_groups],
                # (fp32 masters whatever the in-memory form: the checkpoint format does not depend on CSM_ADAMW_SPLIT)
                "state": {n: {"master": self.master(n).cpu(), "m": st["m"].cpu(), "v": st["v"].cpu()} for n, st in self.state.items()}}

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
            for k, v in st.items():
                if k == "master":
                    self.set_master(n, v)
                else:
                    self.state[n][k].copy_(v)
        for g in self.param_groups:
            saved = saved_by_name[g["name"]]
            g["lr"], g["weight_decay"] = saved["lr"], saved["weight_decay"]
