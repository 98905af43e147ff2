"""ZeRO-1 optimiser for data-parallel full-parameter training (SURVEY 8e's alternative exchange; new capability - the
reference has no distributed code): every rank keeps the fp32 master / m / v of, and runs AdamW on, only ITS shards of the
parameter arena (``GradSync.pieces``, training/dp.py), whose summed gradients arrive by reduce-scatter; the updated bf16
shards are all-gathered in place behind the step (``GradSync.gather_params``), overlapped with the next forward.

At N ranks the optimiser pass (26 B/param of HBM traffic, 7 ms of a 64 ms step at CSM-1B) and the optimiser state (12 B/param)
shrink to 1/N per rank; the bytes on the wire are those of the all-reduce.  The update itself is the same kernel on the same
values (``csm_adamw_step_split``), element-wise - so with identical reduced gradients the parameters are bit-identical to the
all-reduce path's.  Replicated pieces (the text-embedding table with its row-list exchange, tails too small to cut) are updated
by every rank identically.  Global-norm clipping: each rank sums the squares of its shards, ONE scalar all-reduce adds them
up, the replicated pieces are added locally (same values everywhere) - every rank gets the same coefficient.

Checkpoints keep ``FusedAdamW``'s format (fp32 master / m / v per learning-rate group): ``state_dict`` is a collective that
gathers the shards (tensors materialise on rank 0 only), ``load_state_dict`` takes each rank's shards out of the full state.
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from ..hip import ops
from .optim import F32, join_master, seed_master, split_master


class ZeroAdamW:
    sharded = True

    def __init__(self, model, group_lrs: Dict[str, float], sync, weight_decay: float = 0.01, betas=(0.9, 0.999),
                 eps: float = 1e-8):
        if not sync.zero1 or not sync.pieces:
            raise ValueError("ZeroAdamW needs a GradSync built with zero1=True (GradSync.for_model(model, zero1=True))")
        self.model, self.sync = model, sync
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self._coef: Optional[torch.Tensor] = None
        self.param_groups: List[dict] = []          # one per piece (name = "<group>@<arena offset>")
        self.state: Dict[str, Dict[str, torch.Tensor]] = {}
        self.group_lrs = {k: float(v) for k, v in group_lrs.items() if v is not None and model.trainable.get(k, False)}
        model.ensure_grads()
        # sharded pieces first: their sums of squares are the part of the norm that needs the all-reduce
        for p in sorted(sync.pieces, key=lambda p: (p.chunk == 0, p.off)):
            if p.group not in self.group_lrs:
                continue
            g = dict(name=f"{p.group}@{p.my_off}", group=p.group, lr=self.group_lrs[p.group], weight_decay=weight_decay,
                     offset=p.my_off, numel=p.my_n, sharded=p.chunk > 0, piece=p,
                     param=model.arena[p.my_off:p.my_off + p.my_n], grad=p.grad)
            if g["numel"] % 8:
                raise ValueError(f"ZeRO-1 piece {g['name']}: {g['numel']} elements is not a multiple of 8")
            self.param_groups.append(g)
            master = seed_master(model, g["offset"], g["numel"], g["param"], from_fp32_source=True)
            lo = split_master(master, g["param"])
            self.state[g["name"]] = dict(lo=lo, m=torch.zeros_like(master), v=torch.zeros_like(master))
        self.n_sharded = sum(1 for g in self.param_groups if g["sharded"])
        nb = ops.sumsq_blocks()
        dev = model.device
        self._partials = torch.zeros(max(1, len(self.param_groups)) * nb, dtype=F32, device=dev)
        self._norm_in = torch.zeros(1 + (len(self.param_groups) - self.n_sharded) * nb, dtype=F32, device=dev)
        self._norm_coef = torch.ones(2, dtype=F32, device=dev)
        import weakref
        if not hasattr(model, "_optimizers"):
            model._optimizers = []
        model._optimizers.append(weakref.ref(self))
        sync.gather_params()        # the masters' rounded upper halves are now the working weights of every rank's shards
        sync.wait_params(None, None)

    # ------------------------------------------------------------------ bookkeeping shared with FusedAdamW's interface
    def num_trainable(self) -> int:
        """Parameters this job trains (the whole model's, not this rank's share)."""
        return sum(p.n for p in self.sync.pieces if p.group in self.group_lrs)

    def num_owned(self) -> int:
        """Parameters THIS rank updates per step (its shards + the replicated pieces)."""
        return sum(g["numel"] for g in self.param_groups)

    def uncount_step(self):
        self.step_count = max(0, self.step_count - 1)

    def _arena_ranges(self, group: str):
        """The arena slices of a learning-rate group that carry gradients (whole slices, all ranks' shards)."""
        return [(p.off, p.n) for p in self.sync.pieces if p.group == group]

    def zero_grad(self, set_to_none: bool = False):
        for grp in self.group_lrs:
            o, n = self.model.group_range(grp)
            self.model.grad_arena[o:o + n].zero_()
            self.model.grad_state[grp] = "zero"
        self._coef = None

    def _settle(self):
        gs = self.model.grad_state
        for grp in self.group_lrs:
            if gs.get(grp) == "stale":
                o, n = self.model.group_range(grp)
                self.model.grad_arena[o:o + n].zero_()
                gs[grp] = "zero"

    # ------------------------------------------------------------------ clip + step
    def clip_grad_norm(self, max_norm: float) -> torch.Tensor:
        """Global L2 norm of the REDUCED gradients: sum over this rank's shards, one scalar all-reduce over the ranks, plus the
        replicated pieces (identical on every rank, counted once).  The coefficient is consumed by the next ``step``."""
        nb = ops.sumsq_blocks()
        for i, g in enumerate(self.param_groups):
            ops.sumsq_bf16(g["grad"], self._partials[i * nb:(i + 1) * nb])
        ns = self.n_sharded
        tot = self._partials[:ns * nb].sum().reshape(1) if ns else torch.zeros(1, dtype=F32, device=self._partials.device)
        if dist.is_initialized() and self.sync.world_size > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.sync.group)
        self._norm_in[:1] = tot
        self._norm_in[1:] = self._partials[ns * nb:len(self.param_groups) * nb]
        ops.clip_coef(self._norm_in, max_norm, self._norm_coef)
        self._coef = self._norm_coef
        return self._norm_coef[0]

    def step(self, zero_grad=False, skip: Optional[torch.Tensor] = None):
        """One AdamW update of this rank's pieces, then the parameter all-gather (asynchronous: the next forward waits bucket
        by bucket).  ``zero_grad`` / ``skip`` as in ``FusedAdamW.step``; gradients of other ranks' shards live in the arena only
        as the reduce-scatter's input, so clearing is done on the arena by group, not inside the kernels."""
        self.step_count += 1
        self.model._fp32_source = None
        if skip is not None:
            coef = self._coef[1] if self._coef is not None else torch.ones((), dtype=F32, device=skip.device)
            self._norm_coef[1] = torch.where(skip.reshape(()) != 0, torch.full_like(coef, -1.0), coef)
            self._coef = self._norm_coef
        b1, b2 = self.betas
        for g in self.param_groups:
            st = self.state[g["name"]]
            ops.adamw_step_split(st["lo"], st["m"], st["v"], g["param"], g["grad"], g["lr"], b1, b2, self.eps,
                                 g["weight_decay"], self.step_count, self._coef, zero_grad=False)
        gs = self.model.grad_state
        for grp in self.group_lrs:
            zero = bool(zero_grad) if zero_grad != "lazy" else (grp == "embeddings" or grp not in gs)
            if zero:
                o, n = self.model.group_range(grp)
                self.model.grad_arena[o:o + n].zero_()
                gs[grp] = "zero"
            elif zero_grad == "lazy" and gs.get(grp) == "live":
                gs[grp] = "stale"
        self._coef = None
        self.sync.gather_params()
        if getattr(self.model.engine, "param_hook", None) != self.sync.wait_params:
            self.sync.wait_params(None, None)        # nobody will wait bucket by bucket (exchange detached): wait here

    # ------------------------------------------------------------------ masters
    def master(self, name: str) -> torch.Tensor:
        """fp32 master of one piece (by its state key)."""
        g = next(g for g in self.param_groups if g["name"] == name)
        return join_master(g["param"], self.state[name]["lo"])

    def set_master(self, name: str, master: torch.Tensor):
        g = next(g for g in self.param_groups if g["name"] == name)
        self.state[name]["lo"].copy_(split_master(master.to(device=g["param"].device, dtype=F32), g["param"]))

    def params_rewritten(self, base: bool = True, lora: bool = True, from_fp32_source: bool = False):
        """The working weights were overwritten from outside (load_state_dict, broadcast): they are the master again."""
        if not base:
            return
        self.sync.wait_params(None, None)
        for g in self.param_groups:
            self.set_master(g["name"], seed_master(self.model, g["offset"], g["numel"], g["param"], from_fp32_source))
        self.sync.gather_params()
        self.sync.wait_params(None, None)

    # ------------------------------------------------------------------ checkpoints (FusedAdamW's format)
    def _full(self, group: str, kind: str) -> torch.Tensor:
        """The whole learning-rate group's fp32 ``kind`` (master / m / v) on this rank's GPU, shards gathered."""
        o, n = self.model.group_range(group)
        full = self.model.arena[o:o + n].float() if kind == "master" else torch.zeros(n, dtype=F32, device=self.model.device)
        world, rank, grp_ = self.sync.world_size, self.sync.rank, self.sync.group
        for g in self.param_groups:
            if g["group"] != group:
                continue
            mine = self.master(g["name"]) if kind == "master" else self.state[g["name"]][kind]
            p = g["piece"]
            if not g["sharded"] or world == 1:
                full[p.off - o:p.off - o + p.n] = mine
                continue
            dst = full[p.off - o:p.off - o + p.n]
            if dist.get_backend(grp_) == "nccl":
                dist.all_gather_into_tensor(dst, mine.contiguous(), group=grp_)
            else:
                dist.all_gather([dst[r * p.chunk:(r + 1) * p.chunk] for r in range(world)], mine.contiguous(), group=grp_)
        return full

    def state_dict(self):
        """COLLECTIVE (every rank must call it).  Same dict as ``FusedAdamW.state_dict``; the tensors are materialised on
        rank 0 only (the others get ``None`` entries and must not write the checkpoint)."""
        self.sync.wait_params(None, None)
        keep = self.sync.rank == 0
        state = {}
        for grp in self.group_lrs:
            state[grp] = {}
            for kind in ("master", "m", "v"):
                t = self._full(grp, kind)
                state[grp][kind] = t.cpu() if keep else None
                del t
        groups = []
        for grp, lr in self.group_lrs.items():
            o, n = self.model.group_range(grp)
            wd = next(g["weight_decay"] for g in self.param_groups if g["group"] == grp)
            groups.append(dict(name=grp, lr=lr, weight_decay=wd, offset=o, numel=n))
        return {"step": self.step_count, "groups": groups, "state": state}

    def load_state_dict(self, sd):
        saved = {g["name"]: g for g in sd["groups"]}
        if sorted(saved) != sorted(self.group_lrs):
            raise ValueError(f"optimizer state was saved for parameter groups {sorted(saved)} but this optimizer has "
                             f"{sorted(self.group_lrs)} (different freeze flags?)")
        self.sync.wait_params(None, None)
        self.step_count = sd["step"]
        for g in self.param_groups:
            sv = saved[g["group"]]
            o, n = self.model.group_range(g["group"])
            if (sv["offset"], sv["numel"]) != (o, n):
                raise ValueError(f"optimizer group {g['group']!r}: saved range {sv['offset']}+{sv['numel']} != {o}+{n}")
            lo_, hi_ = g["offset"] - o, g["offset"] - o + g["numel"]
            st = sd["state"][g["group"]]
            self.set_master(g["name"], st["master"][lo_:hi_])
            self.state[g["name"]]["m"].copy_(st["m"][lo_:hi_])
            self.state[g["name"]]["v"].copy_(st["v"][lo_:hi_])
            g["lr"], g["weight_decay"] = sv["lr"], sv["weight_decay"]
        for grp in self.group_lrs:
            self.group_lrs[grp] = saved[grp]["lr"]
        self.sync.gather_params()
        self.sync.wait_params(None, None)
