"""Data parallelism for the train step: one process per GPU, bf16 gradient all-reduce over RCCL/xGMI, bucketed per
layer and overlapped with the rest of the backward on a side stream.  New capability - the reference has no
distributed code; the schedule is designed for MI355X's point-to-point xGMI mesh rather than copied
from anything: a backbone layer's gradients are one contiguous 122 MB slice of the arena, which is already a good
ring-collective payload per launch, so buckets are arena slices (no flatten / copy-in / copy-out).

The sum (not mean) is reduced; callers pre-scale the loss gradient by 1 / world_size so the result is the mean.
Device-agnostic on purpose: with the ``gloo`` backend and CPU tensors the same bucket logic is unit-tested.

The text-embedding gradient is the exception to "all-reduce the slice": it is the largest tensor (128256 x 2048 = 525 MB
in bf16), it is final only when the whole backward is (nothing left to hide its all-reduce behind), and a step touches a
few hundred of its rows.  On the GPU the ranks therefore exchange (row id, gradient row) lists with an all-gather of a few
MB and add them in rank order - every rank ends up with bit-identical gradients, as after an all-reduce.
"""
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Read RANK / WORLD_SIZE / LOCAL_RANK (torchrun) and create the process group.  Returns (rank, world, local)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL buffer sharing); too late if HIP is already up
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            init_nccl(rank, world, local)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def init_nccl(rank: int, world: int, local: int):
    """RCCL process group bound to this rank's GPU, its collectives on a high-priority stream: the all-reduce's few
    workgroups should get CUs as GEMM workgroups retire instead of queueing behind the whole backward."""
    kw = dict(rank=rank, world_size=world, device_id=torch.device("cuda", local))
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        dist.init_process_group("nccl", pg_options=opts, **kw)
    except (AttributeError, TypeError):
        dist.init_process_group("nccl", **kw)


class GradSync:
    """All-reduces slices of a flat gradient tensor as they become final.

    ``buckets``: ordered mapping  ready-key -> list of (offset, numel) slices.  ``on_ready(key...)`` launches the
    collective for that key on the communication stream (after the compute stream's work so far); ``finish()``
    launches whatever was not announced and makes the compute stream wait for all of it.
    """

    def __init__(self, flat_grad: torch.Tensor, buckets: Dict[tuple, List[Tuple[int, int]]], group=None,
                 extra: Optional[List[torch.Tensor]] = None):
        self.flat, self.buckets, self.group = flat_grad, buckets, group
        self.extra = extra or []
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = flat_grad.is_cuda
        # high priority: the collective's few workgroups should get CUs as soon as GEMM workgroups retire, not queue behind them
        self.comm_stream = torch.cuda.Stream(priority=-1) if self.cuda else None
        self.armed = False
        self.done = set()
        self.handles = []
        self.launch_log: List[tuple] = []
        self.sparse: Optional[dict] = None        # text-embedding rows exchanged as lists (set by for_model on the GPU)

    @staticmethod
    def active() -> bool:
        return dist.is_initialized() and dist.get_world_size() > 1

    @classmethod
    def for_model(cls, model, group=None) -> "GradSync":
        """Buckets = one per transformer layer (in backward order they complete back to front), plus the final norms,
        the head group and the embeddings.  Registers itself as the engine's gradient-ready hook."""
        model.ensure_grads()
        slots = model._slots
        buckets: Dict[tuple, List[Tuple[int, int]]] = {}
        for name, s in slots.items():
            if not model.trainable.get(s.group, False):
                continue
            parts = name.split(".")
            if s.group in ("backbone", "decoder") and parts[1] == "layers":
                key = (s.group, int(parts[2]))
            elif s.group in ("backbone", "decoder"):
                key = (s.group, model.bb.num_layers - 1 if s.group == "backbone" else model.dc.num_layers - 1)  # final norm: first done
            elif s.group == "embeddings":
                key = ("embeddings", -1)
            else:
                key = ("other", -1)
            buckets.setdefault(key, []).append((s.offset, s.numel))
        sparse = None
        if ("embeddings", -1) in buckets and model.grad_arena.is_cuda and os.environ.get("CSM_DP_DENSE_EMBEDDINGS") != "1":
            t = slots["text_embeddings.weight"]
            buckets[("embeddings", -1)] = [sl for sl in buckets[("embeddings", -1)] if sl != (t.offset, t.numel)]
            sparse = dict(slice=(t.offset, t.numel), D=model.bb.embed_dim, seen=[], rows=None, counts=None)
        merged = {k: cls._merge(v) for k, v in buckets.items()}
        extra = [model.lora.grad_arena] if model.lora is not None else []
        gs = cls(model.grad_arena, merged, group, extra)
        gs.sparse = sparse
        model.engine.grad_hook = gs.on_ready
        return gs

    @staticmethod
    def _merge(slices: List[Tuple[int, int]]) -> List[Tuple[int, int]]:
        """Coalesce adjacent slices (allowing the <=63-element alignment gaps of the arena, which hold zeros)."""
        out: List[Tuple[int, int]] = []
        for off, n in sorted(slices):
            if out and off - (out[-1][0] + out[-1][1]) < 64:
                out[-1] = (out[-1][0], off + n - out[-1][0])
            else:
                out.append((off, n))
        return out

    SPARSE_MAX_ROWS = 16384       # beyond this many touched text rows per rank the dense all-reduce is used instead

    def note_batch(self, tokens: torch.Tensor, masks: torch.Tensor):
        """Remember which text-embedding rows this micro-batch touches (column K of the live text slots)."""
        if self.sparse is None:
            return
        k = tokens.shape[-1] - 1
        t, mk = tokens[..., k].reshape(-1), masks[..., k].reshape(-1).bool()
        self.sparse["seen"].append(t[mk].to(self.flat.device, torch.int64))

    def arm(self, enabled: bool = True):
        """Call before a backward: ``enabled`` only on the micro-batch that ends an accumulation window."""
        self.armed = enabled and self.world_size > 1
        self.done = set()
        self.handles = []
        if self.armed and self.sparse is not None:
            # union of the window's text rows and every rank's count, exchanged now - before the backward - so that the
            # one host sync this needs does not sit at the end of the step behind all the queued collectives
            sp = self.sparse
            rows = torch.unique(torch.cat(sp["seen"])) if sp["seen"] else torch.empty(0, dtype=torch.int64, device=self.flat.device)
            sp["seen"] = []
            cnt = torch.tensor([rows.numel()], dtype=torch.int64, device=self.flat.device)
            got = [torch.zeros_like(cnt) for _ in range(self.world_size)]
            dist.all_gather(got, cnt, group=self.group)
            sp["rows"], sp["counts"] = rows.to(torch.int32), [int(x) for x in got]

    def _launch_text_rows(self):
        """All-gather (row id, gradient row) of the text-embedding rows each rank touched and add them in rank order."""
        sp = self.sparse
        o, n = sp["slice"]
        counts, D = sp["counts"], sp["D"]
        if max(counts) > self.SPARSE_MAX_ROWS:
            self._launch([self.flat[o:o + n]], ("embeddings", "text-dense"))
            return
        from ..hip import ops
        self.launch_log.append(("embeddings", "text-rows"))
        cap = max(64, (max(counts) + 63) // 64 * 64)
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            g = self.flat[o:o + n].view(-1, D)
            rows, mine = sp["rows"], sp["rows"].numel()
            send_r = torch.full((cap,), -1, dtype=torch.int32, device=g.device)
            send_g = torch.zeros(cap, D, dtype=g.dtype, device=g.device)
            if mine:
                send_r[:mine] = rows
                send_g[:mine] = g[rows.long()]
            all_r = [torch.empty_like(send_r) for _ in range(self.world_size)]
            all_g = [torch.empty_like(send_g) for _ in range(self.world_size)]
            dist.all_gather(all_r, send_r, group=self.group)
            dist.all_gather(all_g, send_g, group=self.group)
            if mine:
                g.index_fill_(0, rows.long(), 0)
            for r in range(self.world_size):                      # fixed order: every rank computes the same bits
                if counts[r]:
                    ops.rows_add_bf16(g, all_r[r][:counts[r]].contiguous(), all_g[r], 1)

    def _launch(self, tensors: List[torch.Tensor], key):
        self.launch_log.append(key)
        if not tensors:
            return
        if self.cuda:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            for t in tensors:
                self.handles.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def on_ready(self, group: str, layer: int):
        if not self.armed:
            return
        key = (group, layer)
        if key in self.buckets and key not in self.done:
            self.done.add(key)
            self._launch([self.flat[o:o + n] for o, n in self.buckets[key]], key)
            if key == ("embeddings", -1) and self.sparse is not None:
                self._launch_text_rows()

    def finish(self):
        """Reduce anything not yet announced, then make the compute stream wait for the communication stream."""
        if not self.armed:
            return
        for key in self.buckets:
            if key not in self.done:
                self.done.add(key)
                self._launch([self.flat[o:o + n] for o, n in self.buckets[key]], key)
                if key == ("embeddings", -1) and self.sparse is not None:
                    self._launch_text_rows()
        if self.extra:
            self._launch(list(self.extra), ("extra", -1))
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for h in self.handles:
                h.wait()
        self.armed = False
