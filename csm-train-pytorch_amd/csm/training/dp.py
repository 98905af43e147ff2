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

ZeRO-1 form (SURVEY 8e's alternative, ``CSM_DP_ZERO1=1`` / ``for_model(..., zero1=True)``): the same buckets, but each
bucket slice is cut into ``world`` equal shards; the gradient exchange is a reduce-scatter (rank r receives the sum of shard
r), every rank runs AdamW on its own shards only (``training/zero.py``: 1/world of the 26 B/param optimiser pass and of the
optimiser state) and the updated bf16 parameters are all-gathered in place, bucket by bucket in the order the next forward
uses them, on the communication stream - the forward waits per bucket (``Engine.param_hook``), so the gather hides behind
it.  Same bytes on the wire as the all-reduce (which is a reduce-scatter + all-gather of gradients).  The text-embedding
rows keep their list exchange and stay replicated, like every slice too small to cut.
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
    opts = None
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
    except AttributeError as e:      # a torch build without the option: say so, do not silently change the schedule
        import warnings
        warnings.warn(f"RCCL process group WITHOUT a high-priority stream ({e!r}): gradient all-reduces will queue behind "
                      "the backward's GEMMs and the exposed communication time grows", RuntimeWarning, stacklevel=2)
    if opts is not None:
        dist.init_process_group("nccl", pg_options=opts, **kw)
    else:
        dist.init_process_group("nccl", **kw)


class Piece:
    """ZeRO-1: one contiguous run of the arenas as the optimiser sees it.  ``chunk > 0``: slice [off, off + n) is cut into
    ``world`` shards of ``chunk`` elements and this rank owns [my_off, my_off + my_n); its reduced gradient arrives in ``grad``
    (a segment of the shard buffer).  ``chunk == 0``: replicated - every rank holds the whole reduced gradient (``grad`` is the
    arena view) and applies the same update."""
    __slots__ = ("key", "off", "n", "chunk", "my_off", "my_n", "grad")

    def __init__(self, key, off, n, chunk, my_off, my_n):
        self.key, self.off, self.n, self.chunk, self.my_off, self.my_n = key, off, n, chunk, my_off, my_n
        self.grad = None

    @property
    def group(self) -> str:
        return self.key[0]


class GradSync:
    """All-reduces slices of a flat gradient tensor as they become final.

    ``buckets``: ordered mapping  ready-key -> list of (offset, numel) slices.  ``on_ready(key...)`` launches the
    collective for that key on the communication stream (after the compute stream's work so far); ``finish()``
    launches whatever was not announced and makes the compute stream wait for all of it.
    """

    def __init__(self, flat_grad: torch.Tensor, buckets: Dict[tuple, List[Tuple[int, int]]], group=None,
                 extra: Optional[List[torch.Tensor]] = None, zero1: bool = False, flat_param: Optional[torch.Tensor] = None):
        self.flat, self.buckets, self.group = flat_grad, buckets, group
        self.extra = extra or []
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.zero1 = bool(zero1)
        self.flat_param = flat_param                 # ZeRO-1: the parameter arena the updated shards are gathered into
        self.pieces: List["Piece"] = []              # ZeRO-1: what this rank's optimiser owns (plan_shards)
        self.shard_grad: Optional[torch.Tensor] = None
        self._param_events: Dict[tuple, object] = {} # ZeRO-1: bucket key -> event after its parameter all-gather
        self.param_gather_ms: List[float] = []
        self._gather_events: List[tuple] = []
        self.cuda = flat_grad.is_cuda
        # high priority: the collective's few workgroups should get CUs as soon as GEMM workgroups retire, not queue behind them
        self.comm_stream = torch.cuda.Stream(priority=-1) if self.cuda else None
        self.armed = False
        self.done = set()
        self.handles = []
        self.launch_log: List[tuple] = []
        self.sparse: Optional[dict] = None        # text-embedding rows exchanged as lists (set by for_model on the GPU)
        self.timing = False                       # record how long the compute stream waits for the collectives (bench.py)
        self.exposed_ms: List[float] = []
        self._wait_events: List[tuple] = []
        self.opt_step = 0                         # optimiser steps whose collectives have been launched
        self.skipped_steps: List[int] = []        # optimiser steps dropped on every rank (text-row exchange over capacity)
        self.on_skip: List[Callable[[int], None]] = []   # called with the step index once the host has seen the flag
        self._restore_persistent: Optional[int] = None

    @staticmethod
    def active() -> bool:
        return dist.is_initialized() and dist.get_world_size() > 1

    @classmethod
    def for_model(cls, model, group=None, zero1: Optional[bool] = None) -> "GradSync":
        """Buckets = one per transformer layer (in backward order they complete back to front), plus the final norms,
        the head group and the embeddings.  Registers itself as the engine's gradient-ready hook.  ``zero1`` (default: the
        ``CSM_DP_ZERO1`` switch; never with LoRA, whose gradients are 2 MB): reduce-scatter + sharded optimiser + parameter
        all-gather instead of the all-reduce - build the optimiser with ``training.zero.ZeroAdamW(model, lrs, sync)``."""
        if zero1 is None:
            zero1 = os.environ.get("CSM_DP_ZERO1", "0") == "1"
        zero1 = bool(zero1) and model.lora is None
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
            cap = int(os.environ.get("CSM_DP_TEXT_ROWS_CAP", cls.TEXT_ROWS_CAP))
            # ``host``: a ring of pinned int32 slots, allocated ONCE (a pinned allocation is a driver call that can serialise with
            # in-flight work; it must not happen inside the overlap window of every step).  Step k writes slot k % len; the
            # host has read step k - OVERFLOW_LAG's slot before step k launches (arm), so OVERFLOW_LAG + 2 slots never collide.
            sparse = dict(slice=(t.offset, t.numel), D=model.bb.embed_dim, seen=[], cap=cap, n_rows=t.shape[0],
                          overflow=torch.zeros(1, dtype=torch.int32, device=model.grad_arena.device),
                          host=torch.zeros(cls.OVERFLOW_LAG + 2, dtype=torch.int32).pin_memory(),
                          pending=[])      # (optimiser step, event, slot of ``host``) per exchange, oldest first
        merged = {k: cls._merge(v) for k, v in buckets.items()}
        extra = [model.lora.grad_arena] if model.lora is not None else []
        gs = cls(model.grad_arena, merged, group, extra, zero1=zero1, flat_param=model.arena)
        gs.sparse = sparse
        gs._model = model
        if zero1:
            gs.plan_shards()
        gs.attach()
        return gs

    def attach(self):
        """Hook into the model's engine (gradient-ready and, for ZeRO-1, parameter-needed callbacks) and switch the GEMM
        schedule for shared CUs; ``close`` undoes both, a later ``attach`` re-does them (trainers call train() repeatedly)."""
        model, gs = self._model, self
        if gs.world_size > 1 and model.grad_arena.is_cuda and "CSM_GEMM256_PERSISTENT" not in os.environ \
                and self._restore_persistent is None:
            # The collectives' workgroups hold a few CUs while the backward runs.  A persistent GEMM launches one workgroup per CU
            # with a fixed tile list each, so the workgroups that find their CU taken would run their whole list late; one tile
            # per workgroup lets the dispatcher pack the remaining CUs instead.  (Single-GPU runs keep the persistent form.)
            from ..hip import lib
            import logging
            gs._restore_persistent = lib.csm_get_gemm256_persistent()
            lib.csm_set_gemm256_persistent(0)
            logging.getLogger("csm_trainer").info(
                "data parallel: 256x256 GEMM switched from persistent workgroups to one tile per workgroup while gradient "
                "collectives share the CUs (CSM_GEMM256_PERSISTENT pins it; GradSync.close() restores the previous setting)")
        model.engine.grad_hook = gs.on_ready
        if self.zero1:
            model.engine.param_hook = gs.wait_params

    def close(self):
        """End of data-parallel training in this process: detach from the engine, look at every outstanding overflow flag and
        give the process-global GEMM schedule switch back (a later single-GPU model in the same process must not inherit it)."""
        self.check_overflow(block=True)
        m = getattr(self, "_model", None)
        if m is not None and getattr(m.engine, "grad_hook", None) == self.on_ready:
            m.engine.grad_hook = None
        if self.zero1:
            self.wait_params(None, None)
            if m is not None and getattr(m.engine, "param_hook", None) == self.wait_params:
                m.engine.param_hook = None
        if self._restore_persistent is not None:
            from ..hip import lib
            lib.csm_set_gemm256_persistent(self._restore_persistent)
            self._restore_persistent = None

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

    TEXT_ROWS_CAP = 2048          # rows of the text-embedding gradient a rank may touch per optimiser step (CSM_DP_TEXT_ROWS_CAP)

    def note_batch(self, tokens: torch.Tensor, masks: torch.Tensor):
        """Remember which text-embedding rows this micro-batch touches (column K of the live text slots).  No host
        sync: masked-out slots become a sentinel id and the sizes stay static."""
        if self.sparse is None:
            return
        sp = self.sparse
        k = tokens.shape[-1] - 1
        t = tokens[..., k].reshape(-1).to(self.flat.device, torch.int64, non_blocking=True)
        mk = masks[..., k].reshape(-1).to(self.flat.device, non_blocking=True).bool()
        sp["seen"].append(torch.where(mk, t, torch.full_like(t, sp["n_rows"])))

    OVERFLOW_LAG = 2      # optimiser steps between an exchange and the host's look at its flag (see check_overflow)

    def arm(self, enabled: bool = True):
        """Call before a backward: ``enabled`` only on the micro-batch that ends an accumulation window."""
        self.armed = enabled and self.world_size > 1
        self.done = set()
        self.handles = []
        self.check_overflow(block=False)

    def skip_flag(self) -> Optional[torch.Tensor]:
        """Device int32[1]: non-zero when the step whose collectives just finished must be dropped (the same value on every
        rank).  Valid on the compute stream after ``finish()``; hand it to ``FusedAdamW.step(skip=...)``."""
        return self.sparse["overflow"] if self.sparse is not None else None

    def check_overflow(self, block: bool):
        """Look at the overflow flags of past text-row exchanges.

        The flag of step k is all-reduced (MAX) on the device, applied on the device (the optimiser step k is dropped on every
        rank: ``skip_flag``), and copied to its own pinned slot - never overwritten, so no flag is lost however far the host
        runs ahead.  The host reads it at a FIXED distance: ``arm`` of step k + OVERFLOW_LAG waits for step k's copy (by then
        it completed long ago: the host never runs two steps ahead of the device, so the wait costs nothing), and
        ``block=True`` (before a checkpoint, at the end of an epoch, in ``close``) drains every outstanding one.  A fixed
        distance matters: the reaction - doubling the capacity - changes the size of the next all-gather and must happen at
        the same step on every rank; ``event.query()`` would make it depend on each host's timing."""
        sp = self.sparse
        if sp is None:
            return
        while sp["pending"] and (block or sp["pending"][0][0] <= self.opt_step - self.OVERFLOW_LAG):
            step, event, slot = sp["pending"].pop(0)
            event.synchronize()
            if int(sp["host"][slot]) > 0:
                self._on_overflow(step)

    def _on_overflow(self, step: int):
        import logging
        sp = self.sparse
        old = sp["cap"]
        sp["cap"] = min(sp["n_rows"], 2 * old)
        self.skipped_steps.append(step)
        for cb in self.on_skip:
            cb(step)
        logging.getLogger("csm_trainer").warning(
            f"data parallel: optimiser step {step} touched more than {old} distinct text-embedding rows on some rank; the step "
            f"was dropped on every rank (weights and moments untouched, replicas identical) and the exchange capacity is now "
            f"{sp['cap']} rows (CSM_DP_TEXT_ROWS_CAP sets the start value, CSM_DP_DENSE_EMBEDDINGS=1 the dense all-reduce)")

    def _launch_text_rows(self):
        """Fixed-capacity exchange of the touched text-embedding gradient rows: every rank sends ``cap`` (row id, gradient
        row) slots (id -1 = unused), all-gathers everybody's, and adds them in rank order - bit-identical tables on every
        rank, as after an all-reduce, for a few MB instead of 525.  Every size is static, so nothing here waits for the
        host: an overflow (more than ``cap`` distinct rows on some rank) raises a device flag that drops the optimiser step
        on every rank (``skip_flag``) and that the host reads two steps later (``check_overflow``)."""
        from ..hip import ops
        sp = self.sparse
        o, n = sp["slice"]
        D, cap, n_rows = sp["D"], sp["cap"], sp["n_rows"]
        self.launch_log.append(("embeddings", "text-rows"))
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            g = self.flat[o:o + n].view(-1, D)
            dev = g.device
            send_r = torch.full((cap + 1,), -1, dtype=torch.int32, device=dev)
            if sp["seen"]:
                for t in sp["seen"]:
                    # allocated on the compute stream, consumed here on the communication stream and dropped below: without
                    # this the caching allocator may hand the block to the next compute-stream allocation before the sort ran
                    t.record_stream(self.comm_stream)
                ids = torch.cat(sp["seen"]).sort().values                       # sentinel n_rows sorts last
                first = torch.ones_like(ids, dtype=torch.bool)
                first[1:] = ids[1:] != ids[:-1]
                first &= ids < n_rows
                pos = torch.cumsum(first, 0) - 1
                slot = torch.where(first & (pos < cap), pos, torch.full_like(pos, cap))   # overflow and duplicates -> scratch slot
                send_r.scatter_(0, slot, ids.to(torch.int32))
                sp["overflow"].copy_((first.sum() > cap).to(torch.int32).reshape(1))
            else:
                sp["overflow"].zero_()
            sp["seen"] = []
            send_r = send_r[:cap].contiguous()
            send_g = torch.empty(cap, D, dtype=g.dtype, device=dev)
            ops.rows_take_bf16(g, send_r, send_g)                                # lift own rows out of the table (zeroes them)
            all_r = torch.empty(self.world_size * cap, dtype=torch.int32, device=dev)
            all_g = torch.empty(self.world_size * cap, D, dtype=g.dtype, device=dev)
            self._all_gather(all_r, send_r)
            self._all_gather(all_g, send_g)
            for r in range(self.world_size):                                     # fixed order: every rank computes the same bits
                ops.rows_add_bf16(g, all_r[r * cap:(r + 1) * cap], all_g[r * cap:(r + 1) * cap], 1)
            dist.all_reduce(sp["overflow"], op=dist.ReduceOp.MAX, group=self.group)
            slot = self.opt_step % sp["host"].numel()                            # this step's own slot of the pinned ring
            if any(s_ == slot for _, _, s_ in sp["pending"]):                   # (cannot happen with arm() before every step)
                self.check_overflow(block=True)
            sp["host"][slot:slot + 1].copy_(sp["overflow"], non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            sp["pending"].append((self.opt_step, event, slot))

    def _all_gather(self, out: torch.Tensor, part: torch.Tensor):
        """out = concatenation over the ranks of ``part`` (one flat collective on RCCL; the list form where the backend -
        gloo in the one-GPU rehearsal tests - lacks it)."""
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, part, group=self.group)
        else:
            n = part.shape[0]
            dist.all_gather([out[r * n:(r + 1) * n] for r in range(self.world_size)], part, group=self.group)

    # ------------------------------------------------------------------ ZeRO-1: shard plan, reduce-scatter, parameter gather
    def plan_shards(self):
        """Cut every bucket slice into ``world`` equal shards of a multiple of 8 elements (the AdamW kernels' vector width);
        what does not divide (a tail of < 8 * world elements) and the text-embedding table (row-list exchange) stay
        replicated: every rank updates them identically.  ``pieces`` is the optimiser's work list, in arena order."""
        world, rank = self.world_size, self.rank
        pieces: List[Piece] = []
        for key, slices in self.buckets.items():
            for off, n in slices:
                if n % 8:
                    raise ValueError(f"ZeRO-1: bucket slice {key} [{off}, +{n}) is not a multiple of 8 elements")
                chunk = (n // (world * 8)) * 8
                if chunk:
                    pieces.append(Piece(key, off, world * chunk, chunk, off + rank * chunk, chunk))
                if n - world * chunk:
                    pieces.append(Piece(key, off + world * chunk, n - world * chunk, 0, off + world * chunk, n - world * chunk))
        if self.sparse is not None:
            o, n = self.sparse["slice"]
            pieces.append(Piece(("embeddings", "text-rows"), o, n, 0, o, n))
        pieces.sort(key=lambda p: p.off)
        total = sum(p.chunk for p in pieces)
        self.shard_grad = torch.zeros(max(total, 8), dtype=self.flat.dtype, device=self.flat.device)
        at = 0
        for p in pieces:
            if p.chunk:
                p.grad = self.shard_grad[at:at + p.chunk]
                at += p.chunk
            else:
                p.grad = self.flat[p.off:p.off + p.n]
        self.pieces = pieces

    def _launch_zero(self, key):
        """One bucket's gradients: reduce-scatter of every cut slice (rank r receives the sum of shard r into its shard
        buffer), all-reduce of the replicated tails."""
        self.launch_log.append(key)
        mine = [p for p in self.pieces if p.key == key]
        if not mine:
            return

        def run():
            for p in mine:
                src = self.flat[p.off:p.off + p.n]
                if not p.chunk:
                    dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.group)
                elif dist.get_backend(self.group) == "nccl":
                    dist.reduce_scatter_tensor(p.grad, src, op=dist.ReduceOp.SUM, group=self.group)
                else:       # gloo (CPU tests, the one-GPU rehearsal): no reduce-scatter; same result from an all-reduce
                    dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.group)
                    p.grad.copy_(src[self.rank * p.chunk:(self.rank + 1) * p.chunk])
        if self.cuda:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                run()
        else:
            run()

    FORWARD_ORDER = ("embeddings", "backbone", "other", "decoder")     # the order a forward first touches the groups

    def gather_params(self):
        """After the sharded optimiser step: all-gather every cut slice of the PARAMETER arena in place (each rank's shard is
        where it updated it), on the communication stream, bucket by bucket in the order the next forward needs them; an event
        per bucket lets the forward wait for exactly the layer it is about to run (``wait_params``)."""
        keys = sorted({p.key for p in self.pieces if p.chunk},
                      key=lambda k: (self.FORWARD_ORDER.index(k[0]) if k[0] in self.FORWARD_ORDER else 9, k[1] if isinstance(k[1], int) else 0))
        nccl = dist.get_backend(self.group) == "nccl"

        def run(key):
            for p in self.pieces:
                if p.key != key or not p.chunk:
                    continue
                full = self.flat_param[p.off:p.off + p.n]
                part = full[self.rank * p.chunk:(self.rank + 1) * p.chunk]
                if nccl:
                    dist.all_gather_into_tensor(full, part, group=self.group)       # in place: part IS full's shard `rank`
                else:
                    dist.all_gather([full[r * p.chunk:(r + 1) * p.chunk] for r in range(self.world_size)], part.clone(), group=self.group)
        if self.cuda:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                for key in keys:
                    run(key)
                    ev = torch.cuda.Event()
                    ev.record()
                    self._param_events[key] = ev
                if self.timing:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._gather_events.append((e0, e1))
        else:
            for key in keys:
                run(key)

    def wait_params(self, group: Optional[str], layer: Optional[int]):
        """``Engine.param_hook``: the compute stream waits for the parameter all-gather of bucket (group, layer); with
        ``group=None`` for all of them (anything that is not the layer-by-layer forward: generation, state_dict, close)."""
        if not self._param_events:
            return
        if group is None:
            keys = list(self._param_events)
        else:
            keys = [(group, layer)] if (group, layer) in self._param_events else []
        cur = torch.cuda.current_stream() if self.cuda else None
        for k in keys:
            ev = self._param_events.pop(k)
            if cur is not None:
                cur.wait_event(ev)

    def _launch(self, tensors: List[torch.Tensor], key):
        if self.zero1 and key[0] != "extra":
            return self._launch_zero(key)
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
            cur = torch.cuda.current_stream()
            if self.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.comm_stream)
                e1.record(cur)
                self._wait_events.append((e0, e1))
            else:
                cur.wait_stream(self.comm_stream)
        else:
            for h in self.handles:
                h.wait()
        self.armed = False
        self.opt_step += 1

    def param_gather_times_ms(self) -> List[float]:
        """ZeRO-1, ``timing`` on: duration of each step's parameter all-gather on the communication stream (it overlaps the
        next forward; what the forward actually waited is part of the step time, not separable here).  Synchronises."""
        if self._gather_events:
            torch.cuda.synchronize()
            self.param_gather_ms += [a.elapsed_time(b) for a, b in self._gather_events]
            self._gather_events = []
        return self.param_gather_ms

    def exposed_comm_ms(self) -> List[float]:
        """Per optimiser step: how long the compute stream sat waiting for the communication stream at the end of the
        backward (``timing`` must be on).  Synchronises."""
        if self._wait_events:
            torch.cuda.synchronize()
            self.exposed_ms += [a.elapsed_time(b) for a, b in self._wait_events]
            self._wait_events = []
        return self.exposed_ms

    # ------------------------------------------------------------------ replica consistency
    @staticmethod
    def broadcast_parameters(model, src: int = 0, group=None):
        """Rank ``src``'s parameters to every rank (start of training / after loading a checkpoint): replicas must not
        depend on every rank having drawn the same random init or read the same file."""
        if not GradSync.active():
            return
        dist.broadcast(model.arena, src=src, group=group)
        if model.lora is not None:
            dist.broadcast(model.lora.arena, src=src, group=group)
        if hasattr(model, "params_rewritten"):
            model.params_rewritten() # an optimiser built earlier keeps part of its fp32 master IN these weights (optim.py)

    @staticmethod
    def assert_replicas_equal(model, group=None, what: str = "parameters"):
        """Cheap divergence check: two checksums of the bf16 arena (sum of the raw 16-bit words, sum of squares of the
        values) must agree on every rank.  One host sync - call it at start-up / checkpoint time, not per step."""
        if not GradSync.active():
            return
        a = model.arena
        chk = torch.stack([a.view(torch.int16).to(torch.int64).sum().double(), a.double().square().sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if not torch.equal(lo, hi):
            raise RuntimeError(f"data-parallel replicas diverged: {what} checksums differ across ranks ({lo.tolist()} .. {hi.tolist()})")

    @staticmethod
    def mean_scalar(x, group=None) -> float:
        """Mean over the ranks of a logged scalar (training / validation loss): SURVEY 8e, C2."""
        if not GradSync.active():
            return float(x)
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return float(t) / dist.get_world_size(group)
