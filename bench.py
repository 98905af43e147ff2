#!/usr/bin/env python3
"""Headline benchmark: train tokens/sec (text+audio positions) of CSM-1B, bf16, seq 2048, on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full optimiser step of ``CSMTrainer.train_step`` on one fixed synthetic interleaved text+audio batch
(BASELINE.json configs[1]: full-param bf16, S=2048, B=4 per GPU): embedding -> 16-layer backbone -> codebook-0 CE
(+ teacher-forced depth decoder on 1/16 of the frames, "mode C" of SURVEY 8d) -> backward -> bf16 gradient all-reduce
(N>1) -> global-norm clip -> fused AdamW.  tokens/s = N * B * S / step time (metric definition of the reference's own
harness, src/csm/training/run_lora_benchmark.py:357-361).  Inputs and weights are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
# the host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (already exported on the
# build and GPU boxes; kept here so that a bare `torchrun bench.py` works too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBPS = 8000.0           # HBM3E (same table)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # SURVEY 8d protocol: >= 10 warm-up, >= 50 timed steps (~5 s of GPU time)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="sequences per GPU")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--mode", choices=["A", "B", "C"], default="C",
                    help="A: semantic CE only (reference loss); B: + decoder on all frames; C: + decoder on 1/16 of frames")
    ap.add_argument("--lora", action="store_true", help="configs[2]: LoRA r=8 q_proj/v_proj instead of full-param")
    ap.add_argument("--zero1", action="store_true", help="N > 1: reduce-scatter + sharded AdamW + parameter all-gather (ZeRO-1) "
                    "instead of the gradient all-reduce (also: CSM_DP_ZERO1=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seq", type=int, default=2048, help="positions of the CPU-baseline sample (SURVEY 8d protocol: B=1, S=2048)")
    ap.add_argument("--cpu-budget", type=float, default=340.0, help="seconds the CPU-baseline leg may take (bounds the sample)")
    ap.add_argument("--gemm-shapes", action="store_true", help="stderr: the instrumented step's GEMM launches grouped by shape")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (loss modes A / B, LoRA B=8, generate 10 s)")
    ap.add_argument("--tiny", action="store_true", help="tiny model (plumbing check only; the number is not the metric)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous plumbing only (gloo on the CPU, no model, no GPU): what tests/test_cpu.py runs")
    return ap.parse_args()


def launch_ranks(a):
    """``--gpus N`` (N > 1) without a torchrun environment: start the N ranks ourselves, one process per GPU, as
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>``
    - children, never an exec, and before this process has touched the GPU - forward rank 0's single JSON line and exit
    non-zero if any rank did.  The line must say ``n_gpus == N`` and name N ranks, or the run fails: an N-GPU request must
    never silently measure one GPU."""
    import socket
    import subprocess
    if not a.dry_run and "CSM_BENCH_FORCE_DEVICE" not in os.environ:
        have = torch.cuda.device_count()          # counts without initialising the GPU runtime
        if have < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)            # stderr passes through
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        raise SystemExit(f"bench.py --gpus {a.gpus}: a rank failed (torch.distributed.run exit code {proc.returncode})")
    if len(lines) != 1:
        sys.stderr.write(proc.stdout)
        raise SystemExit(f"bench.py --gpus {a.gpus}: expected ONE JSON line from rank 0, got {len(lines)}")
    out = json.loads(lines[0])
    ranks = (out.get("rccl") or {}).get("world")
    if out.get("n_gpus") != a.gpus or ranks != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: the result line reports n_gpus={out.get('n_gpus')}, rccl.world={ranks}")
    print(lines[0], flush=True)


class GemmTimer:
    """HIP-event timing of every GEMM launch of ONE step inside the timed region (events are recorded on the stream
    the kernels run on: torch's current stream).  Gives per-variant algorithmic FLOP / measured time."""

    def __init__(self):
        self.records = []

    def _kernel(self):
        """rocprofv3's name of the kernel the library just launched (csm_gemm_last_kernel), 'unsigned short' spelled u16."""
        from csm.hip import lib
        return lib.csm_gemm_last_kernel().decode().replace("unsigned short", "u16")

    def __enter__(self):
        from csm.hip import ops
        self.ops, self.orig = ops, ops.gemm
        self.orig_fwd, self.orig_bwd = ops.linear_swiglu_fwd, ops.linear_dx_swiglu_bwd
        timer = self

        def ev():
            return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def fused_fwd(x, w13, gu, act):           # gu[M,2F] = x w13^T, act = SwiGLU(gu) in the epilogue
            e0, e1 = ev()
            e0.record(); timer.orig_fwd(x, w13, gu, act); e1.record()
            M, K = x.shape
            N = w13.shape[0]
            timer.records.append(("nt_fwd_bf16", 2.0 * M * N * K, e0, e1, 2.0 * (M * K + N * K + M * N + M * N // 2), timer._kernel()))

        def fused_bwd(dy, w2, gu, dgu):           # dgu[M,2F] = SwiGLU'(gu) . (dy w2) in the epilogue
            e0, e1 = ev()
            e0.record(); timer.orig_bwd(dy, w2, gu, dgu); e1.record()
            M, K = dy.shape
            F = w2.shape[1]
            timer.records.append(("nn_dgrad_bf16", 2.0 * M * F * K, e0, e1, 2.0 * (M * K + F * K + 4 * M * F), timer._kernel()))

        def timed(A, B, C, R=None, transA=False, transB=False, alpha=1.0, batch=1, sA=0, sB=0, sC=0, sR=0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = timer.orig(A, B, C, R, transA, transB, alpha, batch, sA, sB, sC, sR)
            e1.record()
            M, N = C.shape
            K = A.shape[0] if transA else A.shape[1]
            kind = {(False, False): "nt_fwd", (False, True): "nn_dgrad", (True, True): "tn_wgrad", (True, False): "tt"}[(transA, transB)]
            kind += "_f32" if C.dtype == torch.float32 else "_bf16"
            esz = C.element_size()
            timer.records.append((kind, 2.0 * M * N * K * batch, e0, e1, batch * (2.0 * (M * K + N * K) + esz * M * N), timer._kernel()))
            return out

        self.orig_rope = ops.linear_rope_fwd

        def fused_rope(x, w, out, table, S, n_rope_cols, head_dim):      # q|k|v projection with RoPE in the epilogue
            e0, e1 = ev()
            e0.record(); r = timer.orig_rope(x, w, out, table, S, n_rope_cols, head_dim); e1.record()
            M, K = x.shape
            N = w.shape[0]
            timer.records.append(("nt_fwd_bf16", 2.0 * M * N * K, e0, e1, 2.0 * (M * K + N * K + M * N), timer._kernel()))
            return r

        ops.linear_rope_fwd = fused_rope
        self.orig_adamw = ops.adamw_step
        self.hbm = []

        def adamw(master, m, v, param, grad, *args, zero_grad=False, **kw):     # HBM-bound side of the step, timed the same way
            e0, e1 = ev()
            e0.record(); r = timer.orig_adamw(master, m, v, param, grad, *args, zero_grad=zero_grad, **kw); e1.record()
            timer.hbm.append(((28 + (2 if zero_grad else 0)) * master.numel(), e0, e1, "adamw_kernel"))
            return r

        self.orig_two = ops.two_linear_dw

        def two_dw(dy1, x1, dw1, dy2, x2, dw2, accumulate=False, alpha=1.0):       # two weight gradients in one launch
            e0, e1 = ev()
            e0.record(); ok = timer.orig_two(dy1, x1, dw1, dy2, x2, dw2, accumulate=accumulate, alpha=alpha); e1.record()
            if ok:
                M = dy1.shape[0]
                fl = 2.0 * M * (dw1.numel() + dw2.numel())
                timer.records.append(("tn_wgrad_bf16", fl, e0, e1, 2.0 * (dy1.numel() + x1.numel() + dy2.numel() + x2.numel() + dw1.numel() + dw2.numel()), timer._kernel()))
            return ok

        ops.two_linear_dw = two_dw
        self.orig_multi = ops.multi_linear_dw

        def multi_dw(problems, accumulate=False, alpha=1.0):                    # several layers' attention weight gradients in one launch
            e0, e1 = ev()
            e0.record(); ok = timer.orig_multi(problems, accumulate=accumulate, alpha=alpha); e1.record()
            if ok:
                M = problems[0][0].shape[0]
                fl = 2.0 * M * sum(p[2].numel() for p in problems)
                timer.records.append(("tn_wgrad_bf16", fl, e0, e1, 2.0 * sum(t.numel() for p in problems for t in p), timer._kernel()))
            return ok

        ops.multi_linear_dw = multi_dw
        self.orig_kext = ops.gemm_kext

        def kext(A, B, C, xA, xB, R=None, transB=False, **kw):        # frozen projection + LoRA group in one product
            e0, e1 = ev()
            e0.record(); r = timer.orig_kext(A, B, C, xA, xB, R=R, transB=transB, **kw); e1.record()
            M, K = A.shape
            N = B.shape[1] if transB else B.shape[0]
            timer.records.append(("nn_dgrad_bf16" if transB else "nt_fwd_bf16", 2.0 * M * N * (K + xA.shape[1]), e0, e1,
                                  2.0 * (M * K + N * K + C.numel() + xA.numel() + xB.numel()), timer._kernel()))
            return r

        ops.gemm_kext = kext
        self.orig_adamw_split = ops.adamw_step_split

        def adamw_split(lo, m, v, param, grad, *args, zero_grad=False, **kw):   # master as bf16 + 16-bit halves: 26 B/param
            e0, e1 = ev()
            e0.record(); r = timer.orig_adamw_split(lo, m, v, param, grad, *args, zero_grad=zero_grad, **kw); e1.record()
            timer.hbm.append(((26 + (2 if zero_grad else 0)) * lo.numel(), e0, e1, "adamw_split_kernel"))
            return r

        ops.adamw_step_split = adamw_split
        ops.gemm, ops.linear_swiglu_fwd, ops.linear_dx_swiglu_bwd, ops.adamw_step = timed, fused_fwd, fused_bwd, adamw
        return self

    def __exit__(self, *a):
        self.ops.gemm, self.ops.linear_swiglu_fwd, self.ops.linear_dx_swiglu_bwd = self.orig, self.orig_fwd, self.orig_bwd
        self.ops.adamw_step = self.orig_adamw
        self.ops.adamw_step_split = self.orig_adamw_split
        self.ops.two_linear_dw = self.orig_two
        self.ops.multi_linear_dw = self.orig_multi
        self.ops.gemm_kext = self.orig_kext
        self.ops.linear_rope_fwd = self.orig_rope

    def adamw_summary(self):
        torch.cuda.synchronize()
        if not self.hbm:
            return None
        nbytes = sum(r[0] for r in self.hbm)
        sec = sum(r[1].elapsed_time(r[2]) for r in self.hbm) * 1e-3
        names = sorted({r[3] for r in self.hbm})             # the symbol rocprofv3 shows (ops.hip)
        return {"kernel": "+".join(names), "launches": len(self.hbm), "bound": "hbm", "achieved": round(nbytes / sec / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBPS, 4), "ms_per_step": round(sec * 1e3, 3),
                "algorithmic_bytes_per_step": nbytes}

    def summary(self, by_kernel=False):
        """Per GEMM kind (operand layout x output type) or, ``by_kernel``, per kernel symbol as rocprofv3 names it."""
        torch.cuda.synchronize()
        agg = {}
        for kind, flop, e0, e1, nbytes, kernel in self.records:
            d = agg.setdefault(kernel if by_kernel else kind, [0.0, 0.0, 0, 0.0, {}])
            d[0] += flop
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
            d[3] += nbytes
            other = kind if by_kernel else kernel
            d[4][other] = d[4].get(other, 0) + 1
        return {k: {"tflops": v[0] / v[1] / 1e12, "time_ms": v[1] * 1e3, "launches": v[2], "avg_us": v[1] / v[2] * 1e6,
                    "flop": v[0], "operand_bytes_per_launch": v[3] / v[2], "split": v[4]} for k, v in agg.items() if v[1] > 0}


    def print_shapes(self):
        """stderr table of the instrumented step's launches grouped by (kind, FLOP, operand bytes, kernel): which shapes lose time."""
        torch.cuda.synchronize()
        agg = {}
        for kind, flop, e0, e1, nbytes, kernel in self.records:
            d = agg.setdefault((kind, flop, nbytes, kernel), [0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += 1
        for (kind, flop, nbytes, kernel), (sec, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            print(f"{sec * 1e3:8.3f} ms  x{n:3d}  avg {sec / n * 1e6:7.1f} us  {flop / (sec / n) / 1e12:7.1f} TF/s  {flop / 1e9:7.1f} GF  "
                  f"{nbytes / 1e6:7.1f} MB  {kind:14s} {kernel}", file=sys.stderr)


def kernel_source_sha16():
    """Identity of the kernels the running library was built from: sha256 over the HIP sources (the built .so is not
    bit-reproducible across toolchain paths, the sources are what a profile must match)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "csm-train-pytorch_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.inc"))
                    + glob.glob(os.path.join(src, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, launches_per_step=None):
    """HBM-side bytes per launch of a GEMM kernel (rocprofv3 symbol) from the committed rocprofv3 PMC passes over this same
    command (profiles/run_pmc_bench_rNN.sh -> the newest profiles/rNN_bench_pmc_traffic.json); None when there is none.
    Counters cannot be read from inside the timed process, so this is the profile's figure, not a live one - and therefore
    only accepted when the profile was taken with THESE kernels: the file records the sha of the kernel sources it ran
    (``_meta.kernel_source_sha16``) and its per-step launch count of the symbol; a mismatch of either returns no traffic and
    says why (a traffic regression must not hide behind a stale profile)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_traffic.json")))
    if not files:
        return None, "no profiles/r*_bench_pmc_traffic.json"
    rel = os.path.relpath(files[-1], ROOT)
    doc = json.load(open(files[-1]))
    meta = doc.get("_meta", {})
    have = kernel_source_sha16()
    if meta.get("kernel_source_sha16") != have:
        return None, f"{rel} REFUSED: taken with kernel sources {meta.get('kernel_source_sha16')}, running {have} - re-run profiles/run_pmc_bench_r04.sh"
    t = doc.get("_kernels", {}).get(kernel)
    if not t:
        return None, f"{rel} REFUSED: no entry for the running dominant kernel {kernel}"
    if launches_per_step and t.get("launches_per_step") and t["launches_per_step"] != launches_per_step:
        return None, f"{rel} REFUSED: {kernel} ran {t['launches_per_step']} times per step in the profile, {launches_per_step} now"
    return t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"], rel


def attention_flops(args_model, B, S, n_dec_frames, factor=3.0):
    """Attention FLOPs of one train step on the causal half.  ``factor`` 3.0 = SURVEY 8d's ALGORITHMIC count (forward + a
    backward of twice the forward: 4 d (S/2) L per position, x3); 3.5 = what the kernels EXECUTE (the backward recomputes S
    in both of its passes)."""
    bb, dc = args_model.bb, args_model.dc
    one = lambda c, b, s: 4.0 * b * c.num_heads * (s * s / 2.0) * c.head_dim * c.num_layers   # noqa: E731
    return factor * (one(bb, B, S) + (one(dc, n_dec_frames, args_model.args.audio_num_codebooks) if n_dec_frames else 0.0))


def cpu_baseline(model, cfg_fn, seq, seed, sw, aw, budget_s):
    """The oracle (CPU restatement of the reference PyTorch path) on this box's host cores, SURVEY 8d / reference
    src/csm/training/benchmark_lora.py:536-571 protocol: full-param fp32 train step (loss -> backward -> clip -> AdamW,
    semantic loss only = what the reference computes), B=1, S=seq, one fixed synthetic batch, 1 warm-up step + timed
    steps on all cores, plus a 1-thread figure.  Bounded: a step at S=2048 is tens of seconds of CPU work, so the number of
    timed steps (1..3) and the size of the 1-thread sample follow from the measured warm-up step and ``budget_s``.
    Uses the SAME weights as the GPU model (its bf16 values widened to fp32), so the two losses are comparable."""
    from oracle import csm_oracle as O
    cfg = cfg_fn()
    params = {k: v.float().cpu().requires_grad_(True) for k, v in model._views(model.arena).items()}
    plist = list(params.values())
    tokens, mask, targets = O.synthetic_batch(cfg, 1, seq, seed=seed)
    threads = torch.get_num_threads()
    opt = torch.optim.AdamW(plist, lr=1e-5, weight_decay=0.01)

    def step(tk, mk, tg):
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        total, _ = O.compute_loss(params, cfg, tk, mk, tg, sw, aw, acoustic_rows="off")
        total.backward()
        torch.nn.utils.clip_grad_norm_(plist, 1.0)
        opt.step()
        return time.time() - t0, float(total.detach())

    t_warm, loss0 = step(tokens, mask, targets)            # warm-up (allocates the AdamW state); its loss is the parity value
    timed = []
    left = budget_s - t_warm
    reserve = 30.0                                         # kept for the 1-thread sample once one timed step exists
    while len(timed) < 3 and left > 1.3 * (min(timed) if timed else t_warm) + (reserve if timed else 0.0):
        dt, _ = step(tokens, mask, targets)
        timed.append(dt)
        left -= dt
    note = ""
    if not timed:                                          # the budget does not even hold one timed step: report the cold one
        timed, note = [t_warm], " (budget exhausted by the warm-up step: this is the COLD step)"
    dt = sum(timed) / len(timed)
    # 1-thread figure on a short sample (a 1-thread step at S=2048 would take the better part of an hour)
    one = None
    if left > 20:
        s1 = max(16, min(seq, 64))
        tk1, mk1, tg1 = O.synthetic_batch(cfg, 1, s1, seed=seed + 1)
        torch.set_num_threads(1)
        try:
            d1, _ = step(tk1, mk1, tg1)
        finally:
            torch.set_num_threads(threads)
        one = {"value": s1 / d1, "unit": "tokens/s", "cores": 1, "sample": f"1 step at B=1,S={s1}, {d1:.1f}s (optimiser pass over 1.55 B "
               "fp32 parameters included, which dominates a sample this short)"}
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=seq / dt, unit="tokens/s", cores=threads, kind="port",
                sample=f"full-param fp32 train step of the oracle (fwd+bwd+clip+AdamW, semantic loss = the reference's compute_loss) at "
                       f"B=1,S={seq}: 1 warm-up step ({t_warm:.1f}s) + {len(timed)} timed step(s), mean {dt:.1f}s{note}",
                os_cpu_count=os.cpu_count(), torch_threads=threads, cpu_model=cpu_model, one_thread=one,
                loss=loss0), (tokens, mask, targets)


def comm_census(world, backend, local):
    """Who is in the job: every rank reports (rank, device index, device name) through the process group itself, so the
    result line proves that N ranks on N different GPUs took part in the collectives (the driver reads ``rccl.world``)."""
    import torch.distributed as dist
    me = {"rank": int(os.environ.get("RANK", 0)), "device": local, "name": torch.cuda.get_device_name(local)}
    if world == 1:
        return {"world": 1, "backend": None, "devices": [local]}
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    ones = torch.ones(1, device="cuda")
    dist.all_reduce(ones)                                            # a real device collective: RCCL carries it when backend == nccl
    devices = [e["device"] for e in sorted(everyone, key=lambda e: e["rank"])]
    if int(ones.item()) != world:
        raise SystemExit(f"bench.py: the all-reduce saw {int(ones.item())} ranks, WORLD_SIZE={world}")
    if len(set(devices)) != world and "CSM_BENCH_FORCE_DEVICE" not in os.environ:
        raise SystemExit(f"bench.py: {world} ranks but devices {devices}: two ranks share a GPU")
    return {"world": world, "backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "devices": devices,
            "ranks_in_allreduce": int(ones.item())}


def dry_run(a, rank, world, real_stdout):
    """The launcher's contract without a GPU: rendezvous (gloo), barrier, MAX over ranks of a timed region, one JSON line
    from rank 0 that names the ranks.  Not a measurement."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    x = torch.ones(1)
    for _ in range(a.steps):
        if world > 1:
            dist.all_reduce(x)
            x /= world
    t = torch.tensor([time.perf_counter() - t0])
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"metric": "dry-run (launcher plumbing only, no model, no GPU)", "value": 0.0, "unit": "tokens/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": float(t) / max(1, a.steps) * 1e3, "dry_run": True,
                          "rccl": {"world": int(seen.item()), "backend": "gloo (dry run)"}}), file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    # stdout carries exactly ONE line (the JSON result): everything incidental - logger handlers created from here on,
    # library prints - goes to stderr
    real_stdout, sys.stdout = sys.stdout, sys.stderr
    if a.gpus < 1:
        raise SystemExit(f"--gpus {a.gpus}")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.stdout = real_stdout
        return launch_ranks(a)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if "CSM_BENCH_FORCE_DEVICE" in os.environ:      # rehearsal of the multi-rank path on a one-GPU box (with gloo)
        local = int(os.environ["CSM_BENCH_FORCE_DEVICE"])
    if world != a.gpus:                             # a launcher that started a different number of ranks than --gpus says
        raise SystemExit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    backend = None
    if a.dry_run:
        return dry_run(a, rank, world, real_stdout)
    torch.cuda.set_device(local)
    if world > 1:
        backend = os.environ.get("CSM_BENCH_BACKEND", "nccl")        # nccl == RCCL on ROCm
        if backend == "nccl":
            from csm.training.dp import init_nccl
            init_nccl(rank, world, local)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    rccl = comm_census(world, backend, local)

    from csm.data import SyntheticCSMDataset, collate_variable_length
    from csm.models.model import Model, ModelArgs
    from csm.training.trainer import CSMTrainer, csm_1b_args
    from csm.training.lora import apply_lora_to_model
    from csm.training.optim import FusedAdamW
    from csm.training.dp import GradSync
    from csm.training.utils import compute_loss

    if a.tiny:
        args = ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", 300, 67, 4)
        a.seq = min(a.seq, 128)
    else:
        args = csm_1b_args()
    import tempfile

    def make_trainer(model, lora):
        tr = CSMTrainer("", tempfile.mkdtemp(prefix=f"csm_bench_rank{rank}_"), device=f"cuda:{local}")   # trainer wants an output dir
        tr.logger.setLevel(30)
        tr.model = model
        tr.zero1 = True if a.zero1 else None         # None: the CSM_DP_ZERO1 switch decides (data parallel only)
        if lora:
            apply_lora_to_model(model, r=8, alpha=16.0, target_modules=["q_proj", "v_proj"])
            if GradSync.active():
                GradSync.broadcast_parameters(model)
            tr.optimizer = FusedAdamW(model, {}, lora_lr=1e-4)
            tr.grad_sync = GradSync.for_model(model) if GradSync.active() else None
        else:
            tr.prepare_optimizer()
        return tr

    def make_batch(nb):
        ds = SyntheticCSMDataset(nb, a.seq, args.text_vocab_size, args.audio_vocab_size, args.audio_num_codebooks, seed=1234 + rank)
        return {k: v.cuda() for k, v in collate_variable_length([ds[i] for i in range(nb)]).items()}

    K_ = args.audio_num_codebooks

    def run(tr, batch, warmup, steps, with_timer=True):
        """W untimed + K timed optimiser steps, barrier + device sync on both sides, MAX over ranks.  Returns a dict."""
        for _ in range(warmup):
            tr.train_step(batch, 1, True, 1.0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gt = None
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]     # per-step spread (no host sync inside the loop)
        marks[0].record()
        for i in range(steps):
            if with_timer and i == steps - 1:
                with GemmTimer() as gt:
                    loss, det = tr.train_step(batch, 1, True, 1.0)
            else:
                loss, det = tr.train_step(batch, 1, True, 1.0)
            marks[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        nb = batch["input_tokens"].shape[0]
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(max(1, steps - 1))) or [dt / steps * 1e3]
        return dict(dt=dt, steps=steps, ms=dt / steps * 1e3, tokens_per_s=world * nb * a.seq / (dt / steps), gt=gt, loss=float(loss),
                    per_step=per_step, nb=nb)

    def step_flops(model, res):
        """GEMM FLOPs of the instrumented step + algorithmic attention FLOPs (causal half, SURVEY 8d: 3x forward for fwd+bwd)."""
        kinds = res["gt"].summary() if res["gt"] is not None else {}
        n_dec = 0
        if model.acoustic_mode != "off":
            nfr = res["nb"] * (a.seq - 1)
            n_dec = nfr if model.acoustic_mode == "all" else max(1, int(round(nfr * model.acoustic_fraction)))
        gemm = sum(v["flop"] for v in kinds.values())
        return kinds, gemm, attention_flops(model, res["nb"], a.seq, n_dec)

    def leg_summary(model, res, what):
        kinds, gemm, att = step_flops(model, res)
        tf = (gemm + att) / (res["ms"] * 1e-3) / 1e12
        return {"workload": what, "ms_per_step": round(res["ms"], 3), "tokens_per_s": round(res["tokens_per_s"], 1), "steps": res["steps"],
                "loss": res["loss"],
                "roofline": {"bound": "mfma", "what": "whole step: GEMM + attention FLOP / step time", "achieved": round(tf, 1),
                             "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4)}}

    model = Model(args, device=f"cuda:{local}", seed=0)          # random init, same on every rank (and broadcast from rank 0)
    model.acoustic_mode = {"A": "off", "B": "all", "C": "amortized"}[a.mode]
    tr = make_trainer(model, a.lora)
    if tr.grad_sync is not None:
        tr.grad_sync.timing = True
    batch = make_batch(a.batch)
    cb_tokens = int(batch["input_masks"][:, :, 0].sum()) * K_ + int(batch["input_masks"][:, :, K_].sum())
    res = run(tr, batch, a.warmup, a.steps)
    dt, ms, tokens_per_s, gt, loss = res["dt"], res["ms"], res["tokens_per_s"], res["gt"], res["loss"]
    exposed = tr.grad_sync.exposed_comm_ms()[-a.steps:] if tr.grad_sync is not None else None

    # ---- N > 1: BASELINE config 4 proper is 8 sequences per GPU (global batch 64 on 8 GPUs).  The headline keeps 4 per GPU
    # at every N so that the driver's scaling curve compares equal per-GPU work; config 4's own number is reported beside it.
    dp_b8 = None
    if world > 1 and not a.tiny and not a.lora and a.batch != 8:
        r8 = run(tr, make_batch(8), 2, 5, with_timer=False)
        dp_b8 = {"workload": f"BASELINE config 4: full-param bf16 DP, seq={a.seq}, 8 sequences/GPU, global batch {8 * world}",
                 "ms_per_step": round(r8["ms"], 3), "tokens_per_s": round(r8["tokens_per_s"], 1), "steps": r8["steps"],
                 "exposed_comm_ms": round(sum(tr.grad_sync.exposed_comm_ms()[-5:]) / 5, 3)}

    if rank == 0:
        per_step = res["per_step"]
        pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 3)   # noqa: E731
        kinds, gemm_flop, att_flop = step_flops(model, res)
        # the dominant kernel = the kernel SYMBOL (as rocprofv3 prints it) with the most time in the instrumented step
        kernels = res["gt"].summary(by_kernel=True) if res["gt"] is not None else {}
        if a.gemm_shapes and res["gt"] is not None and rank == 0:
            res["gt"].print_shapes()
        dom = max(kernels.items(), key=lambda kv: kv[1]["time_ms"]) if kernels else (None, None)
        roof = None
        if dom[0] is not None:
            traffic, traffic_src = (pmc_traffic(dom[0], dom[1]["launches"]) if (a.batch, a.seq, a.mode, a.lora, a.tiny) == (4, 2048, "C", False, False)
                                    else (None, "not the headline configuration"))
            roof = {"bound": "mfma", "kernel": dom[0], "kinds": dom[1]["split"], "achieved": round(dom[1]["tflops"], 2),
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dom[1]["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                    "operand_bytes_per_launch": round(dom[1]["operand_bytes_per_launch"]),
                    "peak_note": "dense bf16 MFMA peak of the micro-architecture guide (2.4 GHz); the chip is power limited on non-zero data: a "
                                 "register-only MFMA loop reaches 1.56-1.85 PFLOP/s on random bf16 operands (tools/probes/mfma_shape_probe.hip, "
                                 "profiles/r02_clock_power.txt)",
                    "avg_launch_us": round(dom[1]["avg_us"], 2), "launches_per_step": dom[1]["launches"],
                    "all_gemm_kernels": {k: {"tflops": round(v["tflops"], 1), "ms_per_step": round(v["time_ms"], 2),
                                             "launches": v["launches"]} for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["time_ms"])},
                    "all_gemm_variants": {k: {"tflops": round(v["tflops"], 1), "ms_per_step": round(v["time_ms"], 2),
                                              "launches": v["launches"], "kernels": v["split"]} for k, v in kinds.items()}}
        out = {
            "metric": "train tokens/sec (text+audio) CSM-1B bf16 seq2048", "value": round(tokens_per_s, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("tiny-plumbing" if a.tiny else "CSM-1B") + (" LoRA r=8 q_proj/v_proj" if a.lora else " full-param")
                       + f" bf16 train step, seq={a.seq}, batch={a.batch}/GPU, loss mode {a.mode}"
                       + (" (semantic CE + depth decoder on 1/16 of frames)" if a.mode == "C" else "")
                       + (f"; weak scaling keeps {a.batch} sequences/GPU at every N - BASELINE config 4's 8/GPU is in extra.dp_config4_b8" if world > 1 else ""),
                       "global_batch": world * a.batch, "seq_len": a.seq, "parallelism": f"dp{world}"},
            "codebook_tokens_per_s": round(world * cb_tokens / (dt / a.steps), 1),   # secondary: 32 per audio frame + 1 per text token
            # GEMM FLOP + ALGORITHMIC attention FLOP (causal half, fwd + bwd = 3x forward) of one step / step time / dense
            # peak; `_executed` counts attention at the 3.5x the kernels run (S recomputed in both backward passes)
            "mfma_utilisation_step": round((gemm_flop + att_flop) / (dt / a.steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if kinds else None,
            "mfma_utilisation_step_executed": round((gemm_flop + att_flop * 3.5 / 3.0) / (dt / a.steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if kinds else None,
            "step_flop": {"gemm": gemm_flop, "attention_algorithmic": att_flop, "attention_executed": att_flop * 3.5 / 3.0},
            "kernel_source_sha16": kernel_source_sha16(),
            "hbm_kernel": gt.adamw_summary() if gt is not None else None,
            "loss": float(loss), "step_ms": {"p10": pct(0.10), "p50": pct(0.50), "p90": pct(0.90)}, "roofline": roof,
            "rccl": rccl,
        }
        if exposed is not None:
            out["exposed_comm_ms"] = {"mean": round(sum(exposed) / len(exposed), 3), "max": round(max(exposed), 3),
                                      "what": "per step: time the compute stream waits for the gradient collectives after the backward"}
            zero = bool(getattr(tr.optimizer, "sharded", False))
            out["dp_exchange"] = {"mode": "zero1 (reduce-scatter, AdamW on 1/N shards, parameter all-gather behind the step)" if zero
                                  else "all-reduce (replicated AdamW)",
                                  "optimizer_ms_per_step": out["hbm_kernel"]["ms_per_step"] if out["hbm_kernel"] else None,
                                  "optimizer_params_this_rank": tr.optimizer.num_owned() if zero else tr.optimizer.num_trainable()}
            if zero:
                g = tr.grad_sync.param_gather_times_ms()[-a.steps:]
                out["dp_exchange"]["param_all_gather_ms"] = {"mean": round(sum(g) / max(1, len(g)), 3), "what": "duration on the "
                                                             "communication stream; it overlaps the next forward, which waits per layer bucket"}
    extra = {}
    if dp_b8 is not None:
        extra["dp_config4_b8"] = dp_b8

    # ---- extra legs (1 GPU only, outside the headline's timed region): the other loss modes of SURVEY 8d, BASELINE
    # configs 3 (LoRA r=8 q/v, B=8) and 5 (generate 10 s of audio)
    if world == 1 and not a.no_extras and not a.tiny and not a.lora and a.mode == "C":
        for mode, (w, k) in (("A", (3, 10)), ("B", (1, 3))):
            model.acoustic_mode = {"A": "off", "B": "all"}[mode]
            r = run(tr, batch, w, k)
            extra[f"mode_{mode}"] = leg_summary(model, r, f"full-param, batch {a.batch}, loss mode {mode} ("
                                                + ("semantic CE only = the reference's compute_loss" if mode == "A" else "depth decoder on every frame") + ")")
        model.acoustic_mode = "amortized"
        # config 2's batch against its own rows: the loss of the B = 4 batch (reference loss, and with the depth decoder on pinned
        # frames) must be the mean of the four single-sequence losses - each of those is the quantity parity_check ties to the
        # CPU oracle below and tests/test_e2e_gpu.py ties to it on an S = 128 prefix
        try:
            per = torch.arange(0, a.seq - 1, 16)
            nb = batch["input_tokens"].shape[0]
            rows = torch.cat([per + b * (a.seq - 1) for b in range(nb)])
            with torch.no_grad():
                t4, d4 = compute_loss(model, batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"], 100.0, 1.0, acoustic_rows=rows)
                ones = [compute_loss(model, batch["input_tokens"][b:b + 1], batch["input_masks"][b:b + 1], batch["target_audio_tokens"][b:b + 1],
                                     100.0, 1.0, acoustic_rows=per) for b in range(nb)]
            m_sem = sum(float(o[1]["semantic_loss"]) for o in ones) / nb
            m_ac = sum(float(o[1]["acoustic_loss"]) for o in ones) / nb
            m_tot = sum(float(o[0]) for o in ones) / nb
            extra["config2_b4_parity"] = {
                "what": f"loss of the B={nb} batch vs the mean of its {nb} single-sequence losses (same weights, pinned decoder frames)",
                "batch_total": float(t4), "mean_of_singles_total": m_tot, "rel_diff_total": abs(float(t4) - m_tot) / abs(m_tot),
                "rel_diff_semantic": abs(float(d4["semantic_loss"]) - m_sem) / abs(m_sem),
                "rel_diff_acoustic": abs(float(d4["acoustic_loss"]) - m_ac) / abs(m_ac)}
        except Exception as e:  # noqa: BLE001 - an extra leg must never take the headline line down
            extra["config2_b4_parity"] = {"error": repr(e)}
        # free the full-param trainer before the LoRA model is built
        del tr, res, gt
        lm = Model(args, device=f"cuda:{local}", seed=0)
        lm.acoustic_mode = "amortized"
        ltr = make_trainer(lm, True)
        r = run(ltr, make_batch(8), 3, 10)
        extra["lora_b8"] = leg_summary(lm, r, f"BASELINE config 3: LoRA r=8 q_proj/v_proj, seq={a.seq}, batch 8, loss mode C")
        del ltr, lm, r
        torch.cuda.empty_cache()
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import generate_bench
            g = generate_bench.run(model, frames=125)
            bytes_frame = 9.1e9                                       # weights streamed per 80 ms frame, batch 1
            extra["generate_10s"] = {"workload": "BASELINE config 5: Mimi encode of 5 s context + 125 AR frames (32 codebooks) + Mimi decode",
                                     **g,
                                     "roofline": {"bound": "hbm", "achieved": round(bytes_frame / (g["ms_per_frame"] * 1e-3) / 1e9, 1),
                                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                  "frac": round(bytes_frame / (g["ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                                  "bytes_per_frame": bytes_frame}}
        except Exception as e:  # noqa: BLE001 - an extra leg must never take the headline line down
            extra["generate_10s"] = {"error": repr(e)}
        try:
            gb = generate_bench.run_batch(model, nb=4, frames=125)
            extra["generate_batch4_10s"] = {"workload": "generate_batch(): 4 utterances x 125 frames decoded together (ragged prompts; "
                                                        "every row's frames equal the frames it samples alone)", **gb}
        except Exception as e:  # noqa: BLE001
            extra["generate_batch4_10s"] = {"error": repr(e)}

    if rank == 0:
        if extra:
            out["extra"] = extra
        if not a.no_cpu_baseline and world == 1:
            from oracle import csm_oracle as O
            cb, (ct, cm, cg) = cpu_baseline(model, (O.tiny_cfg if a.tiny else O.csm_1b_cfg), min(a.cpu_seq, a.seq), 4321, 100.0, 1.0,
                                            a.cpu_budget)
            mode = model.acoustic_mode
            model.acoustic_mode = "off"
            with torch.no_grad():
                gl, _ = compute_loss(model, ct, cm, cg, 100.0, 1.0)
            model.acoustic_mode = mode
            cpu_loss = cb.pop("loss")
            out["cpu_baseline"] = cb
            out["parity_check"] = {"what": f"loss of the same weights/batch (B=1, S={ct.shape[1]}): HIP bf16 path vs CPU fp32 oracle", "cpu_loss": cpu_loss,
                                   "gpu_loss": float(gl), "rel_diff": abs(float(gl) - cpu_loss) / abs(cpu_loss)}
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
