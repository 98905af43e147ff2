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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seq", type=int, default=256, help="positions of the bounded CPU-baseline sample")
    ap.add_argument("--tiny", action="store_true", help="tiny model (plumbing check only; the number is not the metric)")
    return ap.parse_args()


class GemmTimer:
    """HIP-event timing of every GEMM launch of ONE step inside the timed region (events are recorded on the stream
    the kernels run on: torch's current stream).  Gives per-variant algorithmic FLOP / measured time."""

    def __init__(self):
        self.records = []

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
            timer.records.append(("nt_fwd_bf16", 2.0 * M * N * K, e0, e1, 2.0 * (M * K + N * K + M * N + M * N // 2)))

        def fused_bwd(dy, w2, gu, dgu):           # dgu[M,2F] = SwiGLU'(gu) . (dy w2) in the epilogue
            e0, e1 = ev()
            e0.record(); timer.orig_bwd(dy, w2, gu, dgu); e1.record()
            M, K = dy.shape
            F = w2.shape[1]
            timer.records.append(("nn_dgrad_bf16", 2.0 * M * F * K, e0, e1, 2.0 * (M * K + F * K + 4 * M * F)))

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
            timer.records.append((kind, 2.0 * M * N * K * batch, e0, e1, batch * (2.0 * (M * K + N * K) + esz * M * N)))
            return out

        self.orig_adamw = ops.adamw_step
        self.hbm = []

        def adamw(master, m, v, param, grad, *args, zero_grad=False, **kw):     # HBM-bound side of the step, timed the same way
            e0, e1 = ev()
            e0.record(); r = timer.orig_adamw(master, m, v, param, grad, *args, zero_grad=zero_grad, **kw); e1.record()
            timer.hbm.append(((28 + (2 if zero_grad else 0)) * master.numel(), e0, e1))
            return r

        ops.gemm, ops.linear_swiglu_fwd, ops.linear_dx_swiglu_bwd, ops.adamw_step = timed, fused_fwd, fused_bwd, adamw
        return self

    def __exit__(self, *a):
        self.ops.gemm, self.ops.linear_swiglu_fwd, self.ops.linear_dx_swiglu_bwd = self.orig, self.orig_fwd, self.orig_bwd
        self.ops.adamw_step = self.orig_adamw

    def adamw_summary(self):
        torch.cuda.synchronize()
        if not self.hbm:
            return None
        nbytes = sum(b for b, _, _ in self.hbm)
        sec = sum(e0.elapsed_time(e1) for _, e0, e1 in self.hbm) * 1e-3
        return {"kernel": "adamw_kernel", "bound": "hbm", "achieved": round(nbytes / sec / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBPS, 4), "ms_per_step": round(sec * 1e3, 3),
                "algorithmic_bytes_per_step": nbytes}

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for kind, flop, e0, e1, nbytes in self.records:
            d = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            d[0] += flop
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
            d[3] += nbytes
        return {k: {"tflops": v[0] / v[1] / 1e12, "time_ms": v[1] * 1e3, "launches": v[2], "avg_us": v[1] / v[2] * 1e6,
                    "flop": v[0], "operand_bytes_per_launch": v[3] / v[2]} for k, v in agg.items() if v[1] > 0}


def pmc_traffic(kind):
    """HBM-side bytes per launch of a GEMM kind from the committed rocprofv3 PMC passes over this same command
    (profiles/run_pmc_bench_r01.sh -> profiles/r01_bench_pmc_traffic.json); None when the file is not there.  Counters
    cannot be read from inside the timed process, so this is the profile's figure, not a live one."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    t = json.load(open(path)).get(kind)
    if not t:
        return None, None
    return t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"], "profiles/r01_bench_pmc_traffic.json"


def cpu_baseline(model, cfg_fn, seq, seed, sw, aw):
    """The oracle (CPU restatement of the reference PyTorch path) on this box's host cores: one full-param fp32 train
    step (loss -> backward -> clip -> AdamW), semantic loss only (what the reference computes), B=1, S=seq.
    Uses the SAME weights as the GPU model (its bf16 values widened to fp32), so the two losses are comparable."""
    from oracle import csm_oracle as O
    cfg = cfg_fn()
    params = {k: v.float().cpu().requires_grad_(True) for k, v in model._views(model.arena).items()}
    tokens, mask, targets = O.synthetic_batch(cfg, 1, seq, seed=seed)
    threads = torch.get_num_threads()
    t0 = time.time()
    total, det = O.compute_loss(params, cfg, tokens, mask, targets, sw, aw, acoustic_rows="off")
    total.backward()
    plist = list(params.values())
    opt = torch.optim.AdamW(plist, lr=1e-5, weight_decay=0.01)
    torch.nn.utils.clip_grad_norm_(plist, 1.0)
    opt.step()
    dt = time.time() - t0
    return dict(value=seq / dt, unit="tokens/s", cores=threads, kind="port",
                sample=f"1 full-param fp32 train step (oracle: fwd+bwd+clip+AdamW, semantic loss) at B=1,S={seq}, {dt:.1f}s",
                loss=float(total)), (tokens, mask, targets)


def main():
    a = parse()
    # stdout carries exactly ONE line (the JSON result): everything incidental - logger handlers created from here on,
    # library prints - goes to stderr
    real_stdout, sys.stdout = sys.stdout, sys.stderr
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if "CSM_BENCH_FORCE_DEVICE" in os.environ:      # rehearsal of the multi-rank path on a one-GPU box (with gloo)
        local = int(os.environ["CSM_BENCH_FORCE_DEVICE"])
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        backend = os.environ.get("CSM_BENCH_BACKEND", "nccl")        # nccl == RCCL on ROCm
        if backend == "nccl":
            from csm.training.dp import init_nccl
            init_nccl(rank, world, local)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

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
    model = Model(args, device=f"cuda:{local}", seed=0)          # random init, same on every rank
    model.acoustic_mode = {"A": "off", "B": "all", "C": "amortized"}[a.mode]
    import tempfile
    tr = CSMTrainer("", tempfile.mkdtemp(prefix=f"csm_bench_rank{rank}_"), device=f"cuda:{local}")   # trainer wants an output dir
    tr.logger.setLevel(30)
    tr.model = model
    if a.lora:
        apply_lora_to_model(model, r=8, alpha=16.0, target_modules=["q_proj", "v_proj"])
        tr.optimizer = FusedAdamW(model, {}, lora_lr=1e-4)
        tr.grad_sync = GradSync.for_model(model) if GradSync.active() else None
    else:
        tr.prepare_optimizer()

    ds = SyntheticCSMDataset(a.batch, a.seq, args.text_vocab_size, args.audio_vocab_size, args.audio_num_codebooks,
                             seed=1234 + rank)
    batch = {k: v.cuda() for k, v in collate_variable_length([ds[i] for i in range(a.batch)]).items()}

    K_ = args.audio_num_codebooks
    cb_tokens = int(batch["input_masks"][:, :, 0].sum()) * K_ + int(batch["input_masks"][:, :, K_].sum())

    def step():
        return tr.train_step(batch, 1, True, 1.0)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gt = None
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # per-step spread (no host sync inside the loop)
    marks[0].record()
    for i in range(a.steps):
        if i == a.steps - 1:
            with GemmTimer() as gt:
                loss, det = step()
        else:
            loss, det = step()
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
    ms = dt / a.steps * 1e3
    tokens_per_s = world * a.batch * a.seq / (dt / a.steps)

    if rank == 0:
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps - 1)) or [ms]   # last step carries the GEMM timers
        pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 3)   # noqa: E731
        kinds = gt.summary() if gt is not None else {}
        bf = {k: v for k, v in kinds.items() if k.endswith("_bf16")}
        dom = max(bf.items(), key=lambda kv: kv[1]["time_ms"]) if bf else (None, None)
        roof = None
        if dom[0] is not None:
            traffic, traffic_src = pmc_traffic(dom[0]) if (a.batch, a.seq, a.mode, a.lora, a.tiny) == (4, 2048, "C", False, False) else (None, None)
            roof = {"bound": "mfma", "kernel": f"gemm_kernel<{dom[0]}>", "achieved": round(dom[1]["tflops"], 2),
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dom[1]["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                    "operand_bytes_per_launch": round(dom[1]["operand_bytes_per_launch"]),
                    "peak_note": "dense bf16 MFMA peak of the micro-architecture guide; the chip is power limited on non-zero data: a "
                                 "register-only MFMA loop reaches 1.56-1.85 PFLOP/s on random bf16 operands (tools/probes/mfma_shape_probe.hip)",
                    "avg_launch_us": round(dom[1]["avg_us"], 2), "launches_per_step": dom[1]["launches"],
                    "all_gemm_variants": {k: {"tflops": round(v["tflops"], 1), "ms_per_step": round(v["time_ms"], 2),
                                              "launches": v["launches"]} for k, v in kinds.items()}}
        out = {
            "metric": "train tokens/sec (text+audio) CSM-1B bf16 seq2048", "value": round(tokens_per_s, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("tiny-plumbing" if a.tiny else "CSM-1B") + (" LoRA r=8 q_proj/v_proj" if a.lora else " full-param")
                       + f" bf16 train step, seq={a.seq}, batch={a.batch}/GPU, loss mode {a.mode}"
                       + (" (semantic CE + depth decoder on 1/16 of frames)" if a.mode == "C" else ""),
                       "global_batch": world * a.batch, "seq_len": a.seq, "parallelism": f"dp{world}"},
            "codebook_tokens_per_s": round(world * cb_tokens / (dt / a.steps), 1),   # secondary: 32 per audio frame + 1 per text token
            "mfma_utilisation_step": round(sum(v["flop"] for v in kinds.values()) / (dt / a.steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
            if kinds else None,                                # GEMM FLOP of one step / step time / dense peak (attention not counted)
            "hbm_kernel": gt.adamw_summary() if gt is not None else None,
            "loss": float(loss), "step_ms": {"p10": pct(0.10), "p50": pct(0.50), "p90": pct(0.90)}, "roofline": roof,
        }
        if not a.no_cpu_baseline and world == 1:
            from oracle import csm_oracle as O
            cb, (ct, cm, cg) = cpu_baseline(model, (O.tiny_cfg if a.tiny else O.csm_1b_cfg), min(a.cpu_seq, a.seq), 4321, 100.0, 1.0)
            mode = model.acoustic_mode
            model.acoustic_mode = "off"
            with torch.no_grad():
                gl, _ = compute_loss(model, ct, cm, cg, 100.0, 1.0)
            model.acoustic_mode = mode
            cpu_loss = cb.pop("loss")
            out["cpu_baseline"] = cb
            out["parity_check"] = {"what": "loss of the same weights/batch: HIP bf16 path vs CPU fp32 oracle", "cpu_loss": cpu_loss,
                                   "gpu_loss": float(gl), "rel_diff": abs(float(gl) - cpu_loss) / abs(cpu_loss)}
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
