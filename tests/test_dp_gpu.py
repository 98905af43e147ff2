"""Data-parallel train step on real hardware: 2 ranks share the one GPU of the test box (gloo transports the CUDA
tensors, so the collective itself is not RCCL here; everything else - hooks, bucket order, side stream, scaling by
1/world, clip on reduced gradients - is the production path)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    for p in (ROOT, os.path.join(ROOT, "csm-train-pytorch_amd")):
        sys.path.insert(0, p)
    from oracle import csm_oracle as O
    from csm.models.model import Model, ModelArgs
    from csm.training.trainer import CSMTrainer
    from csm.training.lora import apply_lora_to_model
    from csm.training.optim import FusedAdamW
    from csm.training.dp import GradSync
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.tiny_cfg()
    m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks), device="cuda:0", seed=3)
    m.acoustic_mode = "all"
    tr = CSMTrainer("", f"/tmp/csm_dp_test_{port}_{rank}", device="cuda:0")
    tr.logger.setLevel(40)
    tr.model = m
    if mode == "lora":
        apply_lora_to_model(m, r=8, alpha=16.0, target_modules=["q_proj", "v_proj"], seed=5)
        tr.optimizer = FusedAdamW(m, {}, lora_lr=1e-3)
        tr.grad_sync = GradSync.for_model(m)
    else:
        tr.prepare_optimizer()
    assert tr.grad_sync is not None and tr.grad_sync.world_size == world
    losses = []
    for step in range(2):
        for micro in range(2):   # accumulation window of 2: only the second micro-batch communicates
            tokens, mask, targets = O.synthetic_batch(cfg, 2, 24, seed=100 + 10 * step + 2 * micro + rank)
            batch = {"input_tokens": tokens, "input_masks": mask, "target_audio_tokens": targets}
            loss, _ = tr.train_step(batch, accumulation_steps=2, is_boundary=(micro == 1), max_grad_norm=1.0)
            losses.append(float(loss))
    torch.cuda.synchronize()
    state = m.lora.arena.float().cpu() if mode == "lora" else m.arena.float().cpu()
    q.put((rank, losses, state.numpy(), list(tr.grad_sync.launch_log)))   # by value: the child may exit first
    dist.barrier()
    dist.destroy_process_group()


def _single(mode):
    for p in (ROOT, os.path.join(ROOT, "csm-train-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import csm_oracle as O
    from csm.models.model import Model, ModelArgs
    from csm.training.trainer import CSMTrainer
    from csm.training.lora import apply_lora_to_model
    from csm.training.optim import FusedAdamW
    cfg = O.tiny_cfg()
    m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks), device="cuda:0", seed=3)
    m.acoustic_mode = "all"
    tr = CSMTrainer("", "/tmp/csm_dp_test_single", device="cuda:0")
    tr.logger.setLevel(40)
    tr.model = m
    if mode == "lora":
        apply_lora_to_model(m, r=8, alpha=16.0, target_modules=["q_proj", "v_proj"], seed=5)
        tr.optimizer = FusedAdamW(m, {}, lora_lr=1e-3)
    else:
        tr.prepare_optimizer()
    for step in range(2):
        # the same 4 micro-batches the two ranks saw, as one accumulation window of 4
        k = 0
        for micro in range(2):
            for rank in range(2):
                tokens, mask, targets = O.synthetic_batch(cfg, 2, 24, seed=100 + 10 * step + 2 * micro + rank)
                batch = {"input_tokens": tokens, "input_masks": mask, "target_audio_tokens": targets}
                k += 1
                tr.train_step(batch, accumulation_steps=4, is_boundary=(k == 4), max_grad_norm=1.0)
    torch.cuda.synchronize()
    return m.lora.arena.float().cpu() if mode == "lora" else m.arena.float().cpu()


@pytest.mark.parametrize("mode", ["full", "lora"])
def test_two_rank_dp_matches_single_process(dev, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 1000 + (0 if mode == "full" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    (r0, l0, s0, log0), (r1, l1, s1, log1) = res
    s0, s1 = torch.from_numpy(s0), torch.from_numpy(s1)
    assert torch.equal(s0, s1), "replicas diverged"
    assert l0 != l1, "ranks must have seen different data"
    if mode == "full":
        # backward order: decoder layers back to front, heads, backbone layers back to front, embeddings; twice (2 steps)
        assert log0[:len(log0) // 2] == log0[len(log0) // 2:]
        first = log0[:len(log0) // 2]
        assert first.index(("decoder", 1)) < first.index(("decoder", 0)) < first.index(("other", -1)) < first.index(("backbone", 1)) \
            < first.index(("backbone", 0)) < first.index(("embeddings", -1))
        assert first[-1] == ("embeddings", "text-rows"), "text-embedding rows go through the row-list exchange, last"
    ref = _single(mode)
    err = (s0 - ref).abs().max().item()
    # Adam moves a weight by ~lr per step whatever the gradient's size, so an element whose (tiny) gradient changes
    # sign under a different bf16 summation order may differ by up to 2*lr per step: 2 steps x 2 x 1e-3
    assert err <= 2e-3 * ref.abs().max().item() + 4.2e-3, f"DP(2) vs single-process on the same 4 micro-batches: max abs diff {err}"
    frac = ((s0 - ref).abs() > 1e-3).float().mean().item()
    assert frac < 0.02, f"{frac:.3%} of the weights differ by more than one Adam step"


def _train_worker(rank, world, port, q, outdir, cap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    if cap:
        os.environ["CSM_DP_TEXT_ROWS_CAP"] = str(cap)
    for p in (ROOT, os.path.join(ROOT, "csm-train-pytorch_amd")):
        sys.path.insert(0, p)
    from oracle import csm_oracle as O
    from csm.data import SyntheticCSMDataset
    from csm.models.model import Model, ModelArgs
    from csm.training.trainer import CSMTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.tiny_cfg()
    # DIFFERENT seeds per rank: prepare_optimizer must broadcast rank 0's parameters
    m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks), device="cuda:0", seed=3 + rank)
    tr = CSMTrainer("", outdir, device="cuda:0")
    tr.model = m
    tr.num_workers = 0
    tr.prepare_optimizer()
    ds = SyntheticCSMDataset(8, 24, cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks, seed=7)
    res = {}
    if cap:
        # NO host sync between the steps: the flag of step k is looked at by arm() of step k + 2 (or by close())
        gs = tr.grad_sync
        snaps = []
        for step in range(8):
            tokens, mask, targets = O.synthetic_batch(cfg, 2, 24, seed=50 + step + 10 * rank)
            tr.train_step({"input_tokens": tokens, "input_masks": mask, "target_audio_tokens": targets}, 1, True, 1.0)
            snaps.append(m.arena.clone())
        gs.close()
        torch.cuda.synchronize()
        res.update(skipped=list(gs.skipped_steps), cap=gs.sparse["cap"], step_count=tr.optimizer.step_count,
                   moved=[bool((snaps[i] != (snaps[i - 1] if i else snaps[0])).any()) for i in range(8)],
                   state=m.arena.float().cpu().numpy(), persistent=__import__("csm.hip", fromlist=["lib"]).lib.csm_get_gemm256_persistent())
    else:
        gs = tr.grad_sync
        gs.timing = True
        tr.train(ds, batch_size=2, accumulation_steps=1, epochs=1, save_every=1)
        res["exposed"] = gs.exposed_comm_ms()
        # train() closes the exchange when it returns: hook detached, process-global GEMM switch given back
        res["closed"] = (tr.grad_sync is None and m.engine.grad_hook is None
                         and __import__("csm.hip", fromlist=["lib"]).lib.csm_get_gemm256_persistent() == 1)
        res["state"] = m.arena.float().cpu().numpy()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_loop_one_writer(dev, tmp_path):
    """CSMTrainer.train under DP: replicas start from rank 0's weights whatever their own seed was, only rank 0 writes
    checkpoints and log lines (both ranks share ONE output directory), nothing half-written stays behind."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30700 + os.getpid() % 1000
    out = str(tmp_path / "run")
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, out, 0)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert (res[0]["state"] == res[1]["state"]).all(), "replicas diverged (initial broadcast missing?)"
    files = sorted(os.listdir(out))
    assert not [f for f in files if f.endswith(".tmp")], files
    # 8 items / (2 ranks x batch 2) = 2 optimiser steps: periodic x2, epoch, final (+ their _latest twins)
    assert {"checkpoint_epoch1_step1.pt", "checkpoint_epoch1_step2.pt", "checkpoint_latest.pt", "epoch_1_epoch1_step2.pt", "final_epoch1_step2.pt"} <= set(files), files
    log = open(os.path.join(out, "training.log")).read()
    assert log.count("Training completed") == 1 and log.count("Epoch 1 completed") == 1, log
    assert len(res[0]["exposed"]) == 2 and all(x >= 0 for x in res[0]["exposed"])
    assert res[0]["closed"] and res[1]["closed"], "CSMTrainer.train must close its GradSync"


def test_text_row_exchange_overflow_drops_the_step_and_grows(dev, tmp_path):
    """More distinct text rows in a step than the exchange's capacity (start value 1 row): the device-side flag - all-reduced,
    so the same on both ranks - drops that optimiser step ON THE DEVICE (weights untouched), each step's flag reaches the
    host in its own pinned slot two steps later WITHOUT any host sync in the loop, the capacity doubles until the step's
    rows fit, training goes on, and the replicas stay bit-identical throughout."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30900 + os.getpid() % 1000
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, str(tmp_path / f"o{r}"), 1)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    a, b = res[0], res[1]
    assert a["skipped"] == b["skipped"] and a["cap"] == b["cap"], (a["skipped"], b["skipped"], a["cap"], b["cap"])
    assert a["skipped"][:2] == [0, 1], "steps 0 and 1 run before the host has seen the first flag (lag 2): both dropped"
    assert len(a["skipped"]) < 8 and a["cap"] >= 2, "the capacity must have grown until a step went through"
    # a dropped step leaves the weights exactly as they were; an applied one moves them
    for i in range(1, 8):
        assert a["moved"][i] == (i not in a["skipped"]), (i, a["moved"], a["skipped"])
    assert a["step_count"] == 8 - len(a["skipped"]), "dropped steps are un-counted (AdamW bias correction)"
    assert (a["state"] == b["state"]).all(), "replicas diverged"
    assert a["persistent"] == 1, "GradSync.close() gives the process-global GEMM schedule switch back"


def _zero_worker(rank, world, port, q, zero1, max_norm, outdir):
    """Two optimiser steps (accumulation window 2) under DP(2), all-reduce or ZeRO-1, then a checkpoint through the trainer's
    own save path; returns parameters and the optimiser state as rank 0 wrote it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    for p in (ROOT, os.path.join(ROOT, "csm-train-pytorch_amd")):
        sys.path.insert(0, p)
    from oracle import csm_oracle as O
    from csm.models.model import Model, ModelArgs
    from csm.training.trainer import CSMTrainer
    from csm.training.utils import save_checkpoint
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.tiny_cfg()
    m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks), device="cuda:0", seed=3 + rank)
    # an fp32 checkpoint, loaded before the optimiser exists: masters must start at fp32 in both modes (also across shard cuts)
    m.load_state_dict({k: v.float() * 1.003 + 1e-5 for k, v in O.init_params(cfg, seed=21).items()})
    m.acoustic_mode = "all"
    tr = CSMTrainer("", outdir, device="cuda:0")
    tr.logger.setLevel(40)
    tr.model = m
    tr.zero1 = zero1
    tr.prepare_optimizer()
    assert getattr(tr.optimizer, "sharded", False) == zero1 and tr.grad_sync.zero1 == zero1
    for step in range(2):
        for micro in range(2):
            tokens, mask, targets = O.synthetic_batch(cfg, 2, 24, seed=100 + 10 * step + 2 * micro + rank)
            tr.train_step({"input_tokens": tokens, "input_masks": mask, "target_audio_tokens": targets}, 2, micro == 1, max_norm)
    # what a later forward sees (goes through the per-bucket parameter waits)
    tokens, mask, targets = O.synthetic_batch(cfg, 2, 24, seed=999)
    with torch.no_grad():
        from csm.training.utils import compute_loss
        probe, _ = compute_loss(m, tokens, mask, targets)
    opt_sd = tr.optimizer.state_dict()               # collective when sharded
    torch.cuda.synchronize()
    res = dict(state=m.arena.float().cpu().numpy(), probe=float(probe), owned=tr.optimizer.num_owned() if zero1 else None,
               trainable=tr.optimizer.num_trainable(), log=list(tr.grad_sync.launch_log)[:6])
    if rank == 0:
        res["opt"] = {g: {k: v.numpy() for k, v in st.items()} for g, st in opt_sd["state"].items()}
        res["opt_step"] = opt_sd["step"]
        path = save_checkpoint(m, tr.optimizer, 1, 2, 0.0, outdir, "zero", optimizer_state=opt_sd if zero1 else None)
        res["ckpt"] = path
    dist.barrier()
    # resume: a fresh trainer in the same mode loads the checkpoint rank 0 wrote and lands on the same state
    m2 = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks), device="cuda:0", seed=77)
    tr2 = CSMTrainer("", outdir + "_b", device="cuda:0")
    tr2.logger.setLevel(40)
    tr2.model = m2
    tr2.zero1 = zero1
    tr2.prepare_optimizer()
    from csm.training.utils import load_checkpoint
    import glob
    load_checkpoint(sorted(glob.glob(os.path.join(outdir, "zero_epoch1_step2.pt")))[0], m2, tr2.optimizer)
    sd2 = tr2.optimizer.state_dict()
    torch.cuda.synchronize()
    res["resumed_params_equal"] = bool(torch.equal(m2.arena, m.arena))
    if rank == 0:
        res["resumed_opt_equal"] = all(torch.equal(sd2["state"][g][k], opt_sd["state"][g][k]) for g in opt_sd["state"] for k in ("master", "m", "v"))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _run_zero(tmp_path, zero1, max_norm, tag):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32100 + os.getpid() % 1000 + (7 if zero1 else 0) + (3 if max_norm else 0)
    procs = [ctx.Process(target=_zero_worker, args=(r, 2, port, q, zero1, max_norm, str(tmp_path / tag))) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(60)
    return res


def test_zero1_matches_allreduce_dp_bit_for_bit(dev, tmp_path):
    """SURVEY 8e's ZeRO-1 option (reduce-scatter -> AdamW on this rank's shards -> parameter all-gather) against the all-reduce
    path, two ranks on the one GPU (gloo carries the tensors).  The update is the same element-wise kernel on the same reduced
    gradients, so WITHOUT clipping parameters, fp32 masters and both moments must be bit-identical between the two modes -
    and the replicas bit-identical to each other; with clipping the norm is summed in a different order (shards + one scalar
    all-reduce), so the result may differ by rounding of one coefficient.  The checkpoint format is the same and resumes."""
    ar = _run_zero(tmp_path, False, 0.0, "ar")
    z = _run_zero(tmp_path, True, 0.0, "z")
    assert (z[0]["state"] == z[1]["state"]).all(), "ZeRO-1 replicas diverged"
    assert z[0]["probe"] == z[1]["probe"]
    assert (z[0]["state"] == ar[0]["state"]).all(), "ZeRO-1 parameters differ from the all-reduce path's"
    assert z[0]["probe"] == ar[0]["probe"], "a forward after the step must see the gathered parameters"
    assert z[0]["opt_step"] == ar[0]["opt_step"] == 2
    for g in ar[0]["opt"]:
        for k in ("master", "m", "v"):
            assert (z[0]["opt"][g][k] == ar[0]["opt"][g][k]).all(), f"optimiser state {g}.{k} differs between the modes"
    # each rank owns about half of the trained parameters (+ the replicated text-embedding table and tails)
    assert z[0]["owned"] < z[0]["trainable"] and z[1]["owned"] < z[1]["trainable"]
    assert z[0]["owned"] + z[1]["owned"] >= z[0]["trainable"]
    assert z[0]["resumed_params_equal"] and z[1]["resumed_params_equal"] and z[0]["resumed_opt_equal"]
    assert ar[0]["resumed_params_equal"] and ar[0]["resumed_opt_equal"]
    # with global-norm clipping: same to within the rounding of the coefficient
    arc = _run_zero(tmp_path, False, 1.0, "arc")
    zc = _run_zero(tmp_path, True, 1.0, "zc")
    assert (zc[0]["state"] == zc[1]["state"]).all(), "ZeRO-1 replicas diverged (clipped)"
    import numpy as np
    d = np.abs(zc[0]["state"] - arc[0]["state"])
    assert d.max() <= 2e-3 * np.abs(arc[0]["state"]).max() + 4.2e-3 and (d > 1e-3).mean() < 0.02, d.max()
    assert (np.abs(arc[0]["state"] - ar[0]["state"]) > 0).any(), "the clip must have been active in this test"


def _rccl_worker(port, q):
    """World size 1 over the REAL RCCL backend: no traffic leaves the GPU, but every RCCL-only code path of training/dp.py
    executes on hardware - init_nccl with the high-priority stream option, all_gather_into_tensor in the text-row exchange,
    the all-reduces on the side stream, the persistent-GEMM switch (forced on here through ``world_size``)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for p in (ROOT, os.path.join(ROOT, "csm-train-pytorch_amd")):
        sys.path.insert(0, p)
    from oracle import csm_oracle as O
    from csm.hip import lib
    from csm.models.model import Model, ModelArgs
    from csm.training import dp
    from csm.training.trainer import CSMTrainer
    torch.cuda.set_device(0)
    dp.init_nccl(0, 1, 0)
    assert dist.get_backend() == "nccl"
    cfg = O.tiny_cfg()

    def build(seed):
        m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks), device="cuda:0", seed=seed)
        m.acoustic_mode = "all"
        tr = CSMTrainer("", f"/tmp/csm_rccl_test_{port}_{seed}", device="cuda:0")
        tr.logger.setLevel(40)
        tr.model = m
        return m, tr

    # reference run: no GradSync at all
    m0, tr0 = build(3)
    tr0.prepare_optimizer()
    assert tr0.grad_sync is None
    # RCCL run: the same model and data, GradSync forced on over the 1-rank RCCL group
    m1, tr1 = build(3)
    real_active = dp.GradSync.active
    dp.GradSync.active = staticmethod(lambda: True)
    try:
        tr1.prepare_optimizer()
        gs = tr1.grad_sync
        gs.world_size = 1
        real_arm = gs.arm
        gs.arm = lambda enabled=True: (real_arm(enabled), setattr(gs, "armed", bool(enabled)))   # armed although world == 1
        for step in range(3):
            tokens, mask, targets = O.synthetic_batch(cfg, 2, 24, seed=200 + step)
            b = {"input_tokens": tokens, "input_masks": mask, "target_audio_tokens": targets}
            l0, _ = tr0.train_step(b, 1, True, 1.0)
            l1, _ = tr1.train_step(b, 1, True, 1.0)
            assert float(l0) == float(l1), (step, float(l0), float(l1))
        gs.close()
        # ZeRO-1 over the real RCCL group: reduce_scatter_tensor / in-place all_gather_into_tensor with one rank are identities
        m2, tr2 = build(3)
        tr2.zero1 = True
        tr2.prepare_optimizer()
        gz = tr2.grad_sync
        assert gz.zero1 and tr2.optimizer.sharded and tr2.optimizer.num_owned() == tr2.optimizer.num_trainable()
        gz.world_size = 1
        real_arm2 = gz.arm
        gz.arm = lambda enabled=True: (real_arm2(enabled), setattr(gz, "armed", bool(enabled)))
        m3, tr3 = build(3)
        dp.GradSync.active = real_active
        tr3.prepare_optimizer()
        dp.GradSync.active = staticmethod(lambda: True)
        for step in range(3):
            tokens, mask, targets = O.synthetic_batch(cfg, 2, 24, seed=300 + step)
            b = {"input_tokens": tokens, "input_masks": mask, "target_audio_tokens": targets}
            l2, _ = tr2.train_step(b, 1, True, 1.0)
            l3, _ = tr3.train_step(b, 1, True, 1.0)
            assert float(l2) == float(l3), ("zero1", step, float(l2), float(l3))
        gz.close()
        torch.cuda.synchronize()
        q.put(dict(equal=bool(torch.equal(m0.arena, m1.arena)), log=list(gs.launch_log)[:12], skipped=list(gs.skipped_steps),
                   backend=dist.get_backend(), persistent_after_close=lib.csm_get_gemm256_persistent(),
                   zero_equal=bool(torch.equal(m2.arena, m3.arena)), zero_log=list(gz.launch_log)[:4]))
    finally:
        dp.GradSync.active = real_active
    dist.destroy_process_group()


def test_rccl_paths_execute_on_one_rank(dev):
    """RCCL itself (not gloo) carries the DP step's collectives - with one rank, which is all a one-GPU box allows (RCCL
    refuses two ranks on one device): the all-reduces and the all_gather_into_tensor of the text-row exchange are identities,
    so three optimiser steps must leave exactly the weights of a run without GradSync."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(31900 + os.getpid() % 1000, q))
    p.start()
    res = q.get(timeout=300)
    p.join(60)
    assert res["backend"] == "nccl"
    assert res["equal"], "a 1-rank RCCL DP step must be the identity on the gradients"
    assert ("embeddings", "text-rows") in res["log"] and not res["skipped"], res
    assert res["persistent_after_close"] == 1
    assert res["zero_equal"], "1-rank ZeRO-1 over RCCL (reduce-scatter, sharded AdamW, all-gather) must equal the plain step"
