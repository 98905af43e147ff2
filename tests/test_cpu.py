"""CPU-only suite: oracle vs golden fixtures, host logic, C-ABI export check, 2-rank gloo data-parallel logic."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import csm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TINY = O.tiny_cfg()


def _gold():
    return np.load(os.path.join(GOLD, "golden_small.npz")), json.load(open(os.path.join(GOLD, "golden_meta.json")))


# ----------------------------------------------------------------------------- oracle vs fixtures
def test_oracle_reproduces_reference_loss():
    z, meta = _gold()
    params = O.init_params(TINY, seed=11)
    tokens, mask, targets = (torch.from_numpy(z[k]) for k in ("tokens", "mask", "targets"))
    total, det = O.compute_loss(params, TINY, tokens, mask, targets, 100.0, 1.0, acoustic_rows="off")
    assert abs(float(total) - meta["compute_loss_ref"]) <= 1e-5 * meta["compute_loss_ref"]
    assert abs(float(det["semantic_loss"]) - meta["semantic_loss_ref"]) <= 1e-5 * meta["semantic_loss_ref"]
    assert torch.equal(O.embed_masked_sum(params, TINY, tokens, mask), torch.from_numpy(z["embed_sum"]))


def test_oracle_train_step_fixture():
    z, meta = _gold()
    params = {k: v.requires_grad_(True) for k, v in O.init_params(TINY, seed=11).items()}
    tokens, mask, targets = (torch.from_numpy(z[k]) for k in ("tokens", "mask", "targets"))
    B, S = tokens.shape[:2]
    rows = torch.arange(0, B * (S - 1), meta["train_step"]["rows_stride"])
    total, det = O.compute_loss(params, TINY, tokens, mask, targets, 100.0, 1.0, acoustic_rows=rows)
    total.backward()
    assert abs(float(total) - meta["train_step"]["total"]) <= 1e-5 * meta["train_step"]["total"]
    assert abs(float(det["acoustic_loss"]) - meta["train_step"]["acoustic"]) <= 1e-5 * meta["train_step"]["acoustic"]
    for key in z.files:
        if key.startswith("grad::"):
            g = params[key[6:]].grad
            g = g[..., :64] if g.dim() > 1 else g
            assert torch.allclose(g, torch.from_numpy(z[key]), rtol=1e-4, atol=1e-6), key


def test_oracle_sampler_and_rvq_fixture():
    z, meta = _gold()
    out = O.sample_topk(torch.from_numpy(z["sampler_logits"]), 50, 0.9, torch.from_numpy(z["sampler_q"]))
    assert torch.equal(out, torch.from_numpy(z["sampler_out"]))
    g = torch.Generator().manual_seed(21)
    cbs = torch.randn(8, 2048, 256, generator=g)
    x = torch.randn(40, 256, generator=g) * 4
    codes = O.rvq_encode(x, cbs)
    assert torch.equal(codes, torch.from_numpy(z["rvq_codes"]))
    dec = O.rvq_decode(codes, cbs)
    assert torch.equal(dec[:4], torch.from_numpy(z["rvq_decode_head"]))
    # domain properties: decoding then re-encoding the semantic layer is a fixed point; residual norm shrinks
    assert torch.equal(O.rvq_encode(cbs[0][codes[0]], cbs[:1], 1)[0], codes[0])
    assert (x - dec).norm() < (x - cbs[0][codes[0]]).norm() + 1e-3


def test_oracle_generate_frames_fixture():
    z, meta = _gold()
    params = O.init_params(TINY, seed=11)
    tokens, mask = torch.from_numpy(z["tokens"]), torch.from_numpy(z["mask"])
    K = TINY.n_codebooks
    all_t, all_m = tokens[:1, :9], mask[:1, :9]
    for step in range(3):
        torch.manual_seed(1000 + step)
        qs = [torch.empty(1, TINY.audio_vocab).exponential_(1) for _ in range(K)]
        with torch.no_grad():
            f = O.generate_frame(params, TINY, all_t, all_m, 0.9, 10, qs)
        assert f[0].tolist() == meta["generate_frames"][step]
        nxt = torch.cat([f.long(), torch.zeros(1, 1, dtype=torch.long)], dim=1).unsqueeze(1)
        nm = torch.cat([torch.ones(1, K, dtype=torch.bool), torch.zeros(1, 1, dtype=torch.bool)], dim=1).unsqueeze(1)
        all_t, all_m = torch.cat([all_t, nxt], 1), torch.cat([all_m, nm], 1)


def test_oracle_rope_properties():
    t = O.rope_table(64, 64)
    x = torch.randn(1, 64, 2, 64)
    pos = torch.arange(64).unsqueeze(0)
    y = O.rope(x, t, pos)
    assert torch.allclose(y.norm(dim=-1), x.norm(dim=-1), rtol=1e-5)          # rotations preserve pair norms
    assert torch.allclose(y[:, 0], x[:, 0])                                   # position 0 is the identity
    f = O.llama3_inv_freq(64)
    assert abs(float(f[0]) - 1.0) < 1e-12 and float(f[-1]) < 1.0 / 500_000 ** (62 / 64) / 31.9  # low freqs / 32


# ----------------------------------------------------------------------------- host logic
def test_library_exports_every_declared_symbol():
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "csm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(csm_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("csm_stream_t")
    from csm import hip
    assert set(hip.EXPORTS) == declared, (sorted(declared - set(hip.EXPORTS)), sorted(set(hip.EXPORTS) - declared))
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.lib.csm_abi_version() == 3


def test_only_one_library_in_the_tree():
    """What ships to the GPU box is the tree: a second shared library in it (an A/B build under tools/probes/build/, a stale copy)
    is one CSM_HIP_LIB typo away from being the library under test.  The probe scripts build theirs into
    tools/probes/build/abl/ and they are deleted before a round ends."""
    import glob
    libs = [p for p in glob.glob(os.path.join(ROOT, "**", "*.so"), recursive=True) if "/.git/" not in p and "/gpurun_out/" not in p]
    assert [os.path.relpath(p, ROOT) for p in libs] == ["csm-train-pytorch_amd/csm/hip/libcsm_hip.so"], libs


def test_product_path_has_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from csm import hip
    from csm.models.model import Model, ModelArgs
    m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", 300, 67, 4))
    with pytest.raises(RuntimeError):
        m.to("cpu")
    with pytest.raises(hip.CsmHipError):
        m.to("cuda")
    pkg = os.path.join(ROOT, "csm-train-pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle's", "").replace("the oracle does", ""), f"{f} mentions the oracle"


def test_model_layout_matches_reference_inventory():
    from csm.models.model import Model, ModelArgs, llama3_rope_table
    m = Model(ModelArgs("llama-1B", "llama-100M", 128256, 2051, 32))
    views = m._views(torch.empty(0, dtype=torch.bfloat16).new_empty(0).expand(0)) if False else None
    shapes = O.param_shapes(O.csm_1b_cfg())
    n_ref = sum(int(np.prod(s)) for s in shapes.values())
    assert n_ref == 1_552_791_552
    assert m._numel >= n_ref and m._numel - n_ref < 3_000_000       # only vocabulary padding / alignment on top
    off, n = m.group_range("backbone")
    assert off == 0 and n >= 973_146_112
    # every slot 128-byte aligned, groups contiguous and ordered like the optimizer's LR groups
    assert all(s.offset % 64 == 0 for s in m._slots.values())
    order = [m.group_range(g)[0] for g in ("backbone", "decoder", "embeddings", "other")]
    assert order == sorted(order)
    assert torch.equal(llama3_rope_table(2048, 64, 500000.0, 32.0), O.rope_table(2048, 64))
    assert torch.equal(llama3_rope_table(64, 128, 500000.0, 32.0), O.rope_table(64, 128))


def test_collate_and_synthetic_dataset():
    from csm.data import SyntheticCSMDataset, collate_variable_length
    ds = SyntheticCSMDataset(4, 64, n_codebooks=32)
    it = ds[1]
    assert it["input_tokens"].shape == (64, 33) and it["input_masks"].dtype == torch.bool
    text_rows = it["input_masks"][:, -1]
    assert (it["input_masks"][text_rows, :-1] == 0).all() and (it["input_masks"][~text_rows, :-1] == 1).all()
    assert torch.equal(ds[1]["input_tokens"], it["input_tokens"])   # deterministic per index
    short = {k: v[:40] for k, v in ds[2].items()}
    b = collate_variable_length([it, short])
    assert b["input_tokens"].shape == (2, 64, 33) and (b["input_tokens"][1, 40:] == 0).all()
    assert b["target_audio_tokens"].shape == (2, 64, 32)
    gb = ds.get_batch(0, 2)
    assert gb["input_tokens"].shape == (2, 64, 33)


# ----------------------------------------------------------------------------- data parallel (gloo, 2 ranks)
def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
    from csm.training.dp import GradSync
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    buckets = {("backbone", 1): [(0, 300)], ("backbone", 0): [(300, 300)], ("other", -1): [(600, 200)], ("embeddings", -1): [(800, 200)]}
    gs = GradSync(flat, buckets)
    # accumulation micro-batch: nothing may be communicated
    gs.arm(False)
    gs.on_ready("backbone", 1)
    gs.finish()
    assert torch.equal(flat, torch.arange(1000, dtype=torch.float32) * (rank + 1))
    # boundary micro-batch: layers announce back to front, the rest is swept up by finish()
    gs.arm(True)
    gs.on_ready("backbone", 1)
    gs.on_ready("backbone", 0)
    gs.on_ready("backbone", 0)   # double announce is ignored
    gs.finish()
    expect = torch.arange(1000, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(flat, expect) and gs.launch_log == [("backbone", 1), ("backbone", 0), ("other", -1), ("embeddings", -1)]
    q.put((rank, ok, gs.launch_log))
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res


def _zero1_worker(rank, world, port, q):
    """ZeRO-1 exchange of training/dp.py on CPU tensors over gloo: reduce-scatter into the shard buffer, a stand-in
    element-wise optimiser on this rank's pieces only, in-place parameter all-gather - must leave every rank with exactly the
    parameters the all-reduce + full update leaves (the update is element-wise, the reduced gradients are the same)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
    from csm.training.dp import GradSync
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000 + 24                                     # slices that do not divide by 8 * world: replicated tails appear
    buckets = {("backbone", 1): [(0, 296)], ("backbone", 0): [(296, 312)], ("other", -1): [(608, 200)], ("embeddings", -1): [(808, 216)]}
    g0 = torch.Generator().manual_seed(5)
    param0 = torch.randn(n, generator=g0)

    def grads(step, r):
        return torch.randn(n, generator=torch.Generator().manual_seed(100 * step + r))

    def update(p, g):                                 # any element-wise rule will do
        p.sub_(0.1 * g + 0.01 * p)

    # reference: all-reduce, every rank updates everything
    ref = param0.clone()
    flat = torch.zeros(n)
    gs = GradSync(flat, buckets)
    for step in range(3):
        flat.copy_(grads(step, rank))
        gs.arm(True)
        gs.on_ready("backbone", 1)
        gs.finish()
        update(ref, flat)
    # ZeRO-1
    par = param0.clone()
    flat2 = torch.zeros(n)
    gz = GradSync(flat2, buckets, zero1=True, flat_param=par)
    gz.plan_shards()
    owned = sum(p.my_n for p in gz.pieces)
    sharded = sum(p.chunk for p in gz.pieces)
    for step in range(3):
        flat2.copy_(grads(step, rank))
        gz.arm(True)
        gz.on_ready("backbone", 1)
        gz.finish()
        for p in gz.pieces:
            update(par[p.my_off:p.my_off + p.my_n], p.grad)
        gz.gather_params()
        gz.wait_params(None, None)
    cover = torch.zeros(n, dtype=torch.int32)
    for p in gz.pieces:
        cover[p.off:p.off + p.n] += 1
    q.put((rank, dict(equal=bool(torch.equal(par, ref)), cover_ok=bool((cover == 1).all()), owned=owned, sharded=sharded,
                      tails=sum(p.n for p in gz.pieces if not p.chunk), log=gz.launch_log[:4])))
    dist.destroy_process_group()


def test_zero1_exchange_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_zero1_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    for r in (0, 1):
        assert res[r]["equal"], "ZeRO-1 (reduce-scatter, sharded update, all-gather) must equal all-reduce + full update"
        assert res[r]["cover_ok"], "the pieces must tile the bucket slices exactly once"
        assert res[r]["tails"] > 0 and res[r]["sharded"] > 0
        assert res[r]["owned"] == res[r]["sharded"] + res[r]["tails"] and 2 * res[r]["sharded"] + res[r]["tails"] == 1024
        assert res[r]["log"] == [("backbone", 1), ("backbone", 0), ("other", -1), ("embeddings", -1)]


def _dp_helpers_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
    from types import SimpleNamespace
    from csm.training.dp import GradSync
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    # replicas that start different are made equal by the broadcast, and the checksum test sees both states
    m = SimpleNamespace(arena=(torch.arange(4096, dtype=torch.float32) * (rank + 1)).to(torch.bfloat16), lora=None)
    try:
        GradSync.assert_replicas_equal(m)
        out["diverged_detected"] = False
    except RuntimeError:
        out["diverged_detected"] = True
    GradSync.broadcast_parameters(m)
    GradSync.assert_replicas_equal(m)
    out["after_broadcast"] = torch.equal(m.arena, torch.arange(4096, dtype=torch.float32).to(torch.bfloat16))
    # logged scalars are averaged over the ranks
    out["mean"] = GradSync.mean_scalar(float(rank + 1))
    q.put((rank, out))
    dist.destroy_process_group()


def test_dp_replica_helpers_two_ranks_gloo(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_helpers_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    for r in (0, 1):
        assert res[r]["diverged_detected"] and res[r]["after_broadcast"] and res[r]["mean"] == 1.5, res


def test_checkpoint_is_written_atomically(tmp_path):
    """save_checkpoint goes through a temporary file + rename and leaves no partial file behind."""
    from types import SimpleNamespace
    from csm.training.utils import save_checkpoint
    model = SimpleNamespace(state_dict=lambda: {"w": torch.ones(3)})
    path = save_checkpoint(model, None, 1, 7, 0.5, str(tmp_path), "ck")
    names = sorted(os.listdir(tmp_path))
    assert names == ["ck_epoch1_step7.pt", "ck_latest.pt"] and os.path.basename(path) == "ck_epoch1_step7.pt"
    assert torch.load(path, weights_only=False)["global_step"] == 7


def test_optimizer_state_groups_match_by_name():
    """FusedAdamW.load_state_dict refuses a state saved for other parameter groups (freeze flags changed)."""
    from csm.training.optim import FusedAdamW
    opt = FusedAdamW.__new__(FusedAdamW)
    opt.param_groups = [dict(name="decoder", lr=1.0, weight_decay=0.0, offset=0, numel=4), dict(name="other", lr=2.0, weight_decay=0.0, offset=4, numel=4)]
    opt.state = {"decoder": {"m": torch.zeros(4)}, "other": {"m": torch.zeros(4)}}
    good = {"step": 3, "groups": [dict(name="other", lr=0.2, weight_decay=0.1, offset=4, numel=4), dict(name="decoder", lr=0.1, weight_decay=0.1, offset=0, numel=4)],
            "state": {"decoder": {"m": torch.ones(4)}, "other": {"m": torch.full((4,), 2.0)}}}
    opt.load_state_dict(good)      # saved in another ORDER: still lands on the right groups
    assert opt.step_count == 3 and opt.param_groups[0]["lr"] == 0.1 and opt.param_groups[1]["lr"] == 0.2
    assert float(opt.state["other"]["m"][0]) == 2.0
    bad = dict(good, groups=[dict(name="backbone", lr=0.2, weight_decay=0.1, offset=4, numel=4), good["groups"][1]])
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)


def test_rope_table_all_positions_matches_fixture():
    """Llama-3 scaled RoPE tables for positions 0..2047 at both head dims: the oracle's table (hash pinned by the fixture,
    which also records its distance to the Hugging Face rotary embedding) and the table the HIP path is fed."""
    import hashlib
    from csm.models.model import llama3_rope_table
    z = np.load(os.path.join(GOLD, "golden_full_layer.npz"))
    meta = json.load(open(os.path.join(GOLD, "golden_full_layer_meta.json")))
    rows = [0, 1, 2, 63, 64, 511, 1024, 2047]
    for hd in (64, 128):
        tab = O.rope_table(2048, hd)
        assert hashlib.sha256(tab.contiguous().numpy().tobytes()).hexdigest()[:16] == meta[f"rope_hd{hd}"]["sha"]
        assert meta[f"rope_hd{hd}"]["max_abs_diff_vs_hf"] < 2e-3
        assert np.array_equal(tab[rows].numpy(), z[f"rope_hd{hd}::rows"])
        prod = llama3_rope_table(2048, hd, 500000.0, 32.0)
        assert prod.shape == tab.shape and float((prod - tab).abs().max()) <= 1e-6


def test_split_master_roundtrip():
    """fp32 master <-> (bf16 working copy, 16-bit lower half): exact both ways, working copy = half-up rounding."""
    from csm.training.optim import join_master, split_master
    g = torch.Generator().manual_seed(0)
    m = torch.randn(50000, generator=g) * torch.randn(50000, generator=g).exp()
    m[:4] = torch.tensor([0.0, -0.0, 1.0, -1.0])
    b = m.view(torch.int32)
    b[10:30] = (b[10:30] & ~0xFFFF) | 0x8000                    # exact ties
    p = torch.empty(50000, dtype=torch.bfloat16)
    lo = split_master(m, p)
    assert lo.dtype == torch.int16 and torch.equal(join_master(p, lo).view(torch.int32), m.view(torch.int32))
    rne = m.to(torch.bfloat16)
    differs = p.view(torch.int16) != rne.view(torch.int16)
    ties = (m.view(torch.int32) & 0xFFFF) == 0x8000
    assert not bool((differs & ~ties).any()), "away from exact ties the working copy is the usual nearest bf16"
    assert float(((p.float() - m).abs() / m.abs().clamp_min(1e-30)).max()) <= 2.0 ** -8


def test_gradsync_bucket_plan_for_model():
    from csm.models.model import Model, ModelArgs
    from csm.training.dp import GradSync
    m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", 300, 67, 4))
    slices = []
    for name, s in m._slots.items():
        slices.append((s.offset, s.numel))
    merged = GradSync._merge(slices)
    assert len(merged) == 1 and merged[0][0] == 0      # the whole arena is one run once alignment gaps are bridged
    # a backbone layer is one contiguous slice
    lay = [(s.offset, s.numel) for n, s in m._slots.items() if n.startswith("backbone.layers.1.")]
    assert len(GradSync._merge(lay)) == 1


def test_bench_launches_and_verifies_its_own_ranks():
    """``python bench.py --gpus N`` without a torchrun environment must start N ranks itself (children of a process that
    has not touched the GPU), forward ONE JSON line that names N ranks, and fail loudly when the environment disagrees with
    ``--gpus`` - the first 8-GPU driver run must not silently measure one GPU.  (``--dry-run``: rendezvous and collectives
    over gloo on the CPU, no model.)"""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl"]["world"] == 2 and out["steps"] == 3 and out["dry_run"] is True
    # a launcher that started a different number of ranks than --gpus says: refuse
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr and not p.stdout.strip()
    # and a single-process run asked for 1 GPU inside a 2-rank environment likewise
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
