"""End-to-end parity of the HIP train / generate path against the CPU oracle and the committed golden fixtures.

Tolerances (stated per north_star): loss within 1e-3 relative of the fp32 CPU path *on the same (bf16-representable)
weights*; gradients (bf16 storage, bf16 activations) within 3e-2 of the tensor's max magnitude; sampled codebook
indices bit-exact.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import csm_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TINY = O.tiny_cfg()


def tiny_model(dev, seed=11):
    from csm.models.model import Model, ModelArgs
    m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", TINY.text_vocab, TINY.audio_vocab, TINY.n_codebooks), device="cuda")
    p32 = O.init_params(TINY, seed=seed)
    m.load_state_dict(p32)
    # what the GPU really holds: bf16-rounded weights, as fp32 for the oracle
    pq = {k: v.to(BF).float() for k, v in p32.items()}
    return m, p32, pq


def rel(a, b):
    return abs(float(a) - float(b)) / max(1e-12, abs(float(b)))


def gclose(name, got, ref, tol=3e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-20
    assert err <= tol * scale, f"{name}: max abs err {err:.4g} vs max |ref| {scale:.4g}"


def test_state_dict_names_and_roundtrip(dev):
    m, p32, pq = tiny_model(dev)
    sd = m.state_dict()
    assert list(sorted(sd.keys())) == sorted(p32.keys())
    for k in p32:
        assert tuple(sd[k].shape) == tuple(p32[k].shape), k
        assert torch.equal(sd[k].float().cpu(), pq[k]), k
    names = [n for n, _ in m.named_parameters()]
    assert sorted(names) == sorted(p32.keys())


def test_compute_loss_reference_mode(dev):
    """Mode A: exactly the reference's loss (semantic CE x100, acoustic placeholder 0)."""
    from csm.training.utils import compute_loss
    m, p32, pq = tiny_model(dev)
    z = np.load(os.path.join(GOLD, "golden_small.npz"))
    meta = json.load(open(os.path.join(GOLD, "golden_meta.json")))
    tokens, mask, targets = (torch.from_numpy(z[k]) for k in ("tokens", "mask", "targets"))
    with torch.no_grad():
        total, det = compute_loss(m, tokens, mask, targets, 100.0, 1.0)
    ref_total, ref_det = O.compute_loss(pq, TINY, tokens, mask, targets, 100.0, 1.0, acoustic_rows="off")
    assert rel(total, ref_total) < 1e-3, (float(total), float(ref_total))
    assert float(det["acoustic_loss"]) == 0.0
    # against the value the REFERENCE's own compute_loss produced in the build container (fp32 weights)
    assert rel(total, meta["compute_loss_ref"]) < 3e-3, (float(total), meta["compute_loss_ref"])
    # embedding kernel against the reference-pinned masked sum
    B, S, K1 = tokens.shape
    from csm.hip import ops
    h0 = torch.empty(B * S, TINY.backbone.dim, dtype=BF, device=dev)
    ops.embed_fwd(tokens.view(B * S, K1).to(dev), mask.view(B * S, K1).to(torch.uint8).to(dev),
                  m.block("text_embeddings.weight"), m.block("audio_embeddings.weight"), h0, TINY.audio_vocab)
    gclose("embed vs reference", h0, torch.from_numpy(z["embed_sum"]).view(B * S, -1), 1e-2)


def test_hidden_states_match_oracle(dev):
    m, p32, pq = tiny_model(dev)
    tokens, mask, _ = O.synthetic_batch(TINY, 2, 100, seed=3)
    hid = m.engine.hidden_states(tokens, mask)
    ref = O.backbone_hidden(pq, TINY, tokens, mask)
    gclose("backbone hidden", hid, ref, 3e-2)


def test_train_step_full(dev):
    """Loss (semantic + teacher-forced acoustic on a row subset), gradients and one AdamW step vs the oracle."""
    from csm.training.utils import compute_loss
    from csm.training.optim import FusedAdamW
    m, p32, pq = tiny_model(dev)
    m.acoustic_mode = "all"
    z = np.load(os.path.join(GOLD, "golden_small.npz"))
    meta = json.load(open(os.path.join(GOLD, "golden_meta.json")))
    tokens, mask, targets = (torch.from_numpy(z[k]) for k in ("tokens", "mask", "targets"))
    B, S = tokens.shape[:2]
    rows = torch.arange(0, B * (S - 1), meta["train_step"]["rows_stride"])
    opt = FusedAdamW(m, {"backbone": 1e-3, "decoder": 1e-3, "embeddings": 1e-3, "other": 1e-3})
    total, det = compute_loss(m, tokens, mask, targets, 100.0, 1.0, acoustic_rows=rows)
    total.backward()
    pt = {k: v.clone().requires_grad_(True) for k, v in pq.items()}
    rt, rd = O.compute_loss(pt, TINY, tokens, mask, targets, 100.0, 1.0, acoustic_rows=rows)
    rt.backward()
    assert rel(det["semantic_loss"], rd["semantic_loss"]) < 1e-3
    assert rel(det["acoustic_loss"], rd["acoustic_loss"]) < 1e-3
    assert rel(total, rt) < 1e-3
    assert rel(total, meta["train_step"]["total"]) < 3e-3          # fp32-weight value frozen in the fixture
    grads = dict(m.named_parameters())
    worst = 0.0
    for k, p in pt.items():
        g = grads[k].grad
        assert g is not None, k
        gclose(f"grad {k}", g, p.grad, 4e-2)
    # fixture gradients (fp32 weights, first 64 columns) agree as well
    for key in z.files:
        if key.startswith("grad::"):
            k = key[6:]
            g = grads[k].grad.float().cpu()
            ref = torch.from_numpy(z[key])
            gclose(f"golden grad {k}", g[..., :64] if g.dim() > 1 else g, ref, 6e-2)
    # clip + AdamW against the oracle restatement
    norm = opt.clip_grad_norm(1.0)
    ref_grads = [pt[k].grad.clone() for k in pt]
    rn, coef = O.clip_grad_norm(ref_grads, 1.0)
    assert rel(norm, rn) < 2e-2, (float(norm), float(rn))
    opt.step()
    masters = dict(opt.named_master())
    assert sorted(masters) == sorted(pt)
    for (k, p), g in zip(pt.items(), ref_grads):
        pr = p.detach().clone()
        O.adamw_step(pr, g, torch.zeros_like(pr), torch.zeros_like(pr), 1, 1e-3)
        # step 1 moves every weight by ~lr*sign(g) (+ decay): compare the fp32 master where the gradient is not ~0
        upd, upd_ref = masters[k].float().cpu() - p32[k], pr - pq[k]   # the master was seeded with the fp32 checkpoint values
        big = g.abs() > 0.2 * g.abs().max()   # where eps=1e-8 and bf16 gradient noise do not matter
        if big.any():
            assert (upd[big] - upd_ref[big]).abs().max().item() < 1e-4, k
        # the bf16 working copy is the rounded master (upper half of the fp32 bits, rounded half-up: optim.py)
        mb = masters[k].contiguous().view(torch.int32).cpu()
        want = (((mb + 0x8000) >> 16) & 0xFFFF).to(torch.int16)
        assert torch.equal(dict(m.named_parameters())[k].detach().contiguous().view(torch.int16).cpu(), want), k


def test_post_adamw_parameters_match_fixture(dev):
    """the parameters AFTER one CSMTrainer step (clip 1.0, four learning-rate groups x0.1 / x1 / x0.5 / x1,
    weight decay 0.01) against values produced by torch's own clip_grad_norm_ + AdamW on the oracle's fp32 gradients."""
    from csm.training.trainer import CSMTrainer
    m, p32, pq = tiny_model(dev)
    m.acoustic_mode = "all"
    z = np.load(os.path.join(GOLD, "golden_small.npz"))
    meta = json.load(open(os.path.join(GOLD, "golden_meta.json")))
    cfgp = meta["train_step"]["post_adamw"]
    tokens, mask, targets = (torch.from_numpy(z[k]) for k in ("tokens", "mask", "targets"))
    B, S = tokens.shape[:2]
    rows = torch.arange(0, B * (S - 1), meta["train_step"]["rows_stride"])
    tr = CSMTrainer("", "/tmp/csm_post_adamw", device=str(dev), learning_rate=cfgp["lr"], backbone_lr_multiplier=cfgp["multipliers"][0],
                    decoder_lr_multiplier=cfgp["multipliers"][1], embedding_lr_multiplier=cfgp["multipliers"][2], weight_decay=cfgp["weight_decay"])
    tr.logger.setLevel(40)
    tr.model = m
    tr.prepare_optimizer()
    from csm.training.utils import compute_loss
    total, _ = compute_loss(m, tokens, mask, targets, 100.0, 1.0, acoustic_rows=rows)
    total.backward()
    tr.optimizer.clip_grad_norm(cfgp["max_grad_norm"])
    tr.optimizer.step()
    masters = dict(tr.optimizer.named_master())
    checked = 0
    for key in z.files:
        if not key.startswith("post::"):
            continue
        k = key[6:]
        ref = torch.from_numpy(z[key])
        got = masters[k].float().cpu()
        got = got[..., :64] if got.dim() > 1 else got
        start = p32[k][..., :64] if p32[k].dim() > 1 else p32[k]
        gref = torch.from_numpy(z["grad::" + k])
        upd, upd_ref = got - start, ref - start
        # step 1 of Adam moves a weight by ~lr * sign(g): compare where the gradient is clearly non-zero (elsewhere eps and
        # the bf16 noise of the HIP gradient decide the sign)
        big = gref.abs() > 0.2 * gref.abs().max()
        assert big.any(), k
        lr_k = cfgp["lr"] * (cfgp["multipliers"][0] if "backbone" in k else cfgp["multipliers"][1] if "decoder" in k else
                             cfgp["multipliers"][2] if "embeddings" in k else cfgp["multipliers"][3])
        assert (upd[big] - upd_ref[big]).abs().max().item() < 0.1 * lr_k, (k, (upd[big] - upd_ref[big]).abs().max().item(), lr_k)
        assert (upd_ref[big].abs() > 0.5 * lr_k).all(), k      # the fixture really moved by about one learning rate
        checked += 1
    assert checked >= 8


def test_grad_accumulation_and_freeze(dev):
    from csm.training.utils import compute_loss
    m, _, _ = tiny_model(dev)
    tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=5)
    m.ensure_grads()
    t1, _ = compute_loss(m, tokens, mask, targets)
    (t1 / 2).backward()
    g1 = m.grad_arena.clone()
    t2, _ = compute_loss(m, tokens, mask, targets)
    (t2 / 2).backward()
    gclose("two half steps == one full", m.grad_arena, 2 * g1.float(), 2e-2)
    m.grad_arena.zero_()
    m.trainable.update(backbone=False, embeddings=False)
    t3, _ = compute_loss(m, tokens, mask, targets)
    t3.backward()
    o, n = m.group_range("backbone")
    assert float(m.grad_arena[o:o + n].abs().max()) == 0.0
    o, n = m.group_range("other")
    assert float(m.grad_arena[o:o + n].abs().max()) > 0.0


def test_ignore_index_padding(dev):
    """Targets padded with IGNORE_INDEX (csm.data.collate_variable_length(target_pad=-100)): padded frames leave both
    loss terms and the mean runs over labelled rows, as torch's cross_entropy(ignore_index=-100) does in the oracle."""
    from csm.data.training_data import IGNORE_INDEX
    from csm.training.utils import compute_loss
    m, p32, pq = tiny_model(dev)
    m.acoustic_mode = "all"
    tokens, mask, targets = O.synthetic_batch(TINY, 3, 24, seed=11)
    targets = targets.clone()
    targets[0, 15:] = IGNORE_INDEX
    targets[2, 7:] = IGNORE_INDEX
    with pytest.raises(ValueError):
        compute_loss(m, tokens, mask, targets)                     # negative labels are an error unless an ignore index is set
    m.target_ignore_index = IGNORE_INDEX
    m.ensure_grads()
    total, det = compute_loss(m, tokens, mask, targets)
    total.backward()
    S = tokens.shape[1]
    valid = (targets[:, :S - 1, 0] >= 0).reshape(-1).nonzero().squeeze(1)
    pr = {k: v.clone().requires_grad_(True) for k, v in pq.items()}
    rt, rdet = O.compute_loss(pr, TINY, tokens, mask, targets, acoustic_rows=valid)
    rt.backward()
    assert rel(total, rt) < 1e-3, (float(total), float(rt))
    assert rel(det["semantic_loss"], rdet["semantic_loss"]) < 1e-3 and rel(det["acoustic_loss"], rdet["acoustic_loss"]) < 1e-3
    grads = dict(m.named_parameters())
    for k in ("codebook0_head.weight", "backbone.layers.0.attn.q_proj.weight", "decoder.layers.1.mlp.w2.weight", "projection.weight"):
        gclose(k, grads[k].grad, pr[k].grad, 5e-2)
    # the same batch with zero padding (reference behaviour) gives a different, larger-denominator loss
    m.target_ignore_index = None
    z, _ = compute_loss(m, tokens, mask, targets.clamp(min=0))
    assert abs(float(z) - float(total)) > 1e-3


def test_lazy_zero_grad_matches_eager(dev):
    """``step(zero_grad="lazy")`` (dense gradient groups are overwritten by the next backward instead of being zeroed) must
    leave exactly the parameters ``zero_grad=True`` leaves - across accumulation windows, a step in which the decoder gets
    no gradient (its stale buffer has to be cleared), a frozen group and a step without any backward."""
    from csm.training.optim import FusedAdamW
    from csm.training.utils import compute_loss
    results = []
    for mode in (True, "lazy"):
        m, _, _ = tiny_model(dev)
        opt = FusedAdamW(m, {"backbone": 1e-3, "decoder": 2e-3, "embeddings": 5e-4, "other": 1e-3})
        plan = [("all", 1, {}), ("off", 1, {}), ("all", 2, {}), ("amortized", 1, {"decoder": False}), ("all", 1, {"decoder": True})]
        for step, (acoustic, n_micro, flags) in enumerate(plan):
            m.acoustic_mode = acoustic
            m.trainable.update(flags)
            for k in range(n_micro):
                tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=100 + 10 * step + k)
                rows = torch.arange(0, 2 * 23, 3) if acoustic == "amortized" else None
                total, _ = compute_loss(m, tokens, mask, targets, acoustic_rows=rows)
                (total / n_micro).backward()
            opt.clip_grad_norm(1.0)
            opt.step(zero_grad=mode)
        opt.clip_grad_norm(1.0)          # a step with no backward in between: gradients are zero either way
        opt.step(zero_grad=mode)
        results.append(m.arena.clone())
    assert torch.equal(results[0], results[1])


def test_lora_step(dev):
    from csm.training.lora import apply_lora_to_model, merge_lora_weights
    from csm.training.utils import compute_loss
    m, p32, pq = tiny_model(dev)
    m.acoustic_mode = "all"
    apply_lora_to_model(m, r=8, alpha=16.0, target_modules=["q_proj", "v_proj", "w2"], seed=1)
    with torch.no_grad():   # make B non-zero so that every gradient path is exercised
        g = torch.Generator(device=dev).manual_seed(2)
        for ad in m.lora.adapters.values():
            ad.B.copy_((torch.randn(ad.B.shape, generator=g, device=dev) * 0.05).to(BF))
    tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=6)
    total, det = compute_loss(m, tokens, mask, targets)
    total.backward()
    lora = {k: v.detach().float().cpu().requires_grad_(True) for k, v in m.get_lora_params().items()}
    rt, _ = O.compute_loss(pq, TINY, tokens, mask, targets, acoustic_rows=None, lora=lora, lora_scaling=2.0)
    rt.backward()
    assert rel(total, rt) < 1e-3, (float(total), float(rt))
    for ad in m.lora.adapters.values():
        gclose(f"{ad.name}.lora_A grad", ad.gA, lora[f"{ad.name}.lora_A"].grad, 5e-2)
        gclose(f"{ad.name}.lora_B grad", ad.gB, lora[f"{ad.name}.lora_B"].grad, 5e-2)
    assert m.grad_arena is None or float(m.grad_arena.abs().max()) == 0.0, "base weights are frozen"
    # merged weights reproduce the adapted forward
    with torch.no_grad():
        before, _ = compute_loss(m, tokens, mask, targets)
        merge_lora_weights(m)
        m.lora = None
        after, _ = compute_loss(m, tokens, mask, targets)
    assert rel(after, before) < 2e-3


def test_lora_fused_groups_match_per_adapter_path_and_oracle(dev):
    """Adapters on all seven projections: the grouped path (one K-extension operand pair per fused projection, RoPE and
    SwiGLU epilogues on the sum, csm_gemm_bf16_kext) against the per-adapter products on the same storage and against the
    oracle's LoRALinear restatement (reference lora.py:85-105)."""
    import csm.engine as E
    from csm.training.lora import apply_lora_to_model
    from csm.training.utils import compute_loss
    mods = ["q_proj", "k_proj", "v_proj", "output_proj", "w1", "w2", "w3"]
    tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=16)
    res = {}
    for fuse in (True, False):
        m, p32, pq = tiny_model(dev)
        m.acoustic_mode = "all"
        apply_lora_to_model(m, r=4, alpha=8.0, target_modules=mods, seed=11)
        with torch.no_grad():
            g = torch.Generator(device=dev).manual_seed(12)
            for ad in m.lora.adapters.values():
                ad.B[:, :4].copy_((torch.randn(ad.B.shape[0], 4, generator=g, device=dev) * 0.05).to(BF))
        old = E.LORA_FUSE
        E.LORA_FUSE = fuse
        try:
            total, _ = compute_loss(m, tokens, mask, targets)
            total.backward()
        finally:
            E.LORA_FUSE = old
        res[fuse] = (float(total), m.lora.grad_arena.float().clone(), m)
    (tf, gf, m), (tu, gu, _) = res[True], res[False]
    assert abs(tf - tu) <= 2e-3 * abs(tu), (tf, tu)
    scale = gu.abs().max().item()
    assert (gf - gu).abs().max().item() <= 4e-2 * scale, ((gf - gu).abs().max().item(), scale)
    lora = {k: v.detach().float().cpu().requires_grad_(True) for k, v in m.get_lora_params().items()}
    rt, _ = O.compute_loss(pq, TINY, tokens, mask, targets, acoustic_rows=None, lora=lora, lora_scaling=2.0)
    rt.backward()
    assert rel(torch.tensor(tf), rt) < 1e-3, (tf, float(rt))
    for ad in m.lora.adapters.values():
        gclose(f"{ad.name}.lora_A grad", ad.gA[:4], lora[f"{ad.name}.lora_A"].grad, 5e-2)
        gclose(f"{ad.name}.lora_B grad", ad.gB[:, :4], lora[f"{ad.name}.lora_B"].grad, 5e-2)
    # nothing outside the adapters' blocks of a group's Bx / At may move: their gradients are exactly zero
    for G in m.lora.groups.values():
        if G.mask is not None:
            assert float((G.gBx.float() * (1 - G.mask.float())).abs().max()) == 0.0, G.name
        used = len(G.adapters) * m.lora.r_pad
        assert float(G.gAt[:, used:].abs().max() if used < G.kx else 0.0) == 0.0 and float(G.gBx[:, used:].abs().max() if used < G.kx else 0.0) == 0.0
        for j in range(len(G.adapters)):
            c0 = j * m.lora.r_pad
            assert float(G.gAt[:, c0 + 4:c0 + 8].abs().max()) == 0.0 and float(G.gBx[:, c0 + 4:c0 + 8].abs().max()) == 0.0, "rank padding"
    # merging every adapter into its (possibly row-interleaved) frozen weight reproduces the adapted forward
    from csm.training.lora import merge_lora_weights
    with torch.no_grad():
        before, _ = compute_loss(m, tokens, mask, targets)
        merge_lora_weights(m)
        m.lora = None
        after, _ = compute_loss(m, tokens, mask, targets)
    assert rel(after, before) < 2e-3, (float(after), float(before))


def test_lora_rank4_mlp_adapters_and_trainer_step(dev, tmp_path):
    """Row a9 + the r = 4 case the reference documents: ``CSMLoRATrainer.train_step`` (loss -> gradients of the LoRA
    parameters only -> clip g * max_norm / (||g|| + 1e-6) when ||g|| > max_norm -> Adam, no weight decay; reference
    lora_trainer.py:374-457, mlx_trainer.py:688-731) against the oracle, with rank 4 (stored padded to 8) and adapters on
    q / v / w1 / w3 - the w1 / w3 pair goes through the SwiGLU epilogue's residual port."""
    from csm.training.lora_trainer import CSMLoRATrainer
    from csm.data import SyntheticCSMDataset
    m, p32, pq = tiny_model(dev)
    m.acoustic_mode = "all"
    lr = 1e-3
    tr = CSMLoRATrainer("", str(tmp_path), learning_rate=lr, lora_r=4, lora_alpha=8.0,
                        target_modules=["q_proj", "v_proj", "w1", "w3"], device=str(dev), model=m)
    tr.logger.setLevel(40)
    lo = m.lora
    assert lo.r == 4 and lo.r_pad == 8
    with torch.no_grad():
        g = torch.Generator(device=dev).manual_seed(12)
        for ad in lo.adapters.values():
            ad.B[:, :4].copy_((torch.randn(ad.B.shape[0], 4, generator=g, device=dev) * 0.05).to(BF))
    names = dict(lo.named_tensors())
    assert all(v.shape[0] == 4 for k, v in names.items() if k.endswith("lora_A")) and all(v.shape[1] == 4 for k, v in names.items() if k.endswith("lora_B"))
    before = {k: v.detach().float().cpu().clone() for k, v in names.items()}
    tr.prepare_optimizer()
    tr.max_grad_norm = 1.0
    tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=16)
    loss = tr.train_step({"input_tokens": tokens.numpy(), "input_masks": mask.numpy(), "target_audio_tokens": targets.numpy()})
    assert loss.dim() == 0
    # oracle: same loss, same clip rule, Adam without decay
    lora = {k: v.clone().requires_grad_(True) for k, v in before.items()}
    rt, _ = O.compute_loss(pq, TINY, tokens, mask, targets, 100.0, 1.0, acoustic_rows=None, lora=lora, lora_scaling=2.0)
    rt.backward()
    assert rel(loss, rt) < 1e-3, (float(loss), float(rt))
    grads = {k: v.grad.clone() for k, v in lora.items()}
    norm = float(torch.sqrt(sum((gv.double() ** 2).sum() for gv in grads.values())))
    assert norm > 1.0, "the clip must be live in this test"
    after = {k: v.detach().float().cpu() for k, v in lo.named_tensors()}
    checked = 0
    for k, gv in grads.items():
        gv = gv * (1.0 / (norm + 1e-6))
        ref = before[k].clone()
        O.adamw_step(ref, gv, torch.zeros_like(ref), torch.zeros_like(ref), 1, lr, weight_decay=0.0)
        big = gv.abs() > 0.2 * gv.abs().max()
        # bf16 working copy of the updated parameter: one learning rate of movement, compared at bf16 resolution
        tol = 0.15 * lr + 2.0 ** -8 * ref[big].abs().max().item()
        assert ((after[k] - before[k])[big] - (ref - before[k])[big]).abs().max().item() <= tol, k
        checked += 1
    assert checked == 2 * len(lo.adapters)
    # the padding of the rank-4 adapters (rows 4..7 of A, columns 4..7 of B) is still exactly zero after the step
    for ad in lo.adapters.values():
        assert float(ad.A[4:].abs().max()) == 0.0 and float(ad.B[:, 4:].abs().max()) == 0.0 and float(ad.gA[4:].abs().max()) == 0.0
    # train() keeps the reference's return contract: best_loss, inf when no validation ran (mlx_trainer.py:733-876)
    ds = SyntheticCSMDataset(4, 24, TINY.text_vocab, TINY.audio_vocab, TINY.n_codebooks, seed=2)
    assert tr.train(ds, batch_size=2, epochs=1, save_every=1) == float("inf")
    assert os.path.exists(os.path.join(str(tmp_path), "checkpoint_step_2.safetensors"))
    from safetensors.torch import load_file
    sd = load_file(os.path.join(str(tmp_path), "checkpoint_step_2.safetensors"))
    assert all(tuple(sd[k].shape) == tuple(v.shape) for k, v in lo.named_tensors())


def test_lora_dropout_and_bias_step(dev):
    """lora_dropout > 0 and lora_use_bias (reference lora.py:85-102): the oracle is given the very masks the HIP path
    drew (regenerated from each adapter's seed), so loss and every adapter gradient can be compared exactly as above."""
    from csm.hip import ops
    from csm.training.lora import apply_lora_to_model
    from csm.training.utils import compute_loss
    m, p32, pq = tiny_model(dev)
    m.acoustic_mode = "all"
    p = 0.25
    apply_lora_to_model(m, r=8, alpha=16.0, dropout=p, target_modules=["q_proj", "output_proj", "w1"], use_bias=True, seed=3)
    with torch.no_grad():
        g = torch.Generator(device=dev).manual_seed(4)
        for ad in m.lora.adapters.values():
            ad.B.copy_((torch.randn(ad.B.shape, generator=g, device=dev) * 0.05).to(BF))
            ad.bias.copy_((torch.randn(ad.bias.shape, generator=g, device=dev) * 0.02).to(BF))
    tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=8)
    total, _ = compute_loss(m, tokens, mask, targets)
    total.backward()
    lora = {k: v.detach().float().cpu().requires_grad_(True) for k, v in m.get_lora_params().items()}
    assert any(k.endswith("lora_bias") for k in lora)
    Bt, St = tokens.shape[:2]
    n_rows = {"backbone": Bt * St, "decoder": Bt * (St - 1) * m.args.audio_num_codebooks}   # decoder: every frame trains
    for ad in m.lora.adapters.values():
        assert ad._draw is not None
        in_f = ad.A.shape[1]
        rows = n_rows[ad.name.split(".")[0]]
        ones = torch.ones(rows, in_f, dtype=BF, device=dev)
        keep = ops.dropout_bf16(ones, torch.empty_like(ones), p, ad._draw) != 0
        frac = keep.float().mean().item()
        assert abs(frac - (1 - p)) < 0.02, (ad.name, frac)
        lora[f"{ad.name}.lora_dropout_scale"] = keep.float().cpu() / (1 - p)
    rt, _ = O.compute_loss(pq, TINY, tokens, mask, targets, acoustic_rows=None, lora=lora, lora_scaling=2.0)
    rt.backward()
    assert rel(total, rt) < 1e-3, (float(total), float(rt))
    for ad in m.lora.adapters.values():
        gclose(f"{ad.name}.lora_A grad", ad.gA, lora[f"{ad.name}.lora_A"].grad, 5e-2)
        gclose(f"{ad.name}.lora_B grad", ad.gB, lora[f"{ad.name}.lora_B"].grad, 5e-2)
        gclose(f"{ad.name}.lora_bias grad", ad.gbias, lora[f"{ad.name}.lora_bias"].grad, 5e-2)
    # a second forward draws fresh masks; with training off the adapters are deterministic
    d0 = m.lora.draws
    compute_loss(m, tokens, mask, targets)
    assert m.lora.draws > d0
    m.lora.training = False
    with torch.no_grad():
        a1, _ = compute_loss(m, tokens, mask, targets)
        a2, _ = compute_loss(m, tokens, mask, targets)
    assert float(a1) == float(a2)


@pytest.mark.parametrize("use_graph", [True, False])
def test_generate_frame_matches_reference_fixture(dev, use_graph):
    """Frames sampled by the REFERENCE's generate_frame (fp32, stand-in stacks) for fixed noise: indices bit-exact, for
    8 frames of one prompt and 6 frames of a batch of two.  With ``use_graph`` (the default of ``generate()``) frame 0 is
    the prefill, frame 1 the eager warm-up, and every later frame a REPLAY of the captured HIP graph whose Exp(1) noise
    was written into the persistent device buffer before the replay."""
    m, p32, pq = tiny_model(dev)
    meta = json.load(open(os.path.join(GOLD, "golden_meta.json")))
    z = np.load(os.path.join(GOLD, "golden_small.npz"))
    tokens, mask = torch.from_numpy(z["tokens"]), torch.from_numpy(z["mask"])
    K = TINY.n_codebooks
    m.use_hip_graph = use_graph
    n_prompt = 9
    for B, seed0, want in ((1, 1000, meta["generate_frames"]), (2, 2000, meta["generate_frames_b2"])):
        m.setup_caches(B)
        m.reset_caches()
        cur_t, cur_m, cur_p = tokens[:B, :n_prompt], mask[:B, :n_prompt], torch.arange(n_prompt).unsqueeze(0).repeat(B, 1)
        want_t = torch.tensor(want).view(len(want), B, K)
        got = []
        for step in range(len(want)):
            torch.manual_seed(seed0 + step)
            qs = [torch.empty(B, TINY.audio_vocab).exponential_(1) for _ in range(K)]
            f = m.generate_frame(cur_t, cur_m, cur_p, 0.9, 10, noise=qs).cpu()
            got.append(f)
            # the REFERENCE's frame is fed back (teacher forcing), so that one near-tie decided differently by bf16
            # arithmetic shows up as one differing frame instead of a different continuation
            nxt = want_t[step]
            cur_t = torch.cat([nxt.long(), torch.zeros(B, 1, dtype=torch.long)], dim=1).unsqueeze(1)
            cur_m = torch.cat([torch.ones(B, K, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.bool)], dim=1).unsqueeze(1)
            cur_p = cur_p[:, -1:] + 1
        got = torch.stack(got).view(len(want), B, K).to(want_t.dtype)
        if B == 1:
            assert torch.equal(got, want_t), (use_graph, got.tolist(), want)
        else:
            # bf16 weights / activations against the reference's fp32: a near-tie in the top-k race can go the other way
            # (which ones do depends on the last bf16 bit of the prefill - tools/probes/rope_ab.py shows two equally
            # accurate RoPE placements flipping different ones); exact on at least 10 of the 12 row-frames
            bad_rows = int((got != want_t).any(dim=2).sum())
            assert bad_rows <= 2, (use_graph, bad_rows, got.tolist(), want)
            assert float((got == want_t).float().mean()) >= 0.9
            # ... and every differing row-frame IS such a near-tie (VERDICT r03 #4b), checked at its first differing codebook,
            # where both sides have seen the same history and the same earlier codes of the frame: on the oracle's logits
            # (fp32 arithmetic on the same bf16 weights) the reference's pick r and ours g must both survive the top-k cut up to
            # the logit error eps of bf16 activations, and their race scores l/T - log q must be within 2 eps / T.
            for step, b in (got != want_t).any(dim=2).nonzero().tolist():
                i_star = int((got[step, b] != want_t[step, b]).nonzero()[0])
                hist_t = torch.cat([tokens[b, :n_prompt]] + [torch.cat([want_t[s_, b].long(), torch.zeros(1, dtype=torch.long)])[None]
                                                                for s_ in range(step)])
                hist_m = torch.cat([mask[b, :n_prompt]] + [torch.cat([torch.ones(K, dtype=torch.bool), torch.zeros(1, dtype=torch.bool)])[None]
                                                              for s_ in range(step)])
                last_h = O.backbone_hidden(pq, TINY, hist_t[None], hist_m[None])[:, -1, :]
                if i_star == 0:
                    lg = last_h @ pq["codebook0_head.weight"].t()
                else:
                    seq = [last_h.unsqueeze(1)] + [pq["audio_embeddings.weight"][int(want_t[step, b, i]) + i * TINY.audio_vocab][None, None]
                                                    for i in range(i_star)]
                    x = torch.cat(seq, dim=1) @ pq["projection.weight"].t()
                    pos = torch.arange(x.shape[1]).unsqueeze(0)
                    dec = O.transformer({k_: v_ for k_, v_ in pq.items() if k_.startswith("decoder.")}, "decoder", TINY.decoder, x, pos)
                    lg = dec[:, -1, :] @ pq["audio_head"][i_star - 1]
                lg = lg[0].double()
                torch.manual_seed(seed0 + step)
                qrow = [torch.empty(B, TINY.audio_vocab).exponential_(1) for _ in range(K)][i_star][b].double()
                r, g = int(want_t[step, b, i_star]), int(got[step, b, i_star])
                assert int(O.sample_topk(lg[None].float(), 10, 0.9, qrow[None].float())[0, 0]) == r, "the oracle reproduces the reference's pick"
                # our pick g must be a POSSIBLE outcome of the reference's sampler when every logit may move by eps (the error
                # of bf16 activations): (1) g can make the top-k cut - fewer than k logits are certainly above it; (2) every
                # token that would certainly beat g in the race l / T - log q can be cut out - at least k others may be above it
                eps = 2.0 ** -7 * max(1.0, float(lg.abs().max()))
                topk_, T_ = 10, 0.9
                assert int((lg - eps > lg[g] + eps).sum()) <= topk_ - 1, (step, b, i_star, "our pick cannot make the top-k cut", float(lg[g]))
                score_lo = (lg - eps) / T_ - qrow.log()
                for j in (score_lo > (float(lg[g]) + eps) / T_ - float(qrow[g].log())).nonzero().flatten().tolist():
                    if j == g:
                        continue
                    others_above = int((lg + eps > lg[j] - eps).sum()) - 1
                    assert others_above >= topk_, (step, b, i_star, f"token {j} beats our pick {g} in the race by more than the logit error "
                                                                      f"and cannot be cut out: not a near-tie", float(lg[j]), float(lg[g]), r)
        if use_graph:
            assert m._decode_state.graph is not None, "frames >= 2 must have gone through the captured graph"
    m.use_hip_graph = True


def test_generate_kv_cache_vs_recompute_and_graph(dev):
    """KV-cache decode path vs the cache-free recompute path on the SAME history and noise: the two use different
    reduction orders (matrix-vector kernels vs MFMA tiles), so a rare near-tie may sample differently; they must agree
    on almost every code.  The captured-graph replay must reproduce the eager KV path exactly (same kernels)."""
    m, _, _ = tiny_model(dev)
    K = TINY.n_codebooks
    tokens, mask, _ = O.synthetic_batch(TINY, 2, 20, seed=12)

    def noise(step):
        g = torch.Generator().manual_seed(500 + step)
        return [torch.empty(2, TINY.audio_vocab).exponential_(1, generator=g) for _ in range(K)]

    def run(use_cache, history):
        m.use_kv_cache = use_cache
        m.setup_caches(2)
        m.reset_caches()
        cur_t, cur_m, cur_p = tokens[:, :11], mask[:, :11], torch.arange(11).unsqueeze(0).repeat(2, 1)
        frames = []
        for step in range(8):
            f = m.generate_frame(cur_t, cur_m, cur_p, 0.8, 12, noise=noise(step)).cpu()
            frames.append(f)
            nxt = history[step] if history is not None else f            # teacher-force the reference history
            cur_t = torch.cat([nxt.long(), torch.zeros(2, 1, dtype=torch.long)], dim=1).unsqueeze(1)
            cur_m = torch.cat([torch.ones(2, K, dtype=torch.bool), torch.zeros(2, 1, dtype=torch.bool)], dim=1).unsqueeze(1)
            cur_p = cur_p[:, -1:] + 1
        return torch.stack(frames)

    kv = run(True, None)
    rc = run(False, kv)
    m.use_kv_cache = True
    agree = (kv == rc).float().mean().item()
    assert agree >= 0.9, f"KV-cache and recompute paths agree on only {agree:.1%} of the sampled codes"
    assert torch.equal(kv[0], rc[0]), "the prefill frame goes through the same kernels in both paths"
    # graph replay == eager decode, frame for frame, given the same noise through the persistent buffer
    outs = []
    for use_graph in (False, True):
        m.use_hip_graph = use_graph
        m.setup_caches(2)
        m.reset_caches()
        cur_t, cur_m, cur_p = tokens[:, :11], mask[:, :11], torch.arange(11).unsqueeze(0).repeat(2, 1)
        frames = []
        for step in range(8):
            f = m.generate_frame(cur_t, cur_m, cur_p, 0.8, 12, noise=noise(step)).cpu()
            frames.append(f)
            cur_t = torch.cat([f.long(), torch.zeros(2, 1, dtype=torch.long)], dim=1).unsqueeze(1)
            cur_m = torch.cat([torch.ones(2, K, dtype=torch.bool), torch.zeros(2, 1, dtype=torch.bool)], dim=1).unsqueeze(1)
            cur_p = cur_p[:, -1:] + 1
        outs.append(torch.stack(frames))
    m.use_hip_graph = True
    assert torch.equal(outs[0], outs[1]), "graph replay must reproduce the eager KV-cache frames bit for bit"
    assert torch.equal(outs[0], kv), "and both equal the first run"
    # fresh draws (noise=None): the replayed graph must still see NEW noise every frame (the buffer is refilled outside it)
    m.setup_caches(2)
    m.reset_caches()
    torch.manual_seed(7)
    torch.cuda.manual_seed(7)
    cur_t, cur_m, cur_p = tokens[:, :11], mask[:, :11], torch.arange(11).unsqueeze(0).repeat(2, 1)
    seen = []
    for step in range(6):
        f = m.generate_frame(cur_t, cur_m, cur_p, 1.5, 50).cpu()
        seen.append(m._decode_state.noise_buf[0, 0, :8].cpu().clone())
        cur_t = torch.cat([f.long(), torch.zeros(2, 1, dtype=torch.long)], dim=1).unsqueeze(1)
        cur_m = torch.cat([torch.ones(2, K, dtype=torch.bool), torch.zeros(2, 1, dtype=torch.bool)], dim=1).unsqueeze(1)
        cur_p = cur_p[:, -1:] + 1
        assert int(f.min()) >= 0 and int(f.max()) < TINY.audio_vocab
    assert all(not torch.equal(seen[i], seen[i + 1]) for i in range(5)), "every frame needs its own noise"


def test_decode_kernels_vs_oracle(dev):
    from csm.hip import ops
    g = torch.Generator().manual_seed(44)
    # gemv / gemv_t
    x, W = (torch.randn(3, 1024, generator=g)).to(BF), (torch.randn(300, 1024, generator=g) * 0.1).to(BF)
    R = torch.randn(3, 300, generator=g).to(BF)
    y = torch.empty(3, 300, dtype=BF, device=dev)
    ops.gemv(x.to(dev), W.to(dev), y, residual=R.to(dev))
    gclose("gemv", y, x.float() @ W.float().t() + R.float(), 1e-2)
    Wt = (torch.randn(512, 2112, generator=g) * 0.1).to(BF)
    x2 = torch.randn(2, 512, generator=g).to(BF)
    y2 = torch.empty(2, 2112, dtype=torch.float32, device=dev)
    ops.gemv_t(x2.to(dev), Wt.to(dev), y2)
    gclose("gemv_t", y2, x2.float() @ Wt.float(), 1e-4)
    # cache attention for both head dims, ragged positions per batch row
    for H, KV, hd in ((4, 2, 64), (2, 1, 128)):
        B, S_max = 2, 96
        qkv = torch.randn(B, (H + 2 * KV) * hd, generator=g).to(BF)
        kc = torch.randn(B, KV, S_max, hd, generator=g).to(BF)
        vc = torch.randn(B, KV, S_max, hd, generator=g).to(BF)
        pos = torch.tensor([70, 13], dtype=torch.int32)
        kd, vd = kc.to(dev), vc.to(dev)
        ops.kv_append(qkv.to(dev), kd, vd, pos.to(dev), H, KV, hd)
        out = torch.empty(B, H * hd, dtype=BF, device=dev)
        ops.attn_decode(qkv.to(dev), kd, vd, out, pos.to(dev), H, KV, hd)
        for b in range(B):
            n = int(pos[b]) + 1
            kk, vv = kc[b].clone().float(), vc[b].clone().float()
            kk[:, n - 1] = qkv[b, H * hd:(H + KV) * hd].view(KV, hd).float()
            vv[:, n - 1] = qkv[b, (H + KV) * hd:].view(KV, hd).float()
            assert torch.equal(kd[b, :, n - 1].cpu().float(), kk[:, n - 1])
            q = qkv[b, :H * hd].view(H, hd).float()
            for h in range(H):
                kvh = h // (H // KV)
                p = torch.softmax(kk[kvh, :n] @ q[h] / hd ** 0.5, dim=0)
                gclose(f"attn_decode b{b} h{h}", out[b, h * hd:(h + 1) * hd], p @ vv[kvh, :n], 1.5e-2)


def test_fused_decode_kernels_match_their_parts(dev):
    """csm_gemv_bf16_ex and csm_attn_decode_rope replace chains of smaller launches in a decode step; they must give the
    same bits as the chains they replace (and, through those, follow the oracle as tested above)."""
    from csm.hip import ops
    from csm.models.model import llama3_rope_table
    g = torch.Generator().manual_seed(45)
    for B, K, N in ((1, 1024, 512), (3, 2048, 384)):
        x = torch.randn(B, K, generator=g).to(BF).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
        w = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(dev)
        R = torch.randn(B, N, generator=g).to(BF).to(dev)
        xn, y_ref, y = torch.empty_like(x), torch.empty(B, N, dtype=BF, device=dev), torch.empty(B, N, dtype=BF, device=dev)
        ops.rmsnorm_fwd(x, w, xn, None, 1e-5)
        ops.gemv(xn, W, y_ref, residual=R)
        ops.gemv_ex(x, W, y, residual=R, norm_scale=w, eps=1e-5)
        assert torch.equal(y, y_ref), "norm + gemv"
        a_ref, a = torch.empty(B, N // 2, dtype=BF, device=dev), torch.empty(B, N // 2, dtype=BF, device=dev)
        gu = torch.empty(B, N, dtype=BF, device=dev)
        ops.gemv(xn, W, gu)
        ops.swiglu_fwd(gu, a_ref)
        ops.gemv_ex(x, W, a, norm_scale=w, eps=1e-5, swiglu=True)
        assert torch.equal(a, a_ref), "norm + gemv + swiglu"
    for H, KV, hd in ((4, 2, 64), (2, 1, 128)):
        B, S_max = 2, 96
        table = llama3_rope_table(S_max, hd, 500000.0, 32.0).to(dev).contiguous()
        qkv = torch.randn(B, (H + 2 * KV) * hd, generator=g).to(BF).to(dev)
        kc = torch.randn(B, KV, S_max, hd, generator=g).to(BF)
        vc = torch.randn(B, KV, S_max, hd, generator=g).to(BF)
        pos = torch.tensor([70, 0], dtype=torch.int32, device=dev)
        k1, v1, k2, v2 = kc.to(dev), vc.to(dev), kc.to(dev), vc.to(dev)
        q1 = qkv.clone()
        ops.rope(q1, table, 1, H + KV, hd, pos=pos)
        ops.kv_append(q1, k1, v1, pos, H, KV, hd)
        o1, o2 = torch.empty(B, H * hd, dtype=BF, device=dev), torch.empty(B, H * hd, dtype=BF, device=dev)
        ops.attn_decode(q1, k1, v1, o1, pos, H, KV, hd)
        ops.attn_decode_rope(qkv, k2, v2, o2, pos, table, H, KV, hd)
        assert torch.equal(k1, k2) and torch.equal(v1, v2), "caches after the fused append"
        assert torch.equal(o1, o2), "fused rope + append + attention"


def test_batched_matrix_vector_kernels_match_single_row(dev):
    """Round 4: two to four batch rows through gemv_regn_kernel (every row of x in registers, two rows per packed multiply-add) -
    every row bit-equal to the one-row launch on that row AND to the LDS kernel it replaces (csm_set_decode_tuning(3, 0)), for
    both widths it takes (K = 1024, 2048), plain / residual / RMSNorm prologue / SwiGLU / fp32 output / rows gathered from a
    table; K = 8192 (not taken) stays consistent too.  Batched generation equals single generation only because of this."""
    from csm.hip import ops
    g = torch.Generator().manual_seed(77)
    for K, N in ((1024, 1536), (2048, 1024), (8192, 512)):
        W = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
        w = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(dev)
        table = torch.randn(64, K, generator=g).to(BF).to(dev)
        for B in (2, 3, 4):
            x = torch.randn(B, K, generator=g).to(BF).to(dev)
            R = torch.randn(B, N, generator=g).to(BF).to(dev)
            idx = torch.randint(0, 32, (B,), generator=g).to(torch.int32).to(dev)

            def run_all(xb, Rb, idxb, Bn):
                outs = []
                y = torch.empty(Bn, N, dtype=BF, device=dev); ops.gemv(xb, W, y); outs.append(y)
                y = torch.empty(Bn, N, dtype=BF, device=dev); ops.gemv(xb, W, y, residual=Rb); outs.append(y)
                y = torch.empty(Bn, N, dtype=BF, device=dev); ops.gemv_ex(xb, W, y, residual=Rb, norm_scale=w, eps=1e-5); outs.append(y)
                y = torch.empty(Bn, N // 2, dtype=BF, device=dev); ops.gemv_ex(xb, W, y, norm_scale=w, eps=1e-5, swiglu=True); outs.append(y)
                y = torch.empty(Bn, N, dtype=torch.float32, device=dev); ops.gemv_ex(xb, W, y, norm_scale=w, eps=1e-5); outs.append(y)
                y = torch.empty(Bn, N, dtype=BF, device=dev); ops.gemv_ex(table, W, y, row_index=idxb, row_offset=7); outs.append(y)
                return outs

            new = run_all(x, R, idx, B)
            try:
                ops.lib.csm_set_decode_tuning(3, 0)
                old = run_all(x, R, idx, B)
            finally:
                ops.lib.csm_set_decode_tuning(3, 1)
            for i, (a_, b_) in enumerate(zip(new, old)):
                assert torch.equal(a_, b_), f"K={K} B={B} form {i}: register kernel vs LDS kernel"
            for r in range(B):
                one = run_all(x[r:r + 1].contiguous(), R[r:r + 1].contiguous(), idx[r:r + 1].contiguous(), 1)
                for i, (a_, b_) in enumerate(zip(new, one)):
                    assert torch.equal(a_[r:r + 1], b_), f"K={K} B={B} row {r} form {i}: batched vs single"


def test_decoder_attention_fused_into_output_projection(dev):
    """csm_gemv_attn_bf16 (a depth-decoder layer's rope + cache append + attention + output projection + residual in one
    launch) against the two launches it replaces, csm_attn_decode_rope + csm_gemv_bf16, at the decoder's own geometry (8 q /
    2 kv heads of 128, 32-slot cache, d = 1024) for every position of a frame and batch 1..4: same bits in the output AND in
    the caches - so frames decoded through it are the frames the reference fixture pins."""
    from csm.hip import ops
    from csm.models.model import llama3_rope_table
    g = torch.Generator().manual_seed(46)
    H, KV, hd, S_max, d = 8, 2, 128, 32, 1024
    table = llama3_rope_table(S_max, hd, 500000.0, 32.0).to(dev).contiguous()
    W = (torch.randn(d, H * hd, generator=g) * 0.05).to(BF).to(dev)
    for B in (1, 2, 4):
        kc = torch.randn(B, KV, S_max, hd, generator=g).to(BF)
        vc = torch.randn(B, KV, S_max, hd, generator=g).to(BF)
        for p0 in (0, 1, 7, 16, 17, 30, 31):
            posv = [(p0 + 5 * b) % S_max for b in range(B)]
            pos = torch.tensor(posv, dtype=torch.int32, device=dev)
            qkv = torch.randn(B, (H + 2 * KV) * hd, generator=g).to(BF).to(dev)
            R = torch.randn(B, d, generator=g).to(BF).to(dev)
            k1, v1, k2, v2 = kc.to(dev), vc.to(dev), kc.to(dev), vc.to(dev)
            o = torch.empty(B, H * hd, dtype=BF, device=dev)
            y1, y2 = torch.empty(B, d, dtype=BF, device=dev), torch.empty(B, d, dtype=BF, device=dev)
            ops.attn_decode_rope(qkv, k1, v1, o, pos, table, H, KV, hd)
            ops.gemv(o, W, y1, residual=R)
            ops.gemv_attn(qkv, k2, v2, pos, table, W, y2, R, H, KV, hd)
            assert torch.equal(k1, k2) and torch.equal(v1, v2), f"caches B={B} pos={posv}"
            assert torch.equal(y1, y2), f"output B={B} pos={posv}: max diff {(y1.float() - y2.float()).abs().max().item()}"
            y3 = torch.empty(B, d, dtype=BF, device=dev)
            ops.gemv_attn(qkv, k2, v2, pos, table, W, y3, None, H, KV, hd)        # (append again: same bits; no residual)
            ops.gemv(o, W, y1)
            assert torch.equal(y1, y3)
            if B > 1:
                # round 4: rows that share a position the host knows (a batch of utterances at decoder step p): the attention-only
                # launch with all loads at once - output and caches bit-equal to csm_attn_decode_rope at every position
                for p in range(S_max):
                    posb = torch.full((B,), p, dtype=torch.int32, device=dev)
                    k3, v3, k4, v4 = kc.to(dev), vc.to(dev), kc.to(dev), vc.to(dev)
                    oa, ob = torch.empty(B, H * hd, dtype=BF, device=dev), torch.empty(B, H * hd, dtype=BF, device=dev)
                    ops.attn_decode_rope(qkv, k3, v3, oa, posb, table, H, KV, hd)
                    ops.attn_decode_rope(qkv, k4, v4, ob, posb, table, H, KV, hd, pos_host=p)
                    assert torch.equal(k3, k4) and torch.equal(v3, v4), f"caches, B={B}, host position {p}"
                    assert torch.equal(oa, ob), f"attention, B={B}, host position {p}"
            if B == 1:
                # round 4: the launch that takes the position from the host (the depth decoder's step i is at position i) and
                # issues every load of its prologue at once - every position of a frame, same bits in output and caches
                for p in range(S_max):
                    pos1 = torch.tensor([p], dtype=torch.int32, device=dev)
                    k3, v3, k4, v4 = kc.to(dev), vc.to(dev), kc.to(dev), vc.to(dev)
                    ya, yb = torch.empty(1, d, dtype=BF, device=dev), torch.empty(1, d, dtype=BF, device=dev)
                    ops.attn_decode_rope(qkv, k3, v3, o, pos1, table, H, KV, hd)
                    ops.gemv(o, W, ya, residual=R)
                    ops.gemv_attn(qkv, k4, v4, pos1, table, W, yb, R, H, KV, hd, pos_host=p)
                    assert torch.equal(k3, k4) and torch.equal(v3, v4), f"caches, host position {p}"
                    assert torch.equal(ya, yb), f"output, host position {p}: {(ya.float() - yb.float()).abs().max().item()}"


def test_batched_generation_matches_single(dev):
    """Two utterances with prompts of different lengths decoded together (ragged prefill into the rows of one KV cache,
    per-row positions) must sample exactly the frames each samples alone, given the same Exp(1) draws."""
    m, _, _ = tiny_model(dev)
    m.setup_caches(2)
    K, V = m.args.audio_num_codebooks, m.args.audio_vocab_size
    g = torch.Generator().manual_seed(31)
    prompts = []
    for S in (11, 19):
        tokens, mask, _ = O.synthetic_batch(TINY, 1, S, seed=40 + S)
        prompts.append((tokens[0], mask[0]))
    n_frames = 4
    noise = [[torch.empty(2, V).exponential_(1.0, generator=g) for _ in range(K)] for _ in range(n_frames)]
    amask = torch.cat([torch.ones(1, K, dtype=torch.bool), torch.zeros(1, 1, dtype=torch.bool)], 1).unsqueeze(1)

    def run(rows):
        B = len(rows)
        m.reset_caches()
        fr = [m.engine.generate_first_frames([prompts[r][0] for r in rows], [prompts[r][1] for r in rows], 0.9, 10,
                                             noise=[n[rows] for n in noise[0]])]
        for f in range(1, n_frames):
            tok = torch.cat([fr[-1].long().cpu(), torch.zeros(B, 1, dtype=torch.long)], 1).unsqueeze(1)
            fr.append(m.generate_frame(tok, amask.expand(B, -1, -1), torch.ones(B, 1, dtype=torch.long), 0.9, 10,
                                       noise=[n[rows] for n in noise[f]]))
        return torch.stack([x.cpu() for x in fr], 1)                # [B, frames, K]

    both = run([0, 1])
    assert both.shape == (2, n_frames, K)
    assert torch.equal(both[0:1], run([0])) and torch.equal(both[1:2], run([1]))
    assert not torch.equal(both[0], both[1])


def test_checkpoint_roundtrip(dev, tmp_path):
    from csm.training.optim import FusedAdamW
    from csm.training.utils import compute_loss, load_checkpoint, save_checkpoint
    m, _, _ = tiny_model(dev)
    opt = FusedAdamW(m, {"backbone": 1e-3, "decoder": 1e-3, "embeddings": 1e-3, "other": 1e-3})
    tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=5)
    t, _ = compute_loss(m, tokens, mask, targets)
    t.backward()
    opt.step()
    path = save_checkpoint(m, opt, 1, 7, float(t), str(tmp_path))
    assert os.path.exists(path) and os.path.exists(os.path.join(str(tmp_path), "checkpoint_latest.pt"))
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "global_step", "loss"}
    m2, _, _ = tiny_model(dev, seed=99)
    opt2 = FusedAdamW(m2, {"backbone": 1e-3, "decoder": 1e-3, "embeddings": 1e-3, "other": 1e-3})
    meta = load_checkpoint(path, m2, opt2)
    assert meta["epoch"] == 1 and meta["global_step"] == 7
    assert torch.equal(m2.arena, m.arena)
    assert torch.equal(opt2.state["backbone"]["m"], opt.state["backbone"]["m"])


def test_cli_entry_points(dev, tmp_path, monkeypatch):
    """csm-train / csm-finetune-lora front ends: reference flag names, tiny model, synthetic data, one epoch."""
    import csm.training.trainer as T
    from csm.models.model import ModelArgs
    monkeypatch.setattr(T, "csm_1b_args", lambda: ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", 128256, 2051, 32))
    from csm.cli import train as cli_train, finetune_lora as cli_lora
    rc = cli_train.main(["--model-path", "", "--output-dir", str(tmp_path / "t"), "--synthetic", "6", "--max-seq-len", "48",
                         "--epochs", "1", "--batch-size", "2", "--accumulation-steps", "1", "--num-workers", "0",
                         "--acoustic-mode", "amortized", "--val-every", "1", "--freeze-embeddings"])
    assert rc == 0 and (tmp_path / "t" / "final_latest.pt").exists()
    rc = cli_lora.main(["--model-path", "", "--output-dir", str(tmp_path / "l"), "--synthetic", "6", "--max-seq-len", "48",
                        "--epochs", "1", "--batch-size", "2", "--lora-r", "8", "--target-modules", "q_proj", "v_proj", "--save-mode", "both"])
    assert rc == 0 and (tmp_path / "l" / "final_lora.safetensors").exists() and (tmp_path / "l" / "final_full.safetensors").exists()
    meta = json.load(open(tmp_path / "l" / "final_lora_metadata.json"))
    assert meta["lora_r"] == 8 and meta["target_modules"] == ["q_proj", "v_proj"]


def test_full_size_model_properties(dev):
    """CSM-1B at BASELINE's sequence length (S=2048), where the CPU oracle is too slow to be the checker: properties that
    do not depend on the size - bit-exact run-to-run determinism of loss and every gradient (no atomics anywhere in the
    backward), invariance of the mean loss under duplicating the batch, linearity of the backward in the loss scale, and
    the 1e-3 north-star bar against the oracle on a prefix that the oracle can still do (S=128)."""
    from csm.data import SyntheticCSMDataset, collate_variable_length
    from csm.models.model import Model
    from csm.training.trainer import csm_1b_args
    from csm.training.utils import compute_loss
    m = Model(csm_1b_args(), device=dev, seed=0)
    m.acoustic_mode = "amortized"
    ds = SyntheticCSMDataset(2, 2048, seed=77)
    one = collate_variable_length([ds[0]])
    rows = torch.arange(0, 2047, 16)                       # fixed decoder rows: the amortised sampler would draw its own
    m.ensure_grads()

    def run(batch, rows, scale=1.0):
        m.grad_arena.zero_()
        m.grad_state.update({k: "zero" for k in m.grad_state})
        total, det = compute_loss(m, batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"], acoustic_rows=rows)
        (total * scale).backward()
        return float(total), float(det["semantic_loss"]), float(det["acoustic_loss"]), m.grad_arena.clone()

    t1, s1, a1, g1 = run(one, rows)
    t2, s2, a2, g2 = run(one, rows)
    assert math.isfinite(t1) and (t1, s1, a1) == (t2, s2, a2), "loss must be bit-reproducible"
    assert torch.equal(g1, g2), "every gradient must be bit-reproducible"
    assert float(g1.float().abs().max()) > 0
    two = {k: torch.cat([v, v], 0) for k, v in one.items()}
    t3, s3, a3, g3 = run(two, torch.cat([rows, rows + 2047]))
    assert rel(s3, s1) < 1e-5 and rel(a3, a1) < 1e-5, "mean loss is invariant under duplicating the batch"
    gclose("gradient of the duplicated batch", g3, g1.float(), 2e-2)
    _, _, _, g4 = run(one, rows, scale=0.25)
    gclose("backward is linear in the loss scale", g4, 0.25 * g1.float(), 2e-2)
    # oracle on a short prefix with the same weights (bf16 values widened to fp32)
    S = 128
    short = {k: v[:, :S] for k, v in one.items()}
    m.acoustic_mode = "off"
    with torch.no_grad():
        tg, _ = compute_loss(m, short["input_tokens"], short["input_masks"], short["target_audio_tokens"])
    params = {k: v.float().cpu() for k, v in m._views(m.arena).items()}
    with torch.no_grad():
        to, _ = O.compute_loss(params, O.csm_1b_cfg(), short["input_tokens"], short["input_masks"], short["target_audio_tokens"],
                               acoustic_rows="off")
    assert rel(tg, to) < 1e-3, (float(tg), float(to))


def test_full_size_acoustic_and_adamw_vs_oracle(dev):
    """CSM-1B (real widths, 16 + 4 layers) at a sequence short enough for the CPU oracle: semantic AND acoustic loss terms,
    the global gradient norm, and one clipped AdamW step with the reference's four learning-rate groups, against the
    oracle run on the same weights (bf16 values widened to fp32)."""
    from csm.data import SyntheticCSMDataset, collate_variable_length
    from csm.models.model import Model
    from csm.training.trainer import CSMTrainer, csm_1b_args
    from csm.training.utils import compute_loss
    m = Model(csm_1b_args(), device=dev, seed=0)
    m.acoustic_mode = "all"
    S = 64
    batch = collate_variable_length([SyntheticCSMDataset(1, S, seed=91)[0]])
    rows = torch.tensor([3, 17, 29, 30, 41, 52, 60, 62])
    lr = 1e-3
    tr = CSMTrainer("", "/tmp/csm_full_adamw", device=str(dev), learning_rate=lr)
    tr.logger.setLevel(40)
    tr.model = m
    params = {k: v.float().cpu() for k, v in m._views(m.arena).items()}          # before the step
    tr.prepare_optimizer()
    total, det = compute_loss(m, batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"], 100.0, 1.0, acoustic_rows=rows)
    total.backward()
    norm = tr.optimizer.clip_grad_norm(1.0)
    tr.optimizer.step()
    masters = dict(tr.optimizer.named_master())
    # ---- oracle
    cfg = O.csm_1b_cfg()
    pt = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    rt, rd = O.compute_loss(pt, cfg, batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"], 100.0, 1.0, acoustic_rows=rows)
    rt.backward()
    assert rel(det["semantic_loss"], rd["semantic_loss"]) < 1e-3, (float(det["semantic_loss"]), float(rd["semantic_loss"]))
    assert rel(det["acoustic_loss"], rd["acoustic_loss"]) < 1e-3, (float(det["acoustic_loss"]), float(rd["acoustic_loss"]))
    assert rel(total, rt) < 1e-3
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in pt.values())))
    assert rel(norm, gn) < 2e-2, (float(norm), gn)
    coef = min(1.0, 1.0 / (gn + 1e-6))
    mult = {"backbone": 0.1, "decoder": 1.0, "embeddings": 0.5, "other": 1.0}
    for k in ["backbone.layers.0.attn.q_proj.weight", "backbone.layers.7.mlp.w1.weight", "backbone.layers.15.mlp.w2.weight", "backbone.norm.scale",
              "decoder.layers.0.attn.v_proj.weight", "decoder.layers.3.mlp.w3.weight", "projection.weight", "codebook0_head.weight",
              "audio_head", "audio_embeddings.weight"]:
        grp = "backbone" if "backbone" in k else "decoder" if "decoder" in k else "embeddings" if "embeddings" in k else "other"
        g = pt[k].grad * coef
        ref = params[k].clone()
        O.adamw_step(ref, g, torch.zeros_like(ref), torch.zeros_like(ref), 1, lr * mult[grp], weight_decay=0.01)
        upd, upd_ref = masters[k].float().cpu() - params[k], ref - params[k]
        big = g.abs() > 0.2 * g.abs().max()
        assert big.any(), k
        err = (upd[big] - upd_ref[big]).abs().max().item()
        assert err < 0.1 * lr * mult[grp], (k, err)
        assert upd_ref[big].abs().min().item() > 0.5 * lr * mult[grp], k


def test_config2_bench_size_loss_and_gradients_vs_oracle(dev):
    """BASELINE config 2's own sequence length (VERDICT r03 #4d): CSM-1B, S = 2048, one sequence (the CPU oracle's step at
    B = 1 is what the box's host cores finish in about a minute; the batch dimension is covered by the bench's config2_b4_parity
    leg: batch loss == mean of its single-sequence losses), loss mode C with the decoder rows pinned: total / semantic /
    acoustic loss within 1e-3 relative (the north star's bar) and a sample of gradients - attention, MLP, norm, both stacks,
    the heads, the audio embeddings - against the oracle's autograd on the same weights (bf16 values widened to fp32)."""
    from csm.data import SyntheticCSMDataset, collate_variable_length
    from csm.models.model import Model
    from csm.training.trainer import csm_1b_args
    from csm.training.utils import compute_loss
    m = Model(csm_1b_args(), device=dev, seed=0)
    m.acoustic_mode = "amortized"
    S = 2048
    batch = collate_variable_length([SyntheticCSMDataset(1, S, seed=4321)[0]])
    rows = torch.arange(0, S - 1, 16)                             # 1/16 of the frames, pinned
    m.ensure_grads()
    total, det = compute_loss(m, batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"], 100.0, 1.0, acoustic_rows=rows)
    total.backward()
    torch.cuda.synchronize()
    names = ["backbone.layers.0.attn.q_proj.weight", "backbone.layers.7.mlp.w1.weight", "backbone.layers.15.mlp.w2.weight", "backbone.norm.scale",
             "decoder.layers.0.attn.v_proj.weight", "decoder.layers.3.mlp.w3.weight", "projection.weight", "codebook0_head.weight",
             "audio_head", "audio_embeddings.weight"]
    gviews = m._views(m.grad_arena)
    got = {k: gviews[k].float().cpu() for k in names}
    # ---- oracle (fp32, all host cores)
    cfg = O.csm_1b_cfg()
    pt = {k: v.float().cpu().requires_grad_(k in names) for k, v in m._views(m.arena).items()}
    ref, rd = O.compute_loss(pt, cfg, batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"], 100.0, 1.0, acoustic_rows=rows)
    ref.backward()
    assert rel(total, ref) < 1e-3, (float(total), float(ref))
    assert rel(det["semantic_loss"], rd["semantic_loss"]) < 1e-3 and rel(det["acoustic_loss"], rd["acoustic_loss"]) < 1e-3
    for k in names:
        g, r = got[k], pt[k].grad
        scale = r.abs().max().item()
        assert scale > 0, k
        err = (g - r).abs().max().item()
        cos = torch.nn.functional.cosine_similarity(g.flatten().double(), r.flatten().double(), dim=0).item()
        assert err <= 5e-2 * scale and cos > 0.999, (k, err, scale, cos)


def test_full_size_lora_config3_properties(dev):
    """BASELINE config 3 at full size (CSM-1B, LoRA r=8 on q_proj / v_proj, S=2048, B=8), where only size-independent
    properties can be checked: fresh adapters (B = 0) leave the loss equal to the bare model's, the base weights
    receive no gradient, dA is exactly 0 while dB is not (the chain rule through B = 0), the step is bit-reproducible,
    and merging the trained adapters reproduces the adapted loss."""
    from csm.data import SyntheticCSMDataset, collate_variable_length
    from csm.models.model import Model
    from csm.training.lora import apply_lora_to_model, merge_lora_weights
    from csm.training.optim import FusedAdamW
    from csm.training.trainer import csm_1b_args
    from csm.training.utils import compute_loss
    m = Model(csm_1b_args(), device=dev, seed=0)
    m.acoustic_mode = "off"
    ds = SyntheticCSMDataset(8, 2048, seed=55)
    batch = {k: v.to(dev) for k, v in collate_variable_length([ds[i] for i in range(8)]).items()}
    args = (batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"])
    with torch.no_grad():
        bare, _ = compute_loss(m, *args)
    apply_lora_to_model(m, r=8, alpha=16.0, target_modules=["q_proj", "v_proj"], seed=1)
    assert m.lora.num_params() == 958464
    opt = FusedAdamW(m, {}, lora_lr=1e-3)

    def step():
        m.lora.grad_arena.zero_()
        total, _ = compute_loss(m, *args)
        total.backward()
        return float(total), m.lora.grad_arena.clone()

    t1, g1 = step()
    t2, g2 = step()
    # (not bit-identical: with adapters on q / v the rotation runs after the adapter's addition as its own kernel, the bare
    #  model rotates the fp32 accumulators inside the projection's epilogue - one bf16 rounding apart)
    assert rel(t1, bare) < 1e-4, "adapters with B = 0 must not change the loss"
    assert t1 == t2 and torch.equal(g1, g2), "bit-reproducible"
    assert m.grad_arena is None or float(m.grad_arena.abs().max()) == 0.0, "base weights are frozen"
    for ad in m.lora.adapters.values():
        assert float(ad.gA.abs().max()) == 0.0, "dA = (dy B)^T x = 0 while B = 0"
    # (loss mode A: only the backbone runs, so only its adapters see a gradient)
    assert all(float(ad.gB.float().abs().max()) > 0 for (prefix, _, _), ad in m.lora.adapters.items() if prefix == "backbone")
    opt.step(zero_grad=True)
    with torch.no_grad():
        adapted, _ = compute_loss(m, *args)
        assert float(adapted) != float(bare)
        merge_lora_weights(m)
        m.lora = None
        merged, _ = compute_loss(m, *args)
    assert rel(merged, adapted) < 2e-3, (float(merged), float(adapted))


def test_full_size_lora_config3_mode_c_decoder_adapters(dev):
    """BASELINE config 3 in the loss mode the bench's LoRA leg runs (mode C: semantic CE + depth decoder on 1/16 of the frames),
    with the decoder rows PINNED so that every evaluation sees the same frames (VERDICT r03 #4c): fresh adapters leave both loss
    terms equal to the bare model's, the DECODER's adapters receive gradients too (dB != 0, dA == 0 while B = 0), the step is
    bit-reproducible, and merging the trained adapters reproduces the adapted semantic AND acoustic losses."""
    from csm.data import SyntheticCSMDataset, collate_variable_length
    from csm.models.model import Model
    from csm.training.lora import apply_lora_to_model, merge_lora_weights
    from csm.training.optim import FusedAdamW
    from csm.training.trainer import csm_1b_args
    from csm.training.utils import compute_loss
    B, S = 8, 2048
    m = Model(csm_1b_args(), device=dev, seed=0)
    m.acoustic_mode = "amortized"
    ds = SyntheticCSMDataset(B, S, seed=56)
    batch = {k: v.to(dev) for k, v in collate_variable_length([ds[i] for i in range(B)]).items()}
    per = torch.arange(0, S - 1, 16)
    rows = torch.cat([per + b * (S - 1) for b in range(B)])       # the same 1/16 of the frames in every call
    args = (batch["input_tokens"], batch["input_masks"], batch["target_audio_tokens"])
    kw = dict(acoustic_rows=rows)
    with torch.no_grad():
        bare, bd = compute_loss(m, *args, **kw)
    apply_lora_to_model(m, r=8, alpha=16.0, target_modules=["q_proj", "v_proj"], seed=1)
    opt = FusedAdamW(m, {}, lora_lr=1e-3)

    def step():
        m.lora.grad_arena.zero_()
        total, det = compute_loss(m, *args, **kw)
        total.backward()
        return float(total), det, m.lora.grad_arena.clone()

    t1, d1, g1 = step()
    t2, d2, g2 = step()
    assert rel(t1, bare) < 1e-4 and rel(d1["acoustic_loss"], bd["acoustic_loss"]) < 1e-4, "adapters with B = 0 must not change either loss term"
    assert t1 == t2 and torch.equal(g1, g2), "bit-reproducible"
    assert float(bd["acoustic_loss"]) > 0
    for (prefix, _, _), ad in m.lora.adapters.items():
        assert float(ad.gA.abs().max()) == 0.0, "dA = (dy B)^T x = 0 while B = 0"
        assert float(ad.gB.float().abs().max()) > 0, f"{prefix} adapter without a gradient in mode C"
    assert any(prefix == "decoder" for (prefix, _, _) in m.lora.adapters), "config 3 adapts both stacks"
    opt.step(zero_grad=True)
    with torch.no_grad():
        adapted, ad_ = compute_loss(m, *args, **kw)
        assert float(adapted) != float(bare) and float(ad_["acoustic_loss"]) != float(bd["acoustic_loss"])
        merge_lora_weights(m)
        m.lora = None
        merged, md = compute_loss(m, *args, **kw)
    assert rel(merged, adapted) < 2e-3, (float(merged), float(adapted))
    assert rel(md["acoustic_loss"], ad_["acoustic_loss"]) < 2e-3 and rel(md["semantic_loss"], ad_["semantic_loss"]) < 2e-3


def test_optimizer_follows_rewritten_weights(dev):
    """The split fp32 master keeps its upper half IN the bf16 arena: anything that rewrites the working weights after the
    optimiser exists - Model.load_state_dict, merge_lora_weights, GradSync.broadcast_parameters - must leave master == new
    weights (no stale lower halves shifting every weight by up to one bf16 ulp), at fp32 where the loaded values were fp32."""
    from csm.training.optim import FusedAdamW
    m, _, _ = tiny_model(dev)
    opt = FusedAdamW(m, {"backbone": 1e-3, "decoder": 1e-3, "embeddings": 1e-3, "other": 1e-3})
    tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=5)
    from csm.training.utils import compute_loss
    t, _ = compute_loss(m, tokens, mask, targets)
    t.backward()
    opt.step()                                                   # lower halves are now non-trivial
    assert any(float(st["lo"].float().abs().max()) > 0 for st in opt.state.values() if "lo" in st) or not any("lo" in st for st in opt.state.values())
    fresh = {k: (v.float() * 1.01 + 1e-4) for k, v in O.init_params(TINY, seed=77).items()}        # fp32 values, not bf16-representable
    m.load_state_dict(fresh)
    for name, master in opt.named_master():
        assert torch.equal(master.cpu(), fresh[name]), f"{name}: master must be the loaded fp32 values"
    bf = {k: v.to(torch.bfloat16) for k, v in fresh.items()}
    m.load_state_dict(bf)
    for name, master in opt.named_master():
        assert torch.equal(master.cpu(), bf[name].float()), f"{name}: master must equal the new bf16 weights exactly"


def test_merge_and_broadcast_do_not_revert_to_the_loaded_checkpoint(dev):
    """a model loaded from an fp32 state dict kept that dict (``_fp32_source``) and every later
    ``params_rewritten`` - merge_lora_weights, broadcast_parameters, save_model's restore - re-seeded the master from it,
    i.e. reverted the weights to the checkpoint.  Only load_state_dict may seed from the fp32 dict; a step retires it."""
    from csm.training.optim import FusedAdamW
    from csm.training.utils import compute_loss
    m, _, _ = tiny_model(dev)
    fresh = {k: (v.float() * 1.01 + 1e-4) for k, v in O.init_params(TINY, seed=78).items()}         # fp32, not bf16-representable
    m.load_state_dict(fresh)
    opt = FusedAdamW(m, {"backbone": 1e-2, "decoder": 1e-2, "embeddings": 1e-2, "other": 1e-2})
    for name, master in opt.named_master():
        assert torch.equal(master.cpu(), fresh[name]), f"{name}: an optimiser built after an fp32 load starts at fp32"
    tokens, mask, targets = O.synthetic_batch(TINY, 2, 24, seed=5)
    t, _ = compute_loss(m, tokens, mask, targets)
    t.backward()
    opt.step()
    assert m._fp32_source is None
    trained = m.arena.clone()
    moved = sum(int((m._views(m.arena)[k].float().cpu() != fresh[k].to(torch.bfloat16).float()).sum()) for k in fresh)
    assert moved > 0, "the step must have moved the weights away from the checkpoint"
    # an outside rewrite that is not a load: new bf16 weights become the master, never the old checkpoint
    new = (trained.float() * 0.5).to(torch.bfloat16)
    m.arena.copy_(new)
    m.params_rewritten()
    assert torch.equal(m.arena, new), "params_rewritten must keep the rewritten weights"
    for g in opt.param_groups:
        assert torch.equal(opt.master(g["name"]), new[g["offset"]:g["offset"] + g["numel"]].float())
    # and a rewrite right after an fp32 load (before any step) must not resurrect the loaded values either
    m.load_state_dict(fresh)
    m.arena.copy_(new)
    m.params_rewritten()
    assert torch.equal(m.arena, new)
    for g in opt.param_groups:
        assert torch.equal(opt.master(g["name"]), new[g["offset"]:g["offset"] + g["numel"]].float())


def test_seed_master_on_shard_ranges_equals_full_range(dev):
    """ZeRO-1 (training/zero.py) seeds the fp32 master of arbitrary arena ranges: a parameter that straddles a shard cut -
    including the interleaved w1 / w3 rows of the fused w13 block - must contribute exactly the part inside the range."""
    from csm.training.optim import seed_master
    m, _, _ = tiny_model(dev)
    fresh = {k: (v.float() * 1.01 + 1e-4) for k, v in O.init_params(TINY, seed=79).items()}
    m.load_state_dict(fresh)
    for grp in ("backbone", "decoder", "embeddings", "other"):
        o, n = m.group_range(grp)
        full = seed_master(m, o, n, m.arena[o:o + n], True)
        assert not torch.equal(full, m.arena[o:o + n].float()), "the fp32 source must have been used"
        cuts = [0, 8, 1000 // 8 * 8, n // 3 // 8 * 8, n // 2 // 8 * 8 + 8, n - 8, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b <= a:
                continue
            part = seed_master(m, o + a, b - a, m.arena[o + a:o + b], True)
            assert torch.equal(part, full[a:b]), (grp, a, b)
    # and without the source flag: the bf16 weights themselves
    o, n = m.group_range("backbone")
    assert torch.equal(seed_master(m, o + 8, 64, m.arena[o + 8:o + 72], False), m.arena[o + 8:o + 72].float())
