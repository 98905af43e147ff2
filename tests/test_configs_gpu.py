"""BASELINE.json's configurations at their own sizes (CSM-1B widths, S = 2048), each through the public API of the
reference module it belongs to.  The CPU oracle cannot run these sizes in test time, so every test pairs (1) oracle checks
of the very kernels / shapes the configuration uses, at a size the oracle does in seconds, with (2) size-independent
properties at the full size: bit-reproducibility, batch-mean identities, schedule A/B bit-equality, cache-path vs
recompute-path agreement, graph replay == eager.

  config 1  CSM-1B through ``CSMLoRATrainer.train`` (reference lora_trainer.py:374-457, mlx_trainer.py:733-876), 2 segments
  config 2  full-param, S = 2048, batch 4 (M = 8192: the persistent 256x256 GEMM's multi-round tile lists)
  config 5  ``generate_frame`` loop at CSM-1B: prefill + 125 frames x 32 codebooks (reference model.py:140-195,
            generator.py:196-207)
"""
import json
import math
import os

import pytest
import torch

from oracle import csm_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    return abs(float(a) - float(b)) / max(1e-12, abs(float(b)))


def gclose(name, got, ref, tol):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    assert math.isfinite(err) and err <= tol * scale, f"{name}: max abs err {err:.4g} vs scale {scale:.4g} (tol {tol})"


@pytest.fixture(scope="module")
def csm1b(dev):
    from csm.models.model import Model
    from csm.training.trainer import csm_1b_args
    return Model(csm_1b_args(), device=dev, seed=0)


def test_config2_full_param_b4_s2048(dev, csm1b):
    """BASELINE config 2 (the headline bench workload): one full-param step's loss and gradients at B = 4, S = 2048, loss
    mode C with pinned decoder rows.  Bit-reproducible; the batch loss is the mean of the four single-sequence losses (each of
    which is the quantity test_full_size_model_properties / bench.py's parity_check tie to the oracle); and the persistent
    256x256 GEMM schedule (tile lists of several rounds at M = 8192) gives the same bits as one tile per workgroup."""
    from csm.data import SyntheticCSMDataset, collate_variable_length
    from csm.hip import lib
    from csm.training.utils import compute_loss
    m = csm1b
    m.acoustic_mode = "amortized"
    ds = SyntheticCSMDataset(4, 2048, seed=1234)
    items = [ds[i] for i in range(4)]
    batch = collate_variable_length(items)
    per = torch.arange(0, 2047, 16)                                        # 128 decoder frames per sequence
    rows4 = torch.cat([per + b * 2047 for b in range(4)])
    m.ensure_grads()

    def run(b, rows, grads=True):
        m.grad_arena.zero_()
        m.grad_state.update({k: "zero" for k in m.grad_state})
        if not grads:
            with torch.no_grad():
                total, det = compute_loss(m, b["input_tokens"], b["input_masks"], b["target_audio_tokens"], acoustic_rows=rows)
            return float(total), float(det["semantic_loss"]), float(det["acoustic_loss"]), None
        total, det = compute_loss(m, b["input_tokens"], b["input_masks"], b["target_audio_tokens"], acoustic_rows=rows)
        total.backward()
        return float(total), float(det["semantic_loss"]), float(det["acoustic_loss"]), m.grad_arena.clone()

    assert lib.csm_get_gemm256_persistent() == 1
    t1, s1, a1, g1 = run(batch, rows4)
    t2, s2, a2, g2 = run(batch, rows4)
    assert math.isfinite(t1) and (t1, s1, a1) == (t2, s2, a2), "loss must be bit-reproducible at B = 4"
    assert torch.equal(g1, g2), "every gradient must be bit-reproducible at B = 4"
    assert float(g1.float().abs().max()) > 0
    singles = [run(collate_variable_length([it]), per, grads=False) for it in items]
    assert rel(s1, sum(x[1] for x in singles) / 4) < 1e-5, (s1, [x[1] for x in singles])
    assert rel(a1, sum(x[2] for x in singles) / 4) < 1e-5, (a1, [x[2] for x in singles])
    assert rel(t1, sum(x[0] for x in singles) / 4) < 1e-5
    try:
        lib.csm_set_gemm256_persistent(0)
        t3, s3, a3, g3 = run(batch, rows4)
    finally:
        lib.csm_set_gemm256_persistent(1)
    assert (t3, s3, a3) == (t1, s1, a1), "one tile per workgroup vs persistent tile lists: loss bits"
    assert torch.equal(g3, g1), "one tile per workgroup vs persistent tile lists: gradient bits"
    # the attention projections' weight gradients: three layers per launch (default) vs one layer per launch - the same tiles,
    # so the same bits; and the per-layer "gradients are final" hook still fires exactly once per layer, layer 0 last
    from csm import engine as E
    assert E.DEFER_ATTN_DW == 3
    seen = []
    eng = m.engine
    old_hook, old_defer = eng.grad_hook, E.DEFER_ATTN_DW
    try:
        eng.grad_hook = lambda prefix, i: seen.append((prefix, i))
        t4, s4, a4, g4 = run(batch, rows4)
        order3 = list(seen)
        seen.clear()
        E.DEFER_ATTN_DW = 1
        t5, s5, a5, g5 = run(batch, rows4)
        order1 = list(seen)
        E.DEFER_NORM_DW = False              # RMSNorm scale gradients: one column-sum launch per norm instead of per group
        t6, s6, a6, g6 = run(batch, rows4)
    finally:
        eng.grad_hook, E.DEFER_ATTN_DW, E.DEFER_NORM_DW = old_hook, old_defer, True
    assert torch.equal(g4, g1), "the default grouping is reproducible"
    ob, nb_ = m.group_range("backbone")
    assert torch.equal(g5[ob:ob + nb_], g1[ob:ob + nb_]), "backbone: deferred attention weight gradients are the same tiles, the same bits"
    # the depth decoder's small weight gradients: deferred = one direct bf16 product each (fp32 sum over all 16384 rows);
    # per layer = fp32 split-K slabs + column sums.  Different summation orders of the same sums.
    od, nd = m.group_range("decoder")
    dd_, d1_ = g5[od:od + nd].float(), g1[od:od + nd].float()
    assert float((dd_ - d1_).abs().max()) <= 2e-2 * float(d1_.abs().max()) and rel(float(dd_.norm()), float(d1_.norm())) < 1e-3
    oo = torch.ones_like(g1, dtype=torch.bool)
    oo[od:od + nd] = False
    assert torch.equal(g5[oo], g1[oo]), "everything outside the decoder's range is bit-identical"
    assert torch.equal(g6, g5), "grouped RMSNorm scale-gradient reductions must not change a bit (same grouping of the GEMMs)"
    bb3 = [i for p_, i in order3 if p_ == "backbone"]
    assert sorted(bb3) == list(range(16)) and bb3[-1] == 0 and sorted(order3) == sorted(order1)
    assert [i for p_, i in order1 if p_ == "backbone"] == list(range(15, -1, -1))


def test_config5_decode_kernels_at_csm1b_shapes_vs_oracle(dev):
    """The decode step's kernels at CSM-1B's own shapes against the oracle's arithmetic (fp32 on the same bf16 inputs):
    matrix-vector products with K = 2048 / 8192 and N = 3072 (q|k|v) / 16384 (w1|w3, with the RMSNorm prologue and the SwiGLU
    epilogue) / 2048 (w2, with residual), and cache attention with RoPE + append against a 2048-slot cache at position ~200,
    backbone (32 heads / 8 kv, hd 64) and decoder (8 / 2, hd 128) geometry."""
    from csm.hip import ops
    from csm.models.model import llama3_rope_table
    g = torch.Generator().manual_seed(4242)
    for B in (1, 2):
        x = torch.randn(B, 2048, generator=g).to(BF)
        w = (1 + 0.1 * torch.randn(2048, generator=g)).to(BF)
        xn = O.rmsnorm(x, w, 1e-5)                                                   # torchtune: normalised in fp32, cast to bf16, scaled
        # q|k|v projection with the norm in the prologue
        W = (torch.randn(3072, 2048, generator=g) * 0.02).to(BF)
        y = torch.empty(B, 3072, dtype=BF, device=dev)
        ops.gemv_ex(x.to(dev), W.to(dev), y, norm_scale=w.to(dev), eps=1e-5)
        gclose("gemv_ex qkv", y, xn.float() @ W.float().t(), 1.5e-2)
        # w1|w3 (gate / up interleaved rows) with norm prologue and SwiGLU epilogue
        W13 = (torch.randn(16384, 2048, generator=g) * 0.02).to(BF)
        act = torch.empty(B, 8192, dtype=BF, device=dev)
        ops.gemv_ex(x.to(dev), W13.to(dev), act, norm_scale=w.to(dev), eps=1e-5, swiglu=True)
        gu = (xn.float() @ W13.float().t()).to(BF).float()                            # the unfused path rounds gate / up to bf16
        ref = torch.nn.functional.silu(gu[:, 0::2]) * gu[:, 1::2]
        gclose("gemv_ex w13 + swiglu", act, ref, 2e-2)
        # w2 with residual, K = 8192
        a_in = (torch.randn(B, 8192, generator=g) * 0.5).to(BF)
        W2 = (torch.randn(2048, 8192, generator=g) * 0.02).to(BF)
        R = torch.randn(B, 2048, generator=g).to(BF)
        y2 = torch.empty(B, 2048, dtype=BF, device=dev)
        ops.gemv(a_in.to(dev), W2.to(dev), y2, residual=R.to(dev))
        gclose("gemv w2 + residual", y2, a_in.float() @ W2.float().t() + R.float(), 1e-2)
        # codebook-0 head, fp32 logits
        Wh = (torch.randn(2112, 2048, generator=g) * 0.02).to(BF)
        lg = torch.empty(B, 2112, dtype=torch.float32, device=dev)
        ops.gemv(x.to(dev), Wh.to(dev), lg)
        gclose("gemv head f32", lg, x.float() @ Wh.float().t(), 1e-4)
    for H, KV, hd, S_max in ((32, 8, 64, 2048), (8, 2, 128, 2048), (8, 2, 128, 32)):
        B = 2
        table = O.rope_table(S_max, hd)
        assert torch.equal(table, llama3_rope_table(S_max, hd, 500000.0, 32.0))
        qkv = torch.randn(B, (H + 2 * KV) * hd, generator=g).to(BF)
        posv = [min(S_max - 1, 201), min(S_max - 1, 17)]
        pos = torch.tensor(posv, dtype=torch.int32)
        kc = torch.zeros(B, KV, S_max, hd).to(BF)
        vc = torch.zeros(B, KV, S_max, hd).to(BF)
        for b in range(B):                                                          # a filled history: post-RoPE keys, values
            kc[b, :, :posv[b]] = torch.randn(KV, posv[b], hd, generator=g).to(BF)
            vc[b, :, :posv[b]] = torch.randn(KV, posv[b], hd, generator=g).to(BF)
        kd, vd = kc.to(dev), vc.to(dev)
        out = torch.empty(B, H * hd, dtype=BF, device=dev)
        ops.attn_decode_rope(qkv.to(dev), kd, vd, out, pos.to(dev), table.to(dev).contiguous(), H, KV, hd)
        q = qkv[:, :H * hd].view(B, 1, H, hd)
        k = qkv[:, H * hd:(H + KV) * hd].view(B, 1, KV, hd)
        v = qkv[:, (H + KV) * hd:].view(B, KV, hd)
        qr = O.rope(q, table, pos.long().view(B, 1))[:, 0].float()                  # [B,H,hd], bf16-rounded like the kernel's
        kr = O.rope(k, table, pos.long().view(B, 1))[:, 0]
        for b in range(B):
            n = posv[b] + 1
            assert torch.equal(kd[b, :, n - 1].cpu(), kr[b]), "appended key row = RoPE of the new key (bf16)"
            assert torch.equal(vd[b, :, n - 1].cpu(), v[b]), "appended value row"
            kk, vv = kc[b].float(), vc[b].float()
            kk[:, n - 1], vv[:, n - 1] = kr[b].float(), v[b].float()
            for h in range(0, H, max(1, H // 8)):
                kvh = h // (H // KV)
                p = torch.softmax(kk[kvh, :n] @ qr[b, h] / hd ** 0.5, dim=0)
                gclose(f"attn_decode_rope H{H} hd{hd} b{b} h{h}", out[b, h * hd:(h + 1) * hd], p @ vv[kvh, :n], 1.5e-2)


def test_config5_generate_csm1b_125_frames(dev, csm1b):
    """BASELINE config 5's loop at CSM-1B size: a prompt of 40 text positions + 5 s of context audio (62 frames + EOS frame),
    then 125 frames (10 s) x 32 codebooks through ``Model.generate_frame`` with pinned Exp(1) draws.  Graph replay == eager,
    bit for bit, over all 125 frames; the frames are valid codes; against the cache-free recompute path (MFMA tiles over the
    whole prefix instead of matrix-vector kernels against the KV caches) the first two frames are equal up to their first
    difference, which must be a sampler near-tie within bf16 logit error (checked on the recompute path's logits)."""
    m = csm1b
    K, V = m.args.audio_num_codebooks, m.args.audio_vocab_size
    g = torch.Generator().manual_seed(99)
    n_text, n_ctx = 40, 63
    S = n_text + n_ctx
    tokens = torch.zeros(1, S, K + 1, dtype=torch.long)
    mask = torch.zeros(1, S, K + 1, dtype=torch.bool)
    tokens[0, :n_text, K] = torch.randint(0, m.args.text_vocab_size, (n_text,), generator=g)
    mask[0, :n_text, K] = True
    tokens[0, n_text:, :K] = torch.randint(0, 2048, (n_ctx, K), generator=g)
    tokens[0, -1, :K] = 0                                                            # the context segment's EOS frame
    mask[0, n_text:, :K] = True
    n_frames = 125
    noise = torch.empty(n_frames, K, 1, V).exponential_(1.0, generator=g)
    amask = torch.cat([torch.ones(1, K, dtype=torch.bool), torch.zeros(1, 1, dtype=torch.bool)], 1).unsqueeze(1)

    def run(use_graph, frames, use_cache=True, history=None):
        m.use_hip_graph, m.use_kv_cache = use_graph, use_cache
        m.setup_caches(1)
        m.reset_caches()
        cur_t, cur_m, cur_p = tokens, mask, torch.arange(S).unsqueeze(0)
        out = []
        for f in range(frames):
            fr = m.generate_frame(cur_t, cur_m, cur_p, 0.9, 50, noise=list(noise[f]))
            out.append(fr)
            nxt = fr if history is None else history[f].to(fr.device)
            cur_t = torch.cat([nxt.long().cpu(), torch.zeros(1, 1, dtype=torch.long)], 1).unsqueeze(1)
            cur_m, cur_p = amask, cur_p[:, -1:] + 1
        return torch.stack([x.cpu() for x in out])                                  # [frames, 1, K]

    try:
        eager = run(False, n_frames)
        graph = run(True, n_frames)
        assert m._decode_state.graph is not None, "frames >= 2 must have gone through the captured graph"
        assert eager.shape == (n_frames, 1, K) and eager.dtype == torch.int32
        assert int(eager.min()) >= 0 and int(eager.max()) < V
        assert torch.equal(eager, graph), "graph replay must reproduce the eager KV-cache frames bit for bit (125 frames)"
        assert len({tuple(f.flatten().tolist()) for f in eager}) > n_frames // 2, "frames must differ (fresh noise, moving state)"
        # The cache-free path (prefix recompute through the training kernels: MFMA tiles over the whole history) against the
        # KV-cache path (matrix-vector kernels) on the first two frames, fed the same history: different reduction orders, so the
        # logits differ by bf16 rounding and a near-tie of the sampler may go the other way - after which the rest of the frame
        # conditions on a different code.  Checked property: the frames are equal up to their first difference, and at that
        # codebook the KV-cache path's pick is a POSSIBLE outcome of the sampler on the recompute path's logits when every logit
        # may move by the error of bf16 activations (it can make the top-k cut, and every token that would certainly beat it in
        # the race l / T - log q can be cut out).
        eng = m._engine
        eng.capture_logits = []
        try:
            rc = run(False, 2, use_cache=False, history=eager)
            logits = [x.float().cpu() for x in eng.capture_logits]                   # 2 frames x K codebooks, [1, V] each
        finally:
            eng.capture_logits = None
        assert len(logits) == 2 * K
        topk_, T_ = 50, 0.9
        n_checked = 0
        for f in range(2):
            diff = (rc[f, 0] != eager[f, 0]).nonzero().flatten()
            if diff.numel() == 0:
                continue
            i_star = int(diff[0])
            lg = logits[f * K + i_star][0].double()
            qrow = noise[f, i_star, 0].double()
            r, gk = int(rc[f, 0, i_star]), int(eager[f, 0, i_star])
            eps = 2.0 ** -7 * max(1.0, float(lg.abs().max()))
            assert int((lg - eps > lg[gk] + eps).sum()) <= topk_ - 1, (f, i_star, "the KV-cache pick cannot make the top-k cut", float(lg[gk]), float(lg[r]))
            score_lo = (lg - eps) / T_ - qrow.log()
            for j in (score_lo > (float(lg[gk]) + eps) / T_ - float(qrow[gk].log())).nonzero().flatten().tolist():
                if j == gk:
                    continue
                others_above = int((lg + eps > lg[j] - eps).sum()) - 1
                assert others_above >= topk_, (f, i_star, f"token {j} beats the KV-cache pick {gk} by more than the logit error and cannot be "
                                                          f"cut out: not a near-tie", float(lg[j]), float(lg[gk]), r)
            n_checked += 1
        print(f"config 5: recompute vs KV-cache, first two frames: {n_checked} near-tie divergence(s) checked, "
              f"agreement {[(rc[f] == eager[f]).float().mean().item() for f in range(2)]}")
    finally:
        m.use_hip_graph, m.use_kv_cache = True, True


def test_config5_batched_generation_csm1b_matches_single(dev, csm1b):
    """generate_batch's kernels at CSM-1B shapes (round 4: two to four rows through the register matrix-vector kernel - K = 1024,
    2048, 8192 - and the depth decoder's attention with the shared host position): three utterances with prompts of different
    lengths decoded together sample exactly the frames each samples alone, given the same Exp(1) draws - prefill frame, first
    decode frame (eager) and the frames replayed from the captured graph."""
    m = csm1b
    K, V = m.args.audio_num_codebooks, m.args.audio_vocab_size
    g = torch.Generator().manual_seed(131)
    prompts = []
    for S in (23, 40, 31):
        tokens = torch.zeros(S, K + 1, dtype=torch.long)
        mask = torch.zeros(S, K + 1, dtype=torch.bool)
        n_text = S // 2
        tokens[:n_text, K] = torch.randint(0, m.args.text_vocab_size, (n_text,), generator=g)
        mask[:n_text, K] = True
        tokens[n_text:, :K] = torch.randint(0, 2048, (S - n_text, K), generator=g)
        mask[n_text:, :K] = True
        prompts.append((tokens, mask))
    n_frames = 5
    noise = [[torch.empty(3, V).exponential_(1.0, generator=g) for _ in range(K)] for _ in range(n_frames)]
    amask = torch.cat([torch.ones(1, K, dtype=torch.bool), torch.zeros(1, 1, dtype=torch.bool)], 1).unsqueeze(1)

    def run(rows):
        B = len(rows)
        m.use_hip_graph, m.use_kv_cache = True, True
        m.setup_caches(B)
        m.reset_caches()
        fr = [m.engine.generate_first_frames([prompts[r][0] for r in rows], [prompts[r][1] for r in rows], 0.9, 50,
                                             noise=[n[rows] for n in noise[0]])]
        for f in range(1, n_frames):
            tok = torch.cat([fr[-1].long().cpu(), torch.zeros(B, 1, dtype=torch.long)], 1).unsqueeze(1)
            fr.append(m.generate_frame(tok, amask.expand(B, -1, -1), torch.ones(B, 1, dtype=torch.long), 0.9, 50,
                                       noise=[n[rows] for n in noise[f]]))
        return torch.stack([x.cpu() for x in fr], 1)                # [B, frames, K]

    try:
        all3 = run([0, 1, 2])
        assert all3.shape == (3, n_frames, K)
        assert m._decode_state.graph is not None, "the later frames went through the captured graph"
        for r in range(3):
            assert torch.equal(all3[r:r + 1], run([r])), f"utterance {r}: batched frames differ from the frames it samples alone"
        pair = run([2, 0])
        assert torch.equal(pair[0], all3[2]) and torch.equal(pair[1], all3[0]), "two rows, other order"
        assert not torch.equal(all3[0], all3[1])
    finally:
        m.setup_caches(1)


def test_config1_lora_trainer_csm1b(dev, tmp_path):
    """BASELINE config 1's plumbing on the device: CSM-1B random-init, a dataset of 2 synthetic text+audio segments per item,
    one epoch of ``CSMLoRATrainer.train`` (get_batch protocol, batch 2), then ``save_model``: returns a float, every step's
    loss is finite, B moved away from 0, the base weights did not move, and the adapter file carries the reference's
    ``{stack}.layers.{i}.attn.{q_proj,v_proj}.lora_{A,B}`` names for all 16 + 4 layers with the reference shapes."""
    from safetensors.torch import load_file
    from csm.data import SyntheticCSMDataset
    from csm.models.model import Model
    from csm.training.lora_trainer import CSMLoRATrainer
    from csm.training.trainer import csm_1b_args
    m = Model(csm_1b_args(), device=dev, seed=0)
    m.acoustic_mode = "amortized"
    base_before = m.arena.clone()
    tr = CSMLoRATrainer("", str(tmp_path / "lora"), model=m, lora_r=8, lora_alpha=16.0, target_modules=["q_proj", "v_proj"])
    tr.logger.setLevel(40)
    assert tr.model is m and m.lora.num_params() == 958464
    ds = SyntheticCSMDataset(4, 256, seed=3, n_segments=2)
    losses = []
    step = tr.train_step
    tr.train_step = lambda b: losses.append(step(b)) or losses[-1]
    best = tr.train(ds, batch_size=2, epochs=1, max_grad_norm=1.0)
    assert isinstance(best, float) and best == float("inf"), "no validation set: best_loss stays inf (reference mlx_trainer.py:876)"
    assert tr.global_step == 2 and len(losses) == 2 and all(math.isfinite(float(x)) for x in losses)
    assert torch.equal(m.arena, base_before), "LoRA training must not touch the base weights"
    assert any(float(ad.B.float().abs().max()) > 0 for ad in m.lora.adapters.values()), "B must have moved away from 0"
    path = tr.save_model(str(tmp_path / "lora" / "adapter"), "lora")
    sd = load_file(path + ".safetensors")
    want = {f"{st}.layers.{i}.attn.{mod}.lora_{ab}" for st, n in (("backbone", 16), ("decoder", 4)) for i in range(n)
            for mod in ("q_proj", "v_proj") for ab in ("A", "B")}
    assert set(sd) == want, sorted(set(sd) ^ want)[:6]
    assert tuple(sd["backbone.layers.0.attn.q_proj.lora_A"].shape) == (8, 2048) and tuple(sd["backbone.layers.0.attn.q_proj.lora_B"].shape) == (2048, 8)
    assert tuple(sd["backbone.layers.0.attn.v_proj.lora_B"].shape) == (512, 8) and tuple(sd["decoder.layers.3.attn.v_proj.lora_B"].shape) == (256, 8)
    meta = json.load(open(path + "_metadata.json"))
    assert meta["lora_r"] == 8 and meta["target_modules"] == ["q_proj", "v_proj"] and meta["params_count"] == 958464
