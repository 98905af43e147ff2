"""Mint the golden fixtures in tests/golden/ (run ONLY in the build container).

    python tests/golden/make_golden.py

Three sources pin the oracle (oracle/csm_oracle.py):

1. The reference's own code, loaded by file path from /root/reference:
   ``src/csm/models/model.py`` (``Model._embed_tokens``, ``_embed_audio``, ``sample_topk``,
   ``generate_frame``) and ``src/csm/training/utils.py`` (``compute_loss``).  torchtune is absent
   here, so - exactly as the reference's own tests do with ``MockTransformer``
   (src/csm/training/test_training.py:73-99) - the two Llama stacks the reference would obtain from
   torchtune are stand-in modules.  The stand-in runs the oracle's transformer, so what these
   fixtures pin is everything AROUND the stacks: embedding offsets, mask-sum, logits slicing, CE
   reduction, loss weighting, frame generation order, sampler maths.
2. The HF ``CsmForConditionalGeneration`` installed in the container (independent implementation of
   the same architecture) pins the stack arithmetic itself: RMSNorm, Llama-3 scaled RoPE, GQA,
   SwiGLU, depth decoder + per-codebook heads.
3. torch's own AdamW / clip_grad_norm_ pin the optimiser restatement.

Outputs are *data only* (inputs are regenerated from seeds by the oracle; expected outputs stored).
The fixtures travel to the GPU box; /root/reference does not.
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import csm_oracle as O  # noqa: E402

REF = "/root/reference/src/csm"
TINY = O.tiny_cfg()


# ----------------------------------------------------------------------------------------------
# stand-in for the torchtune stacks (see module docstring, item 1)
class StandInStack(nn.Module):
    """Has the surface ``Model`` touches: tok_embeddings/output, caches API, forward(h,input_pos,mask)."""

    def __init__(self, c: O.StackCfg, prefix: str):
        super().__init__()
        self.c, self.prefix = c, prefix
        self.max_seq_len = c.max_seq_len
        self.tok_embeddings = nn.Embedding(4, c.dim)
        self.output = nn.Identity()
        self.flat = nn.ParameterDict()
        for name, shape in O.stack_param_shapes(prefix, c).items():
            self.flat[name.replace(".", "/")] = nn.Parameter(torch.zeros(shape))
        self._hist = None
        self._caches = False

    def params(self):
        return {k.replace("/", "."): v for k, v in self.flat.items()}

    def setup_caches(self, *a, **k):
        self._caches = True

    def caches_are_enabled(self):
        return self._caches

    def reset_caches(self):
        self._hist = None

    def forward(self, h, input_pos=None, mask=None):
        # stateful "KV cache": keep every input seen since reset and recompute the prefix.
        if self._caches and input_pos is not None and self._hist is not None and int(input_pos[0, 0]) > 0:
            full = torch.cat([self._hist, h], dim=1)
        else:
            full = h
        if self._caches:
            self._hist = full.detach()
        B, S, _ = full.shape
        pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
        out = O.transformer(self.params(), self.prefix, self.c, full, pos)
        return out[:, -h.shape[1]:]


def load_reference():
    made = []

    def factory(**kw):
        c = TINY.backbone if kw["embed_dim"] == 2048 else TINY.decoder
        st = StandInStack(c, "backbone" if kw["embed_dim"] == 2048 else "decoder")
        made.append(st)
        return st

    tt = types.ModuleType("torchtune")
    tt.models = types.ModuleType("torchtune.models")
    tt.models.llama3_2 = types.ModuleType("torchtune.models.llama3_2")
    tt.models.llama3_2.llama3_2 = factory
    tt.modules = types.ModuleType("torchtune.modules")
    tt.modules.transformer = types.ModuleType("torchtune.modules.transformer")
    tt.modules.transformer.TransformerDecoder = StandInStack
    for k, v in {"torchtune": tt, "torchtune.models": tt.models, "torchtune.models.llama3_2": tt.models.llama3_2,
                 "torchtune.modules": tt.modules, "torchtune.modules.transformer": tt.modules.transformer}.items():
        sys.modules[k] = v

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    ref_model = load("ref_model", f"{REF}/models/model.py")
    ref_utils = load("ref_utils", f"{REF}/training/utils.py")
    return ref_model, ref_utils


def build_ref_model(ref_model, params):
    args = ref_model.ModelArgs("llama-1B", "llama-100M", TINY.text_vocab, TINY.audio_vocab, TINY.n_codebooks)
    m = ref_model.Model(args)
    # the stand-in keeps its weights under flat/<name with slashes>; everything else uses reference names
    sd = {}
    for k, v in params.items():
        if k.startswith("backbone.") or k.startswith("decoder."):
            top = k.split(".")[0]
            sd[f"{top}.flat.{k.replace('.', '/')}"] = v
        else:
            sd[k] = v
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("tok_embeddings" in k for k in missing), missing
    return m


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()[:16]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    meta = {}
    ref_model, ref_utils = load_reference()
    params = O.init_params(TINY, seed=11)
    model = build_ref_model(ref_model, params)
    # reference state-dict names (minus the stand-in internals) must equal the oracle's inventory
    ref_names = sorted(k for k in model.state_dict().keys() if ".flat." not in k and "tok_embeddings" not in k
                       and "causal_mask" not in k)
    assert ref_names == sorted(k for k in params if not (k.startswith("backbone.") or k.startswith("decoder."))), ref_names

    B, S = 2, 24
    tokens, mask, targets = O.synthetic_batch(TINY, B, S, seed=5)

    # -- 1a. embedding -------------------------------------------------------------------------
    e_ref = model._embed_tokens(tokens)
    e_or = O.embed_tokens(params, TINY, tokens)
    assert torch.equal(e_ref, e_or), "embed_tokens differs from reference"
    h_ref = (e_ref * mask.unsqueeze(-1)).sum(dim=2)
    assert torch.equal(h_ref, O.embed_masked_sum(params, TINY, tokens, mask))
    ea_ref = model._embed_audio(2, tokens[:, :, 2])
    assert torch.equal(ea_ref, params["audio_embeddings.weight"][tokens[:, :, 2] + 2 * TINY.audio_vocab])

    # -- 1b. compute_loss (reference defect C.1 patched the way its own tests do) ---------------
    model.setup_caches(B)
    model._index_causal_mask = ref_model._index_causal_mask
    for st in (model.backbone, model.decoder):
        st._caches = False  # training call: plain causal forward (appendix A, last bullet)
    loss_ref, det_ref = ref_utils.compute_loss(model, tokens, mask, targets, 100.0, 1.0)
    loss_or, det_or = O.compute_loss(params, TINY, tokens, mask, targets, 100.0, 1.0, acoustic_rows="off")
    d = abs(float(loss_ref) - float(loss_or))
    assert d <= 1e-5 * abs(float(loss_ref)), (float(loss_ref), float(loss_or))
    assert float(det_ref["acoustic_loss"]) == 0.0
    meta["compute_loss_ref"] = float(loss_ref)
    meta["compute_loss_oracle"] = float(loss_or)
    meta["semantic_loss_ref"] = float(det_ref["semantic_loss"])

    # -- 1c. sampler with injected noise ----------------------------------------------------------
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(5, TINY.audio_vocab, generator=g) * 3
    torch.manual_seed(77)
    s_ref = ref_model.sample_topk(logits.clone(), 10, 0.9)
    torch.manual_seed(77)
    q = torch.empty_like(logits).exponential_(1)
    s_or = O.sample_topk(logits, 10, 0.9, q)
    assert torch.equal(s_ref, s_or) and s_ref.dtype == s_or.dtype
    # full-size vocabulary vectors for the HIP sampler test
    logits_big = torch.randn(8, 2051, generator=g) * 2.5
    q_big = torch.empty(8, 2051).exponential_(1, generator=g)
    s_big = O.sample_topk(logits_big, 50, 0.9, q_big)

    # -- 1d. generate_frame ---------------------------------------------------------------------------
    for st in (model.backbone, model.decoder):
        st._caches = True
    model.reset_caches()
    n_prompt = 9
    pt, pm = tokens[:1, :n_prompt], mask[:1, :n_prompt]
    pos = torch.arange(n_prompt).unsqueeze(0)
    frames_ref, frames_or = [], []
    cur_t, cur_m, cur_p = pt, pm, pos
    all_t, all_m = pt, pm
    K = TINY.n_codebooks
    for step in range(8):
        torch.manual_seed(1000 + step)
        with torch.no_grad():
            f_ref = model.generate_frame(cur_t, cur_m, cur_p, 0.9, 10)
        torch.manual_seed(1000 + step)
        qs = [torch.empty(1, TINY.audio_vocab).exponential_(1) for _ in range(K)]
        with torch.no_grad():
            f_or = O.generate_frame(params, TINY, all_t, all_m, 0.9, 10, qs)
        assert torch.equal(f_ref, f_or), (step, f_ref, f_or)
        frames_ref.append(f_ref)
        nxt = torch.cat([f_ref.long(), torch.zeros(1, 1, dtype=torch.long)], dim=1).unsqueeze(1)
        nm = torch.cat([torch.ones(1, K, dtype=torch.bool), torch.zeros(1, 1, dtype=torch.bool)], dim=1).unsqueeze(1)
        all_t, all_m = torch.cat([all_t, nxt], 1), torch.cat([all_m, nm], 1)
        cur_t, cur_m, cur_p = nxt, nm, cur_p[:, -1:] + 1
    meta["generate_frames"] = torch.cat(frames_ref).tolist()
    # the same for a batch of two prompts (rows decode independently; the noise of a frame is drawn [B, V] per codebook)
    model.reset_caches()
    cur_t, cur_m, cur_p = tokens[:, :n_prompt], mask[:, :n_prompt], torch.arange(n_prompt).unsqueeze(0).repeat(2, 1)
    all_t, all_m = cur_t, cur_m
    frames_b2 = []
    for step in range(6):
        torch.manual_seed(2000 + step)
        with torch.no_grad():
            f_ref = model.generate_frame(cur_t, cur_m, cur_p, 0.9, 10)
        torch.manual_seed(2000 + step)
        qs = [torch.empty(2, TINY.audio_vocab).exponential_(1) for _ in range(K)]
        with torch.no_grad():
            f_or = O.generate_frame(params, TINY, all_t, all_m, 0.9, 10, qs)
        assert torch.equal(f_ref, f_or), (step, f_ref, f_or)
        frames_b2.append(f_ref.tolist())
        nxt = torch.cat([f_ref.long(), torch.zeros(2, 1, dtype=torch.long)], dim=1).unsqueeze(1)
        nm = torch.cat([torch.ones(2, K, dtype=torch.bool), torch.zeros(2, 1, dtype=torch.bool)], dim=1).unsqueeze(1)
        all_t, all_m = torch.cat([all_t, nxt], 1), torch.cat([all_m, nm], 1)
        cur_t, cur_m, cur_p = nxt, nm, cur_p[:, -1:] + 1
    meta["generate_frames_b2"] = frames_b2

    # -- 2. HF cross-check of the stack arithmetic -----------------------------------------------------
    meta["hf_crosscheck"] = hf_crosscheck(params)

    # -- 3. optimiser restatement vs torch ---------------------------------------------------------------
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(1000, generator=g)
    grads = [torch.randn(1000, generator=g) for _ in range(3)]
    pt_ = nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt_], lr=1e-3, weight_decay=0.01)
    po, m_, v_ = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for i, gr in enumerate(grads):
        pt_.grad = gr.clone()
        opt.step()
        O.adamw_step(po, gr, m_, v_, i + 1, 1e-3)
    assert torch.allclose(pt_.detach(), po, rtol=1e-6, atol=1e-7)
    gl = [torch.randn(50, generator=g) * 3 for _ in range(4)]
    pl = [nn.Parameter(torch.zeros(50)) for _ in range(4)]
    for p_, g_ in zip(pl, gl):
        p_.grad = g_.clone()
    n_t = torch.nn.utils.clip_grad_norm_(pl, 1.0)
    go = [g_.clone() for g_ in gl]
    n_o, _ = O.clip_grad_norm(go, 1.0)
    assert torch.allclose(n_t, n_o) and all(torch.allclose(a.grad, b, rtol=1e-6) for a, b in zip(pl, go))

    # -- 4. tiny-model full train step (oracle values; HIP path is compared to these on the GPU) --------
    ptrain = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    rows = torch.arange(0, B * (S - 1), 3)
    total, det = O.compute_loss(ptrain, TINY, tokens, mask, targets, 100.0, 1.0, acoustic_rows=rows)
    total.backward()
    gsel = {}
    for k in ["backbone.layers.0.attn.q_proj.weight", "backbone.layers.1.mlp.w2.weight", "backbone.norm.scale",
              "decoder.layers.0.attn.k_proj.weight", "decoder.layers.1.mlp.w1.weight", "projection.weight",
              "codebook0_head.weight", "audio_head", "text_embeddings.weight", "audio_embeddings.weight"]:
        gsel[k] = ptrain[k].grad.clone()
    meta["train_step"] = {"total": float(total), "semantic": float(det["semantic_loss"]),
                          "acoustic": float(det["acoustic_loss"]), "rows_stride": 3,
                          "grad_norm": float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ptrain.values())))}

    # -- 4b. the parameters AFTER that step: global-norm clip to 1.0, then torch's own AdamW with the four
    # learning-rate groups of reference trainer.py:143-159 (backbone x0.1, decoder x1.0, embeddings x0.5, other x1; wd 0.01)
    lr = 1e-3
    groups = {"backbone": [], "decoder": [], "embeddings": [], "other": []}
    for k, v in ptrain.items():
        groups["backbone" if "backbone" in k else "decoder" if "decoder" in k else "embeddings" if "embeddings" in k else "other"].append(v)
    torch.nn.utils.clip_grad_norm_(list(ptrain.values()), 1.0)
    opt = torch.optim.AdamW([{"params": groups["backbone"], "lr": lr * 0.1}, {"params": groups["decoder"], "lr": lr * 1.0},
                             {"params": groups["embeddings"], "lr": lr * 0.5}, {"params": groups["other"], "lr": lr}],
                            lr=lr, weight_decay=0.01)
    opt.step()
    post = {k: ptrain[k].detach().clone() for k in gsel}
    meta["train_step"]["post_adamw"] = {"lr": lr, "multipliers": [0.1, 1.0, 0.5, 1.0], "weight_decay": 0.01, "max_grad_norm": 1.0}

    # -- 5. RVQ ----------------------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(21)
    cbs = torch.randn(8, 2048, 256, generator=g)
    x = torch.randn(40, 256, generator=g) * 4
    codes = O.rvq_encode(x, cbs)
    dec = O.rvq_decode(codes, cbs)
    meta["rvq"] = {"codes_sha": sha(codes), "decode_sha": sha(dec)}

    np.savez_compressed(
        os.path.join(HERE, "golden_small.npz"),
        tokens=tokens.numpy(), mask=mask.numpy(), targets=targets.numpy(),
        embed_sum=h_ref.detach().numpy(),
        sampler_logits=logits_big.numpy(), sampler_q=q_big.numpy(), sampler_out=s_big.numpy(),
        rvq_codes=codes.numpy(), rvq_decode_head=dec[:4].numpy(),
        **{"grad::" + k: v.detach().numpy()[..., :64].copy() if v.dim() > 1 else v.detach().numpy()
           for k, v in gsel.items()},
        **{"post::" + k: v.numpy()[..., :64].copy() if v.dim() > 1 else v.numpy() for k, v in post.items()},
    )
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


def hf_crosscheck(params):
    """Oracle stacks vs HF CsmForConditionalGeneration built from config (no downloads)."""
    from transformers import CsmConfig, CsmDepthDecoderConfig, CsmForConditionalGeneration

    bb, dc = TINY.backbone, TINY.decoder
    rope_params = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
                   "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
    dcfg = CsmDepthDecoderConfig(
        num_codebooks=TINY.n_codebooks, backbone_hidden_size=bb.dim, vocab_size=TINY.audio_vocab,
        hidden_size=dc.dim, intermediate_size=dc.ffn, num_hidden_layers=dc.n_layers,
        num_attention_heads=dc.n_heads, num_key_value_heads=dc.n_kv_heads, head_dim=dc.head_dim,
        max_position_embeddings=TINY.n_codebooks + 1, rms_norm_eps=1e-5, rope_parameters=dict(rope_params))
    cfg = CsmConfig(
        num_codebooks=TINY.n_codebooks, vocab_size=TINY.audio_vocab, text_vocab_size=TINY.text_vocab,
        hidden_size=bb.dim, intermediate_size=bb.ffn, num_hidden_layers=bb.n_layers,
        num_attention_heads=bb.n_heads, num_key_value_heads=bb.n_kv_heads, head_dim=bb.head_dim,
        max_position_embeddings=2048, rms_norm_eps=1e-5, rope_parameters=dict(rope_params),
        depth_decoder_config=dcfg, codec_config=None, tie_codebooks_embeddings=False,
        attn_implementation="eager")
    hf = CsmForConditionalGeneration(cfg).eval()

    def perm_rows(w, n_heads, hd):
        # torchtune rotates interleaved pairs (2i,2i+1); HF rotates halves (i, i+hd/2)
        w = w.view(n_heads, hd // 2, 2, -1)
        return torch.cat([w[:, :, 0], w[:, :, 1]], dim=1).reshape(n_heads * hd, -1)

    def load_stack(layers, norm, prefix, c):
        for i, L in enumerate(layers):
            p = f"{prefix}.layers.{i}"
            L.input_layernorm.weight.data.copy_(params[f"{p}.sa_norm.scale"])
            L.post_attention_layernorm.weight.data.copy_(params[f"{p}.mlp_norm.scale"])
            L.self_attn.q_proj.weight.data.copy_(perm_rows(params[f"{p}.attn.q_proj.weight"], c.n_heads, c.head_dim))
            L.self_attn.k_proj.weight.data.copy_(perm_rows(params[f"{p}.attn.k_proj.weight"], c.n_kv_heads, c.head_dim))
            L.self_attn.v_proj.weight.data.copy_(params[f"{p}.attn.v_proj.weight"])
            L.self_attn.o_proj.weight.data.copy_(params[f"{p}.attn.output_proj.weight"])
            L.mlp.gate_proj.weight.data.copy_(params[f"{p}.mlp.w1.weight"])
            L.mlp.up_proj.weight.data.copy_(params[f"{p}.mlp.w3.weight"])
            L.mlp.down_proj.weight.data.copy_(params[f"{p}.mlp.w2.weight"])
        norm.weight.data.copy_(params[f"{prefix}.norm.scale"])

    # perturb norm scales so that they are actually exercised
    g = torch.Generator().manual_seed(4)
    params = dict(params)
    for k in list(params):
        if k.endswith(".scale"):
            params[k] = 1.0 + 0.1 * torch.randn(params[k].shape, generator=g)
    load_stack(hf.backbone_model.layers, hf.backbone_model.norm, "backbone", bb)
    dd = hf.depth_decoder.model
    load_stack(dd.layers, dd.norm, "decoder", dc)
    dd.inputs_embeds_projector.weight.data.copy_(params["projection.weight"])
    dd.embed_tokens.weight.data.copy_(params["audio_embeddings.weight"])
    hf.depth_decoder.codebooks_head.weight.data.copy_(params["audio_head"])

    B, S = 2, 40
    tokens, mask, targets = O.synthetic_batch(TINY, B, S, seed=8)
    h0 = O.embed_masked_sum(params, TINY, tokens, mask)
    with torch.no_grad():
        hid_or = O.transformer(params, "backbone", bb, h0, torch.arange(S).unsqueeze(0).repeat(B, 1))
        hid_hf = hf.backbone_model(inputs_embeds=h0).last_hidden_state
        K = TINY.n_codebooks
        hsel = hid_or[:, :-1].reshape(-1, bb.dim)
        codes = targets[:, :S - 1].reshape(-1, K)
        _, logits_or = O.acoustic_loss(params, TINY, hid_or, targets, None, return_logits=True)
        ids = torch.cat([torch.zeros(codes.shape[0], 1, dtype=torch.long), codes[:, :K - 1]], dim=1)
        out = hf.depth_decoder(input_ids=ids, backbone_last_hidden_state=hsel)
        logits_hf = out.logits  # HF already drops position 0
    d1 = float((hid_or - hid_hf).abs().max())
    d2 = float((logits_or - logits_hf).abs().max())
    assert d1 < 2e-4 and d2 < 2e-4, (d1, d2)
    return {"backbone_hidden_max_abs_diff": d1, "decoder_logits_max_abs_diff": d2,
            "hidden_scale": float(hid_or.abs().mean()), "logits_scale": float(logits_or.abs().mean())}


if __name__ == "__main__":
    main()
