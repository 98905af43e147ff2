"""one FULL-SHAPE backbone layer (d=2048, hd=64, S=2048) and one decoder layer (d=1024, hd=128, S=32),
forward and backward in fp32, oracle vs the installed Hugging Face Csm modules at those widths, plus the Llama-3
scaled RoPE tables for positions 0..2047 at both head dims.  (Run ONLY in the build container:

    python tests/golden/make_golden_full.py

The stack arithmetic lives in torchtune 0.4.0, which the reference pins but does not vendor (pyproject.toml:18; call
sites src/csm/models/model.py:13-25,30-42): the HF port is the only independent implementation of it available here,
and the tiny-shape cross-check of make_golden.py never reaches the real widths or positions > 40.)  Stored: hashes and
512 sampled elements per tensor of the ORACLE's outputs (which the HF modules reproduce to the printed tolerance) -
tests/test_full_shape_gpu.py compares the HIP path with them at full width.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import csm_oracle as O  # noqa: E402
import full_layer_common as C  # noqa: E402


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()[:16]


def hf_stack(c, params, prefix):
    from transformers import CsmConfig
    from transformers.models.csm.modeling_csm import CsmBackboneModel
    rope_params = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
                   "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
    cfg = CsmConfig(num_codebooks=2, vocab_size=8, text_vocab_size=8, hidden_size=c.dim, intermediate_size=c.ffn,
                    num_hidden_layers=c.n_layers, num_attention_heads=c.n_heads, num_key_value_heads=c.n_kv_heads,
                    head_dim=c.head_dim, max_position_embeddings=2048, rms_norm_eps=1e-5, rope_parameters=rope_params,
                    codec_config=None, attn_implementation="eager")
    hf = CsmBackboneModel(cfg).eval()

    def perm_rows(w, n_heads, hd):      # torchtune rotates interleaved pairs (2i,2i+1); HF rotates halves (i, i+hd/2)
        w = w.view(n_heads, hd // 2, 2, -1)
        return torch.cat([w[:, :, 0], w[:, :, 1]], dim=1).reshape(n_heads * hd, -1)

    L, p = hf.layers[0], f"{prefix}.layers.0"
    with torch.no_grad():
        L.input_layernorm.weight.copy_(params[f"{p}.sa_norm.scale"])
        L.post_attention_layernorm.weight.copy_(params[f"{p}.mlp_norm.scale"])
        L.self_attn.q_proj.weight.copy_(perm_rows(params[f"{p}.attn.q_proj.weight"], c.n_heads, c.head_dim))
        L.self_attn.k_proj.weight.copy_(perm_rows(params[f"{p}.attn.k_proj.weight"], c.n_kv_heads, c.head_dim))
        L.self_attn.v_proj.weight.copy_(params[f"{p}.attn.v_proj.weight"])
        L.self_attn.o_proj.weight.copy_(params[f"{p}.attn.output_proj.weight"])
        L.mlp.gate_proj.weight.copy_(params[f"{p}.mlp.w1.weight"])
        L.mlp.up_proj.weight.copy_(params[f"{p}.mlp.w3.weight"])
        L.mlp.down_proj.weight.copy_(params[f"{p}.mlp.w2.weight"])
        hf.norm.weight.copy_(params[f"{prefix}.norm.scale"])
    names = {"attn.q_proj.weight": (L.self_attn.q_proj.weight, lambda g: unperm(g, c.n_heads, c.head_dim)),
             "attn.k_proj.weight": (L.self_attn.k_proj.weight, lambda g: unperm(g, c.n_kv_heads, c.head_dim)),
             "attn.v_proj.weight": (L.self_attn.v_proj.weight, None), "attn.output_proj.weight": (L.self_attn.o_proj.weight, None),
             "mlp.w1.weight": (L.mlp.gate_proj.weight, None), "mlp.w3.weight": (L.mlp.up_proj.weight, None),
             "mlp.w2.weight": (L.mlp.down_proj.weight, None), "sa_norm.scale": (L.input_layernorm.weight, None),
             "mlp_norm.scale": (L.post_attention_layernorm.weight, None)}
    return hf, names


def unperm(g, n_heads, hd):
    g = g.view(n_heads, 2, hd // 2, -1)
    return torch.stack([g[:, 0], g[:, 1]], dim=2).reshape(n_heads * hd, -1)


def one_stack(which, params, meta, store):
    c = C.BB_L1 if which == "backbone" else C.DC_L1
    h, gout = C.inputs(which)
    B, S, _ = h.shape
    pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
    keys = [k for k in params if k.startswith(f"{which}.")]
    pt = {k: params[k].clone().requires_grad_(True) for k in keys}
    hin = h.clone().requires_grad_(True)
    out = O.transformer(pt, which, c, hin, pos)
    (out * gout).sum().backward()
    hf, names = hf_stack(c, params, which)
    hin2 = h.clone().requires_grad_(True)
    out_hf = hf(inputs_embeds=hin2).last_hidden_state
    (out_hf * gout).sum().backward()
    diffs = {"hidden": float((out - out_hf).abs().max()), "dx": float((hin.grad - hin2.grad).abs().max())}
    scales = {"hidden": float(out.abs().mean()), "dx": float(hin.grad.abs().mean())}
    for n, (w, fix) in names.items():
        g_hf = w.grad if fix is None else fix(w.grad)
        g_or = pt[f"{which}.layers.0.{n}"].grad
        diffs["g:" + n] = float((g_or - g_hf).abs().max())
        scales["g:" + n] = float(g_or.abs().mean())
    rel = {k: diffs[k] / max(scales[k], 1e-30) for k in diffs}
    assert max(rel.values()) < 5e-4, rel        # fp32 round-off of two summation orders at K up to 8192
    meta[which] = {"max_abs_diff_vs_hf": diffs, "mean_abs": scales, "shape": list(h.shape),
                   "sha": {"hidden": sha(out), "dx": sha(hin.grad)}}
    for name, t in [("hidden", out.detach()), ("dx", hin.grad)] + [("g:" + n, pt[f"{which}.layers.0.{n}"].grad) for n in C.GRAD_NAMES]:
        idx = C.sample_idx(t.numel())
        store[f"{which}::{name}"] = t.reshape(-1)[idx].numpy()
    print(which, json.dumps(rel, indent=1))


def rope_tables(meta, store):
    from transformers import CsmConfig
    from transformers.models.csm.modeling_csm import CsmRotaryEmbedding
    rope_params = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
                   "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
    for hd, heads in ((64, 32), (128, 8)):
        tab = O.rope_table(2048, hd)                                         # [2048, hd/2, 2]
        cfg = CsmConfig(num_codebooks=2, vocab_size=8, text_vocab_size=8, hidden_size=hd * heads, num_attention_heads=heads,
                        num_key_value_heads=heads, head_dim=hd, num_hidden_layers=1, intermediate_size=64,
                        max_position_embeddings=2048, rope_parameters=rope_params, codec_config=None)
        cos, sin = CsmRotaryEmbedding(cfg)(torch.zeros(1, 1, hd), torch.arange(2048).unsqueeze(0))
        d = max(float((cos[0, :, :hd // 2] - tab[..., 0]).abs().max()), float((sin[0, :, :hd // 2] - tab[..., 1]).abs().max()))
        assert d < 2e-3, d       # fp32 angle pos*theta at pos ~ 2000: a few ulp of the argument
        meta[f"rope_hd{hd}"] = {"sha": sha(tab), "max_abs_diff_vs_hf": d}
        store[f"rope_hd{hd}::rows"] = tab[[0, 1, 2, 63, 64, 511, 1024, 2047]].numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    params = C.params()
    meta, store = {}, {}
    rope_tables(meta, store)
    one_stack("decoder", params, meta, store)
    one_stack("backbone", params, meta, store)
    np.savez_compressed(os.path.join(HERE, "golden_full_layer.npz"), **store)
    with open(os.path.join(HERE, "golden_full_layer_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
