"""Full-WIDTH parity: one backbone layer at d=2048 / hd=64 / S=2048 and one decoder layer at d=1024 /
hd=128 / S=32, forward and backward on the HIP path, against the fixture minted by tests/golden/make_golden_full.py
(the oracle's fp32 values, which the Hugging Face Csm modules reproduce to <= 2e-4 relative at these widths)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import full_layer_common as C  # noqa: E402

BF = torch.bfloat16


def _close(what, got, ref, tol):
    got, ref = got.float().cpu(), torch.as_tensor(ref).float()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e} (tol {tol})"
    # and no systematic drift: the mean error is an order of magnitude below the bound
    assert (got - ref).abs().mean().item() <= 0.25 * tol * scale, what


@pytest.mark.parametrize("which", ["decoder", "backbone"])
def test_full_shape_layer_matches_fixture(dev, which):
    from csm.models.model import Model, ModelArgs
    z = np.load(os.path.join(HERE, "golden", "golden_full_layer.npz"))
    meta = json.load(open(os.path.join(HERE, "golden", "golden_full_layer_meta.json")))
    m = Model(ModelArgs("llama-1B-L1", "llama-100M-L1", C.CFG.text_vocab, C.CFG.audio_vocab, C.CFG.n_codebooks), device=dev)
    m.load_state_dict(C.params())
    m.ensure_grads()
    stack = m.engine.backbone if which == "backbone" else m.engine.decoder
    h, gout = C.inputs(which)
    B, S, D = h.shape
    assert [B, S, D] == meta[which]["shape"]
    xf = stack.forward(h.view(B * S, D).to(BF).to(dev), B, S, True)
    _close(f"{which} hidden", xf.reshape(-1)[C.sample_idx(xf.numel()).to(dev)], z[f"{which}::hidden"], 2e-2)
    m.grad_arena.zero_()
    dx = stack.backward(gout.view(B * S, D).to(BF).to(dev), B, S, True, 1.0, acc=False)
    _close(f"{which} dx", dx.reshape(-1)[C.sample_idx(dx.numel()).to(dev)], z[f"{which}::dx"], 2e-2)
    gv = m._views(m.grad_arena)
    for n in C.GRAD_NAMES:
        g = gv[f"{which}.layers.0.{n}"].reshape(-1)
        _close(f"{which} d({n})", g[C.sample_idx(g.numel()).to(dev)], z[f"{which}::g:{n}"], 3e-2)
