"""Shared by make_golden_full.py (build container) and tests/test_full_shape_gpu.py (GPU box): the seeded inputs of the
full-width single-layer fixtures.  Everything is regenerated from seeds by the oracle's own initialiser; only expected
OUTPUTS are stored in the fixture."""
import torch

from oracle import csm_oracle as O

BB_L1 = O.StackCfg(2048, 1, 32, 8, 8192)            # one backbone layer at CSM-1B width (reference model.py:11-25)
DC_L1 = O.StackCfg(1024, 1, 8, 2, 8192)             # one decoder layer at CSM-100M width (reference model.py:28-42)
CFG = O.CsmCfg(backbone=BB_L1, decoder=DC_L1, text_vocab=300, audio_vocab=67, n_codebooks=4)
S_BB, S_DC, N_DC = 2048, 32, 48


def bf(x):
    """bf16-representable fp32 (what the GPU path will hold), so the fixture isolates the kernels' arithmetic."""
    return x.to(torch.bfloat16).float()


def params(seed=21):
    p = O.init_params(CFG, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in list(p):
        if k.endswith(".scale"):
            p[k] = 1.0 + 0.1 * torch.randn(p[k].shape, generator=g)       # exercise the norm scales
        p[k] = bf(p[k])
    return p


def inputs(which, seed=33):
    g = torch.Generator().manual_seed(seed + (0 if which == "backbone" else 1))
    c, B, S = (BB_L1, 1, S_BB) if which == "backbone" else (DC_L1, N_DC, S_DC)
    h = bf(torch.randn(B, S, c.dim, generator=g) * 0.5)
    gout = bf(torch.randn(B, S, c.dim, generator=g) * 0.01)
    return h, gout


def sample_idx(numel, n=512, seed=5):
    g = torch.Generator().manual_seed(seed + numel % 1000)
    return torch.randint(0, numel, (n,), generator=g)


GRAD_NAMES = ["attn.q_proj.weight", "attn.k_proj.weight", "attn.v_proj.weight", "attn.output_proj.weight", "mlp.w1.weight",
              "mlp.w2.weight", "mlp.w3.weight", "sa_norm.scale", "mlp_norm.scale"]
