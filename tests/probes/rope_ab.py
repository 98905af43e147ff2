import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "csm-train-pytorch_amd"))
import torch
from oracle import csm_oracle as O
from csm.models.model import Model, ModelArgs
import csm.engine as E
cfg = O.tiny_cfg()
m = Model(ModelArgs("llama-tiny-backbone", "llama-tiny-decoder", cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks), device="cuda:0")
p32 = O.init_params(cfg, seed=11); m.load_state_dict(p32)
pq = {k: v.to(torch.bfloat16).float() for k, v in p32.items()}
for seed in (3, 4, 5):
    tokens, mask, _ = O.synthetic_batch(cfg, 2, 100, seed=seed)
    ref = O.backbone_hidden(pq, cfg, tokens, mask)
    out = {}
    for f in (False, True):
        E.FUSE_ROPE_FWD = f
        hid = m.engine.hidden_states(tokens, mask).float().cpu()
        out[f] = hid
        print(seed, "fused" if f else "plain", "max err", float((hid - ref).abs().max()), "rms err", float((hid - ref).pow(2).mean().sqrt()))
    print("   fused vs plain max diff", float((out[True] - out[False]).abs().max()))
