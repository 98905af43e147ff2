#!/usr/bin/env python3
"""Per-shape GEMM time inside one real train step (HIP events around every csm_gemm launch)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.hip import ops
from csm.data import SyntheticCSMDataset, collate_variable_length
from csm.models.model import Model
from csm.training.trainer import CSMTrainer, csm_1b_args

model = Model(csm_1b_args(), device="cuda:0", seed=0)
model.acoustic_mode = "amortized"
tr = CSMTrainer("", "/tmp/gs", device="cuda:0"); tr.logger.setLevel(40); tr.model = model; tr.prepare_optimizer()
ds = SyntheticCSMDataset(4, 2048)
batch = {k: v.cuda() for k, v in collate_variable_length([ds[i] for i in range(4)]).items()}
for _ in range(2): tr.train_step(batch)
recs = []
orig_g, orig_ex = ops.lib.csm_gemm_bf16, ops.lib.csm_gemm_bf16_ex
def wrap(fn, name):
    def f(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a); e1.record()
        M, N, K, tA, tB, batch_ = a[4], a[5], a[6], a[11], a[12], a[15]
        recs.append(((name, M, N, K, tA, tB, batch_), e0, e1)); return r
    return f
ops.lib.csm_gemm_bf16 = wrap(orig_g, "plain"); ops.lib.csm_gemm_bf16_ex = wrap(orig_ex, "fused")
tr.train_step(batch); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for key, e0, e1 in recs:
    agg[key][0] += e0.elapsed_time(e1); agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"total GEMM {tot:.2f} ms in {len(recs)} launches")
for key, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    name, M, N, K, tA, tB, b = key
    fl = 2.0 * M * N * K * b * n
    print(f"{t:7.3f} ms  x{n:3d}  {name:5s} M={M:6d} N={N:6d} K={K:6d} tA={tA} tB={tB} batch={b:2d}  {fl / t / 1e9:7.1f} TF/s")
