#!/usr/bin/env python3
"""Would replaying the whole train step as ONE HIP graph shrink the ~3 ms of GPU-side gaps between its 528 kernels?
Timing probe only: the captured AdamW launches carry the capture-time step number (their bias correction is a host scalar)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd"))
import torch
from csm.data import SyntheticCSMDataset, collate_variable_length
from csm.models.model import Model
from csm.training.trainer import CSMTrainer, csm_1b_args

model = Model(csm_1b_args(), device="cuda:0", seed=0)
model.acoustic_mode = "amortized"
tr = CSMTrainer("", "/tmp/gsp", device="cuda:0"); tr.logger.setLevel(40); tr.model = model; tr.prepare_optimizer()
ds = SyntheticCSMDataset(4, 2048)
batch = {k: v.cuda() for k, v in collate_variable_length([ds[i] for i in range(4)]).items()}

def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

eager = timed(lambda: tr.train_step(batch, 1, True, 1.0))
print(f"eager  {eager:.3f} ms/step")
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): tr.train_step(batch, 1, True, 1.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tr.train_step(batch, 1, True, 1.0)
    graph = timed(g.replay)
    print(f"graph  {graph:.3f} ms/step")
except Exception as e:
    print("graph capture failed:", repr(e)[:400])
