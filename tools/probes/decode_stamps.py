"""Where does a decode kernel's time go?  Needs a library built with -DCSM_DECODE_STAMPS (tools/probes/decode_stamps.sh): the first
and last workgroup of every matrix-vector kernel record the 100 MHz wall clock at a few points.  Prints the timeline of a run of
consecutive launches in the middle of a captured frame (depth-decoder steps)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "csm-train-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import generate_bench as GB
from csm.models.model import Model
from csm.training.trainer import csm_1b_args
from csm.hip import lib
from csm.generator import Generator, Segment

os.environ.setdefault("GEN_CODEC", "rvq")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Model(csm_1b_args(), device="cuda:0", seed=0)
gen = Generator(model, text_tokenizer=GB.ByteTokenizer(), audio_tokenizer=GB.make_codec(dev))
ctx = [Segment(0, "hello there", torch.randn(5 * 24000, device=dev) * 0.1)]
gen.generate("the quick brown fox", 1, ctx, max_audio_length_ms=80 * 6)          # warm-up + capture
f = lib.csm_decode_stamps; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int]
f(None, 0)                                                                        # reset
gen.generate("the quick brown fox", 1, ctx, max_audio_length_ms=80 * 2)          # two frames (replays)
buf = np.zeros(819 * 10, dtype=np.uint64)
n = f(buf.ctypes.data, 819)
rec = buf[:n * 10].reshape(n, 10).astype(np.int64)
print("records", n)
names = {200: "attn+wo", 300: "sampler"}
def nm(i):
    if i in names: return names[i]
    i -= 100; kch, sw, nw = i // 4, (i >> 1) & 1, i & 1
    return f"gemv K={512 * kch}{' swiglu' if sw else ''}{' norm' if nw else ''}"
# pair the first / last workgroup records of one launch (same id, adjacent start times), order by start
rec = rec[np.argsort(rec[:, 2])]
t0 = rec[0, 2]
rows = []
i = 0
while i < n:
    r = rec[i]
    grp = [r]
    if i + 1 < n and rec[i + 1, 0] == r[0] and abs(rec[i + 1, 2] - r[2]) < 300 and rec[i + 1, 1] != r[1]:
        grp.append(rec[i + 1]); i += 1
    rows.append(grp); i += 1
print("launch: kernel | per workgroup (first / last): start [us since first record], then deltas in us between stamps | gap to the next launch's first start")
lo = max(0, len(rows) // 2 - 24)
for k in range(lo, min(len(rows), lo + 48)):
    grp = rows[k]
    out = []
    end = 0
    for r in grp:
        st = [int(x) for x in r[2:] if x != 0]
        d = [(b - a) / 100.0 for a, b in zip(st[:-1], st[1:])]
        out.append(f"wg{int(r[1]):4d} @{(st[0] - t0) / 100.0:9.2f} " + " ".join(f"{x:5.2f}" for x in d) + f" | total {(st[-1] - st[0]) / 100.0:5.2f}")
        end = max(end, st[-1])
    nxt = min(int(r[2]) for r in rows[k + 1]) if k + 1 < len(rows) else end
    print(f"{nm(int(grp[0][0])):24s} " + "  ||  ".join(out) + f"  -> next starts {(nxt - end) / 100.0:5.2f} us after my last stamp")
