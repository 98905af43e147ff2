#!/usr/bin/env python3
"""HBM traffic per launch of every kernel of one bench.py run, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Units / corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of a wide coalesced stream at 64 B, so the read side is doubled; WRITE_SIZE is exact for 16-B stores.
Writes a table to stdout and, with a third argument, a JSON keyed by bench.py's GEMM kinds (nt_fwd_bf16, ...) that
bench.py reads back into ``roofline.traffic``.
The JSON records which kernels it measured (``_meta.kernel_source_sha16`` = bench.py's hash of csrc/, and per symbol the
launches per train step): bench.py refuses the file when either disagrees with the running library.
Usage: python tools/pmc_bench_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json [train steps in the run]]
"""
import collections, csv, json, os, re, sys


def load(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "u16")
        rows[name.split("(")[0]].append(float(r["Counter_Value"]))
    return rows


def kind(name):
    if name.startswith("gemm256two_tn_kernel"):      # two bf16 weight gradients in one launch
        return "tn_wgrad_bf16"
    m = re.match(r"gemm(256p|256s|256)?_kernel<(\d), (\d), (u16|float)", name)
    if not m:
        return None
    k = {("0", "0"): "nt_fwd", ("0", "1"): "nn_dgrad", ("1", "1"): "tn_wgrad", ("1", "0"): "tt"}[(m.group(2), m.group(3))]
    return k + ("_f32" if m.group(4) == "float" else "_bf16")


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
tot = {k: (2.0 * sum(f[k]) * 1024, sum(w.get(k, [0.0])) * 1024, len(f[k])) for k in f}
print(f"{'kernel':60s} {'launches':>8s} {'read MB/launch (x2 corr.)':>26s} {'write MB/launch':>16s} {'total GB':>9s}")
for k, (rd, wr, n) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:40]:
    print(f"{k[:60]:60s} {n:8d} {rd / n / 1e6:26.1f} {wr / n / 1e6:16.1f} {(rd + wr) / 1e9:9.2f}")
print(f"all kernels: read {sum(v[0] for v in tot.values()) / 1e9:.1f} GB, write {sum(v[1] for v in tot.values()) / 1e9:.1f} GB")
if len(sys.argv) > 3:
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k, (rd, wr, n) in tot.items():
        kd = kind(k)
        if kd:
            agg[kd][0] += rd; agg[kd][1] += wr; agg[kd][2] += n
    out = {kd: {"hbm_read_bytes_per_launch": v[0] / v[2], "hbm_write_bytes_per_launch": v[1] / v[2], "launches_profiled": v[2]}
           for kd, v in agg.items()}
    # per kernel symbol (template arguments kept, argument list dropped): what bench.py's roofline.kernel names
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out["_kernels"] = {k: {"hbm_read_bytes_per_launch": rd / n, "hbm_write_bytes_per_launch": wr / n, "launches_profiled": n,
                           "launches_per_step": (n // steps if steps and n % steps == 0 else None)}
                       for k, (rd, wr, n) in tot.items() if k.startswith("gemm")}
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    out["_meta"] = {"kernel_source_sha16": bench.kernel_source_sha16(), "train_steps_profiled": steps}
    out["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --steps 2 --warmup 1 " \
                     "--no-cpu-baseline`; FETCH_SIZE x2 (gfx950 128-B request correction), KiB -> bytes"
    json.dump(out, open(sys.argv[3], "w"), indent=1)
