#!/usr/bin/env python3
"""HBM traffic per launch of the GEMM kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Units / corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of a wide coalesced stream at 64 B, so the read side is doubled; WRITE_SIZE is exact for 16-B stores.
Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv>
"""
import csv, sys, collections

def load(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "gemm" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16")
        key = (name.split("(")[0], int(r["Grid_Size"]))
        rows[key].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
    return rows

f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print(f"{'kernel':55s} {'grid':>9s} {'launches':>8s} {'read MB (x2 corrected)':>22s} {'write MB':>9s} {'avg us':>8s}")
for key in sorted(f):
    fv = [v for v, _ in f[key]]
    wv = [v for v, _ in w.get(key, [(0.0, 0.0)])]
    us = [t for _, t in f[key]]
    rd = 2.0 * sum(fv) / len(fv) * 1024 / 1e6
    wr = sum(wv) / len(wv) * 1024 / 1e6
    print(f"{key[0][:55]:55s} {key[1]:9d} {len(fv):8d} {rd:22.1f} {wr:9.1f} {sum(us)/len(us):8.1f}")
