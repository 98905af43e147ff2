# SQ counters of the attention kernels (one rocprofv3 --pmc pass each set; kernel-trace only, never with hip/hsa traces)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export AB_VARIANTS=${AB_VARIANTS:-0}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02/attn_pmc1 -- python3 tools/attn_bench.py > gpurun_out/r02/attn_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/r02/attn_pmc2 -- python3 tools/attn_bench.py > gpurun_out/r02/attn_pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("attn_pmc1", "attn_pmc2"):
    f = glob.glob(f"gpurun_out/r02/{d}/*/*counter_collection.csv")
    if not f:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, c in agg.items():
        if "attn" not in k: continue
        print(k)
        for n, v in sorted(c.items()):
            print(f"   {n:28s} {v / cnt[(k, n)]:16.0f} per launch")
PY
