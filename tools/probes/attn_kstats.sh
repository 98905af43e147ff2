# per-kernel durations of tools/attn_bench.py (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export AB_VARIANTS=${AB_VARIANTS:-758,0}
rm -rf gpurun_out/r02/attn_ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/attn_ks -- python3 tools/attn_bench.py > gpurun_out/r02/attn_ks.log 2>&1
tail -3 gpurun_out/r02/attn_ks.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02/attn_ks/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "attn" in r["Name"]:
        print(f"{float(r['AverageNs'])/1e3:8.1f} us avg  x{r['Calls']:>4s}  min {float(r['MinNs'])/1e3:7.1f}  {r['Name'][:90]}")
PY
