# per-kernel times of the attention backward at the backbone's shape (B=4, S=2048, 32/8 heads): asm dK/dV vs second generation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/a64prof -- python3 tools/probes/a64_time.py > gpurun_out/r04/a64prof.log 2>&1
f=$(ls gpurun_out/r04/a64prof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
rm -rf gpurun_out/r04/a64prof
