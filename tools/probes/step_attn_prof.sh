# in-step times of the attention kernels (rocprofv3 kernel trace over a short bench run): default vs CSM_ATTN_VARIANT=5366 (gen2 backward)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 5366; do
  echo "== CSM_ATTN_VARIANT=$v"
  CSM_ATTN_VARIANT=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/sap -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/r04/sap/*/*kernel_stats.csv")[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if "attn" in r["Name"]:
        print(f'   {r["Name"][23:52]:30s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
print(f"   all kernels: {tot/1e6:.1f} ms")
PY
  rm -rf gpurun_out/r04/sap
done
