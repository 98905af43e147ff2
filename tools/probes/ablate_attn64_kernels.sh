# Per-kernel view of tools/probes/ablate_attn64.sh: libraries with -DCSM_ATT64_ABLATE=<bits> on attention64.hip (1 no exp2,
# 2 no barrier / DMA wait, 4 no LDS fragment reads, 8 no MFMA), `build` here, then on the GPU box each one under
# rocprofv3 --kernel-trace --stats: average duration of attn64_fwd / attn64_dq / attn64_dkv.
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=csm-train-pytorch_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/probes/build/abl
  for b in 1 2 4 8 5; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DCSM_ATT64_ABLATE=$b -c $C/attention64.hip -o tools/probes/build/abl/a64_$b.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/build/abl/libcsm_a$b.so $(ls $C/build/*.o | grep -v attention64) tools/probes/build/abl/a64_$b.o
  done
  exit
fi
export TMPDIR=/tmp
for v in ${ABL_LIST:-0 1 2 4 5}; do
  lib=tools/probes/build/abl/libcsm_a$v.so
  [ $v = 0 ] && lib=csm-train-pytorch_amd/csm/hip/libcsm_hip.so
  rm -rf /tmp/abl_prof
  CSM_HIP_LIB=$PWD/$lib AB_VARIANTS=0 timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_prof -- python3 tools/attn_bench.py > /dev/null 2>&1
  f=$(find /tmp/abl_prof -name "*kernel_stats.csv" | head -1)
  [ -z "$f" ] && { echo "== ablate $v: no kernel_stats.csv"; continue; }
  python3 - "$v" "$f" <<'PY'
import csv, sys
import re
rows = {re.search(r"attn64_\w+", r["Name"]).group(): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[2])) if "attn64_" in r["Name"]}
print(f"== ablate {sys.argv[1]}: " + " | ".join(f"{k} {v:.1f} us" for k, v in sorted(rows.items())))
PY
done
