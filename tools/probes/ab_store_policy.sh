# Cache policy of the four-wave GEMM's epilogue stores (VERDICT r03 #2b): same-box A/B on the headline step.
# Build (container): bash tools/probes/ab_store_policy.sh build     Run (GPU box): bash tools/probes/ab_store_policy.sh
C=csm-train-pytorch_amd/csrc
D=tools/probes/build/abl
if [ "$1" = build ]; then
  mkdir -p $D
  i=0
  for pol in ' nt' ' sc1' ' sc0 sc1' ' sc0 sc1 nt'; do
    i=$((i+1))
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -mllvm -amdgpu-spill-vgpr-to-agpr=0 "-DCSM_W4_STORE_POLICY=\"$pol\"" -c $C/gemm256w4.hip -o $D/w4pol$i.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libab_w4pol$i.so $(ls $C/build/*.o | grep -v gemm256w4) $D/w4pol$i.o || exit 1
  done
  rm -f $D/*.o
  exit 0
fi
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in csm-train-pytorch_amd/csm/hip/libcsm_hip.so $D/libab_w4pol1.so $D/libab_w4pol2.so $D/libab_w4pol3.so $D/libab_w4pol4.so; do
    ms=$(CSM_HIP_LIB=$lib python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$lib $ms"
  done
done | tee /tmp/ab_pol.txt
echo "(1 = nt, 2 = sc1, 3 = sc0 sc1, 4 = sc0 sc1 nt)"
python - <<'PY'
import collections
d = collections.defaultdict(list)
for line in open("/tmp/ab_pol.txt"):
    k, v = line.split()
    d[k].append(float(v))
for k, v in d.items():
    v.sort(); print(f"median {v[len(v) // 2]:.3f}  min {v[0]:.3f}  max {v[-1]:.3f}   {k}")
PY
