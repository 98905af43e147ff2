# What bounds the generated dQ loop?  Libraries with one ingredient removed (results are wrong; timing only).
# Build (in the container):  bash tools/probes/ablate_a64dq.sh build      Run (GPU box): bash tools/probes/ablate_a64dq.sh
C=csm-train-pytorch_amd/csrc
D=tools/probes/build/abl
if [ "$1" = build ]; then
  mkdir -p $D
  for b in 1 2 4 8 16 6 23; do
    CSM_A64DQ_ABLATE=$b python3 tools/gen/gen_attn64_dq_loop.py $D/attn64_dq_loop.inc &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$C -mllvm -amdgpu-spill-vgpr-to-agpr=0 -include $D/attn64_dq_loop.inc -c $C/attention64_asm.hip -o $D/a64asm_$b.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libab_a64dq_$b.so $(ls $C/build/*.o | grep -v attention64_asm) $D/a64asm_$b.o || exit 1
  done
  rm -f $D/*.o $D/attn64_dq_loop.inc
  exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in csm-train-pytorch_amd/csm/hip/libcsm_hip.so $D/libab_a64dq_1.so $D/libab_a64dq_2.so $D/libab_a64dq_4.so $D/libab_a64dq_8.so $D/libab_a64dq_16.so $D/libab_a64dq_6.so $D/libab_a64dq_23.so; do
  echo "== $lib"
  CSM_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/abq -- python3 tools/probes/a64_time.py > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob("gpurun_out/r04/abq/*/*kernel_stats.csv")[0])):
    if "attn64_dq_asm" in r["Name"] or "attn64_dkv_asm" in r["Name"]:
        print(f'   {r["Name"][23:50]:28s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}')
PY
  rm -rf gpurun_out/r04/abq
done
