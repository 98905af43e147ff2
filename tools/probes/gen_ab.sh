# generate(): previous library build vs in-tree, same box (tools/probes/build/abl/libcsm_prev.so)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in tools/probes/build/abl/libcsm_prev.so csm-train-pytorch_amd/csm/hip/libcsm_hip.so; do
  echo -n "$(basename $lib): "; CSM_HIP_LIB=$PWD/$lib GEN_BATCH=1 python tools/generate_bench.py 2>/dev/null | tail -1
done; done
