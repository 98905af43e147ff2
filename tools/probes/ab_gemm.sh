# Same-box A/B of two builds of libcsm_hip.so on the train-step GEMM shapes (tools/gemm_bench.py, auto variant), then on the
# headline step: tools/probes/build/abl/libcsm_prev.so (previous commit, built out of tree) against the in-tree library.
cd $GRAFT_REPO_ROOT
for lib in tools/probes/build/abl/libcsm_prev.so csm-train-pytorch_amd/csm/hip/libcsm_hip.so; do
  echo "== $lib"
  CSM_HIP_LIB=$PWD/$lib GB_VARIANTS=2 GB_NOCHECK=1 GB_ROUNDS=${GB_ROUNDS:-5} timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep -E "^v2"
done
