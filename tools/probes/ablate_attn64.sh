# What bounds the head_dim-64 forward loop?  Libraries built with -DCSM_ATT64_ABLATE=<bits> (see attention64.hip) timed
# with tools/attn_bench.py: bit0 no exp2, bit1 no barrier / DMA wait, bit2 no LDS fragment reads, bit3 no MFMA, bit4 no max.
cd $GRAFT_REPO_ROOT
for v in ${ABL_LIST:-0 1 2 4 8}; do
  lib=tools/probes/build/abl/libcsm_a$v.so
  [ $v = 0 ] && lib=csm-train-pytorch_amd/csm/hip/libcsm_hip.so
  echo "== ablate $v"
  CSM_HIP_LIB=$PWD/$lib AB_VARIANTS=0,0 timeout -k 10 100 python tools/attn_bench.py 2>&1 | tail -1
done
