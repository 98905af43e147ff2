# What bounds the 256x256 main loop now?  Libraries prebuilt in the build container with -DCSM_ABLATE=<bits> on gemm256.hip
# (bit0: no LDS-DMA after the prologue; bit1: no fragment reads; bit2: no MFMA; bit3: no barrier) -> tools/probes/build/abl/libcsm_g<bits>.so
cd $GRAFT_REPO_ROOT
for a in ${ABL_LIST:-0 1 2 4 8}; do
  lib=tools/probes/build/abl/libcsm_g$a.so
  [ $a = 0 ] && lib=csm-train-pytorch_amd/csm/hip/libcsm_hip.so
  echo "== CSM_ABLATE=$a"
  CSM_HIP_LIB=$PWD/$lib GB_VARIANTS=2 GB_NOCHECK=1 timeout -k 10 120 python tools/gemm_bench.py 2>&1 | grep -E "^v2"
done
