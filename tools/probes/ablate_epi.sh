# What does the fused SwiGLU-backward epilogue of the w2 dgrad GEMM pay for?  Libraries built with -DCSM_ABLATE_EPI=<bits> on
# gemm256.hip (1: no gate/up loads, 2: no stores, 4: no sigmoid arithmetic) against the in-tree one; the bit checks in
# tools/probes/touch_ab.py fail for the ablated builds by design, hence TOUCH_NOCHECK.
cd $GRAFT_REPO_ROOT
for lib in csm-train-pytorch_amd/csm/hip/libcsm_hip.so tools/probes/build/abl/libcsm_e1.so; do
  echo "== $lib"
  CSM_HIP_LIB=$PWD/$lib TOUCH_NOCHECK=1 timeout -k 10 120 python tools/probes/touch_ab.py 2>&1 | grep swiglu
done
