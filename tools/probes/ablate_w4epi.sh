# What does the four-wave GEMM's fused SwiGLU-backward epilogue pay for?  Libraries built with -DCSM_ABLATE_W4EPI=<bits> on
# gemm256w4.hip (1: no gate/up loads, 2: no stores, 4: no sigmoid, 8: no epilogue at all), built HERE by
# `bash tools/probes/ablate_w4epi.sh build`, run on the GPU box with no argument: the isolated product, then the headline step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=csm-train-pytorch_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/probes/build/abl
  for b in 1 2 3 4 8; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -mllvm -amdgpu-spill-vgpr-to-agpr=0 -DCSM_ABLATE_W4EPI=$b -c $C/gemm256w4.hip -o tools/probes/build/abl/w4e$b.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/build/abl/libcsm_w4e$b.so $(ls $C/build/*.o | grep -v gemm256w4) tools/probes/build/abl/w4e$b.o
  done
  exit
fi
for lib in csm-train-pytorch_amd/csm/hip/libcsm_hip.so tools/probes/build/abl/libcsm_w4e1.so tools/probes/build/abl/libcsm_w4e2.so tools/probes/build/abl/libcsm_w4e3.so tools/probes/build/abl/libcsm_w4e4.so tools/probes/build/abl/libcsm_w4e8.so; do
  echo "== $lib"
  CSM_HIP_LIB=$PWD/$lib TOUCH_NOCHECK=1 timeout -k 10 120 python tools/probes/touch_ab.py 2>&1 | grep "swiglu.*touch=0"
  CSM_HIP_LIB=$PWD/$lib python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --gemm-shapes 2>&1 >/dev/null | grep -E "503.3 MB|604.0 MB  nn_dgrad" | cut -c1-100
done
