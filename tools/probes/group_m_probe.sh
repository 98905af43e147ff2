set -e
make -C csm-train-pytorch_amd/csrc -j8 >/dev/null 2>&1
cd csm-train-pytorch_amd/csrc
for a in 4 2 8 16; do
  mkdir -p /tmp/gm$a
  for f in gemm.hip gemm256.hip attention.hip ops.hip generate.hip codec.hip csm_api.cpp; do
    if [ $f = gemm256.hip ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DCSM_GROUP_M=$a -c $f -o /tmp/gm$a/$f.o
    else cp build/$f.o /tmp/gm$a/$f.o; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/gm$a/libcsm_hip.so /tmp/gm$a/*.o
done
cd ../..
for a in 4 2 8 16 4; do echo "== GROUP_M=$a"; CSM_HIP_LIB=/tmp/gm$a/libcsm_hip.so GB_VARIANTS=2 GB_ROUNDS=5 GB_NOCHECK=1 python tools/gemm_bench.py 2>&1 | grep -E "^v2" | cut -c1-62; done
