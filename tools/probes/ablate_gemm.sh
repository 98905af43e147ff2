#!/bin/bash
# Build ablated copies of the library (see CSM_ABLATE in gemm256.hip) and time the long-K GEMM shapes with each.
# Run from the repo root on a GPU box:  bash tools/probes/ablate_gemm.sh "0 1 2 3 4 8"
set -e
make -C csm-train-pytorch_amd/csrc -j8 >/dev/null 2>&1
cd csm-train-pytorch_amd/csrc
for a in ${1:-0 1 2 4 8}; do
  mkdir -p /tmp/abl$a
  for f in gemm.hip gemm256.hip attention.hip ops.hip generate.hip codec.hip csm_api.cpp; do
    if [ $f = gemm256.hip ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DCSM_ABLATE=$a -c $f -o /tmp/abl$a/$f.o
    else
      cp build/$f.o /tmp/abl$a/$f.o
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abl$a/libcsm_hip.so /tmp/abl$a/*.o
done
cd ../..
for a in ${1:-0 1 2 4 8}; do
  echo "== CSM_ABLATE=$a"
  CSM_HIP_LIB=/tmp/abl$a/libcsm_hip.so GB_VARIANTS=2 GB_NOCHECK=1 python tools/gemm_bench.py 2>&1 | grep -E "^v2"
done
if [ -n "$2" ]; then
  echo "== parity of CSM_ABLATE=$2"
  CSM_HIP_LIB=/tmp/abl$2/libcsm_hip.so python -m pytest tests/test_ops_gpu.py -q -m gpu -k gemm -p no:cacheprovider 2>&1 | tail -3
fi
