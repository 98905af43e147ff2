#!/bin/bash
# Build ablated copies of the library (CSM_ATT_ABLATE in attention.hip) and time attention backward with each.
set -e
make -C csm-train-pytorch_amd/csrc -j8 >/dev/null 2>&1
cd csm-train-pytorch_amd/csrc
for a in ${1:-0 1 2 3}; do
  mkdir -p /tmp/aabl$a
  for f in gemm.hip gemm256.hip attention.hip ops.hip generate.hip codec.hip csm_api.cpp; do
    if [ $f = attention.hip ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DCSM_ATT_ABLATE=$a -c $f -o /tmp/aabl$a/$f.o
    else
      cp build/$f.o /tmp/aabl$a/$f.o
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/aabl$a/libcsm_hip.so /tmp/aabl$a/*.o
done
cd ../..
for a in ${1:-0 1 2 3}; do
  echo "== CSM_ATT_ABLATE=$a"
  CSM_HIP_LIB=/tmp/aabl$a/libcsm_hip.so python tools/attn_bench.py 2>&1 | tail -2 | head -1
done
