set -e
make -C csm-train-pytorch_amd/csrc -j8 >/dev/null 2>&1
cd csm-train-pytorch_amd/csrc
mkdir -p /tmp/gm4
for f in gemm.hip gemm256.hip attention.hip ops.hip generate.hip codec.hip csm_api.cpp; do
  if [ $f = gemm256.hip ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DCSM_GROUP_M=4 -c $f -o /tmp/gm4/$f.o
  else cp build/$f.o /tmp/gm4/$f.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/gm4/libcsm_hip.so /tmp/gm4/*.o
cd ../..
for i in 1 2 3; do
  for lib in /tmp/gm4/libcsm_hip.so ""; do
    CSM_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib', d['ms_per_step'], d['roofline']['all_gemm_variants']['nt_fwd_bf16']['ms_per_step'])"
  done
done
