# raster group size of the 256x256 GEMM (m-tiles per group) inside the train step, same box: libraries prebuilt with -DCSM_GROUP_M=2 / 8
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in csm-train-pytorch_amd/csm/hip/libcsm_hip.so tools/probes/build/abl/libcsm_gm2.so tools/probes/build/abl/libcsm_gm8.so; do
  CSM_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.split('/')[-1], d['ms_per_step'], 'ms/step')"
done; done
