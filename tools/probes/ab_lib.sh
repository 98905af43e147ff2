# Same-box A/B of two builds of libcsm_hip.so on the headline step: tools/probes/build/abl/libcsm_prev.so (e.g. the
# previous commit, built out of tree) against the in-tree library, interleaved twice.  Usage: bash tools/probes/ab_lib.sh [bench args]
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in tools/probes/build/abl/libcsm_prev.so csm-train-pytorch_amd/csm/hip/libcsm_hip.so; do
    CSM_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.split('/')[-1], d['ms_per_step'], 'ms/step', d['value'], d['unit'])"
  done
done
