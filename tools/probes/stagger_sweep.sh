# In-step sweep of the four-wave GEMM's start offsets (CSM_GEMM_STAGGER = groups,fwd,bwd,other in 10 ns ticks):
# ms/step plus the two fused-SwiGLU products of a backbone layer (w13 forward 549.8 GF / 503 MB, w2 dgrad 274.9 GF / 604 MB)
cd $GRAFT_REPO_ROOT
for st in "$@"; do
  CSM_GEMM_STAGGER=$st python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --gemm-shapes 2> /tmp/shapes.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger=$st', d['ms_per_step'], 'ms/step', d['mfma_utilisation_step'])"
  grep -E "503.3 MB|604.0 MB  nn_dgrad|872.4 MB|1124.1 MB" /tmp/shapes.txt | cut -c1-120
done
