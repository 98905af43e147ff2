# 256x256 GEMM: persistent workgroups (default) vs one tile per workgroup, same box: per-shape table and the train step.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm or linear or k_ext or swiglu or paired or grouped" 2>&1 | tail -3
for v in 0 1; do
  echo "== CSM_GEMM256_PERSISTENT=$v"
  CSM_GEMM256_PERSISTENT=$v python tools/gemm_shapes.py 2>/dev/null | head -22 | tee gpurun_out/r02/gemm_shapes_persist$v.txt | head -14
done
for rep in 1 2; do for v in 0 1; do
  CSM_GEMM256_PERSISTENT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persistent=$v', d['ms_per_step'], 'ms/step', d['value'], d['unit'], 'loss', d['loss'])"
done; done
