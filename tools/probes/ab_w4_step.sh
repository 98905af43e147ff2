# Same-box A/B on the headline step: eight-wave 256x256 kernel only (CSM_GEMM_W4=0) vs the four-wave kernel where it applies
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for w4 in 0 1; do
    CSM_GEMM_W4=$w4 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_gemm_kernels']
print('w4=$w4', d['ms_per_step'], 'ms/step', d['mfma_utilisation_step'], ' | '.join(f\"{n[:28]} {v['ms_per_step']} ({v['tflops']})\" for n,v in list(k.items())[:6]))"
  done
done
