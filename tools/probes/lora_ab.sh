# BASELINE config 3 (LoRA r=8 q_proj/v_proj, B=8, S=2048): adapters as K-extension operands of the frozen projections'
# GEMMs (default) vs the per-adapter products (CSM_LORA_FUSE=0), same box, interleaved.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for f in 0 1; do
    CSM_LORA_FUSE=$f python bench.py --lora --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CSM_LORA_FUSE=$f', d['ms_per_step'], 'ms/step', d['value'], d['unit'], 'loss', d['loss'])"
  done
done
