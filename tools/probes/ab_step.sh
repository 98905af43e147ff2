# Same-box A/B on the headline step: old main loop (gemm256.hip of commit 5881a4a) / round-3 loop without / with the epilogue-read
# prefetch, interleaved twice.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "tools/probes/build/abl/libcsm_oldloop.so 1" "csm-train-pytorch_amd/csm/hip/libcsm_hip.so 0" "csm-train-pytorch_amd/csm/hip/libcsm_hip.so 1"; do
    set -- $cfg
    CSM_HIP_LIB=$PWD/$1 CSM_GEMM_TOUCH=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_gemm_kernels']
print('$1'.split('/')[-1], 'touch=$2', d['ms_per_step'], 'ms/step', ' '.join(f\"{n.split('<')[0][-8:]}<{n.split('<')[1][:4] if '<' in n else ''} {v['ms_per_step']}\" for n,v in list(k.items())[:5]))"
  done
done
