# A/B of the paired dgrad+wgrad launches inside the real train step (bits: 1 w2, 2 w13, 4 output_proj, 8 qkv)
cd $GRAFT_REPO_ROOT
for v in 0 15 1 2 4 8 3; do
  echo -n "CSM_PAIR_DX_DW=$v  "
  CSM_PAIR_DX_DW=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['value'], d['mfma_utilisation_step'], d['loss'])"
done
