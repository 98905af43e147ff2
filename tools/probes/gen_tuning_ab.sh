# generate() at B = 1: matrix-vector kernels with x in LDS (0,0), in registers (1,0), + non-temporal backbone weights (1,1); same box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for t in 0,0 1,0 1,1; do
  echo -n "CSM_DECODE_TUNING=$t: "; CSM_DECODE_TUNING=$t GEN_BATCH=1 python tools/generate_bench.py 2>/dev/null | tail -1
done; done
