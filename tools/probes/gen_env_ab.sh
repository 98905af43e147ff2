# generate() at B = 1 under environment settings, interleaved on one box: bash tools/probes/gen_env_ab.sh REPS "ENV_A" "ENV_B" ... ("-" = none)
cd $GRAFT_REPO_ROOT
reps=$1; shift
for rep in $(seq $reps); do for e in "$@"; do
  [ "$e" = "-" ] && ev="" || ev="$e"
  echo -n "$e: "; env $ev GEN_BATCH=1 python tools/generate_bench.py 2>/dev/null | tail -1
done; done
