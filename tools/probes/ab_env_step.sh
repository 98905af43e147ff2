# Same-box A/B of environment settings on the headline step, interleaved: bash tools/probes/ab_env_step.sh REPS "ENV_A" "ENV_B" ...
# (each argument a space-separated list of VAR=value, "-" for none); prints ms/step per run and the median per setting.
cd $GRAFT_REPO_ROOT
reps=$1; shift
for rep in $(seq $reps); do
  for e in "$@"; do
    [ "$e" = "-" ] && ev="" || ev="$e"
    ms=$(env $ev python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$e $ms"
  done
done | tee /tmp/ab_env.txt
python - <<'PY'
import collections
d = collections.defaultdict(list)
for line in open("/tmp/ab_env.txt"):
    *k, v = line.split()
    d[" ".join(k)].append(float(v))
for k, v in d.items():
    v.sort(); print(f"median {v[len(v) // 2]:.3f}  min {v[0]:.3f}  max {v[-1]:.3f}   {k}")
PY
