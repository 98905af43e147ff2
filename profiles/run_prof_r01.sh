set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_e2e_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k train_step_full 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
ls -R gpurun_out/prof_r01 | head -30
