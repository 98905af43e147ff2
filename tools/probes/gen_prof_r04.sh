# per-kernel time of one generated frame (rocprofv3 kernel stats over tools/generate_bench.py, batch 1: 5 + 125 frames)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04
rm -rf gpurun_out/r04/prof_gen
GEN_BATCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_gen -- python3 tools/generate_bench.py > gpurun_out/r04/prof_gen.log 2>&1
tail -1 gpurun_out/r04/prof_gen.log
python3 tools/prof_summary.py $(ls gpurun_out/r04/prof_gen/*/*kernel_stats.csv | head -1) 130 14
rm -rf gpurun_out/r04/prof_gen
