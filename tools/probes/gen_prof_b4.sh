# per-kernel time of generate_batch() with 4 utterances (rocprofv3 kernel stats over tools/generate_bench.py, GEN_BATCH=4)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04
rm -rf gpurun_out/r04/prof_gen4
GEN_BATCH=4 GEN_FRAMES=60 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_gen4 -- python3 tools/generate_bench.py > gpurun_out/r04/prof_gen4.log 2>&1
tail -2 gpurun_out/r04/prof_gen4.log
python3 tools/prof_summary.py $(ls gpurun_out/r04/prof_gen4/*/*kernel_stats.csv | head -1) 185 24
rm -rf gpurun_out/r04/prof_gen4
