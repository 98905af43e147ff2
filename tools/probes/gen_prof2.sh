# per-kernel time of one generated frame (rocprofv3 kernel stats over tools/generate_bench.py, batch 1), for CSM_DECODE_FUSE_ATTN=0 / 1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03
for f in 0 1; do
  rm -rf gpurun_out/r03/prof_gen$f
  CSM_DECODE_FUSE_ATTN=$f GEN_BATCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_gen$f -- python3 tools/generate_bench.py > gpurun_out/r03/prof_gen$f.log 2>&1
  echo "== CSM_DECODE_FUSE_ATTN=$f"; tail -1 gpurun_out/r03/prof_gen$f.log
  python3 tools/prof_summary.py $(ls gpurun_out/r03/prof_gen$f/*/*kernel_stats.csv | head -1) 130 12
  rm -rf gpurun_out/r03/prof_gen$f
done
