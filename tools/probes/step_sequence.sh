# kernel sequence of one headline train step (small kernels and the gaps in front of them): rocprofv3 kernel trace of 3 steps
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04
rm -rf gpurun_out/r04/prof_seq
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/prof_seq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04/prof_seq.log 2>&1
python3 tools/probes/step_sequence.py $(ls gpurun_out/r04/prof_seq/*/*kernel_trace.csv | head -1) 4 40
rm -rf gpurun_out/r04/prof_seq
