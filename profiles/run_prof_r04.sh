# Round-4 profiles (all through gpurun on one MI355X): kernel-trace stats of the headline bench, of the LoRA config and of generate().
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -rf gpurun_out/r04/prof_bench gpurun_out/r04/prof_lora gpurun_out/r04/prof_gen
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04/prof_bench.log 2>&1
python3 tools/prof_summary.py $(ls gpurun_out/r04/prof_bench/*/*kernel_stats.csv | head -1) 4 40 > gpurun_out/r04/r04_bench_b4_s2048_step_summary.txt
cp $(ls gpurun_out/r04/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/r04/r04_bench_b4_s2048_kernel_stats.csv
python3 tools/trace_gaps.py $(ls gpurun_out/r04/prof_bench/*/*kernel_trace.csv | head -1) > gpurun_out/r04/r04_bench_trace_gaps.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_lora -- python3 bench.py --lora --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04/prof_lora.log 2>&1
python3 tools/prof_summary.py $(ls gpurun_out/r04/prof_lora/*/*kernel_stats.csv | head -1) 4 30 > gpurun_out/r04/r04_lora_b8_s2048_step_summary.txt
GEN_BATCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_gen -- python3 tools/generate_bench.py > gpurun_out/r04/prof_gen.log 2>&1
python3 tools/prof_summary.py $(ls gpurun_out/r04/prof_gen/*/*kernel_stats.csv | head -1) 130 30 > gpurun_out/r04/r04_generate_10s_frame_summary.txt
tail -3 gpurun_out/r04/prof_gen.log
head -32 gpurun_out/r04/r04_bench_b4_s2048_step_summary.txt
head -24 gpurun_out/r04/r04_generate_10s_frame_summary.txt
