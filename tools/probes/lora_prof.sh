# kernel-trace summary of the LoRA config-3 step (one step cut out of the trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02 && rm -rf gpurun_out/r02/prof_lora2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_lora2 -- python3 bench.py --lora --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02/prof_lora2.log 2>&1
python3 tools/trace_gaps.py $(ls gpurun_out/r02/prof_lora2/*/*kernel_trace.csv | head -1) 1 > gpurun_out/r02/lora_step_cut.txt 2>&1
head -45 gpurun_out/r02/lora_step_cut.txt
rm -rf gpurun_out/r02/prof_lora2
