# HBM traffic per launch of the kernels of the headline bench step, each counter in its own pass
# (MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots"); summary -> profiles/r04_bench_pmc_traffic.{txt,json}
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r04/pmcb_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04/pmcb_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r04/pmcb_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04/pmcb_write.log 2>&1
python3 tools/pmc_bench_summary.py $(ls gpurun_out/r04/pmcb_fetch/*/*counter_collection.csv | head -1) $(ls gpurun_out/r04/pmcb_write/*/*counter_collection.csv | head -1) gpurun_out/r04/r04_bench_pmc_traffic.json 3 > gpurun_out/r04/r04_bench_pmc_traffic.txt
head -30 gpurun_out/r04/r04_bench_pmc_traffic.txt
rm -rf gpurun_out/r04/pmcb_fetch gpurun_out/r04/pmcb_write
