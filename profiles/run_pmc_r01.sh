# HBM traffic of the dominant GEMM kernels via PMC counters, each counter in its own pass (MI355X_MICROARCH.md,
# "HBM" + "rocprofv3 PMC slots"): FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE counts 128-B requests at
# 64 B on gfx950 for wide coalesced streams -> doubled in tools/pmc_summary.py.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GB_VARIANTS=2 GB_ROUNDS=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/gemm_bench.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/gemm_bench.py > gpurun_out/pmc_write.log 2>&1
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/
head -3 gpurun_out/pmc_fetch/*/*counter_collection.csv | cut -c1-400
