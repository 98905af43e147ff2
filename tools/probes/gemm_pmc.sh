# SQ counters of the 256x256 GEMM main loop (one shape, GP_MODE/GP_M/GP_N/GP_K), one rocprofv3 --pmc pass per counter group
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03/pmc
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/r03/pmc/g$i -- python3 tools/probes/gemm_pmc.py > gpurun_out/r03/pmc/g$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/r03/pmc/g$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r03/pmc/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm256" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        v = v[1:] if len(v) > 1 else v            # drop the first (cold) launch
        print(f"  {c:32s} {sum(v)/len(v):16.0f}")
PY
rm -rf gpurun_out/r03/pmc/g1 gpurun_out/r03/pmc/g2 gpurun_out/r03/pmc/g3
