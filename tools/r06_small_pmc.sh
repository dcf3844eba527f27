#!/bin/bash
# SQ counters of k_small_tail at a configuration (two rocprofv3 --pmc passes; run through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/small_pmc; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
ARGS="--config ${1:-c2} --steps 5 --warmup 3 --no-cpu-baseline --timed-events none"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq -o p -- python $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- python $R/bench.py $ARGS > /dev/null 2>&1
cd $R
python - <<PY
import csv, collections
d = {}
for f in ("sq", "sq2"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/small_pmc/%s/p_counter_collection.csv" % f)):
        if "k_small_tail" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): d[k] = sum(v) / len(v)
for k in sorted(d): print("%-32s %.4g" % (k, d[k]))
cyc = d["GRBM_GUI_ACTIVE"] / 8
print("kernel cycles %.4g  (%.1f us at 2.4 GHz)" % (cyc, cyc / 2400))
print("MFMA busy %.1f %%" % (100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)))
print("MFMA instructions per SIMD %.0f -> %.0f cycles of 64" % (d["SQ_INSTS_MFMA"] / 1024, 64 * d["SQ_INSTS_MFMA"] / 1024))
print("non-MFMA VALU per MFMA %.2f" % ((d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / d["SQ_INSTS_MFMA"]))
PY
rm -rf $O/sq $O/sq2
