#!/bin/bash
# Collect rocprofv3 PMC counters for the c4 bench (run on the GPU box via gpurun).  One counter group per pass.
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="${@:---steps 2 --warmup 2 --no-cpu-baseline}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o p -- python $R/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python $R/bench.py $ARGS > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python $R/bench.py $ARGS > $OUT/write.log 2>&1
ls -R $OUT | head -40
