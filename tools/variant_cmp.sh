#!/bin/bash
# Developer tool: A/B per-stage times of library variants built by tools/build_variant.sh.
#   tools/variant_cmp.sh <config> <tag> [<tag> ...]     ("base" = the default library)
cfg=$1; shift
mkdir -p gpurun_out; : > gpurun_out/variant_cmp.log
for rep in 1 2; do
for v in "$@"; do
  if [ $v == base ]; then lib=""; else lib=$PWD/gpz_amd/lib/libgpz_hip_$v.so; fi
  echo "== $v" >> gpurun_out/variant_cmp.log
  GPZ_HIP_LIB=$lib python tools/stage_times.py $cfg 2>&1 | grep ms/eval | sed 's/.* ms\/eval/ms\/eval/' >> gpurun_out/variant_cmp.log
done
done
cat gpurun_out/variant_cmp.log
