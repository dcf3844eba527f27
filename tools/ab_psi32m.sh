#!/bin/bash
# Developer tool (GPU box): A/B variants of k_psi32m.hip by -D flags; each variant rebuilds that one object, relinks and runs the
# config-5 shard once.  usage: tools/ab_psi32m.sh <outdir> "<flags of variant 1>" "<flags of variant 2>" ...
set -u
O=gpurun_out/$1; shift
mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Igpz_amd/csrc"
objs=$(ls build/*.o | grep -v k_psi32m.o | tr '\n' ' ')
i=0
for v in "$@"; do
  i=$((i+1))
  hipcc $FLAGS $v -c gpz_amd/csrc/k_psi32m.hip -o build/k_psi32m.o 2> $O/v$i.build.err || { echo "variant $i failed to build"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC $objs build/k_psi32m.o -ldl -lpthread -o gpz_amd/lib/libgpz_hip.so
  python bench.py --config c5 --rows 250000 --steps 3 --warmup 1 --no-cpu-baseline > $O/v$i.json 2> $O/v$i.err
  echo "variant $i [$v]: $(python - <<PY
import json
d=json.load(open("$O/v$i.json"))
print("moments %.1f ms  step %.1f ms"%(d["kernels"]["stage_ms_per_eval"]["moments"], d["ms_per_step"]))
PY
)"
done
