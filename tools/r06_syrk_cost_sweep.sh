#!/bin/bash
# k_syrk: cost per row of the edge-tile classes (developer build: GPZ_SYRK_C2 / _C3, x 1000) against the kernel's time at c4 / c2
export GPZ_HIP_LIB=$PWD/gpz_amd/lib/libgpz_hip_dev.so
for cfg in c4 c2; do
for c2 in 0 850 905 950 1000 1050; do for c3 in 0 470 600; do
  GPZ_SYRK_C2=$c2 GPZ_SYRK_C3=$c3 python bench.py --config $cfg --no-cpu-baseline --steps 8 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readlines()[-1]); print('$cfg c2=$c2 c3=$c3 syrk %.4f ms  step %.4f' % (o['kernels']['syrk_avg_ms'], o['ms_per_step']))"
done; done; done
