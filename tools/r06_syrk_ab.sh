#!/bin/bash
# A/B of k_syrk on one box: the library of the previous commit (gpz_amd/lib/libgpz_hip_old.so) against the working tree's, alternating
run() { # label lib env...
  label=$1; lib=$2; shift 2
  env GPZ_HIP_LIB=$lib "$@" python bench.py --config $cfg --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readlines()[-1]); print('$cfg %-22s syrk %.4f ms  tgemm %.4f  step %.4f' % ('$label', o['kernels']['syrk_avg_ms'], o['roofline']['avg_ms'], o['ms_per_step']))"
}
for cfg in c4 c3 c2; do for rep in 1 2 3; do
  run old $PWD/gpz_amd/lib/libgpz_hip_old.so
  run new $PWD/gpz_amd/lib/libgpz_hip.so
  run new_c950_600 $PWD/gpz_amd/lib/libgpz_hip_dev.so GPZ_SYRK_C2=950 GPZ_SYRK_C3=600
  run new_c1000_650 $PWD/gpz_amd/lib/libgpz_hip_dev.so GPZ_SYRK_C2=1000 GPZ_SYRK_C3=650
done; done
