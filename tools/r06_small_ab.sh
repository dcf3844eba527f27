#!/bin/bash
# A/B of the one-kernel tail for few basis functions (k_small_tail) against k_tgemm + k_row_scalars + k_moments_fused (developer build,
# GPZ_SMALL_TAIL_OFF=1), same box, alternating.  Usage: tools/r06_small_ab.sh [config ...]
run() { label=$1; shift
  env GPZ_HIP_LIB=$PWD/gpz_amd/lib/libgpz_hip_dev.so "$@" python bench.py --config $cfg $extra --no-cpu-baseline --steps 20 --timed-events none 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readlines()[-1]); st=o['kernels']['stage_ms_per_eval']
print('$cfg $extra %-10s step %.4f ms | %s' % ('$label', o['ms_per_step'], ' '.join('%s %.4f' % (k, v) for k, v in st.items() if v > 0.012)))"
}
for cfg in ${@:-c2}; do for rep in 1 2; do
  extra=""; run small; run separate GPZ_SMALL_TAIL_OFF=1
done; done
