#!/bin/bash
# k_small_tail: start delay of the second workgroup of a compute unit (developer build, GPZ_SMALL_STAGGER x 8128 cycles) against the stage time
for rep in 1 2; do for s in 1 2 3 4 5 6 8; do
  GPZ_HIP_LIB=$PWD/gpz_amd/lib/libgpz_hip_dev.so GPZ_SMALL_STAGGER=$s python bench.py --config ${1:-c2} --no-cpu-baseline --steps 20 --timed-events none 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readlines()[-1]); print('stagger $s: tail_small %.4f ms  step %.4f ms' % (o['kernels']['stage_ms_per_eval']['tail_small'], o['ms_per_step']))"
done; done
