#!/bin/bash
# Round-3 side measurements (via gpurun): cost of the runtime-d route, prediction branches, k > 1, fp64 input-noise path (the VC psi row at d = 10 of the sweep).
# usage: tools/measure_extras.sh <tag>   -> gpurun_out/<tag>/extras.txt
set -u
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
{
echo "# sweep_timing.py 100000 256 VD,VC 10,20,24,32   (d > 20: runtime-d kernels of k_wide.hip; VC psi at d > 10 takes dtype f32 up to d = 20)"
python tools/sweep_timing.py 100000 256 VD,VC 10,20,24,32 2>&1 | grep -v amdgpu.ids
echo "# GPZ_SWEEP_F64=1 sweep_timing.py 100000 256 VC 12,16,20,24,28,32,36,40,48,64   (VC psi in fp64: k_cpsi4.hip up to d = 32, k_cpsi4w.hip up to 48, k_cpsi.hip up to 64)"
GPZ_SWEEP_F64=1 python tools/sweep_timing.py 100000 256 VC 12,16,20,24,28,32,36,40,48,64 2>&1 | grep " psi "
echo "# the same at d = 16, 20, 24 on the 16 x 16 tile kernels (GPZ_CPSI4_OFF=1) and on the general kernels of k_gen.hip (GPZ_CPSI_OFF=1)"
GPZ_CPSI4_OFF=1 GPZ_SWEEP_F64=1 python tools/sweep_timing.py 100000 256 VC 16,20,24 2>&1 | grep " psi "
GPZ_CPSI_OFF=1 GPZ_SWEEP_F64=1 python tools/sweep_timing.py 100000 256 VC 16,20,24 2>&1 | grep " psi "
echo "# predict_noisy_timing.py 100000 500 10 {GC,VC,VD}"
for mth in GC VC VD; do python tools/predict_noisy_timing.py 100000 500 10 $mth 2>&1 | grep -v amdgpu.ids; done
echo "# predict_missing_profile.py {500 200 10 GC | 500 200 10 VC | 5000 500 10 VD}"
python tools/predict_missing_profile.py 500 200 10 GC 2>&1 | grep -v amdgpu.ids
python tools/predict_missing_profile.py 500 200 10 VC 2>&1 | grep -v amdgpu.ids
python tools/predict_missing_profile.py 5000 500 10 VD 2>&1 | grep -v amdgpu.ids
echo "# k outputs, n=100000 d=10 m=200 VD / VC (ms per evaluation)"
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, gpz_amd
from helpers import make_problem
for method in ("VD", "VC"):
    for k in (1, 2, 4):
        model, theta, X, Y, _, rng = make_problem(100000, 10, 200, k, method, True, seed=3)
        ctx = gpz_amd.GPzContext(model, X, Y)
        ctx.eval(theta)
        t0 = time.perf_counter()
        for _ in range(5): ctx.eval(theta)
        print(f"{method} k={k}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/eval", flush=True)
        ctx.close()
PY
} > $O/extras.txt 2>&1
cat $O/extras.txt
