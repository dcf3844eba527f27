#!/bin/bash
# Missing values on the k_small_tail route (diagonal kinds, 3 d <= 32 features): the developer build with and without GPZ_SMALL_TAIL_OFF
# on the same box; the "nan" rows are the ones that change, "plain" is the control, "psi" never takes the route.
O=gpurun_out/r06_small_nan_ab.txt; : > $O
export GPZ_HIP_LIB=$PWD/gpz_amd/lib/libgpz_hip_dev.so
for spec in "100000 64 VD 1,2,5,10" "100000 200 VD 5,10" "100000 255 VD 10" "1000000 128 VD 10"; do
  set -- $spec
  for off in 1 0; do
    echo "== n=$1 m=$2 small_tail_off=$off" >> $O
    if [ $off == 1 ]; then export GPZ_SMALL_TAIL_OFF=1; else unset GPZ_SMALL_TAIL_OFF; fi   # (the switch is "present in the environment")
    python tools/sweep_timing.py $1 $2 $3 $4 2>&1 | grep -v amdgpu.ids >> $O
  done
done
python - > gpurun_out/r06_small_nan_ab_summary.txt <<'PY'
import re
rows = {}; cur = None
for l in open("gpurun_out/r06_small_nan_ab.txt"):
    m = re.match(r"== (n=\d+ m=\d+) small_tail_off=(\d)", l)
    if m: cur = (m.group(1), m.group(2)); continue
    m = re.match(r"(\w\w) d=(\d+)\s+(\w+)\s+([\d.]+) ms", l)
    if m and cur: rows.setdefault((cur[0], m.group(1), m.group(2), m.group(3)), {})[cur[1]] = float(m.group(4))
print("%-18s %-3s %-3s %-6s %12s %12s %7s" % ("shape", "", "d", "case", "separate ms", "one kernel", "ratio"))
for k, v in rows.items():
    if "0" in v and "1" in v and k[3] != "psi":
        print("%-18s %-3s %-3s %-6s %12.3f %12.3f %7.2f" % (k[0], k[1], k[2], k[3], v["1"], v["0"], v["0"] / v["1"]))
PY
cat gpurun_out/r06_small_nan_ab_summary.txt
