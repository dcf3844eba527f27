#!/bin/bash
# Shape sweep, the library at the end of round 5 (gpz_amd/lib/libgpz_hip_r05.so, built from commit 2d226ce) against this round's, same box:
# few basis functions (where k_small_tail applies: m + 1 <= 256 columns, d <= 15 / 6) and the shapes around its limits
O=gpurun_out/r06_sweep.txt; : > $O
for spec in "100000 64 GL,VL,GD,VD 1,2,5,10" "100000 200 VD,GD 5,10,12,16" "100000 255 VD 10" "100000 256 VD 10" "100000 100 GC,VC 2,5,6,8" "20000 50 VL,VD,VC 1,3,5" "1000000 128 VD 10"; do
  set -- $spec
  for lib in r05 now; do
    if [ $lib == r05 ]; then export GPZ_HIP_LIB=$PWD/gpz_amd/lib/libgpz_hip_r05.so; else unset GPZ_HIP_LIB; fi
    echo "== n=$1 m=$2 methods=$3 widths=$4 library=$lib" >> $O
    python tools/sweep_timing.py $1 $2 $3 $4 2>&1 | grep -v amdgpu.ids >> $O
  done
done
python - <<'PY'
import re
rows = {}; cur = None
for l in open("gpurun_out/r06_sweep.txt"):
    m = re.match(r"== (n=\d+ m=\d+) .* library=(\w+)", l)
    if m: cur = (m.group(1), m.group(2)); continue
    m = re.match(r"(\w\w) d=(\d+)\s+(\w+)\s+([\d.]+) ms", l)
    if m and cur: rows.setdefault((cur[0], m.group(1), m.group(2), m.group(3)), {})[cur[1]] = float(m.group(4))
print("%-18s %-3s %-3s %-6s %10s %10s %7s" % ("shape", "", "d", "case", "r05 ms", "now ms", "ratio"))
worst = 0
for k, v in rows.items():
    if "r05" in v and "now" in v:
        print("%-18s %-3s %-3s %-6s %10.3f %10.3f %7.2f" % (k[0], k[1], k[2], k[3], v["r05"], v["now"], v["now"] / v["r05"]))
        worst = max(worst, v["now"] / v["r05"])
print("worst now / r05 ratio: %.2f" % worst)
PY
