#!/bin/bash
# k_syrk_small against k_syrk's 128 x 128 tiles on one box: (1) the two kernels alone at a range of shapes (tools/syrk_small_bench.hip:
# product + record sum, and the largest difference of the two results), (2) whole evaluations on the developer build with and without
# GPZ_SYRK_SMALL_OFF (ms per evaluation, graph replayed): plain / input noise / missing values, one and two outputs.
O=gpurun_out/r06_syrk_small_ab.txt; : > $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/syrk_small_bench.hip -o build/syrk_small_bench 2> /dev/null
echo "## kernels alone (the tiled kernel with 85 row splits for every tile: not the production split of each shape)" >> $O
for a in "100000 208" "100000 256" "100000 224" "100000 128" "100000 64" "100000 16" "20000 208" "1000000 208" "1000000 128" "3000 208"; do
  build/syrk_small_bench $a | grep -v "^status\|alone" >> $O
done
echo "## whole evaluations" >> $O
export GPZ_HIP_LIB=$PWD/gpz_amd/lib/libgpz_hip_dev.so
T=gpurun_out/r06_syrk_small_ab_raw.txt; : > $T
for spec in "1000 100 VD 1,5" "3000 200 VD 5" "100000 64 VD 2,10" "100000 200 VD,VC 5,10" "100000 255 VD 10" "20000 100 VD,VC 5" "1000000 128 VD 10"; do
  set -- $spec
  for off in 1 0; do
    if [ $off == 1 ]; then export GPZ_SYRK_SMALL_OFF=1; else unset GPZ_SYRK_SMALL_OFF; fi   # (the switch is "present in the environment")
    echo "== n=$1 m=$2 syrk_small_off=$off" >> $T
    python tools/sweep_timing.py $1 $2 $3 $4 2>&1 | grep -v amdgpu.ids >> $T
  done
done
python - >> $O <<'PY'
import re
rows = {}; cur = None
for l in open("gpurun_out/r06_syrk_small_ab_raw.txt"):
    m = re.match(r"== (n=\d+ m=\d+) syrk_small_off=(\d)", l)
    if m: cur = (m.group(1), m.group(2)); continue
    m = re.match(r"(\w\w) d=(\d+)\s+(\w+)\s+([\d.]+) ms", l)
    if m and cur: rows.setdefault((cur[0], m.group(1), m.group(2), m.group(3)), {})[cur[1]] = float(m.group(4))
print("%-18s %-3s %-3s %-6s %12s %14s %7s" % ("shape", "", "d", "case", "k_syrk ms", "k_syrk_small", "ratio"))
for k, v in rows.items():
    if "0" in v and "1" in v:
        print("%-18s %-3s %-3s %-6s %12.3f %14.3f %7.2f" % (k[0], k[1], k[2], k[3], v["1"], v["0"], v["0"] / v["1"]))
PY
cat $O
