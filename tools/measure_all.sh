#!/bin/bash
# Round-end measurement pass on the GPU box (via gpurun): bench lines of every configuration, the rocprofv3 kernel
# trace and the PMC passes of the default workload.  usage: tools/measure_all.sh <tag>      -> gpurun_out/<tag>/
set -u
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python bench.py > $O/c4.json 2> $O/c4.err
python bench.py --config c2 > $O/c2.json 2> $O/c2.err
python bench.py --config c3 > $O/c3.json 2> $O/c3.err
python bench.py --rows 125000 --no-cpu-baseline > $O/c4_shard125k.json 2> $O/c4_shard125k.err
python bench.py --config c5 --rows 250000 --steps 3 --warmup 1 > $O/c5s.json 2> $O/c5s.err
python bench.py --config c2 --validation 0.15 > $O/c2_validation15.json 2> $O/c2_validation15.err
tools/pmc_run.sh $tag/pmc > $O/pmc_run.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/$tag/pmc gpurun_out/$tag/pmc_summary.txt --constants gpurun_out/$tag/pmc_constants.json --config c4 > /dev/null 2>&1
find gpurun_out/$tag/pmc/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
tools/pmc_run.sh $tag/pmc_c5 --config c5 --rows 250000 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_c5_run.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/$tag/pmc_c5 gpurun_out/$tag/pmc_c5_summary.txt "--config c5 --rows 250000 --steps 2 --warmup 1 --no-cpu-baseline
# (c5 shard: n=250000 of 2e6, d=20 m=2000 VC hetero + diagonal Psi cubes, dtype f32, 1 x MI355X" --constants gpurun_out/$tag/pmc_constants.json --config c5 > /dev/null 2>&1
find gpurun_out/$tag/pmc_c5/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c5.csv \;
# keep the merged-back payload small: the raw counter CSVs are large
find gpurun_out/$tag/pmc -name "*.csv" -size +4M -delete
# round 4: config 5 with the moment sums on 4 x 4 MFMA tiles (opt-in route), config 5 in fp64, the loopback-8 run of c4 with per-rank
# stage times (the line a first real 8-GPU run is read against), and the issue-rate microbenchmarks behind DESIGN.md section 8
GPZ_PSI32_MFMA=1 python bench.py --config c5 --rows 250000 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5s_mfma_route.json 2> $O/c5s_mfma_route.err
python bench.py --config c5_f64 --rows 250000 --steps 2 --warmup 1 --no-cpu-baseline > $O/c5s_f64.json 2> $O/c5s_f64.err
python bench.py --native-mgpu 8 --no-cpu-baseline --steps 5 > $O/c4_native_mgpu8_loopback.json 2> $O/c4_native_mgpu8_loopback.err
for t in mfma_f32_4x4_rate mfma_valu_overlap mfma_f32_16x16_overlap mfma_f64_valu_overlap pk_fma_rate; do
  hipcc --offload-arch=gfx950 -O3 tools/$t.hip -o build/$t 2> /dev/null && build/$t > $O/ubench_$t.txt 2>&1
done
# row-tile streaming: c4's model on 2e7 rows (PHI + T beyond the HBM: the library picks the tiles) and c4 itself forced into two tiles
python bench.py --config c4 --rows 20000000 --no-cpu-baseline --steps 3 --warmup 1 > $O/c4_n2e7_streamed.json 2> $O/c4_n2e7_streamed.err
GPZ_ROW_TILE=524288 python bench.py --config c4 --no-cpu-baseline > $O/c4_streamed_tile512k.json 2> $O/c4_streamed_tile512k.err
python tools/pm_wide_timing.py 2> /dev/null | grep " d=" > $O/extras_predict_missing_wide.txt
tail -c 600 $O/c4.json; echo; tail -c 300 $O/c2.json; echo; tail -c 300 $O/c3.json; echo; tail -c 300 $O/c4_shard125k.json
