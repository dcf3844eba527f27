#!/bin/bash
# Round-end measurement pass on the GPU box (via gpurun): bench lines of every configuration, the rocprofv3 kernel
# trace and the PMC passes of the default workload, the microbenchmarks behind DESIGN.md.  usage: tools/measure_all.sh <tag>  -> gpurun_out/<tag>/
set -u
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python bench.py > $O/c4.json 2> $O/c4.err
# latency-bound configurations: the whole evaluation as ONE graph in the timed region (--timed-events none); stage times from the stage pass
python bench.py --config c2 --timed-events none > $O/c2.json 2> $O/c2.err
python bench.py --config c3 --timed-events none > $O/c3.json 2> $O/c3.err
python bench.py --config c2 --validation 0.15 --timed-events none > $O/c2_validation15.json 2> $O/c2_validation15.err
# the rows one of 2 / 4 / 8 ranks holds of c4 (tools/scaling_prediction.py)
python bench.py --rows 500000 --no-cpu-baseline > $O/c4_shard500k.json 2> $O/c4_shard500k.err
python bench.py --rows 250000 --no-cpu-baseline > $O/c4_shard250k.json 2> $O/c4_shard250k.err
python bench.py --rows 125000 --no-cpu-baseline > $O/c4_shard125k.json 2> $O/c4_shard125k.err
python bench.py --config c5 --rows 250000 --steps 3 --warmup 2 > $O/c5s.json 2> $O/c5s.err
python bench.py --config c5_f64 --rows 250000 --steps 2 --warmup 2 --no-cpu-baseline > $O/c5s_f64.json 2> $O/c5s_f64.err
# config 5 at its full n = 2e6 on ONE GPU (the only such line was round 1's)
python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 2 > $O/c5_full_1gpu.json 2> $O/c5_full_1gpu.err
# the threaded multi-device driver with 8 shards on this box's one GPU (loopback reducer): graph segments with the all-reduce between them
python bench.py --native-mgpu 8 --no-cpu-baseline --steps 5 > $O/c4_native_mgpu8_loopback.json 2> $O/c4_native_mgpu8_loopback.err
# row-tile streaming
python bench.py --config c4 --rows 20000000 --no-cpu-baseline --steps 3 --warmup 2 > $O/c4_n2e7_streamed.json 2> $O/c4_n2e7_streamed.err
python tools/scaling_prediction.py gpurun_out/$tag $O/scaling_prediction.json > $O/scaling_prediction.log 2>&1
# counters: c4, then the config-5 shard
tools/pmc_run.sh $tag/pmc > $O/pmc_run.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/$tag/pmc gpurun_out/$tag/pmc_summary.txt --constants gpurun_out/$tag/pmc_constants.json --config c4 > /dev/null 2>&1
find gpurun_out/$tag/pmc/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
tools/pmc_run.sh $tag/pmc_c5 --config c5 --rows 250000 --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_c5_run.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/$tag/pmc_c5 gpurun_out/$tag/pmc_c5_summary.txt "--config c5 --rows 250000 --steps 2 --warmup 2 --no-cpu-baseline
# (c5 shard: n=250000 of 2e6, d=20 m=2000 VC hetero + diagonal Psi cubes, dtype f32, 1 x MI355X" --constants gpurun_out/$tag/pmc_constants.json --config c5 > /dev/null 2>&1
find gpurun_out/$tag/pmc_c5/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c5.csv \;
# counters of the latency-bound configurations (c2 / c3 had kernel traces but no FETCH_SIZE / WRITE_SIZE)
for cfg in c2 c3; do
  tools/pmc_run.sh $tag/pmc_$cfg --config $cfg --steps 5 --warmup 3 --no-cpu-baseline --timed-events none > $O/pmc_${cfg}_run.log 2>&1
  cd $R
  python tools/pmc_summary.py gpurun_out/$tag/pmc_$cfg gpurun_out/$tag/pmc_${cfg}_summary.txt "--config $cfg --steps 5 --warmup 3 --no-cpu-baseline --timed-events none
# ($cfg, 1 x MI355X" > /dev/null 2>&1
done
find gpurun_out/$tag/pmc gpurun_out/$tag/pmc_c5 gpurun_out/$tag/pmc_c2 gpurun_out/$tag/pmc_c3 -name "*.csv" -size +4M -delete
# microbenchmarks / probes of the round
for t in graph_event_probe; do
  hipcc --offload-arch=gfx950 -O3 tools/$t.hip -o build/$t 2> /dev/null && build/$t > $O/ubench_$t.txt 2>&1
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/gemm_trace.hip gpz_amd/csrc/gpz_options.hip -o build/gemm_trace 2> /dev/null && build/gemm_trace > $O/tgemm_timeline.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/small_trace.hip -o build/small_trace 2> /dev/null && (build/small_trace 100000 200 10 4 2; build/small_trace 100000 200 10 4 1) > $O/small_tail_timeline.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/chol_trace.hip gpz_amd/csrc/gpz_options.hip -o build/chol_trace 2> /dev/null && (build/chol_trace 1000; build/chol_trace 500; build/chol_trace 200) > $O/chol_step_timeline.txt 2>&1
# kernel statistics of the latency-bound configurations
for cfg in c2 c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python bench.py --config $cfg --timed-events none --no-cpu-baseline --steps 50 > /dev/null 2>&1
  find $O/prof_$cfg -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$cfg.csv \;
  rm -rf $O/prof_$cfg
done
tail -c 400 $O/c4.json; echo; cat $O/scaling_prediction.log | tail -30
