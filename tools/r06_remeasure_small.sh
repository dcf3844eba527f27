#!/bin/bash
# after the last changes to k_small_tail / k_syrk_small: the c2 files of tools/measure_all.sh again (bench lines, PMC passes, kernel statistics,
# phase timeline, the A/B runs of both kernels)
tag=r06; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
rm -rf $O/pmc_c2; tools/pmc_run.sh $tag/pmc_c2 --config c2 --steps 5 --warmup 3 --no-cpu-baseline --timed-events none > $O/pmc_c2_run.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/$tag/pmc_c2 gpurun_out/$tag/pmc_c2_summary.txt "--config c2 --steps 5 --warmup 3 --no-cpu-baseline --timed-events none
# (c2, 1 x MI355X" --constants profiles/pmc_constants.json --config c2 > /dev/null 2>&1 || true
cp profiles/pmc_constants.json gpurun_out/$tag/pmc_c2_summary.txt $O/ 2> /dev/null   # (the bench lines below read the constants of THIS pass)
python bench.py --config c2 --timed-events none > $O/c2.json 2> $O/c2.err
python bench.py --config c2 > $O/c2_events.json 2> $O/c2_events.err
python bench.py --config c2 --validation 0.15 --timed-events none > $O/c2_validation15.json 2> $O/c2_validation15.err
python bench.py --config c3 --timed-events none > $O/c3.json 2> $O/c3.err   # (c3 shares k_ltl_small and the launch merges, not the two big kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python bench.py --config c2 --timed-events none --no-cpu-baseline --steps 50 > /dev/null 2>&1
find $O/prof_c2 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c2.csv \; ; rm -rf $O/prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python bench.py --config c3 --timed-events none --no-cpu-baseline --steps 50 > /dev/null 2>&1
find $O/prof_c3 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c3.csv \; ; rm -rf $O/prof_c3
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/small_trace.hip -o build/small_trace 2> /dev/null && (build/small_trace 100000 200 10 1 2; build/small_trace 100000 200 10 1 1) > $O/small_tail_timeline.txt 2>&1
tools/r06_small_ab.sh c2 > $O/small_tail_ab.txt 2>&1
bash tools/r06_syrk_small_ab.sh > /dev/null 2>&1
bash tools/r06_small_nan_ab.sh > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python -m pytest tests -q -m gpu > $O/gputests.txt 2>&1; grep -E "passed|failed" $O/gputests.txt | tail -1
tail -c 300 $O/c2.json
