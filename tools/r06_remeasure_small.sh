#!/bin/bash
# after the last changes to k_small_tail: the c2 files of tools/measure_all.sh again (bench lines, PMC passes, kernel statistics, phase timeline)
tag=r06; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
python bench.py --config c2 --timed-events none > $O/c2.json 2> $O/c2.err
python bench.py --config c2 > $O/c2_events.json 2> $O/c2_events.err
python bench.py --config c2 --validation 0.15 --timed-events none > $O/c2_validation15.json 2> $O/c2_validation15.err
rm -rf $O/pmc_c2; tools/pmc_run.sh $tag/pmc_c2 --config c2 --steps 5 --warmup 3 --no-cpu-baseline --timed-events none > $O/pmc_c2_run.log 2>&1
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python bench.py --config c2 --timed-events none --no-cpu-baseline --steps 50 > /dev/null 2>&1
find $O/prof_c2 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c2.csv \; ; rm -rf $O/prof_c2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/small_trace.hip -o build/small_trace 2> /dev/null && (build/small_trace 100000 200 10 1 2; build/small_trace 100000 200 10 1 1) > $O/small_tail_timeline.txt 2>&1
tools/r06_small_ab.sh c2 > $O/small_tail_ab.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python -m pytest tests -q -m gpu > $O/gputests.txt 2>&1; grep -E "passed|failed" $O/gputests.txt | tail -1
tail -c 300 $O/c2.json
