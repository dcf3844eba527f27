#!/bin/bash
# rocprofv3 kernel statistics of tools/predict_missing_profile.py (run on the GPU box via gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd /tmp
python $R/tools/predict_missing_profile.py "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ppm -o t -- python $R/tools/predict_missing_profile.py "$@" > /tmp/ppm.log 2>&1
tail -1 /tmp/ppm.log
f=$(find /tmp/ppm -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv,sys
rows=list(csv.DictReader(open('$f')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6, 'launches', sum(int(r['Calls']) for r in rows))
for r in rows[:12]: print(r['Name'][:60], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1),'us')
"
