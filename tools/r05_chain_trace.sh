# Kernel-trace timestamps of the m x m chain (start/end per launch): replayed graph of c2 and of the 125 000-row c4 shard
O=gpurun_out/r05t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for spec in "c2:--config c2" "c3:--config c3" "shard:--rows 125000"; do
  tag=${spec%%:*}; args=${spec#*:}
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python bench.py $args --timed-events none --no-cpu-baseline --steps 3 --warmup 2 > $O/$tag.json 2> $O/$tag.err
  f=$(find $O/$tag -name "*kernel_trace.csv" | head -1)
  python - "$f" $O/${tag}_trace.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last evaluation only: keep the last 400 launches
rows=rows[-400:]
t0=int(rows[0]['Start_Timestamp'])
with open(sys.argv[2],'w') as f:
    for r in rows:
        f.write("%s,%d,%d,%s\n"%(r['Kernel_Name'][:60].replace(',',';'),int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0,r.get('Grid_Size_X','')))
PY
  rm -rf $O/$tag
  tail -c 300 $O/$tag.json
done
