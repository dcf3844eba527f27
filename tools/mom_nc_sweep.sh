# Developer tool (GPU box): replayed ms per evaluation of c2 / c3 / c4 over GPZ_MOM_NC = row chunks of the moment kernel (times the
# number of 256-wide basis-function groups = workgroups per launch).  usage: bash tools/mom_nc_sweep.sh
for nc in 2048 1024 768 512; do for c in c2 c3 c4; do
GPZ_MOM_NC=$nc python bench.py --config $c --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$nc', '$c', d['ms_per_step'], d.get('graph_pass',{}).get('ms_per_step'), {k:round(v,3) for k,v in d.get('stages_ms',{}).items() if 'mom' in k or 'grad' in k})"
done; done
