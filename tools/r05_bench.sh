mkdir -p gpurun_out/r05c
python bench.py > gpurun_out/r05c/c4.json 2> gpurun_out/r05c/c4.err; tail -c 2500 gpurun_out/r05c/c4.json; tail -3 gpurun_out/r05c/c4.err
python bench.py --config c2 --no-cpu-baseline > gpurun_out/r05c/c2.json 2> gpurun_out/r05c/c2.err; python -c "
import json; d=json.loads(open('gpurun_out/r05c/c2.json').read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'], d['stage_pass'], d['route'])"
python bench.py --native-mgpu 8 --no-cpu-baseline --steps 5 > gpurun_out/r05c/c4_mgpu8.json 2> gpurun_out/r05c/c4_mgpu8.err; python -c "
import json; d=json.loads(open('gpurun_out/r05c/c4_mgpu8.json').read().strip().splitlines()[-1]); print('mgpu8', d['value'], d['ms_per_step'], d['stage_pass'], d['per_rank'][0]['route'], d['per_rank'][1])"; tail -3 gpurun_out/r05c/c4_mgpu8.err
python bench.py --rows 125000 --no-cpu-baseline > gpurun_out/r05c/c4_shard125k.json 2>gpurun_out/r05c/c4_shard125k.err;  python -c "
import json; d=json.loads(open('gpurun_out/r05c/c4_shard125k.json').read().strip().splitlines()[-1]); print('shard125k', d['value'], d['ms_per_step'], d['stage_pass'])"
