mkdir -p gpurun_out/r05d
python -m pytest tests -m gpu -x -q > gpurun_out/r05d/pytest_gpu.log 2>&1; tail -2 gpurun_out/r05d/pytest_gpu.log | head -1
python bench.py --config c2 --timed-events none --no-cpu-baseline > gpurun_out/r05d/c2.json 2> gpurun_out/r05d/c2.err
python bench.py --config c3 --timed-events none --no-cpu-baseline > gpurun_out/r05d/c3.json 2> gpurun_out/r05d/c3.err
python bench.py --rows 125000 --no-cpu-baseline > gpurun_out/r05d/c4_shard125k.json 2> gpurun_out/r05d/c4_shard125k.err
python bench.py --no-cpu-baseline > gpurun_out/r05d/c4.json 2> gpurun_out/r05d/c4.err
for f in c2 c3 c4_shard125k c4; do python -c "
import json,sys; d=json.loads(open('gpurun_out/r05d/$f.json').read().strip().splitlines()[-1]); s=d['kernels']['stage_ms_per_eval']; print('$f', d['value'], d['ms_per_step'], 'row_scalars', s['row_scalars'], 'solve_vectors', s['solve_vectors'])"; done
