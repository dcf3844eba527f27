# Round-5 long fuzz pass (GPU box): fresh seeds on the final code of the round
mkdir -p gpurun_out/r05f2
for spec in "fuzz_parity.py 1000 70601" "fuzz_parity.py 600 70602 wide" "fuzz_predict.py 600 70603" "fuzz_predict.py 300 70604 wide" "fuzz_f32.py 150 70605" "fuzz_mgpu.py 300 70606" "fuzz_sharded.py 80 70607 2"; do
  set -- $spec
  timeout 2400 python tools/$@ 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r05f2/$1_$3.txt
  echo "== $spec"; tail -1 gpurun_out/r05f2/$1_$3.txt
done
