# Round-6 long fuzz pass (GPU box): fresh seeds on the final code of the round
mkdir -p gpurun_out/r06g
for spec in "fuzz_parity.py 2000 81601" "fuzz_parity.py 600 81602 wide" "fuzz_predict.py 600 81603" "fuzz_predict.py 300 81604 wide" "fuzz_f32.py 150 81605" "fuzz_mgpu.py 300 81606" "fuzz_sharded.py 80 81607 2"; do
  set -- $spec
  timeout 2400 python tools/$@ 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06g/$1_$3.txt
  echo "== $spec"; tail -1 gpurun_out/r06g/$1_$3.txt
done
