# Round-6 long fuzz pass (GPU box): fresh seeds on the final code of the round
mkdir -p gpurun_out/r06f
for spec in "fuzz_parity.py 1000 71601" "fuzz_parity.py 600 71602 wide" "fuzz_predict.py 600 71603" "fuzz_predict.py 300 71604 wide" "fuzz_f32.py 150 71605" "fuzz_mgpu.py 300 71606" "fuzz_sharded.py 80 71607 2"; do
  set -- $spec
  timeout 2400 python tools/$@ 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06f/$1_$3.txt
  echo "== $spec"; tail -1 gpurun_out/r06f/$1_$3.txt
done
