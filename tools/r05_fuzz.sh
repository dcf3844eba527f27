mkdir -p gpurun_out/r05f
for spec in "fuzz_parity.py 300 50501" "fuzz_parity.py 200 50502 wide" "fuzz_predict.py 200 50503" "fuzz_predict.py 120 50504 wide" "fuzz_f32.py 60 50505" "fuzz_mgpu.py 120 50506" "fuzz_sharded.py 40 50507 2"; do
  set -- $spec
  timeout 1500 python tools/$@ 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r05f/$1_$3.txt
  echo "== $spec"; tail -2 gpurun_out/r05f/$1_$3.txt
done
