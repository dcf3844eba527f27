# Round-5 closing pass on the GPU box: the GPU suite as the driver runs it, smoke(), then the long fuzz pass with fresh seeds
mkdir -p gpurun_out/r05z
python -m pytest tests -m gpu -x -q > gpurun_out/r05z/pytest_gpu.log 2>&1; tail -3 gpurun_out/r05z/pytest_gpu.log | head -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05z/smoke.log 2>&1; tail -1 gpurun_out/r05z/smoke.log
sed -i 's/r05f2/r05z/g' tools/r05_fuzz.sh
bash tools/r05_fuzz.sh
