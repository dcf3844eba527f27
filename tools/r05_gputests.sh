# Round-5 GPU test pass: pytest -m gpu (release library; the route-comparison tests start the developer build in a subprocess)
mkdir -p gpurun_out/r05b
python -m pytest tests -m gpu -x -q > gpurun_out/r05b/pytest_gpu.log 2>&1
tail -15 gpurun_out/r05b/pytest_gpu.log
