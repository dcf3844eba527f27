# Developer measurement (GPU box): c4 on the int8-sliced T-GEMM route of the developer build - kernel trace + bench line
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf $R/gpurun_out/r05oz
export GPZ_HIP_LIB=$R/gpz_amd/lib/libgpz_hip_dev.so
GPZ_TGEMM_INT8=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05oz/trace -o t -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 > $R/gpurun_out/r05oz_bench_trace.json 2>/dev/null
GPZ_TGEMM_INT8=1 python $R/bench.py --steps 10 --no-cpu-baseline > $R/gpurun_out/r05oz_bench.json 2>/dev/null
find $R/gpurun_out/r05oz -name "*kernel_stats.csv" | head -1 | xargs head -9 | cut -c1-110
cd $R; python tools/oz_check.py 20000 1000 10 VC 2>&1 | grep "int8 vs\|ms/eval"
