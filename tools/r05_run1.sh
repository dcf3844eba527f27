set -x
mkdir -p gpurun_out/r05a
build/ozaki_probe > gpurun_out/r05a/ozaki_probe.txt 2>&1
cat gpurun_out/r05a/ozaki_probe.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r05a/pmc_ubench -o p -- $R/tools/mfma_f64_bench 1000 > $R/gpurun_out/r05a/pmc_ubench.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/r05a/pmc_ubench/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:40]][r['Counter_Name']] += float(r['Counter_Value']); 
    for k, v in agg.items(): print(k, dict(v))
PY
python bench.py > gpurun_out/r05a/c4.json 2> gpurun_out/r05a/c4.err; tail -c 1500 gpurun_out/r05a/c4.json
