#!/bin/bash
# r04g: exact-table buckets as ONE 16-byte load (were four word loads): sketch tests, kernel trace, bench
tag=${1:-r04g}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -4 gpurun_out/$tag/tests_gpu.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python3 bench.py --steps 20 --warmup 2 --cpu-sample 0 > gpurun_out/$tag/prof.log 2>&1
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/$tag/kernel_stats.csv
rm -rf gpurun_out/$tag/prof
cut -d, -f1-4 gpurun_out/$tag/kernel_stats.csv | cut -c1-120 | head -12
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python3 - <<PY
import json
j = json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
print("ms_per_step %.4f" % j["ms_per_step"], "value", j["value"], j["kernels"])
PY
timeout 600 python bench.py --workload fastq --cpu-sample 0 --parity-reads 0 > gpurun_out/$tag/bench_fastq.json 2> gpurun_out/$tag/bench_fastq.err
timeout 600 python bench.py --workload mammal --genomes 8 --cpu-sample 0 > gpurun_out/$tag/bench_mammal8.json 2> gpurun_out/$tag/bench_mammal8.err
python3 - <<PY
import json
for f in ("bench_fastq", "bench_mammal8"):
    try:
        j = json.loads(open("gpurun_out/$tag/%s.json" % f).read().strip().splitlines()[-1])
        print(f, j["value"], j["unit"], "ms_per_step %.4f" % j["ms_per_step"], "frac %.4f" % j["roofline"]["frac"])
    except Exception as e:
        print(f, "failed", e)
PY
