#!/bin/bash
# r04x: the exact table's bit filter in sketch_exact_kernel too (batches with a large genome): gpu suite, configs[3] / [4] / [2] benches
tag=${1:-r04x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -4 gpurun_out/$tag/tests_gpu.log
timeout 600 python bench.py --workload fastq --cpu-sample 0 --parity-reads 0 > gpurun_out/$tag/bench_fastq.json 2> gpurun_out/$tag/bench_fastq.err
timeout 600 python bench.py --workload mammal --genomes 8 --cpu-sample 0 > gpurun_out/$tag/bench_mammal8.json 2> gpurun_out/$tag/bench_mammal8.err
timeout 900 python bench.py --genomes 10000 --clades 500 --cpu-sample 0 --steps 10 --warmup 2 > gpurun_out/$tag/bench_config3.json 2> gpurun_out/$tag/bench_config3.err
python3 - <<PY
import json
for f in ("bench_fastq", "bench_mammal8", "bench_config3"):
    try:
        j = json.loads(open("gpurun_out/$tag/%s.json" % f).read().strip().splitlines()[-1])
        print(f, j["value"], j["unit"], "ms_per_step %.4f" % j["ms_per_step"], "frac %.4f" % j["roofline"]["frac"], "dist", j.get("dist_ms_per_step"), json.dumps(j.get("kernels"))[:300])
    except Exception as e:
        print(f, "failed", e)
PY
