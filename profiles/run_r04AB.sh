#!/bin/bash
# r04AB: configs[2] / [3] / [4] with the scan's LDS block queue
tag=${1:-r04AB}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 600 python bench.py --workload fastq --cpu-sample 0 --parity-reads 0 > gpurun_out/$tag/bench_fastq.json 2> gpurun_out/$tag/bench_fastq.err
timeout 600 python bench.py --workload mammal --genomes 8 --cpu-sample 0 > gpurun_out/$tag/bench_mammal8.json 2> gpurun_out/$tag/bench_mammal8.err
timeout 900 python bench.py --genomes 10000 --clades 500 --cpu-sample 0 --steps 10 --warmup 2 > gpurun_out/$tag/bench_config3.json 2> gpurun_out/$tag/bench_config3.err
python3 - <<PY
import json
for f in ("bench_fastq", "bench_mammal8", "bench_config3"):
    try:
        j = json.loads(open("gpurun_out/$tag/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "%.0f" % j["value"], j["unit"], "ms_per_step %.4f" % j["ms_per_step"], "frac %.4f" % j["roofline"]["frac"], "dist", j.get("dist_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
