#!/bin/bash
# r03y: the cleaned gather-based scan: whole gpu suite, default bench, configs[3] and configs[4] lines
mkdir -p gpurun_out/r03y
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03y/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/r03y/tests_gpu.log
tail -4 gpurun_out/r03y/tests_gpu.log
timeout 900 python bench.py > gpurun_out/r03y/bench.json 2> gpurun_out/r03y/bench.err
timeout 900 python bench.py --workload fastq --cpu-sample 0 --parity-reads 0 > gpurun_out/r03y/bench_fastq.json 2> gpurun_out/r03y/bench_fastq.err
timeout 900 python bench.py --workload mammal --genomes 8 --cpu-sample 0 > gpurun_out/r03y/bench_mammal8.json 2> gpurun_out/r03y/bench_mammal8.err
timeout 900 python bench.py --genomes 10000 --clades 500 --cpu-sample 0 --steps 5 > gpurun_out/r03y/bench_config3.json 2> gpurun_out/r03y/bench_config3.err
python3 - <<PY
import json
for f in ("bench", "bench_fastq", "bench_mammal8", "bench_config3"):
    try:
        j = json.loads(open("gpurun_out/r03y/%s.json" % f).read().strip().splitlines()[-1])
        print(f, j["value"], j["unit"], "ms_per_step %.4f" % j["ms_per_step"], "frac %.4f" % j["roofline"]["frac"])
    except Exception as e:
        print(f, "failed", e)
PY
