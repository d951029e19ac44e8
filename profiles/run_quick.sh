#!/bin/bash
# quick check (GPU box): dist / index tests, the default bench line, its rocprofv3 kernel trace summary
tag=${1:-quick}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_set.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python3 - <<PY
import json
j=json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["kernels"])
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/$tag/prof.log 2>&1
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native" "$f" > gpurun_out/$tag/kernel_stats.csv
rm -rf gpurun_out/$tag/prof
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kernel_stats.csv")):
    if "rocclr" in r["Name"]: continue
    print("%-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
