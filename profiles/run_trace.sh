#!/bin/bash
# usage (GPU box): profiles/run_trace.sh <tag> <bench args...>   rocprofv3 kernel trace summary of one bench command
tag=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python bench.py --cpu-sample 0 "$@" > gpurun_out/$tag/prof.log 2>&1
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native" "$f" > gpurun_out/$tag/kernel_stats.csv
rm -rf gpurun_out/$tag/prof
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kernel_stats.csv")):
    if "rocclr" in r["Name"]: continue
    print("%-60s calls %5s avg %9.1f us  max %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
PY
