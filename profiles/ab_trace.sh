#!/bin/bash
# usage (GPU box): profiles/ab_trace.sh <tag> <kernel regex>   kernel averages of the default bench step, product library (A)
# against profiles/libkssd_gpu_alt.so (B), rocprofv3 kernel trace
tag=$1; rx=$2
cd $GRAFT_REPO_ROOT
for v in A B; do
  if [ $v = B ]; then export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_alt.so; else unset KSSD_GPU_LIB; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_$v -- python bench.py --steps 20 --warmup 2 --cpu-sample 0 > gpurun_out/$tag/prof_$v.log 2>&1
  f=$(find gpurun_out/$tag/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$rx" $v <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(sys.argv[3], "%-50s calls %4s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/$tag/prof_$v
done
