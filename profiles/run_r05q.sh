#!/bin/bash
# the exact-evaluation kernel of configs[3] by parts (development build, ablations give wrong results)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05q; mkdir -p $o
export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_dev.so
for v in 0 1 2 3 4; do
  export KSSD_DEV_EXACT_NO_ATOMIC=$v
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --workload fastq --reads 100000000 --steps 10 --warmup 2 --parity-reads 0 > $o/prof_$v.json 2> $o/prof_$v.err
  f=$(find $o/prof -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $v <<PY
import csv, sys
what = {"0": "whole kernel", "1": "no cursor atomic", "2": "no exact-table read (nothing staged)", "3": "no staging (evaluation only)", "4": "launch + filter copy only"}
for r in csv.DictReader(open(sys.argv[1])):
    if "sketch_exact" in r["Name"]:
        print("%-40s sketch_exact_kernel avg %8.1f us min %8.1f" % (what[sys.argv[2]], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
  rm -rf $o/prof
done 2>&1 | tee $o/exact_by_parts.txt
