#!/bin/bash
# round 5: the exact-evaluation kernel of configs[3] (one read set = one genome): no chunk -> genome lookup; candidates per thread 2 / 4 / 8;
# without the cursor atomic (development build, wrong results): what bounds it
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05p; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -2
for v in product per2 per8 dev dev_noatomic; do
  unset KSSD_GPU_LIB KSSD_DEV_EXACT_NO_ATOMIC
  case $v in per2|per8|dev) export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_$v.so;; dev_noatomic) export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_dev.so KSSD_DEV_EXACT_NO_ATOMIC=1;; esac
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --workload fastq --reads 100000000 --steps 10 --warmup 2 --parity-reads 0 > $o/prof_$v.json 2> $o/prof_$v.err
  f=$(find $o/prof -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $v <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sketch_exact" in r["Name"]:
        print("%-14s sketch_exact_kernel calls %4s avg %8.1f us min %8.1f" % (sys.argv[2], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
  rm -rf $o/prof
done 2>&1 | tee $o/exact_variants.txt
