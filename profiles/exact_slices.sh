#!/bin/bash
# sketch_exact_kernel with 6 / 3 / 2 / 1 workgroups per slice (KSSD_EXACT_PER_SLICE): kernel times of the configs[3] step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for ps in 6 3 2 1; do
  export KSSD_EXACT_PER_SLICE=$ps
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ex_$ps -- python3 bench.py --workload fastq --steps 10 --warmup 2 --cpu-sample 0 --parity-reads 0 > /dev/null 2>&1
  f=$(find gpurun_out/ex_$ps -name '*kernel_stats.csv' | head -1)
  echo "per_slice $ps"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "at::native" in n or "rocclr" in n: continue
    print("  %-60s calls %5s avg %9.1f us min %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
  rm -rf gpurun_out/ex_$ps
done
