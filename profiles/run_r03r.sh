#!/bin/bash
# r03r: what bounds sketch_exact_kernel at configs[3] size (8.6 M candidates of ONE genome)?  A/B in the dev build: the
# workgroups' reservation at the genome's one cursor left out (results wrong, time only)
mkdir -p gpurun_out/r03r
cd $GRAFT_REPO_ROOT
export KSSD_GPU_LIB=$PWD/public_kssd_amd/libkssd_gpu_dev.so
for v in product no_atomic; do
  [ $v = no_atomic ] && export KSSD_DEV_EXACT_NO_ATOMIC=1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03r/prof_$v -- python bench.py --workload fastq --steps 5 --warmup 2 --cpu-sample 0 --reads 20000000 > gpurun_out/r03r/$v.log 2>&1
  f=$(find gpurun_out/r03r/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  cp "$f" gpurun_out/r03r/kernel_stats_$v.csv
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("sketch_exact", "sketch_scan", "big_rng", "sketch_dedup")):
        print("%-56s calls %4s avg %9.1f us  max %9.1f" % (n[:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf gpurun_out/r03r/prof_$v
done
