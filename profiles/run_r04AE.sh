#!/bin/bash
# r04AE: buckets of the per-genome LDS bucket sort: as many as key slots (f44), half (f63), a quarter (f62)
tag=${1:-r04AE}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_f62.so timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q 2>&1 | tail -1
for v in f44 f63 f62 f44 f63 f62; do
  KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_$v -- python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/err_$v.log
  f=$(find gpurun_out/$tag/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v $(python3 -c "import json;j=json.loads(open('gpurun_out/$tag/bench_$v.json').read().strip().splitlines()[-1]);print('ms_per_step %.4f'%j['ms_per_step'])")"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sketch_dedup" in r["Name"] or "sketch_scan" in r["Name"] or "sketch_gather" in r["Name"]:
        print("   %-40s calls %s avg %.2f us min %.2f max %.2f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf gpurun_out/$tag/prof_$v
done 2>&1 | tee gpurun_out/$tag/ab.txt
