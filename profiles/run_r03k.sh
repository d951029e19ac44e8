#!/bin/bash
# round 3: the per-genome kernel's sort by bucket counting in LDS against the bitonic network (A/B in the development build),
# parity suite, default line, configs[4]
tag=${1:-r03k}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for g in 400 1000; do
  echo "== genomes $g, bucket sort"; KSSD_DEV_DEDUPTIME=1 timeout 300 profiles/scanbench $g 5000000 10 | grep -v "^stats"
  echo "== genomes $g, bitonic"; KSSD_DEV_NO_BUCKET_SORT=1 KSSD_DEV_DEDUPTIME=1 timeout 300 profiles/scanbench $g 5000000 10 | grep -v "^stats"
done
} > gpurun_out/${tag}_scanbench.txt 2>&1
cat gpurun_out/${tag}_scanbench.txt
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${tag}_pytest.log
tail -4 gpurun_out/${tag}_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${tag}_bench_quick.json 2> gpurun_out/${tag}_bench_quick.err
cut -c1-1200 gpurun_out/${tag}_bench_quick.json
timeout 1500 python bench.py --workload mammal --genomes 50 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${tag}_bench_mammal50.json 2> gpurun_out/${tag}_bench_mammal50.err
echo "mammal50 rc=$?"; cut -c1-1300 gpurun_out/${tag}_bench_mammal50.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/${tag}_prof.log 2>&1
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))[:12]:
    print("%-70s calls %4d avg %9.1f us" % (r['Name'][:70], int(r['Calls']), float(r['AverageNs']) / 1e3))
PY
