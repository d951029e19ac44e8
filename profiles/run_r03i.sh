#!/bin/bash
# round 3: genomes sorted in LDS in parts (DEDUP_PARTS) -- parity suite, configs[4] with 8 and 50 records, kernel trace
tag=${1:-r03i}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${tag}_pytest.log
tail -8 gpurun_out/${tag}_pytest.log
timeout 1500 python bench.py --workload mammal --steps 5 --warmup 2 > gpurun_out/${tag}_bench_mammal8.json 2> gpurun_out/${tag}_bench_mammal8.err
echo "mammal8 rc=$?"; cut -c1-1700 gpurun_out/${tag}_bench_mammal8.json; tail -3 gpurun_out/${tag}_bench_mammal8.err
timeout 1500 python bench.py --workload mammal --genomes 50 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${tag}_bench_mammal50.json 2> gpurun_out/${tag}_bench_mammal50.err
echo "mammal50 rc=$?"; cut -c1-1500 gpurun_out/${tag}_bench_mammal50.json; tail -3 gpurun_out/${tag}_bench_mammal50.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --workload mammal --genomes 8 --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/${tag}_prof.log 2>&1
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/${tag}_mammal_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${tag}_mammal_kernel_stats.csv")))[:14]:
    print("%-70s calls %4d avg %9.1f us" % (r['Name'][:70], int(r['Calls']), float(r['AverageNs']) / 1e3))
PY
