#!/bin/bash
# round 3: where the per-genome kernel's time goes (phase time stamps, development build), the scan's table copy with all
# nine pieces requested up front, parts path after its two fixes (configs[4]), parity suite
tag=${1:-r03j}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for g in 400 1000; do
  echo "== genomes $g"; KSSD_DEV_WAVETIME=1 KSSD_DEV_DEDUPTIME=1 timeout 300 profiles/scanbench $g 5000000 10 | grep -v "^stats"
done
} > gpurun_out/${tag}_scanbench.txt 2>&1
cat gpurun_out/${tag}_scanbench.txt
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${tag}_pytest.log
tail -4 gpurun_out/${tag}_pytest.log
timeout 1500 python bench.py --workload mammal --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/${tag}_bench_mammal8.json 2> gpurun_out/${tag}_bench_mammal8.err
echo "mammal8 rc=$?"; cut -c1-1300 gpurun_out/${tag}_bench_mammal8.json
timeout 1500 python bench.py --workload mammal --genomes 50 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${tag}_bench_mammal50.json 2> gpurun_out/${tag}_bench_mammal50.err
echo "mammal50 rc=$?"; cut -c1-1300 gpurun_out/${tag}_bench_mammal50.json
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${tag}_bench_quick.json 2> gpurun_out/${tag}_bench_quick.err
cut -c1-900 gpurun_out/${tag}_bench_quick.json
