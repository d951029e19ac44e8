#!/bin/bash
# round 3, third GPU call: how the scan launch scales with the batch (fixed cost or tail?), per-wave time stamps, and a
# first pass of the configs[4] workload
tag=${1:-r03c}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for g in 200 400 1000 2000 4000; do
  echo "== genomes $g"
  KSSD_DEV_WAVETIME=1 timeout 300 profiles/scanbench $g 5000000 10 | grep -v "^stats\|^whole"
done
echo "== ablation 1 (loads only), 1000 genomes"; KSSD_DEV_ABLATE=1 KSSD_DEV_WAVETIME=1 timeout 300 profiles/scanbench 1000 5000000 10 | grep -v "^stats\|^whole"
echo "== ablation 2 (stage 1 only), 1000 genomes"; KSSD_DEV_ABLATE=2 KSSD_DEV_WAVETIME=1 timeout 300 profiles/scanbench 1000 5000000 10 | grep -v "^stats\|^whole"
} > gpurun_out/${tag}_scan_scaling.txt 2>&1
cat gpurun_out/${tag}_scan_scaling.txt
timeout 1500 python bench.py --workload mammal --genomes 2 --steps 3 --warmup 1 > gpurun_out/${tag}_bench_mammal2.json 2> gpurun_out/${tag}_bench_mammal2.err
echo "mammal rc=$?"; cut -c1-3000 gpurun_out/${tag}_bench_mammal2.json; tail -15 gpurun_out/${tag}_bench_mammal2.err
