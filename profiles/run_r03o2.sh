#!/bin/bash
# r03o: rocprofv3 kernel trace of the configs[3] bench (read set: the large-genome path)
mkdir -p gpurun_out/r03o
cd /root/repo
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03o/prof_fq -- python bench.py --workload fastq --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r03o/prof_bench.log 2>&1
f=$(find gpurun_out/r03o/prof_fq -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r03o/fastq_kernel_stats.csv
rm -rf gpurun_out/r03o/prof_fq
cut -c1-200 gpurun_out/r03o/fastq_kernel_stats.csv | head -40
