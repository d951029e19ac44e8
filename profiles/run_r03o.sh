#!/bin/bash
# r03o: large genomes sorted by key ranges in LDS (DEDUP_RANGES) instead of the device radix sort: tests, the configs[3] bench line
mkdir -p gpurun_out/r03o
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q -k "ranges or parts or counts or by_pos or min_occ" > gpurun_out/r03o/tests_focus.log 2>&1
echo "focus rc=$?" >> gpurun_out/r03o/tests_focus.log
tail -15 gpurun_out/r03o/tests_focus.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03o/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/r03o/tests_gpu.log
tail -5 gpurun_out/r03o/tests_gpu.log
timeout 900 python bench.py --workload fastq > gpurun_out/r03o/bench_fastq.json 2> gpurun_out/r03o/bench_fastq.err
tail -c 1200 gpurun_out/r03o/bench_fastq.json
