#!/bin/bash
# r03m: read starts on the device (--byread), nccl single-rank gather test, then the full gpu suite
mkdir -p gpurun_out/r03m
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_tokenise.py tests/test_gpu_cli.py tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/r03m/tests_focus.log 2>&1
echo "focus rc=$?" >> gpurun_out/r03m/tests_focus.log
tail -5 gpurun_out/r03m/tests_focus.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03m/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/r03m/tests_gpu.log
tail -5 gpurun_out/r03m/tests_gpu.log
timeout 300 python bench.py > gpurun_out/r03m/bench.json 2> gpurun_out/r03m/bench.err
tail -c 1500 gpurun_out/r03m/bench.json
