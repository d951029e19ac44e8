#!/bin/bash
mkdir -p gpurun_out/r03o
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q > gpurun_out/r03o/tests_focus.log 2>&1
echo "focus rc=$?" >> gpurun_out/r03o/tests_focus.log
tail -5 gpurun_out/r03o/tests_focus.log
bash profiles/run_r03o2.sh | grep -v "at::native\|rocclr" | cut -c1-150
