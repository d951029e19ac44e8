#!/bin/bash
# r04X: the scan's blocks from a queue whose head is a word of LDS (q1) against blocks dealt out in turn (q0): sketch tests, A/B of the
# default line, per-wave times of the queue build
tag=${1:-r04X}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_q1.so timeout 1500 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
for v in q1 q0 q1 q0 q1 q0; do
  KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_$v.so timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/err_$v.log
  python3 -c "
import json
j=json.loads(open('gpurun_out/$tag/bench_$v.json').read().strip().splitlines()[-1])
print('$v: ms_per_step %.4f  scan %.4f ms (frac %.4f)' % (j['ms_per_step'], j['kernels']['sketch_scan_ms'], j['roofline']['frac']))"
done 2>&1 | tee gpurun_out/$tag/ab.txt
KSSD_DEV_WAVETIME=1 timeout 300 profiles/scanbench 1000 5000000 10 | grep -v "^stats\|^whole" | tee -a gpurun_out/$tag/ab.txt
