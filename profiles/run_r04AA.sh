#!/bin/bash
# r04AA: windows per table read of the scan's group filter again, now that the blocks come from a queue (5: product; 6 and 4)
tag=${1:-r04AA}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for v in gw6 gw4; do KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_$v.so timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q -k "not telemetry" 2>&1 | tail -1; done
for v in gw5 gw6 gw4 gw5 gw6 gw4; do
  KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_$v.so timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/err_$v.log
  python3 -c "
import json
j=json.loads(open('gpurun_out/$tag/bench_$v.json').read().strip().splitlines()[-1])
print('$v: ms_per_step %.4f  scan %.4f ms (frac %.4f) past stage 1 %.4f %%' % (j['ms_per_step'], j['kernels']['sketch_scan_ms'], j['roofline']['frac'], 100*j['kernels']['scan_positions_past_stage1']))"
done 2>&1 | tee gpurun_out/$tag/ab.txt
