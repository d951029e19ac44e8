#!/bin/bash
# r04Z2: wave priorities of the scan again, now that its blocks come from a queue (p1: as tuned in round 3, p0: none, p2: only the issue phase raised)
tag=${1:-r04Z2}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for v in p1 p0 p2 p1 p0 p2; do
  KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_$v.so timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/err_$v.log
  python3 -c "
import json
j=json.loads(open('gpurun_out/$tag/bench_$v.json').read().strip().splitlines()[-1])
print('$v: ms_per_step %.4f  scan %.4f ms (frac %.4f)' % (j['ms_per_step'], j['kernels']['sketch_scan_ms'], j['roofline']['frac']))"
done 2>&1 | tee gpurun_out/$tag/ab.txt
