#!/bin/bash
# r04AF: waves per scan workgroup under the block queue: 16 (product), 12, 8
tag=${1:-r04AF}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_t768.so timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q 2>&1 | tail -1
for v in t1024 t768 t512 t1024 t768 t512; do
  KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_$v.so timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/err_$v.log
  python3 -c "
import json
j=json.loads(open('gpurun_out/$tag/bench_$v.json').read().strip().splitlines()[-1])
print('$v: ms_per_step %.4f  scan %.4f ms (frac %.4f)' % (j['ms_per_step'], j['kernels']['sketch_scan_ms'], j['roofline']['frac']))" || tail -3 gpurun_out/$tag/err_$v.log
done 2>&1 | tee gpurun_out/$tag/ab.txt
