#!/bin/bash
# r04AC: alignment B asked only where alignment A left a position standing (bwa1: address 0 = one broadcast for the lanes whose group
# cannot matter) against the product (bwa0), both with the blocks from the LDS queue -- round 3 lost 5.5 % with it under fixed shares
tag=${1:-r04AC}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_bwa1.so timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q 2>&1 | tail -1
for v in bwa1 bwa0 bwa1 bwa0 bwa1 bwa0; do
  KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_$v.so timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/err_$v.log
  python3 -c "
import json
j=json.loads(open('gpurun_out/$tag/bench_$v.json').read().strip().splitlines()[-1])
print('$v: ms_per_step %.4f  scan %.4f ms (frac %.4f)' % (j['ms_per_step'], j['kernels']['sketch_scan_ms'], j['roofline']['frac']))"
done 2>&1 | tee gpurun_out/$tag/ab.txt
