#!/bin/bash
# r04K: A/B of the scan's halo word: out of the next lane's register (DPP wave_shl:1; halo1) or loaded (halo0); sketch tests on halo1 first
tag=${1:-r04K}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_halo1.so timeout 1200 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q 2>&1 | tail -3
for v in halo1 halo0 halo1 halo0; do
  KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_$v.so timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/err_$v.log
  python3 -c "
import json
j=json.loads(open('gpurun_out/$tag/bench_$v.json').read().strip().splitlines()[-1])
print('$v: ms_per_step %.4f  scan %.4f ms (frac %.4f)' % (j['ms_per_step'], j['kernels']['sketch_scan_ms'], j['roofline']['frac']))"
done 2>&1 | tee gpurun_out/$tag/ab.txt
