#!/bin/bash
# r04F: what the events around the scan and rows launches cost the step: every launch bracketed, every 4th, every 20th
tag=${1:-r04F}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for n in 1 4 20 1 4 20; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 --kernel-timing $n > gpurun_out/$tag/bench_$n.json 2> gpurun_out/$tag/err_$n.log
  python3 -c "
import json
j=json.loads(open('gpurun_out/$tag/bench_$n.json').read().strip().splitlines()[-1])
print('timing every $n: ms_per_step %.4f  dist %.4f  scan %.4f ms rows %.4f ms timed %s' % (j['ms_per_step'], j['dist_ms_per_step'], j['kernels']['sketch_scan_ms'], j['kernels']['dist_rows_ms'], j['kernels']['launches_timed']))"
done 2>&1 | tee gpurun_out/$tag/ab.txt
