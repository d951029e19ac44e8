#!/bin/bash
# r04Q: the driver's command three times on one box (spread of the line), then a longer fuzz of both paths
tag=${1:-r04Q}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for i in 1 2 3; do
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_$i.json 2> /dev/null
  python3 -c "
import json
j=json.loads(open('gpurun_out/$tag/bench_$i.json').read().strip().splitlines()[-1])
print('run $i: value %.0f ms_per_step %.4f scan %.4f frac %.4f | dist %.4f ms rows %.4f ms frac %.4f | traffic %s' % (j['value'], j['ms_per_step'], j['kernels']['sketch_scan_ms'], j['roofline']['frac'], j['dist_ms_per_step'], j['kernels']['dist_rows_ms'], j['roofline_dist']['frac'], j['roofline']['traffic']))"
done 2>&1 | tee gpurun_out/$tag/three_runs.txt
sed -i 's/for seed in range(12):/for seed in range(40):/' profiles/fuzz_sketch.py
sed -i 's/for seed in range(300):/for seed in range(1500):/' profiles/fuzz_dist.py
timeout 1800 python3 profiles/fuzz_sketch.py 2>&1 | tail -1 | tee -a gpurun_out/$tag/three_runs.txt
timeout 1800 python3 profiles/fuzz_dist.py 2>&1 | tail -1 | tee -a gpurun_out/$tag/three_runs.txt
