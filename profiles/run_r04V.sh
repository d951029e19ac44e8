#!/bin/bash
# r04V: the bench's N > 1 flow on one GPU (two ranks on cuda:0, gloo)
tag=${1:-r04V}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/$tag/out.txt
KSSD_BENCH_ONE_DEVICE=1 KSSD_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 --spinup 0 --genomes 200 --length 1000000 --clades 10 > gpurun_out/$tag/bench_2ranks.json 2> gpurun_out/$tag/err.log
tail -3 gpurun_out/$tag/err.log; cut -c1-1500 gpurun_out/$tag/bench_2ranks.json
