#!/bin/bash
# r04b: the one-command all-pairs flow, the in-process exchange, the torch path on two gloo ranks, rows-kernel phases
tag=${1:-r04b}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_allpairs_flow.py tests/test_bench_launch.py tests/test_gpu_dist.py tests/test_wide_tuples.py -m gpu -x -q > gpurun_out/$tag/tests_new.log 2>&1
echo "rc=$?" >> gpurun_out/$tag/tests_new.log
tail -12 gpurun_out/$tag/tests_new.log
KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DISTTIME=1 timeout 600 python3 profiles/dist_phases.py > gpurun_out/$tag/dist_phases.txt 2>&1
cat gpurun_out/$tag/dist_phases.txt | tail -12
KSSD_BENCH_ONE_DEVICE=1 KSSD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --genomes 400 --cpu-sample 0 \
  > gpurun_out/$tag/bench_n2_onegpu.json 2> gpurun_out/$tag/bench_n2_onegpu.err
echo "n2 rc=$?"; cut -c1-400 gpurun_out/$tag/bench_n2_onegpu.json; tail -3 gpurun_out/$tag/bench_n2_onegpu.err
timeout 600 python bench.py --gpus 1 --exchange c --steps 20 --warmup 3 > gpurun_out/$tag/bench_exchange_c.json 2> gpurun_out/$tag/bench_exchange_c.err
echo "xc rc=$?"; cut -c1-1500 gpurun_out/$tag/bench_exchange_c.json; tail -3 gpurun_out/$tag/bench_exchange_c.err
