#!/bin/bash
# r04S: the multi-tile partition kernel with the wave search and the pass's genome offsets in LDS: dist + configs tests, configs[2] trace
tag=${1:-r04S}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
bash profiles/run_r04L.sh $tag
