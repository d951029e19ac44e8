#!/bin/bash
# r04N: fuzz of the sketch and distance paths against the oracle after the round's kernel changes (8-byte records, bit filter, two slots
# per probe step, 8-byte index slots, partition / build kernels)
tag=${1:-r04N}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1500 python3 profiles/fuzz_sketch.py > gpurun_out/$tag/fuzz_sketch.txt 2>&1; echo "rc=$?" >> gpurun_out/$tag/fuzz_sketch.txt
timeout 1500 python3 profiles/fuzz_dist.py > gpurun_out/$tag/fuzz_dist.txt 2>&1; echo "rc=$?" >> gpurun_out/$tag/fuzz_dist.txt
tail -4 gpurun_out/$tag/fuzz_sketch.txt; tail -4 gpurun_out/$tag/fuzz_dist.txt
