#!/bin/bash
# r04O: A/B of the rows kernel's work lists: one per WAVE (wl1: no workgroup barrier between probing and walking) or one per row (wl0)
tag=${1:-r04O}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_dev_wl1.so timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
for v in wl1 wl0 wl1 wl0; do
  echo "=== $v"
  KSSD_GPU_LIB=build/variants/libkssd_gpu_dev_$v.so KSSD_DEV_DISTTIME=1 timeout 600 python3 profiles/dist_phases.py 2>&1 | grep -v amdgpu.ids | grep -E "avg ms|start ->|postings|epilogue|whole|ends"
done > gpurun_out/$tag/ab.txt 2>&1
cat gpurun_out/$tag/ab.txt
