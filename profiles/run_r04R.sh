#!/bin/bash
# r04R: A/B of the rows kernel's epilogue: the references' sketch sizes asked for in front of the last barrier (pre1) or inside the epilogue (pre0)
tag=${1:-r04R}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
KSSD_GPU_LIB=$GRAFT_REPO_ROOT/build/variants/libkssd_gpu_dev_pre1.so timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
for v in pre1 pre0 pre1 pre0 pre1 pre0; do
  echo "=== $v"
  KSSD_GPU_LIB=build/variants/libkssd_gpu_dev_$v.so KSSD_DEV_DISTTIME=1 timeout 600 python3 profiles/dist_phases.py 2>&1 | grep -v amdgpu.ids | grep -E "avg ms|epilogue|whole|ends"
done > gpurun_out/$tag/ab.txt 2>&1
cat gpurun_out/$tag/ab.txt
