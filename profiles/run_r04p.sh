#!/bin/bash
# r04p: A/B of the rows kernel's probe step: 2, 4 or 6 slots (1, 2, 3 loads of 16 bytes) per id and step; per-row phases of each
tag=${1:-r04p}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for nl in 1 2 3 1 2 3; do
  echo "=== DIST_PROBE_LOADS=$nl"
  KSSD_GPU_LIB=build/variants/libkssd_gpu_dev_nl$nl.so KSSD_DEV_DISTTIME=1 timeout 600 python3 profiles/dist_phases.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/$tag/dist_phases_ab.txt 2>&1
grep -E "===|avg ms|start ->|postings|epilogue|whole|ends" gpurun_out/$tag/dist_phases_ab.txt
