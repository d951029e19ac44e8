#!/bin/bash
# r04w: per-workgroup phases of the per-genome kernel after the round's changes (bit filter, 8-byte records, six candidates per thread)
tag=${1:-r04w}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
( KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DEDUPTIME=1 timeout 600 python3 profiles/dedup_phases.py
  KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DEDUPTIME=1 KSSD_DEV_GATHERSPLIT=1 timeout 600 python3 profiles/dedup_phases.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/$tag/dedup_phases.txt
cat gpurun_out/$tag/dedup_phases.txt
