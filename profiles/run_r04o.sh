#!/bin/bash
# r04o: per-row phases of dist_rows_kernel after the two-slot probes, with the rows' list / holder counts beside them
tag=${1:-r04o}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DISTTIME=1 timeout 600 python3 profiles/dist_phases.py > gpurun_out/$tag/dist_phases.txt 2>&1
cat gpurun_out/$tag/dist_phases.txt
