#!/bin/bash
# r04I: the end-to-end pipeline's knobs (sketch workers per device, text buffers) on 1 024 files
tag=${1:-r04I}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1200 python3 profiles/e2e_knobs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/e2e_knobs.txt
