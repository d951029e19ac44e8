#!/bin/bash
# r03n: dist_rows_kernel A/B inside one call (old = libkssd_gpu_old.so built from the commit before)
mkdir -p gpurun_out/r03n
cd /root/repo
O=gpurun_out/r03n/dist_ab.txt
: > $O
for rep in 1 2; do
echo "== old" >> $O
KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_old.so python3 profiles/dist_phases.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== new" >> $O
KSSD_GPU_LIB=$PWD/public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DISTTIME=1 python3 profiles/dist_phases.py 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== 10000 rows: old / new" >> $O
KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_old.so python3 profiles/dist_phases.py 10000 2>&1 | grep -v amdgpu.ids >> $O
KSSD_GPU_LIB=$PWD/public_kssd_amd/libkssd_gpu_dev.so python3 profiles/dist_phases.py 10000 2>&1 | grep -v amdgpu.ids >> $O
cat $O
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
