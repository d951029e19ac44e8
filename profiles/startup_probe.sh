#!/bin/bash
cd $GRAFT_REPO_ROOT
hipcc -O2 -o /tmp/startup_probe profiles/src/startup_probe.cpp -ldl || exit 1
for i in 1 2 3; do echo "--- run $i"; /tmp/startup_probe public_kssd_amd/libkssd_gpu.so; done
echo "--- HSA_ENABLE_INTERRUPT=0"; HSA_ENABLE_INTERRUPT=0 /tmp/startup_probe public_kssd_amd/libkssd_gpu.so
echo "--- HIP_VISIBLE_DEVICES=0 ROCR_VISIBLE_DEVICES=0"; HIP_VISIBLE_DEVICES=0 ROCR_VISIBLE_DEVICES=0 /tmp/startup_probe public_kssd_amd/libkssd_gpu.so
echo "--- GPU_MAX_HW_QUEUES=1"; GPU_MAX_HW_QUEUES=1 /tmp/startup_probe public_kssd_amd/libkssd_gpu.so
echo "--- strace -c"; strace -f -c -o /tmp/st.txt /tmp/startup_probe public_kssd_amd/libkssd_gpu.so > /dev/null 2>&1; head -15 /tmp/st.txt
nproc; free -g | head -2
