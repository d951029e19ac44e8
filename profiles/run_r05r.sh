cd $GRAFT_REPO_ROOT
o=gpurun_out/r05r; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -15
python3 profiles/rows_x_probe.py 2>&1 | grep -v amdgpu.ids | tee $o/rows_x_probe.txt
