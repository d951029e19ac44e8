cd $GRAFT_REPO_ROOT
o=gpurun_out/r05t; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -k "eight_parts" 2>&1 | tail -2
python3 profiles/rows_x_probe.py 2>&1 | grep -v amdgpu.ids | tee $o/rows_x_probe.txt
