cd $GRAFT_REPO_ROOT
KSSD_DEV_TRACE=1 timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q -s -k "low_complexity" 2>&1 | grep -E "kssd_gpu\]|passed|failed" | cut -c1-200
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
