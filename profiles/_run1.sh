cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q -k "round_trip" 2>&1 | tail -12
