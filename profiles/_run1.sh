cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q -k "low_complexity" 2>&1 | grep -E "^E|assert|Error" | head -12
