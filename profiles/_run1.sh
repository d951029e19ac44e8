cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --cpu-sample 0 2>/dev/null | cut -c1-1200
