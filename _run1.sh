#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "by_position or byread" 2>&1 | tail -30
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
