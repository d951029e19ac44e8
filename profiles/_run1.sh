cd $GRAFT_REPO_ROOT
timeout 600 python profiles/pcie_rate.py 2>&1 | grep -E "tokeniser|kssd_gpu_sketch_batch|Error|error" | head
