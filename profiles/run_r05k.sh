#!/bin/bash
# round 5: bucket rooms that are no powers of two, the two-level partition of large indexes: GPU suite, index build at both sizes with a trace
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05k; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q -x > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -4 $o/tests_gpu.log
python3 profiles/index_sizes_probe.py 2>&1 | grep -v amdgpu.ids | tee $o/index_sizes.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 profiles/index_sizes_probe.py > /dev/null 2>&1
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); cp "$f" $o/index_kernel_stats.csv; rm -rf $o/prof
grep "idx_" $o/index_kernel_stats.csv | cut -c1-60,100-400 | head
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
j = json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print("bench ms_per_step %.4f" % j["ms_per_step"], "scan %.4f" % j["kernels"]["sketch_scan_ms"], "frac %.4f" % j["roofline"]["frac"], j.get("dist_halves_ms"))
PY
