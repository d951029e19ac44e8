#!/bin/bash
# round 5: the per-genome kernel writes the CSR itself (five launches per step): GPU suite, default line, kernel traces of the
# default step and of one GPU as one rank of eight
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05c; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q -x > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -8 $o/tests_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
for w in default emu8; do
  if [ $w = emu8 ]; then args="--emulate-world 8 --rank 3 --partition own"; else args=""; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$w -- python3 bench.py $args --steps 20 --warmup 5 --cpu-sample 0 > $o/prof_$w.json 2> $o/prof_$w.err; echo "prof $w rc=$?"
  f=$(find $o/prof_$w -name '*kernel_stats.csv' | head -1); cp "$f" $o/${w}_kernel_stats.csv; rm -rf $o/prof_$w
  head -12 $o/${w}_kernel_stats.csv | cut -c1-150
done
python3 - <<PY
import json
for f in ("bench.json",):
    j = json.loads(open("$o/" + f).read().strip().splitlines()[-1])
    print(f, "ms_per_step %.4f" % j["ms_per_step"], "scan %.4f" % j["kernels"]["sketch_scan_ms"], j["kernels"]["sketch_scan_spread"], "frac %.4f" % j["roofline"]["frac"], j.get("dist_halves_ms"))
PY
