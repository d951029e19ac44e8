#!/bin/bash
# round 5, first contact: the GPU suite, the default line, one GPU as one rank of eight (both partitions), and the one-command
# flow as eight ranks on one device
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05a; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -5 $o/tests_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --emulate-world 8 --rank 3 --steps 20 --warmup 5 --cpu-sample 0 > $o/bench_emu8.json 2> $o/bench_emu8.err; echo "emu rc=$?"
tail -3 $o/bench_emu8.err
python3 - <<PY
import json
for f in ("bench.json", "bench_emu8.json"):
    try:
        j = json.loads(open("$o/" + f).read().strip().splitlines()[-1])
        print(f, "ms_per_step %.4f" % j["ms_per_step"], "scan %.4f" % j["kernels"]["sketch_scan_ms"], "frac %.4f" % j["roofline"]["frac"], j.get("emulated"), j.get("dist_halves_ms"))
    except Exception as e:
        print(f, "no line:", e)
PY
