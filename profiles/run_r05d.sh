#!/bin/bash
# round 5: look-back with relaxed atomics, the transposing metrics kernel in tiles of 256 queries x 32 (A) / 16 (B) references
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05d; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_sketch.py tests/test_allpairs_flow.py -m gpu -q -x > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -4 $o/tests_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
for v in A B; do
  if [ $v = B ]; then export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_alt.so; else unset KSSD_GPU_LIB; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$v -- python3 bench.py --emulate-world 8 --rank 3 --partition own --steps 20 --warmup 5 --cpu-sample 0 > $o/prof_$v.json 2> $o/prof_$v.err; echo "prof $v rc=$?"
  f=$(find $o/prof_$v -name '*kernel_stats.csv' | head -1); cp "$f" $o/emu8_${v}_kernel_stats.csv; rm -rf $o/prof_$v
done
unset KSSD_GPU_LIB
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_d -- python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > $o/prof_d.json 2> $o/prof_d.err
f=$(find $o/prof_d -name '*kernel_stats.csv' | head -1); cp "$f" $o/default_kernel_stats.csv; rm -rf $o/prof_d
python3 - <<PY
import json, csv
j = json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print("bench ms_per_step %.4f" % j["ms_per_step"], "scan %.4f" % j["kernels"]["sketch_scan_ms"], j["kernels"]["sketch_scan_spread"], "frac %.4f" % j["roofline"]["frac"], j.get("dist_halves_ms"))
for v in ("A", "B"):
    j = json.loads(open("$o/prof_%s.json" % v).read().strip().splitlines()[-1])
    print(v, j["emulated"]["per_rank_ms"], j["emulated"]["index_ms"], j["emulated"]["rows_ms"])
for f in ("emu8_A", "emu8_B", "default"):
    print(f)
    for r in csv.DictReader(open("$o/%s_kernel_stats.csv" % f)):
        n = r["Name"]
        if "at::native" in n or "rocclr" in n: continue
        print("  %-60s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
