#!/bin/bash
# round 5: configs[3] (100 M reads) after the cheap fixes behind the scan, with a kernel trace; the sketch tests first
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05o; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_configs.py tests/test_gpu_cli.py tests/test_gpu_tokenise.py -m gpu -q -x > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -3 $o/tests_gpu.log
timeout 1500 python bench.py --workload fastq --steps 20 --warmup 3 > $o/bench_fastq.json 2> $o/bench_fastq.err; echo "fastq rc=$?"
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --workload fastq --steps 20 --warmup 3 --parity-reads 0 > $o/prof_fastq.json 2> $o/prof_fastq.err
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); cp "$f" $o/fastq_kernel_stats.csv; rm -rf $o/prof
python3 - <<PY
import json, csv
j = json.loads(open("$o/bench_fastq.json").read().strip().splitlines()[-1])
print("fastq ms_per_step", j["ms_per_step"], j["value"], j["unit"], j.get("roofline", {}).get("frac"))
for r in csv.DictReader(open("$o/fastq_kernel_stats.csv")):
    n = r["Name"]
    if "at::native" in n or "rocclr" in n: continue
    print("  %-60s calls %5s avg %9.1f us" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
