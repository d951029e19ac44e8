#!/bin/bash
# round 6: the closing pass's core once more on the round's FINAL kernel sources (r06ZZ: after r06Z, the tokeniser's te_compose for both header
# states at once + its look-back timeout, the exact stage's reordered chain): GPU suite, the driver's command, its kernel trace, the PMC
# passes (the stamp bench.py checks), the line again with the traffic stamped, the fuzzers that touch what changed
# (r06ZZ2: once more after the tokeniser learnt to write the mask's summary words -- `kssd dist` scans with them)
# (r06ZZ3: the round's last tree -- parts_skew in the per-genome kernel's parts, the tutorial in full among the GPU tests; + configs[4], whose records are sorted in parts)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06ZZ3; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -3 $o/tests_gpu.log
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $o/prof.json 2> $o/prof.err
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); cp "$f" $o/kernel_stats.csv; rm -rf $o/prof
timeout 1500 python3 profiles/pmc_refresh.py r06ZZ3 > $o/pmc_refresh.txt 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_traffic.json $o/pmc_traffic.json 2>/dev/null; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
rm -rf gpurun_out/pmc_r06ZZ3_FETCH_SIZE gpurun_out/pmc_r06ZZ3_WRITE_SIZE
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $o/bench_after_refresh.json 2> $o/bench_after_refresh.err
timeout 900 python3 bench.py --workload mammal --genomes 8 --steps 10 --warmup 2 > $o/bench_mammal8.json 2> $o/bench_mammal8.err; echo "mammal rc=$?"
timeout 600 python3 profiles/fuzz_cli.py 150 60000 > $o/fuzz_cli.txt 2>&1; tail -1 $o/fuzz_cli.txt
timeout 400 python3 profiles/fuzz_fastq.py 400 > $o/fuzz_fastq.txt 2>&1; tail -1 $o/fuzz_fastq.txt
KSSD_MASK_SUMMARY=1 timeout 400 python3 profiles/fuzz_fastq.py 400 9000 > $o/fuzz_fastq_summary.txt 2>&1; tail -1 $o/fuzz_fastq_summary.txt
python3 - <<PY
import json, csv
def last(f):
    return json.loads(open("$o/" + f).read().strip().splitlines()[-1])
j = last("bench.json")
print("bench ms_per_step %.4f value %.0f" % (j["ms_per_step"], j["value"]), "scan", j["kernels"]["sketch_scan_ms"], "frac %.4f" % j["roofline"]["frac"], "dist", j["dist_ms_per_step"], j["roofline_dist"]["frac"])
t = j["roofline_tok"]; print("tok", t["kernel"], t["kernel_ms"], "frac %.4f" % t["frac"], "call", t["call_ms"])
e = j["end_to_end"]
print("e2e", e["value"], e["seconds_runs"], "gz", e["gzip"]["value"], "search", e["search"]["value"], e["search"].get("speedup_vs_reference"), "4k", e["search_4096"]["value"], e["search_4096"]["speedup_vs_reference"], "allpairs", e["allpairs"]["value"])
for k in e:
    if k.startswith("sketch_"): print(k, e[k]["value"], e[k]["seconds"], e[k].get("per_job"))
print("cpu", j["cpu_baseline"]["value"], j.get("cpu_baseline_gz", {}).get("value"), j["cpu_baseline_dist"]["value"])
j = last("bench_after_refresh.json"); print("after refresh", j["ms_per_step"], j["roofline"]["traffic"], j["roofline_dist"]["traffic"], j["roofline_tok"]["traffic"], (j["roofline"]["traffic_source"] or "")[:60])
for r in csv.DictReader(open("$o/kernel_stats.csv")):
    n = r["Name"]
    if "at::native" in n or "rocclr" in n: continue
    print("  %-60s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
python3 -c "import json; j=json.loads(open('$o/bench_mammal8.json').read().strip().splitlines()[-1]); print('mammal8', j['ms_per_step'], j['value'], j['unit'])"
tail -12 $o/pmc_refresh.txt
