#!/bin/bash
# round 6, THE closing pass (one per round: VERDICT r05 item 7) on the committed tree: GPU suite, the driver's command, its kernel trace,
# the PMC passes (scan, rows, tokeniser), the line again (traffic stamped), one GPU as a rank of eight, configs[2] / [3] / [4] for the
# record, the fuzzers once at their long setting
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06Z; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -4 $o/tests_gpu.log
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $o/prof.json 2> $o/prof.err
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); cp "$f" $o/kernel_stats.csv; rm -rf $o/prof
timeout 1500 python3 profiles/pmc_refresh.py r06Z > $o/pmc_refresh.txt 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_traffic.json $o/pmc_traffic.json 2>/dev/null; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
rm -rf gpurun_out/pmc_r06Z_FETCH_SIZE gpurun_out/pmc_r06Z_WRITE_SIZE
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $o/bench_after_refresh.json 2> $o/bench_after_refresh.err
timeout 900 python3 bench.py --emulate-world 8 --rank 3 --steps 20 --warmup 5 --cpu-sample 0 > $o/bench_emu8_rank3.json 2> $o/bench_emu8_rank3.err
timeout 1500 python3 bench.py --genomes 10000 --clades 500 --cpu-sample 0 --steps 10 --warmup 2 --no-tok-leg > $o/bench_config2.json 2> $o/bench_config2.err; echo "config2 rc=$?"
timeout 1500 python3 bench.py --workload fastq --steps 10 --warmup 3 > $o/bench_fastq.json 2> $o/bench_fastq.err; echo "fastq rc=$?"
timeout 1500 python3 bench.py --workload mammal --genomes 8 --steps 10 --warmup 2 > $o/bench_mammal8.json 2> $o/bench_mammal8.err; echo "mammal rc=$?"
timeout 900 python3 profiles/fuzz_cli.py 300 20000 > $o/fuzz_cli.txt 2>&1; tail -1 $o/fuzz_cli.txt
timeout 600 python3 profiles/fuzz_sketch.py 300 30000 > $o/fuzz_sketch.txt 2>&1; tail -1 $o/fuzz_sketch.txt
KSSD_MASK_SUMMARY=1 timeout 600 python3 profiles/fuzz_sketch.py 300 40000 > $o/fuzz_sketch_summary.txt 2>&1; tail -1 $o/fuzz_sketch_summary.txt
timeout 600 python3 profiles/fuzz_dist_device.py 3000 > $o/fuzz_dist_device.txt 2>&1; tail -1 $o/fuzz_dist_device.txt
python3 - <<PY
import json, csv
def last(f):
    return json.loads(open("$o/" + f).read().strip().splitlines()[-1])
j = last("bench.json")
print("bench ms_per_step %.4f value %.0f" % (j["ms_per_step"], j["value"]), "scan", j["kernels"]["sketch_scan_ms"], j["kernels"]["sketch_scan_spread"], "frac %.4f" % j["roofline"]["frac"], "moved", j["roofline"]["bytes_moved"], "dist", j["dist_ms_per_step"], j["roofline_dist"]["frac"])
t = j["roofline_tok"]; print("tok", t["kernel"], t["kernel_ms"], "frac %.4f" % t["frac"], "call", t["call_ms"])
e = j["end_to_end"]
print("e2e", e["value"], e["seconds_runs"]); print("gz", e["gzip"]["value"], e["gzip"]["seconds_runs"]); print("search", e["search"]["value"], e["search"]["seconds_runs"], e["search"].get("speedup_vs_reference"))
print("search4k", {k: v for k, v in e.get("search_4096", {}).items() if k not in ("what", "stages")}); print("allpairs", e["allpairs"]["value"], e["allpairs"]["seconds_runs"])
for k in e:
    if k.startswith("sketch_"): print(k, e[k]["value"], e[k]["seconds"], e[k].get("per_job"))
print("cpu", j["cpu_baseline"]["value"], j.get("cpu_baseline_gz", {}).get("value"), j["cpu_baseline_dist"]["value"])
j = last("bench_after_refresh.json"); print("after refresh", j["ms_per_step"], j["roofline"]["traffic"], j["roofline_dist"]["traffic"], j["roofline_tok"]["traffic"], (j["roofline"]["traffic_source"] or "")[:60])
j = last("bench_emu8_rank3.json"); print("emu8", j["emulated"]["per_rank_ms"], j["emulated"]["index_ms"], j["emulated"]["rows_ms"], j["emulated"]["partition_query"])
j = last("bench_config2.json"); print("config2 ms_per_step", j["ms_per_step"], j["value"], j["dist_ms_per_step"], j.get("dist_halves_ms"), j["config"]["workload"][:40])
j = last("bench_fastq.json"); print("fastq ms_per_step", j["ms_per_step"], j["value"], j["n1"]["phase_ms"])
j = last("bench_mammal8.json"); print("mammal8 ms_per_step", j["ms_per_step"], j["value"], j["unit"])
for r in csv.DictReader(open("$o/kernel_stats.csv")):
    n = r["Name"]
    if "at::native" in n or "rocclr" in n: continue
    print("  %-60s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
tail -14 $o/pmc_refresh.txt
