#!/bin/bash
# r04z: the round's closing run on the sources as committed: gpu suite, smoke, the driver's bench command, kernel trace, counter refresh
tag=${1:-r04z}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -5 gpurun_out/$tag/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -2 gpurun_out/$tag/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/prof.log 2>&1
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/$tag/kernel_stats.csv
rm -rf gpurun_out/$tag/prof
timeout 900 python3 profiles/pmc_refresh.py $tag > gpurun_out/$tag/pmc_refresh.txt 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/$tag/ 2>/dev/null
rm -rf gpurun_out/pmc_${tag}_*/
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench_after_refresh.json 2> /dev/null
python3 - <<PY
import json
for f in ("bench", "bench_after_refresh"):
    try:
        j = json.loads(open("gpurun_out/$tag/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value", j["value"], "ms_per_step %.4f" % j["ms_per_step"], "frac %.4f" % j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], "| dist", "%.4f ms" % j["dist_ms_per_step"],
              "%.3g pairs/s" % j["pairs_per_s_dist"], "rows frac %.4f" % j["roofline_dist"]["frac"], "traffic", j["roofline_dist"]["traffic"])
        e = j.get("end_to_end") or {}
        if e:
            print("   e2e", e.get("value"), e.get("seconds_runs"), "| allpairs", (e.get("allpairs") or {}).get("seconds_runs"), (e.get("allpairs") or {}).get("two_commands_seconds"),
                  "| search", (e.get("search") or {}).get("value"), "| gz", (e.get("gzip") or {}).get("value"), "| ref", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline_dist") or {}).get("value"))
            print("   pipelined", json.dumps(j.get("pipelined"))[:300])
    except Exception as ex:
        print(f, "failed", ex)
PY
cut -d, -f1-4 gpurun_out/$tag/kernel_stats.csv | cut -c1-100 | head -10
cat gpurun_out/$tag/pmc_refresh.txt | cut -c1-160
