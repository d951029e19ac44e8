#!/bin/bash
# r04a: capped index build + cooperative rows kernel + scan without the reset launch: gpu suite, default bench, kernel trace, PMC refresh
tag=${1:-r04a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -15 gpurun_out/$tag/tests_gpu.log
timeout 900 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -3 gpurun_out/$tag/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python3 bench.py --steps 20 --warmup 2 --cpu-sample 0 > gpurun_out/$tag/prof.log 2>&1
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/$tag/kernel_stats.csv
rm -rf gpurun_out/$tag/prof
timeout 900 python3 profiles/pmc_refresh.py $tag > gpurun_out/$tag/pmc_refresh.txt 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/$tag/ 2>/dev/null
rm -rf gpurun_out/pmc_${tag}_*/
python3 - <<PY
import json
try:
    j = json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
    print("value", j["value"], "ms_per_step %.4f" % j["ms_per_step"], "frac %.4f" % j["roofline"]["frac"], "dist ms/step %.4f" % j["dist_ms_per_step"],
          "pairs/s dist %.3g" % j["pairs_per_s_dist"], "rows frac %.4f" % j["roofline_dist"]["frac"], "rows ms %.4f" % j["roofline_dist"]["launch_ms"])
    print(json.dumps(j.get("kernels")))
    print("e2e", json.dumps(j.get("end_to_end"))[:600])
except Exception as e:
    print("bench failed", e)
PY
head -20 gpurun_out/$tag/kernel_stats.csv | cut -c1-200
cat gpurun_out/$tag/pmc_refresh.txt | cut -c1-200
