#!/bin/bash
# r04L: kernel trace of configs[2] (10 000 genomes on one GPU: the index of ten thousand sketches = what every rank builds at N = 8)
tag=${1:-r04L}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python3 bench.py --genomes 10000 --clades 500 --cpu-sample 0 --steps 10 --warmup 2 > gpurun_out/$tag/bench_config3.json 2> gpurun_out/$tag/err.log
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native" "$f" > gpurun_out/$tag/kernel_stats_config3.csv
rm -rf gpurun_out/$tag/prof
python3 - <<PY
import csv, json
for r in csv.DictReader(open("gpurun_out/$tag/kernel_stats_config3.csv")):
    if float(r["AverageNs"]) > 3000 and "rocclr" not in r["Name"]:
        print("   %-44s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
j = json.loads(open("gpurun_out/$tag/bench_config3.json").read().strip().splitlines()[-1])
print("ms_per_step %.4f dist %.4f" % (j["ms_per_step"], j["dist_ms_per_step"]))
PY
