#!/bin/bash
# round 3: parity suite with the new device paths (-Q, -A, in-process exchange), the default line on the kernels as they stand
# (blocks taken in turn, no queue, alignment B on every lane), rocprofv3 kernel trace of the same command
tag=${1:-r03g}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${tag}_pytest.log
tail -8 gpurun_out/${tag}_pytest.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value", "ms_per_step", "kernels", "roofline", "pipelined", "cpu_baseline", "cpu_baseline_dist", "cpu_baseline_gz"):
    print(k, json.dumps(d.get(k))[:500])
e = d.get("end_to_end", {})
print("e2e", e.get("value"), e.get("seconds_runs"), json.dumps(e.get("stages"))[:700])
print("gz", json.dumps(e.get("gzip"))[:700])
print("search", json.dumps(e.get("search"))[:600])
PY
tail -5 gpurun_out/${tag}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/${tag}_prof.log 2>&1
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
cut -d, -f1-8 gpurun_out/${tag}_kernel_stats.csv | head -30
