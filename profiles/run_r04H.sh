#!/bin/bash
# r04H: context creation after the exact-table placement with one move (20 - 60 ms of multiplier lottery before) and events made on demand:
# sketch + dist tests, then the end-to-end legs of the default bench line with their stage times
tag=${1:-r04H}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_dist.py tests/test_gpu_cli.py -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -3 gpurun_out/$tag/tests_gpu.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python3 - <<PY
import json
j = json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
e = j["end_to_end"]
print("value", j["value"], "ms_per_step %.4f" % j["ms_per_step"])
print("e2e", e["value"], e["seconds_runs"], "stages", json.dumps(e["stages"]))
print("allpairs", json.dumps(e.get("allpairs"))[:600])
print("search", json.dumps(e.get("search"))[:600])
PY
