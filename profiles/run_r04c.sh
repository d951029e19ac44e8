#!/bin/bash
# r04c: after the source split: whole gpu suite, rows-kernel phases, the mmap / hipHostRegister probe, default bench with the all-pairs leg
tag=${1:-r04c}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -6 gpurun_out/$tag/tests_gpu.log
KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DISTTIME=1 timeout 600 python3 profiles/dist_phases.py > gpurun_out/$tag/dist_phases.txt 2>&1
tail -12 gpurun_out/$tag/dist_phases.txt
timeout 300 profiles/mmap_probe 1024 > gpurun_out/$tag/mmap_probe.txt 2>&1
cat gpurun_out/$tag/mmap_probe.txt
timeout 900 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -3 gpurun_out/$tag/bench.err
python3 - <<PY
import json
try:
    j = json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
    print("value", j["value"], "ms_per_step %.4f" % j["ms_per_step"], "frac %.4f" % j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline_dist"]["traffic"])
    e = j.get("end_to_end", {})
    print("e2e", e.get("value"), e.get("seconds_runs"), json.dumps(e.get("stages"))[:700])
    print("allpairs", json.dumps(e.get("allpairs"))[:1500])
    print("search", json.dumps(e.get("search"))[:500])
except Exception as ex:
    print("bench failed", ex)
PY
