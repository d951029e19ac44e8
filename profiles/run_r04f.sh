#!/bin/bash
# r04f: rows kernel with four lanes per list and 16-byte posting loads; one-device all-pairs without RCCL; dist + flow tests, phases, bench
tag=${1:-r04f}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_allpairs_flow.py tests/test_gpu_configs.py tests/test_gpu_cli.py tests/test_wide_tuples.py tests/test_bench_launch.py -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -6 gpurun_out/$tag/tests_gpu.log
KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DISTTIME=1 timeout 600 python3 profiles/dist_phases.py > gpurun_out/$tag/dist_phases.txt 2>&1
tail -9 gpurun_out/$tag/dist_phases.txt
timeout 900 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -3 gpurun_out/$tag/bench.err
python3 - <<PY
import json
for f in ("bench",):
    try:
        j = json.loads(open("gpurun_out/$tag/%s.json" % f).read().strip().splitlines()[-1])
        e = j.get("end_to_end", {})
        print(f, "ms_per_step %.4f" % j["ms_per_step"], "rows ms %.4f" % j["roofline_dist"]["launch_ms"], "frac", j["roofline_dist"]["frac"], "dist ms/step %.4f" % j["dist_ms_per_step"], j["kernels"])
        print("   e2e", e.get("value"), e.get("seconds_runs"))
        print("   allpairs", json.dumps(e.get("allpairs"))[:900])
    except Exception as ex:
        print(f, "failed", ex)
PY
