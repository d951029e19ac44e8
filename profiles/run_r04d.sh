#!/bin/bash
# r04d: inputs mapped from the page cache (no page-locked buffers, no reader copy), runtime start-up under the first mappings;
# eight lists in flight in the rows kernel; k12 modes.  CLI + tokeniser + wide-tuple tests, rows phases, default bench, 9 984-file leg
tag=${1:-r04d}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -6 gpurun_out/$tag/tests_gpu.log
KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DISTTIME=1 timeout 600 python3 profiles/dist_phases.py > gpurun_out/$tag/dist_phases.txt 2>&1
tail -9 gpurun_out/$tag/dist_phases.txt
timeout 900 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -3 gpurun_out/$tag/bench.err
KSSD_NO_MMAP=1 timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/$tag/bench_nommap.json 2> gpurun_out/$tag/bench_nommap.err
timeout 1500 python bench.py --steps 5 --warmup 1 --e2e-files 9984 > gpurun_out/$tag/bench_e2e10000.json 2> gpurun_out/$tag/bench_e2e10000.err
python3 - <<PY
import json
for f in ("bench", "bench_nommap", "bench_e2e10000"):
    try:
        j = json.loads(open("gpurun_out/$tag/%s.json" % f).read().strip().splitlines()[-1])
        e = j.get("end_to_end", {})
        print(f, "ms_per_step %.4f" % j["ms_per_step"], "rows ms %.4f" % j["roofline_dist"]["launch_ms"], "dist ms/step %.4f" % j["dist_ms_per_step"])
        print("   e2e", e.get("value"), e.get("seconds_runs"), json.dumps(e.get("stages"))[:900])
        print("   allpairs", (e.get("allpairs") or {}).get("seconds_runs"), (e.get("allpairs") or {}).get("two_commands_seconds"), "search", (e.get("search") or {}).get("seconds_runs"))
        print("   gz", (e.get("gzip") or {}).get("value"), "ref", (j.get("cpu_baseline") or {}).get("value"))
    except Exception as ex:
        print(f, "failed", ex)
PY
