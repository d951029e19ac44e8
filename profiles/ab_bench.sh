#!/bin/bash
# usage (GPU box): profiles/ab_bench.sh <tag> [bench args]     A/B of two builds of the library inside one call:
#   A = public_kssd_amd/libkssd_gpu.so, B = profiles/libkssd_gpu_alt.so (built in the dev container with other -D switches),
# alternating, three runs each; prints ms per step and the scan's own time
tag=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = B ]; then export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_alt.so; else unset KSSD_GPU_LIB; fi
    timeout 600 python bench.py --cpu-sample 0 --steps 20 "$@" > gpurun_out/$tag/bench_$v$rep.json 2> gpurun_out/$tag/bench_$v$rep.err
    python3 - <<PY
import json
j=json.loads(open("gpurun_out/$tag/bench_$v$rep.json").read().strip().splitlines()[-1])
print("$v$rep", "ms_per_step %.4f" % j["ms_per_step"], "scan_ms %.4f" % j["kernels"]["sketch_scan_ms"], "frac %.4f" % j["roofline"]["frac"])
PY
  done
done
