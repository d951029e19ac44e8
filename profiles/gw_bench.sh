#!/bin/bash
# usage (GPU box): profiles/gw_bench.sh <tag>   windows per table read (KSSD_GW = 5 product, 6, 7, 8 builds) inside one call
tag=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for rep in 1 2; do
  for v in 5 6 7 8; do
    if [ $v = 5 ]; then unset KSSD_GPU_LIB; else export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_gw$v.so; fi
    timeout 600 python bench.py --cpu-sample 0 --steps 20 > gpurun_out/$tag/bench_$v$rep.json 2> gpurun_out/$tag/bench_$v$rep.err
    python3 - <<PY
import json
try:
    j=json.loads(open("gpurun_out/$tag/bench_$v$rep.json").read().strip().splitlines()[-1])
    print("GW=$v rep $rep", "ms_per_step %.4f" % j["ms_per_step"], "scan_ms %.4f" % j["kernels"]["sketch_scan_ms"], "frac %.4f" % j["roofline"]["frac"], "past stage 1 %.5f" % j["kernels"]["scan_positions_past_stage1"])
except Exception as e:
    print("GW=$v failed", e, open("gpurun_out/$tag/bench_$v$rep.err").read()[-400:])
PY
  done
done
