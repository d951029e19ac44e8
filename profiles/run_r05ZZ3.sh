#!/bin/bash
# round 5, third closing run: the host side changed after r05ZZ (the restart with the passive wait policy works under the GPU hosts' own
# preloaded guard again; job texts in registered ordinary memory) -- the kernel sources did not, the counters of r05ZZ stand.
# GPU suite + the driver's command twice.
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05ZZ3; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q > $o/tests_gpu.log 2>&1; tail -2 $o/tests_gpu.log
for i in 1 2; do
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench$i.json 2> $o/bench$i.err
  tail -1 $o/bench$i.json | python3 -c '
import json, sys
j = json.loads(sys.stdin.read()); e = j["end_to_end"]
print(j["ms_per_step"], j["value"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"])
print("  e2e", e["value"], e["seconds_runs"], "gz", e["gzip"]["value"], e["gzip"]["seconds_runs"])
print("  search", e["search"]["seconds_runs"], e["search"].get("speedup_vs_reference"), "s4k", e["search_4096"]["seconds_runs"], e["search_4096"]["speedup_vs_reference"], "allpairs", e["allpairs"]["seconds_runs"])
print("  cpu", j["cpu_baseline"]["value"], j["cpu_baseline_gz"]["value"])'
done
