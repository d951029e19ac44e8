#!/bin/bash
# round 5: read-ahead under the runtime's start, the own gzip decoder, the 4 096 x 4 096 search leg: GPU suite (the CLI tests) and the default line
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05i; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q -x > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -4 $o/tests_gpu.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
tail -5 $o/bench.err
python3 - <<PY
import json
j = json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print("ms_per_step %.4f" % j["ms_per_step"], "frac %.4f" % j["roofline"]["frac"])
e = j["end_to_end"]
print("e2e", e["value"], e["seconds_runs"], {k: v for k, v in e["stages"].items() if k.startswith("s_")})
print("gz", e["gzip"]["value"], e["gzip"]["seconds_runs"], {k: v for k, v in e["gzip"]["stages"].items() if k.startswith("s_")})
print("search", e["search"]["value"], e["search"]["seconds_runs"], e["search"].get("speedup_vs_reference"))
print("search4k", {k: v for k, v in e.get("search_4096", {}).items() if k not in ("what", "stages")})
print("allpairs", e["allpairs"]["value"], e["allpairs"]["seconds_runs"])
print("cpu", j["cpu_baseline"]["value"], j.get("cpu_baseline_gz", {}).get("value"), j["cpu_baseline_dist"]["value"])
PY
