#!/bin/bash
# round 5: the global-memory sort of keys that do not spread is a bitonic network of our own (no rocPRIM in the library): the sketch tests;
# one GPU as rank 0 and rank 7 of eight
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05n; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q -x > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -4 $o/tests_gpu.log
for r in 0 7; do
  timeout 900 python3 bench.py --emulate-world 8 --rank $r --steps 20 --warmup 5 --cpu-sample 0 > $o/emu8_rank$r.json 2> $o/emu8_rank$r.err; echo "emu rank $r rc=$?"
done
timeout 900 python3 bench.py --emulate-world 2 --rank 1 --steps 20 --warmup 5 --cpu-sample 0 > $o/emu2_rank1.json 2> $o/emu2_rank1.err
timeout 900 python3 bench.py --emulate-world 4 --rank 2 --steps 20 --warmup 5 --cpu-sample 0 > $o/emu4_rank2.json 2> $o/emu4_rank2.err
python3 - <<PY
import json
for f in ("emu8_rank0", "emu8_rank7", "emu2_rank1", "emu4_rank2"):
    j = json.loads(open("$o/%s.json" % f).read().strip().splitlines()[-1])
    e = j["emulated"]
    print(f, "own: %.4f ms (index %.4f rows %.4f)" % (e["per_rank_ms"], e["index_ms"], e["rows_ms"]), "query:", e["partition_query"], "exchange copies us", e["exchange_copy_us"])
PY
