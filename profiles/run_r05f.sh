#!/bin/bash
# round 5: the other devices' rows walked flat behind the negative filter (A) against one workgroup per row (B: KSSD_DIST_NO_FLAT=1),
# one transposing launch per step; GPU suite first
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05f; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q -x > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -4 $o/tests_gpu.log
for v in A B; do
  if [ $v = B ]; then export KSSD_DIST_NO_FLAT=1; else unset KSSD_DIST_NO_FLAT; fi
  timeout 900 python3 bench.py --emulate-world 8 --rank 3 --steps 20 --warmup 5 --cpu-sample 0 > $o/emu8_$v.json 2> $o/emu8_$v.err; echo "emu $v rc=$?"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$v -- python3 bench.py --emulate-world 8 --rank 3 --partition own --steps 20 --warmup 5 --cpu-sample 0 > $o/prof_$v.json 2> $o/prof_$v.err; echo "prof $v rc=$?"
  f=$(find $o/prof_$v -name '*kernel_stats.csv' | head -1); cp "$f" $o/emu8_${v}_kernel_stats.csv; rm -rf $o/prof_$v
done
unset KSSD_DIST_NO_FLAT
python3 - <<PY
import json, csv
for v in ("A", "B"):
    j = json.loads(open("$o/emu8_%s.json" % v).read().strip().splitlines()[-1])
    print(v, j["emulated"])
for f in ("emu8_A", "emu8_B"):
    print(f)
    for r in csv.DictReader(open("$o/%s_kernel_stats.csv" % f)):
        n = r["Name"]
        if "at::native" in n or "rocclr" in n: continue
        print("  %-60s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
