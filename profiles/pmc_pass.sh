#!/bin/bash
# usage: profiles/pmc_pass.sh <tag> <kernel-regex> <counters...>   (run on the GPU box through gpurun)
# One rocprofv3 --pmc pass (counters only, kernel-trace only) of a short bench run; prints per-kernel sums.
tag=$1; shift; rx=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag
timeout 280 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$rx" --output-format csv -d $out -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 > $out.log 2>&1
echo "rc=$? ($tag)"
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(n[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-28s per-dispatch %.4g" % (c, v / len(n[k])))
PY
