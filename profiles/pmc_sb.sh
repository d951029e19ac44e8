#!/bin/bash
# usage (GPU box): [KSSD_DEV_SCAN=m KSSD_DEV_ABLATE=a] profiles/pmc_sb.sh <tag> <counters...>
# one rocprofv3 --pmc pass over profiles/scanbench (scan kernel only); prints per-dispatch averages
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcsb_$tag
timeout 200 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "sketch_scan|calib_read" --output-format csv -d $out -- profiles/scanbench ${SB_ARGS:-400 5000000 3} > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(sys.argv[2], k, "dispatches", len(n[k]), " ".join("%s=%.4g" % (c, v / len(n[k])) for c, v in sorted(acc[k].items())))
PY
rm -rf $out
