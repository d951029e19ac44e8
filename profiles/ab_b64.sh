#!/bin/bash
# usage (GPU box): profiles/ab_b64.sh <tag>   -- A/B of the stage-1 table read: ds_read_u8 (product) against an aligned
# ds_read_b64 + byte select, same random batch, time + LDS / VALU counters of the scan kernel
tag=${1:-ab}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in ${AB_VARIANTS:-scanbench scanbench_b64}; do
  echo "== $v"; timeout 120 profiles/$v 400 5000000 10 | grep -i "variant\|ms"
  out=gpurun_out/pmc_${tag}_$v
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-include-regex "sketch_scan" --output-format csv -d $out -- profiles/$v 400 5000000 3 > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = set()
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print("   per dispatch:", " ".join("%s=%.4g" % (c, v / len(n)) for c, v in sorted(acc.items())))
PY
  rm -rf $out
done
