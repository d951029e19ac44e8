#!/bin/bash
# r04y: what the partition kernel's returning atomics cost (development build, KSSD_DEV_NO_CURSOR_ATOMIC: plain loads, wrong results)
tag=${1:-r04y}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for v in atomics no_atomics atomics no_atomics; do
  if [ $v = no_atomics ]; then export KSSD_DEV_NO_CURSOR_ATOMIC=1; else unset KSSD_DEV_NO_CURSOR_ATOMIC; fi
  KSSD_GPU_LIB=$GRAFT_REPO_ROOT/public_kssd_amd/libkssd_gpu_dev.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_$v -- python3 profiles/dist_phases.py > gpurun_out/$tag/out_$v.log 2>&1
  f=$(find gpurun_out/$tag/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "idx_" in r["Name"] or "dist_rows" in r["Name"]:
        print("   %-40s calls %s avg %.2f us min %.2f max %.2f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf gpurun_out/$tag/prof_$v
done 2>&1 | tee gpurun_out/$tag/ab.txt
