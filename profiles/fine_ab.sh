#!/bin/bash
# the scan with the candidates' validity settled in the Bloom rounds (FINE) against the coarse bit (KSSD_SCAN_COARSE_VALIDITY=1): configs[3] kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in fine coarse fine coarse; do
  if [ $v = coarse ]; then export KSSD_SCAN_COARSE_VALIDITY=1; else unset KSSD_SCAN_COARSE_VALIDITY; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fa_$v -- python3 bench.py --workload fastq --steps 10 --warmup 2 --cpu-sample 0 --parity-reads 0 > gpurun_out/fa_$v.json 2>/dev/null
  f=$(find gpurun_out/fa_$v -name '*kernel_stats.csv' | head -1)
  echo "$v: $(tail -1 gpurun_out/fa_$v.json | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"], j["unit"])')"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "at::native" in n or "rocclr" in n or "tok_" in n: continue
    if int(r["Calls"]) < 5: continue
    print("  %-60s calls %5s avg %9.1f us min %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
  rm -rf gpurun_out/fa_$v gpurun_out/fa_$v.json
done
