#!/bin/bash
# r04P: kernel trace of configs[3] (100 M reads of 150 bp as one read set) and configs[4] (8 records of 3 Gb) after the round's changes
tag=${1:-r04P}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for w in fastq mammal; do
  extra=""; [ $w = fastq ] && extra="--parity-reads 0"; [ $w = mammal ] && extra="--genomes 8"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_$w -- python3 bench.py --workload $w --cpu-sample 0 $extra > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/err_$w.log
  f=$(find gpurun_out/$tag/prof_$w -name "*kernel_stats.csv" | head -1)
  grep -v "at::native" "$f" > gpurun_out/$tag/kernel_stats_$w.csv
  rm -rf gpurun_out/$tag/prof_$w
  echo "== $w"
  python3 - <<PY
import csv, json
for r in csv.DictReader(open("gpurun_out/$tag/kernel_stats_$w.csv")):
    if float(r["AverageNs"]) > 4000 and "rocclr" not in r["Name"] and int(r["Calls"]) >= 10:
        print("   %-50s calls %s avg %.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
j = json.loads(open("gpurun_out/$tag/bench_$w.json").read().strip().splitlines()[-1])
print("   ms_per_step %.4f value %.0f %s" % (j["ms_per_step"], j["value"], j["unit"]))
PY
done 2>&1 | tee gpurun_out/$tag/summary.txt
