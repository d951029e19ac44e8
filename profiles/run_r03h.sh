#!/bin/bash
# round 3: the other BASELINE configs at size on the kernels as they stand -- configs[4] (8 and 50 records of 3 Gb) with a
# kernel trace and the FETCH_SIZE pass of sketch_scan_kernel<7>, configs[3] (100 M reads), configs[2] (10 000 genomes),
# and the end-to-end legs at 10 000 files / a 4 096 x 4 096 search
tag=${1:-r03h}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python bench.py --workload mammal --steps 5 --warmup 2 > gpurun_out/${tag}_bench_mammal8.json 2> gpurun_out/${tag}_bench_mammal8.err
echo "mammal8 rc=$?"; cut -c1-1800 gpurun_out/${tag}_bench_mammal8.json; tail -3 gpurun_out/${tag}_bench_mammal8.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --workload mammal --genomes 4 --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/${tag}_prof.log 2>&1
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/${tag}_mammal_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_${tag}_$ctr
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "sketch_scan_kernel|sketch_exact_kernel" --output-format csv -d $out -- python bench.py --workload mammal --genomes 4 --steps 2 --warmup 1 --cpu-sample 0 > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python3 - "$f" $ctr <<'PY' >> gpurun_out/${tag}_mammal_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:44]
    acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(sys.argv[2], k, "dispatches", len(n[k]), "per dispatch %.4g (4 records x 3 Gb per launch)" % (acc[k] / len(n[k])))
PY
  rm -rf $out
done
cat gpurun_out/${tag}_mammal_pmc.txt
timeout 1500 python bench.py --workload mammal --genomes 50 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${tag}_bench_mammal50.json 2> gpurun_out/${tag}_bench_mammal50.err
echo "mammal50 rc=$?"; cut -c1-1500 gpurun_out/${tag}_bench_mammal50.json; tail -3 gpurun_out/${tag}_bench_mammal50.err
timeout 1500 python bench.py --workload fastq > gpurun_out/${tag}_bench_config4_fastq.json 2> gpurun_out/${tag}_bench_config4_fastq.err
echo "fastq rc=$?"; cut -c1-1500 gpurun_out/${tag}_bench_config4_fastq.json; tail -3 gpurun_out/${tag}_bench_config4_fastq.err
timeout 1500 python bench.py --genomes 10000 --clades 500 --cpu-sample 0 --steps 5 --warmup 2 --spinup 5 > gpurun_out/${tag}_bench_config3_10000genomes.json 2> gpurun_out/${tag}_bench_config3_10000genomes.err
echo "c3 rc=$?"; cut -c1-1500 gpurun_out/${tag}_bench_config3_10000genomes.json; tail -3 gpurun_out/${tag}_bench_config3_10000genomes.err
timeout 2400 python bench.py --steps 10 --warmup 2 --e2e-files 10000 --e2e-search 4096 > gpurun_out/${tag}_bench_e2e10000.json 2> gpurun_out/${tag}_bench_e2e10000.err
echo "e2e rc=$?"
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_e2e10000.json"))
for k in ("value", "cpu_baseline", "cpu_baseline_dist", "cpu_baseline_gz"):
    print(k, json.dumps(d.get(k))[:500])
e = d.get("end_to_end", {})
print("e2e", e.get("value"), e.get("seconds_runs"), json.dumps(e.get("stages"))[:700])
print("gz", json.dumps(e.get("gzip"))[:500])
print("search", json.dumps(e.get("search"))[:700])
PY
tail -5 gpurun_out/${tag}_bench_e2e10000.err
