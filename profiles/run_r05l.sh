cd $GRAFT_REPO_ROOT
o=gpurun_out/r05l; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -2
for cs in 4 0 2; do echo "cursor shift $cs"; KSSD_INDEX_CURSOR_SHIFT=$cs python3 profiles/index_sizes_probe.py 2>&1 | grep "10000 sketches" | grep -v ONE_LEVEL; done | tee $o/index_cursor_stride.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 profiles/index_sizes_probe.py > /dev/null 2>&1
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); cp "$f" $o/index_kernel_stats.csv; rm -rf $o/prof
python3 - <<PY
import csv
for r in csv.DictReader(open("$o/index_kernel_stats.csv")):
    if "idx_" in r["Name"]:
        print("%-40s calls %4s avg %8.1f min %8.1f max %8.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
