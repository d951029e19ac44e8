cd $GRAFT_REPO_ROOT
o=gpurun_out/r05m; mkdir -p $o
for a in 4 6 7 8 9 10; do echo "first level: $a bits"; KSSD_INDEX_LGA=$a python3 profiles/index_sizes_probe.py 2>&1 | grep "10000 sketches" | grep -v ONE_LEVEL
  KSSD_INDEX_LGA=$a timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 profiles/index_sizes_probe.py > /dev/null 2>&1
  f=$(find $o/prof -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "idx_scatter_tile" in r["Name"] or "idx_scatter_sub" in r["Name"]:
        print("   %-40s calls %4s max %8.1f" % (r["Name"][:40], r["Calls"], float(r["MaxNs"])/1e3))
PY
  rm -rf $o/prof
done 2>&1 | tee $o/index_first_level_bits.txt
