#!/bin/bash
# round 6, after the parts_skew repair: the fuzzers on seed bases no earlier run used
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06fz2; mkdir -p $o
timeout 700 python3 profiles/fuzz_sketch.py 450 200000 > $o/fuzz_sketch.txt 2>&1; tail -1 $o/fuzz_sketch.txt
KSSD_MASK_SUMMARY=1 timeout 700 python3 profiles/fuzz_sketch.py 450 300000 > $o/fuzz_sketch_summary.txt 2>&1; tail -1 $o/fuzz_sketch_summary.txt
timeout 400 python3 profiles/fuzz_fastq.py 500 50000 > $o/fuzz_fastq.txt 2>&1; tail -1 $o/fuzz_fastq.txt
timeout 800 python3 profiles/fuzz_cli.py 300 400000 > $o/fuzz_cli.txt 2>&1; tail -1 $o/fuzz_cli.txt
timeout 500 python3 profiles/fuzz_allpairs.py 100 7000 > $o/fuzz_allpairs.txt 2>&1; tail -1 $o/fuzz_allpairs.txt
timeout 300 python3 profiles/fuzz_dist.py 1500 > $o/fuzz_dist.txt 2>&1; tail -1 $o/fuzz_dist.txt
grep -v "not counted" $o/fuzz_cli.txt | tail -5
grep -i "error\|differ\|bad" $o/fuzz_sketch.txt $o/fuzz_sketch_summary.txt $o/fuzz_fastq.txt $o/fuzz_allpairs.txt $o/fuzz_dist.txt | tail -10
