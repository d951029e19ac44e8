#!/bin/bash
# round 6, the last tree: the fuzzers once more on further seed bases
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06fz3; mkdir -p $o
timeout 800 python3 profiles/fuzz_cli.py 400 500000 > $o/fuzz_cli.txt 2>&1; tail -1 $o/fuzz_cli.txt
timeout 700 python3 profiles/fuzz_sketch.py 450 500000 > $o/fuzz_sketch.txt 2>&1; tail -1 $o/fuzz_sketch.txt
KSSD_MASK_SUMMARY=1 timeout 700 python3 profiles/fuzz_sketch.py 450 600000 > $o/fuzz_sketch_summary.txt 2>&1; tail -1 $o/fuzz_sketch_summary.txt
timeout 400 python3 profiles/fuzz_fastq.py 500 70000 > $o/fuzz_fastq.txt 2>&1; tail -1 $o/fuzz_fastq.txt
KSSD_MASK_SUMMARY=1 timeout 400 python3 profiles/fuzz_fastq.py 500 80000 > $o/fuzz_fastq_summary.txt 2>&1; tail -1 $o/fuzz_fastq_summary.txt
timeout 600 python3 profiles/fuzz_allpairs.py 150 9000 > $o/fuzz_allpairs.txt 2>&1; tail -1 $o/fuzz_allpairs.txt
grep -v "not counted" $o/fuzz_cli.txt | tail -5
grep -i "error\|differ\|bad [1-9]" $o/fuzz_sketch.txt $o/fuzz_sketch_summary.txt $o/fuzz_fastq.txt $o/fuzz_fastq_summary.txt $o/fuzz_allpairs.txt | tail -10
