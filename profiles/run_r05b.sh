#!/bin/bash
# round 5: the GPU suite again, and a kernel trace of one GPU as one rank of eight (own-index partition) -- what the rows cost
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q > $o/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $o/tests_gpu.log
tail -8 $o/tests_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$o/prof_emu8 -o emu8 -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-world 8 --rank 3 --partition own --steps 20 --warmup 5 --cpu-sample 0 > $GRAFT_REPO_ROOT/$o/prof_emu8.json 2> $GRAFT_REPO_ROOT/$o/prof_emu8.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $o/prof_emu8 -name '*kernel_stats.csv' | head -1); echo $f; head -14 "$f" | cut -c1-160
cp "$f" $o/emu8_kernel_stats.csv 2>/dev/null
find $o/prof_emu8 -name '*.csv' ! -name '*kernel_stats.csv' -delete; find $o/prof_emu8 -name '*.db' -delete
