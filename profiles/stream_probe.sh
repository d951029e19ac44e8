#!/bin/bash
# usage (GPU box): profiles/stream_probe.sh -- the streamed upload of one 3 GB FASTQ file through `kssd dist`, with the
# stage split (KSSD_TIMING) for a few ring shapes
cd $GRAFT_REPO_ROOT
d=$(mktemp -d /dev/shm/kssd_sp_XXXX)
python3 - "$d" <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from synth import fastq_records
import public_kssd_amd as K
d = sys.argv[1]
rng = np.random.default_rng(3)
fq = fastq_records(rng.integers(0, 4, (10_000_000, 150), dtype=np.uint8))
open(os.path.join(d, "reads.fastq"), "wb").write(fq)
K.Shuf.generate(10, 6, 3, seed=20260101).write(os.path.join(d, "L3K10.shuf"))
PY
cd $d
for slice in 8388608; do
  for i in 1 2; do
    echo "== slice $slice"
    t0=$(date +%s.%N)
    KSSD_TIMING=1 KSSD_STREAM_SLICE=$slice $GRAFT_REPO_ROOT/public_kssd_amd/kssd dist -L L3K10.shuf -o o_$slice reads.fastq 2>&1 | tr '\r' '\n' | grep "kssd_timing"
    t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc -l 2>/dev/null || python3 -c "print($t1-$t0)") s"
  done
done
cd /; rm -rf $d
