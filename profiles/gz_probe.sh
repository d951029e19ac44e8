#!/bin/bash
# usage (GPU box): profiles/gz_probe.sh -- one gzip'ed read set (2 M reads, 628 MB of text) through `kssd dist`:
# streamed inflate (the default for a file of this size) against inflate-then-copy, and the reference binary (zcat pipe)
cd $GRAFT_REPO_ROOT
d=$(mktemp -d /dev/shm/kssd_gz_XXXX)
python3 - "$d" <<'PY'
import sys, os, numpy as np, subprocess
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from synth import fastq_records
import public_kssd_amd as K
d = sys.argv[1]
rng = np.random.default_rng(3)
fq = fastq_records(rng.integers(0, 4, (2_000_000, 150), dtype=np.uint8))
open(os.path.join(d, "reads.fastq"), "wb").write(fq)
subprocess.check_call(["gzip", "-1", os.path.join(d, "reads.fastq")])
K.Shuf.generate(10, 6, 3, seed=20260101).write(os.path.join(d, "L3K10.shuf"))
print("compressed bytes:", os.path.getsize(os.path.join(d, "reads.fastq.gz")))
PY
cd $d
run() { t0=$(date +%s.%N); "$@" 2>&1 | tr '\r' '\n' | grep "kssd_timing" | cut -c1-260; t1=$(date +%s.%N); python3 -c "print('wall %.3f s' % ($t1-$t0))"; }
echo "== streamed inflate"; KSSD_TIMING=1 run $GRAFT_REPO_ROOT/public_kssd_amd/kssd dist -L L3K10.shuf -o o1 reads.fastq.gz
echo "== streamed inflate (again)"; KSSD_TIMING=1 run $GRAFT_REPO_ROOT/public_kssd_amd/kssd dist -L L3K10.shuf -o o2 reads.fastq.gz
echo "== inflate into memory, then copy"; KSSD_TIMING=1 KSSD_STREAM_MIN_GZ=99999999999 run $GRAFT_REPO_ROOT/public_kssd_amd/kssd dist -L L3K10.shuf -o o3 reads.fastq.gz
cmp o1/combco.0 o3/combco.0 && echo "combco.0 identical"
if [ -x $GRAFT_REPO_ROOT/oracle/_ref/kssd ]; then echo "== reference"; t0=$(date +%s.%N); (exec -a kssd env -u LD_PRELOAD $GRAFT_REPO_ROOT/oracle/_ref/kssd dist -L L3K10.shuf -o oref reads.fastq.gz >/dev/null 2>&1); t1=$(date +%s.%N); python3 -c "print('wall %.3f s' % ($t1-$t0))"; cmp o1/combco.0 oref/combco.0 && echo "combco.0 identical to the reference's"; fi
cd /; rm -rf $d
