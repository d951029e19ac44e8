#!/bin/bash
# round 3: the scan with blocks from per-workgroup queues against a static round-robin of the same blocks (A/B in the
# development build), parity suite, default line
tag=${1:-r03d}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${tag}_pytest.log
tail -5 gpurun_out/${tag}_pytest.log
{
for g in 400 1000 2000; do
  echo "== genomes $g, queue"; KSSD_DEV_WAVETIME=1 timeout 300 profiles/scanbench $g 5000000 10 | grep -v "^stats"
  echo "== genomes $g, static round-robin"; KSSD_DEV_STATIC=1 KSSD_DEV_WAVETIME=1 timeout 300 profiles/scanbench $g 5000000 10 | grep -v "^stats"
done
KSSD_DEV_ABLATE=0 profiles/pmc_sb.sh ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
KSSD_DEV_ABLATE=0 profiles/pmc_sb.sh ${tag}_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
} > gpurun_out/${tag}_scan_queue.txt 2>&1
cat gpurun_out/${tag}_scan_queue.txt
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value", "ms_per_step", "kernels", "roofline", "pipelined"):
    print(k, json.dumps(d.get(k))[:600])
e = d.get("end_to_end", {})
print("e2e", e.get("value"), e.get("seconds_runs"))
print("search", json.dumps(e.get("search"))[:600])
PY
tail -5 gpurun_out/${tag}_bench.err
