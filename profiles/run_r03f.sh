#!/bin/bash
# round 3: alignment B read only where alignment A left a position standing, as a broadcast-predicated read (no branch),
# against the round-2 reads of every lane; the block queue once more on top; parity suite (incl. -Q on the device); default line
tag=${1:-r03f}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for g in 400 1000 2000; do
  echo "== genomes $g, B where A (product)"; timeout 300 profiles/scanbench $g 5000000 10 | grep -v "^stats"
  echo "== genomes $g, B on every lane"; timeout 300 profiles/scanbench_nopred $g 5000000 10 | grep -v "^stats"
  echo "== genomes $g, B where A + block queue"; KSSD_DEV_QUEUE=1 timeout 300 profiles/scanbench $g 5000000 10 | grep -v "^stats"
done
echo "== mammal-like: 8 x 250 Mb at -s 7 -l 5"; SB_SUBK=7 SB_DRL=5 timeout 300 profiles/scanbench 8 250000000 5 | grep -v "^stats"
echo "== the same, B on every lane"; SB_SUBK=7 SB_DRL=5 timeout 300 profiles/scanbench_nopred 8 250000000 5 | grep -v "^stats"
profiles/pmc_sb.sh ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
profiles/pmc_sb.sh ${tag}_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
} > gpurun_out/${tag}_scan_ab.txt 2>&1
cat gpurun_out/${tag}_scan_ab.txt
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${tag}_pytest.log
tail -5 gpurun_out/${tag}_pytest.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value", "ms_per_step", "kernels", "roofline", "pipelined"):
    print(k, json.dumps(d.get(k))[:600])
e = d.get("end_to_end", {})
print("e2e", e.get("value"), e.get("seconds_runs"))
print("search", json.dumps(e.get("search"))[:400])
PY
tail -5 gpurun_out/${tag}_bench.err
