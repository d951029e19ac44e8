#!/bin/bash
# round 3, second GPU call: v_cndmask-free candidate loop (A/B on scanbench with counters), extended instruction-cost probe,
# parity suite, the default line with the new report writer and the search stage split
tag=${1:-r03b}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 profiles/valu_probe > gpurun_out/${tag}_valu_probe.txt 2>&1
{
ABL="2 3 0" profiles/sb_modes.sh
for a in 3 0; do
  KSSD_DEV_ABLATE=$a profiles/pmc_sb.sh ${tag}_abl${a}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
  KSSD_DEV_ABLATE=$a profiles/pmc_sb.sh ${tag}_abl${a}_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
done
} > gpurun_out/${tag}_scan_ablations.txt 2>&1
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${tag}_pytest.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_valu_probe.txt; cat gpurun_out/${tag}_scan_ablations.txt; tail -3 gpurun_out/${tag}_pytest.log
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
for k in ("value", "ms_per_step", "kernels", "roofline", "cpu_baseline", "cpu_baseline_dist", "pipelined"):
    print(k, json.dumps(d.get(k))[:600])
e = d.get("end_to_end", {})
print("e2e", e.get("value"), e.get("seconds_runs"), json.dumps(e.get("stages"))[:900])
print("search", json.dumps(e.get("search"))[:1500])
PY
tail -5 gpurun_out/${tag}_bench.err
