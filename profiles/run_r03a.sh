#!/bin/bash
# round 3, first GPU call: parity suite after the scan clean-up + periodic drain, instruction-cost probe, scan ablations
# with SQ counters, the self-launched N = 2 line on one device (gloo), the default line
tag=${1:-r03a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${tag}_pytest.log
timeout 300 profiles/valu_probe > gpurun_out/${tag}_valu_probe.txt 2>&1
{
ABL="1 2 3 0" profiles/sb_modes.sh
for a in 1 2 3 0; do
  KSSD_DEV_ABLATE=$a profiles/pmc_sb.sh ${tag}_abl${a}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
  KSSD_DEV_ABLATE=$a profiles/pmc_sb.sh ${tag}_abl${a}_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
done
} > gpurun_out/${tag}_scan_ablations.txt 2>&1
KSSD_BENCH_ONE_DEVICE=1 KSSD_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 6 --warmup 2 --genomes 400 --cpu-sample 0 --spinup 5 \
  > gpurun_out/${tag}_bench_n2_selflaunch.json 2> gpurun_out/${tag}_bench_n2_selflaunch.err
echo "n2 rc=$?" >> gpurun_out/${tag}_bench_n2_selflaunch.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 \
  > gpurun_out/${tag}_bench_n1_launcher.json 2> gpurun_out/${tag}_bench_n1_launcher.err
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -3 gpurun_out/${tag}_pytest.log; cat gpurun_out/${tag}_valu_probe.txt; cat gpurun_out/${tag}_scan_ablations.txt
cut -c1-1500 gpurun_out/${tag}_bench_n2_selflaunch.json; tail -5 gpurun_out/${tag}_bench_n2_selflaunch.err
cut -c1-400 gpurun_out/${tag}_bench_n1_launcher.json; cut -c1-700 gpurun_out/${tag}_bench.json
