#!/bin/bash
# usage (GPU box, through gpurun): profiles/run_round.sh <tag>
# GPU parity tests, the default bench line, a rocprofv3 kernel-trace summary, the PMC passes of the scan kernel
# (separate passes, counters + kernel trace only) and the FETCH_SIZE calibration of the three read widths.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${tag}_pytest.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout 300 python bench.py --inflight 3 --steps 30 --warmup 3 --cpu-sample 0 > gpurun_out/${tag}_bench_inflight3.json 2> gpurun_out/${tag}_bench_inflight3.err
# the N = 2 flow on this one GPU (both ranks on cuda:0, gloo instead of RCCL): checks the multi-rank path of the default schedule
KSSD_BENCH_ONE_DEVICE=1 KSSD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --genomes 400 --cpu-sample 0 \
  > gpurun_out/${tag}_bench_n2_onegpu.json 2> gpurun_out/${tag}_bench_n2_onegpu.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/${tag}_prof.log 2>&1
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/${tag}_kernel_stats.csv
rx='sketch_scan_kernel|sketch_exact_kernel'
{
profiles/pmc_pass.sh ${tag}_sq1 "$rx" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
profiles/pmc_pass.sh ${tag}_sq2 "$rx" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
rx2='sketch_scan_kernel|sketch_dedup_kernel|idx_|dist_rows_kernel'
profiles/pmc_pass.sh ${tag}_fetch "$rx2" FETCH_SIZE
profiles/pmc_pass.sh ${tag}_write "$rx2" WRITE_SIZE
echo "--- FETCH_SIZE calibration: every calib_read* launch reads exactly 2^30 bytes (profiles/scanbench calib)"
SB_ARGS=calib profiles/pmc_sb.sh ${tag}_calib FETCH_SIZE
} > gpurun_out/${tag}_pmc_scan.txt 2>&1
rm -rf gpurun_out/${tag}_prof gpurun_out/pmc_${tag}_*/
tail -3 gpurun_out/${tag}_pytest.log; cut -c1-700 gpurun_out/${tag}_bench.json; cut -c1-300 gpurun_out/${tag}_bench_inflight3.json; cut -c1-300 gpurun_out/${tag}_bench_n2_onegpu.json; tail -3 gpurun_out/${tag}_bench_n2_onegpu.err; cat gpurun_out/${tag}_pmc_scan.txt
