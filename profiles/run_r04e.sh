#!/bin/bash
# r04e: what bounds dist_rows_kernel: counter passes (TA / TCP / TCC / SQ) on the rows kernel alone
tag=${1:-r04e}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 -L > gpurun_out/$tag/counters.txt 2>&1
grep -c . gpurun_out/$tag/counters.txt
grep -o -E "\b(TA|TCP|TCC|TD)_[A-Z0-9_]+" gpurun_out/$tag/counters.txt | sort -u | tr '\n' ' ' | cut -c1-6000
echo
rx='dist_rows_kernel|idx_build_kernel|idx_scatter|sketch_dedup_kernel'
{
profiles/pmc_pass.sh ${tag}_sq1 "$rx" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
profiles/pmc_pass.sh ${tag}_sq2 "$rx" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
profiles/pmc_pass.sh ${tag}_ta "$rx" TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
profiles/pmc_pass.sh ${tag}_tcp "$rx" TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
profiles/pmc_pass.sh ${tag}_tcc "$rx" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
profiles/pmc_pass.sh ${tag}_tcc2 "$rx" TCC_READ_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum
} > gpurun_out/$tag/pmc_rows.txt 2>&1
rm -rf gpurun_out/pmc_${tag}_*/
cat gpurun_out/$tag/pmc_rows.txt | cut -c1-160
