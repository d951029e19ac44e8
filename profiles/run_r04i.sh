#!/bin/bash
# r04i: the scan's structural experiment: stage 1 as a 2^20-bit table (three windows per read, one alignment) against the byte table of
# five windows per read and two alignments -- parity first, then time and counters, both in the development build, same box
tag=${1:-r04i}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
export KSSD_GPU_LIB=$GRAFT_REPO_ROOT/public_kssd_amd/libkssd_gpu_dev.so
KSSD_DEV_BITTAB=1 timeout 900 python -m pytest tests/test_gpu_sketch.py -m gpu -x -q -k "not fastq and not byread" > gpurun_out/$tag/tests_bittab.log 2>&1
echo "rc=$?" >> gpurun_out/$tag/tests_bittab.log
tail -4 gpurun_out/$tag/tests_bittab.log
for v in byte bit; do
  if [ $v = bit ]; then export KSSD_DEV_BITTAB=1; else unset KSSD_DEV_BITTAB; fi
  timeout 600 python bench.py --cpu-sample 0 --steps 50 > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/bench_$v.err
  timeout 600 python bench.py --cpu-sample 0 --steps 50 > gpurun_out/$tag/bench_${v}_2.json 2>> gpurun_out/$tag/bench_$v.err
  rx='sketch_scan_kernel|sketch_dedup_kernel'
  {
  profiles/pmc_pass.sh ${tag}_${v}_sq "$rx" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  profiles/pmc_pass.sh ${tag}_${v}_fetch "$rx" FETCH_SIZE
  } > gpurun_out/$tag/pmc_$v.txt 2>&1
done
rm -rf gpurun_out/pmc_${tag}_*/
python3 - <<PY
import json
for v in ("byte", "byte_2", "bit", "bit_2"):
    try:
        j = json.loads(open("gpurun_out/$tag/bench_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "ms_per_step %.4f" % j["ms_per_step"], "scan ms %.4f" % j["kernels"]["sketch_scan_ms"], "past stage1 %.5f" % j["kernels"]["scan_positions_past_stage1"],
              "past bloom %.6f" % j["kernels"]["scan_positions_past_bloom"])
    except Exception as e:
        print(v, "failed", e)
PY
for v in byte bit; do echo "== $v"; grep -A9 "sketch_scan_kernel" gpurun_out/$tag/pmc_$v.txt | cut -c1-100 | head -24; grep -A9 "sketch_dedup" gpurun_out/$tag/pmc_$v.txt | grep -E "dedup|VALU|VMEM_RD" | head -4; done
