#!/bin/bash
# r04h: index build with coalesced partition stores (tile sorted by bucket in LDS) and postings staged in LDS: dist tests, trace, bench, TCC counters
tag=${1:-r04h}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py tests/test_allpairs_flow.py -m gpu -x -q > gpurun_out/$tag/tests_gpu.log 2>&1
echo "gpu rc=$?" >> gpurun_out/$tag/tests_gpu.log
tail -4 gpurun_out/$tag/tests_gpu.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python3 bench.py --steps 20 --warmup 2 --cpu-sample 0 > gpurun_out/$tag/prof.log 2>&1
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/$tag/kernel_stats.csv
rm -rf gpurun_out/$tag/prof
cut -d, -f1-4 gpurun_out/$tag/kernel_stats.csv | cut -c1-110 | head -12
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python3 - <<PY
import json
j = json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
print("ms_per_step %.4f" % j["ms_per_step"], "value", j["value"], "dist ms/step %.4f" % j["dist_ms_per_step"], j["kernels"])
PY
{
profiles/pmc_pass.sh ${tag}_tcc 'idx_' TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
} > gpurun_out/$tag/pmc_idx.txt 2>&1
rm -rf gpurun_out/pmc_${tag}_*/
cat gpurun_out/$tag/pmc_idx.txt | cut -c1-120
timeout 900 python bench.py --genomes 10000 --clades 500 --cpu-sample 0 --steps 5 > gpurun_out/$tag/bench_config3.json 2> gpurun_out/$tag/bench_config3.err
python3 - <<PY
import json
j = json.loads(open("gpurun_out/$tag/bench_config3.json").read().strip().splitlines()[-1])
print("config3 ms_per_step %.4f" % j["ms_per_step"], "value", j["value"], "dist ms/step %.4f" % j["dist_ms_per_step"], j["kernels"])
PY
