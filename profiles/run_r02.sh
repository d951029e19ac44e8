#!/bin/bash
# usage (GPU box, through gpurun): profiles/run_r02.sh <tag> [tests|bench|fastq|c3|prof|pmc ...]
# Round-2 measurement driver: GPU parity tests, the default bench line (configs[1]), configs[2] (10 000 genomes) and
# configs[3] (FASTQ) lines, a rocprofv3 kernel-trace summary and the PMC passes (separate passes, counters + kernel trace only).
tag=${1:-r02}; shift
what=${@:-tests bench}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in $what; do
case $w in
tests) ( timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -40 ) > gpurun_out/${tag}_pytest.log; tail -5 gpurun_out/${tag}_pytest.log ;;
quick) ( timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_cli.py tests/test_gpu_sketch.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${tag}_pytest_quick.log; tail -4 gpurun_out/${tag}_pytest_quick.log ;;
benchq) timeout 600 python bench.py --cpu-sample 0 > gpurun_out/${tag}_benchq.json 2> gpurun_out/${tag}_benchq.err; tail -2 gpurun_out/${tag}_benchq.err; cut -c1-900 gpurun_out/${tag}_benchq.json ;;
e2e) timeout 900 python bench.py --steps 5 --warmup 1 --spinup 5 > gpurun_out/${tag}_bench_e2e.json 2> gpurun_out/${tag}_bench_e2e.err; tail -2 gpurun_out/${tag}_bench_e2e.err; python3 -c "import json;r=json.load(open('gpurun_out/${tag}_bench_e2e.json'));print(json.dumps(r.get('end_to_end'))[:1500]);print(r.get('cpu_baseline'));print(r.get('cpu_baseline_dist'))" ;;
tok) ( timeout 1200 python -m pytest tests/test_gpu_tokenise.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/${tag}_pytest_tok.log; tail -30 gpurun_out/${tag}_pytest_tok.log ;;
tokprof) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_tokprof -- python3 profiles/tok_probe.py > gpurun_out/${tag}_tokprof.log 2>&1
    f=$(find gpurun_out/${tag}_tokprof -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/${tag}_tok_kernel_stats.csv
    rm -rf gpurun_out/${tag}_tokprof; grep -v "^W\|^E" gpurun_out/${tag}_tokprof.log | tail -4; cat gpurun_out/${tag}_tok_kernel_stats.csv ;;
proffq) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_proffq -- python3 bench.py --workload fastq --genomes 10000 --clades 500 --steps 5 --warmup 1 --parity-reads 0 > gpurun_out/${tag}_proffq.log 2>&1
    f=$(find gpurun_out/${tag}_proffq -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/${tag}_fastq_kernel_stats.csv
    rm -rf gpurun_out/${tag}_proffq; tail -2 gpurun_out/${tag}_proffq.log | cut -c1-300; python3 - gpurun_out/${tag}_fastq_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r[0] == "Name": continue
    print("%-70s calls %4s avg %9.1f us" % (r[0][:70], r[1], float(r[3]) / 1e3))
PY
    ;;
smoke) timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 ;;
inflight) timeout 600 python bench.py --inflight 3 --steps 30 --warmup 3 --cpu-sample 0 > gpurun_out/${tag}_bench_inflight3.json 2> gpurun_out/${tag}_bench_inflight3.err; tail -2 gpurun_out/${tag}_bench_inflight3.err; cut -c1-400 gpurun_out/${tag}_bench_inflight3.json ;;
info) { nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; grep -c processor /proc/cpuinfo; grep "model name" /proc/cpuinfo | head -1; free -g | head -2; python3 -c "import os;print(len(os.sched_getaffinity(0)))"; } 2>&1 | tee gpurun_out/${tag}_info.txt ;;
probe) profiles/pinned_probe 2>&1 | tee gpurun_out/${tag}_pinned_probe.txt ;;
bench) timeout 1500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -3 gpurun_out/${tag}_bench.err; cut -c1-1500 gpurun_out/${tag}_bench.json ;;
fastq) timeout 1800 python bench.py --workload fastq --genomes 10000 --clades 500 --steps 10 --warmup 2 > gpurun_out/${tag}_bench_fastq.json 2> gpurun_out/${tag}_bench_fastq.err; tail -3 gpurun_out/${tag}_bench_fastq.err; cut -c1-1500 gpurun_out/${tag}_bench_fastq.json ;;
c3) timeout 900 python bench.py --genomes 10000 --clades 500 --steps 10 --warmup 2 --spinup 5 --cpu-sample 0 > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err; tail -3 gpurun_out/${tag}_bench_c3.err; cut -c1-1200 gpurun_out/${tag}_bench_c3.json ;;
n2) KSSD_BENCH_ONE_DEVICE=1 KSSD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --genomes 400 --cpu-sample 0 \
      > gpurun_out/${tag}_bench_n2_onegpu.json 2> gpurun_out/${tag}_bench_n2_onegpu.err
    KSSD_BENCH_ONE_DEVICE=1 KSSD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 6 --warmup 2 --genomes 400 --cpu-sample 0 --partition query \
      > gpurun_out/${tag}_bench_n2q_onegpu.json 2> gpurun_out/${tag}_bench_n2q_onegpu.err
    cut -c1-300 gpurun_out/${tag}_bench_n2_onegpu.json; tail -2 gpurun_out/${tag}_bench_n2_onegpu.err; cut -c1-300 gpurun_out/${tag}_bench_n2q_onegpu.json; tail -2 gpurun_out/${tag}_bench_n2q_onegpu.err ;;
prof) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/${tag}_prof.log 2>&1
    f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && grep -v "at::native" "$f" > gpurun_out/${tag}_kernel_stats.csv
    rm -rf gpurun_out/${tag}_prof; cat gpurun_out/${tag}_kernel_stats.csv ;;
pmc) rx='sketch_scan_kernel|sketch_exact_kernel|sketch_dedup_kernel|idx_|dist_rows'
    {
    profiles/pmc_pass.sh ${tag}_sq2 "$rx" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
    profiles/pmc_pass.sh ${tag}_fetch "$rx" FETCH_SIZE
    profiles/pmc_pass.sh ${tag}_write "$rx" WRITE_SIZE
    } > gpurun_out/${tag}_pmc.txt 2>&1
    rm -rf gpurun_out/pmc_${tag}_*/; cat gpurun_out/${tag}_pmc.txt ;;
esac
done
