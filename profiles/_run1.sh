cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tmp_prof -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/tmp_prof.log 2>&1
f=$(find gpurun_out/tmp_prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native" "$f" | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    print(r[0][:48].ljust(48), r[1], r[3][:9])
" | grep -i "idx_\|dist_"
rm -rf gpurun_out/tmp_prof
grep '"metric"' gpurun_out/tmp_prof.log | cut -c100-260
