cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --genomes 10000 --clades 500 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -2 | cut -c1-1500
