cd $GRAFT_REPO_ROOT
export KSSD_BENCH_ONE_DEVICE=1 KSSD_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --genomes 200 2>&1 | tail -12 | cut -c1-600
