"""development probe (GPU box): can RCCL run TWO ranks on ONE device?  Both ranks of a torch.distributed.run launch take cuda:0 and try an
all_reduce and an all_gather_into_tensor over the nccl backend.  python3 -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 profiles/rccl_two_ranks_one_gpu_probe.py"""
import datetime, os, sys
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
    x = torch.ones(4, device=dev) * (rank + 1)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    out = torch.zeros(world * 4, device=dev)
    dist.all_gather_into_tensor(out, torch.arange(4, device=dev, dtype=torch.float32) + 10 * rank)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce", x.tolist(), "all_gather", out.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", rank, "FAILED:", str(e)[:600].replace("\n", " | "), flush=True)
    sys.exit(3)
