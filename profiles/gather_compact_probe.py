"""development probe (GPU box): what does the device-side compaction of the all-gathered sketch units cost at N = 8?
(public_kssd_amd/shard.py SketchGather.compact: torch ops on world x cap elements)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from public_kssd_amd.shard import SketchGather
dev = torch.device("cuda", 0)
for world in (2, 4, 8):
    G, cap = 1000, int(1000 * 5_000_000 / 4096 * 1.25) + 4096
    g = SketchGather(world, G, cap, dev)
    sizes = torch.randint(1150, 1300, (world, G), device=dev)
    off = torch.zeros(world, G + 1, dtype=torch.int64, device=dev)
    off[:, 1:] = torch.cumsum(sizes, 1)
    g.off_all.copy_(off.view(-1))
    g.ids_all.random_(0, 1 << 28)
    for _ in range(3):
        g.compact()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.compact()
    e1.record()
    torch.cuda.synchronize()
    t_torch = e0.elapsed_time(e1) / 20 * 1e3
    want_off, want_ids = g.roff.clone(), g.rids.clone()
    import public_kssd_amd as K
    ctx = K.GpuCtx(kmerlen=20)
    g.roff.zero_(); g.rids.zero_()
    for _ in range(3):
        ctx.concat_units_device(g.off_all, g.ids_all, world, G, cap, g.roff, g.rids)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ctx.concat_units_device(g.off_all, g.ids_all, world, G, cap, g.roff, g.rids)
    e1.record()
    torch.cuda.synchronize()
    n = int(want_off[-1])
    assert torch.equal(g.roff, want_off) and torch.equal(g.rids[:n], want_ids[:n])
    ctx.close()
    print("world %d: tensor-op compaction %.1f us, kssd_gpu_concat_units_device %.1f us per call (%.1f M id slots), same CSR"
          % (world, t_torch, e0.elapsed_time(e1) / 20 * 1e3, world * cap / 1e6))
