import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import public_kssd_amd as K, bench
dev = torch.device("cuda", 0)
G, L = 1000, 5_000_000
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
ctx = K.GpuCtx(shuf, 0)
cap = int(G * L / 4096 * 1.25) + 4096
sk = []
for seed in (1, 2):
    packed, mask, chunk_off, _ = bench.make_batch(G, L, 50, seed, dev)
    off = torch.zeros(G + 1, dtype=torch.int64, device=dev); ids = torch.zeros(cap, dtype=torch.int32, device=dev)
    ctx.sketch_device(packed, mask, chunk_off, off, ids, cap); rc, total, bad = ctx.sketch_status(); assert rc == 0
    sk.append((off, ids, int(total)))
    del packed, mask
(offA, idsA, tA), (offB, idsB, tB) = sk
shared = torch.zeros(G * G, dtype=torch.int32, device=dev)
planes = [torch.zeros(G * G, dtype=torch.float64, device=dev) for _ in range(4)]
for filt in (False, True):
    ctx.index_set_filter(filt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.index_build_device(offA, idsA, G, tA)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ctx.index_build_device(offA, idsA, G, tA)
    e1.record()
    torch.cuda.synchronize()
    print("negative filter %s: index build %.1f us" % ("ON" if filt else "off", e0.elapsed_time(e1) / 20 * 1e3))
    sums = {}
    for name, (qo, qi) in (("own", (offA, idsA)), ("foreign", (offB, idsB))):
        for pl, tag in ((planes, "planes"), ([None] * 4, "counts only")):
            ctx.kernel_time(1, reset=True)
            for _ in range(20):
                ctx.dist_device(qo, qi, G, 0, G, shared, *pl)
            torch.cuda.synchronize()
            ms, n = ctx.kernel_time(1)
            print("   rows %-8s %-12s %.1f us (%d launches), shared sum %d" % (name, tag, ms * 1e3, n, int(shared.sum())))
