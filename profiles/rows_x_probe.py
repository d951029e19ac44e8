"""(The kernel this measured -- dist_rows_x_kernel, csrc/kssd_distx.inc -- is in the tree of commit 95bf0b6 only: measured, not kept.)
dist_rows_kernel against dist_rows_x_kernel (a row as eight parts, one per XCD) on the bench's all-pairs: 1 000 sketches of the
bench generator's clade structure (here: random ids with clade sharing), index + rows, the rows kernel's own time by the dispatch events."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import public_kssd_amd as K
from benchlib.workloads import make_batch
dev = torch.device("cuda", 0)
G, L = 1000, 5_000_000
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
packed, mask, chunk_off, _ = make_batch(G, L, 50, 20260101, dev)
ctx = K.GpuCtx(shuf, 0)
cap = int(G * L / 4096 * 1.25) + 4096
off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
ids = torch.zeros(cap, dtype=torch.int32, device=dev)
for _ in range(3):
    ctx.sketch_device(packed, mask, chunk_off, off, ids, cap)
    rc, total, bad = ctx.sketch_status()
del packed, mask
ctx.index_build_device(off, ids, G, int(total) + 1024)
xids = torch.zeros(cap, dtype=torch.int32, device=dev)
xoff = torch.zeros(9 * G, dtype=torch.int32, device=dev)
ctx.xorder_device(off, ids, G, xids, xoff)
sh = [torch.zeros(G * G, dtype=torch.int32, device=dev) for _ in range(2)]
pl = [[torch.zeros(G * G, dtype=torch.float64, device=dev) for _ in range(4)] for _ in range(2)]
ctx.dist_device(off, ids, G, 0, G, sh[0], *pl[0])
assert ctx.dist_device_x(off, xids, xoff, G, 0, G, sh[1], *pl[1])
torch.cuda.synchronize()
assert os.environ.get("KSSD_DISTX_ABLATE") or (torch.equal(sh[0], sh[1]) and all(torch.equal(a.view(torch.int64), b.view(torch.int64)) for a, b in zip(pl[0], pl[1])))
print("same counts and plane bits; shared sum", int(sh[0].to(torch.int64).sum()))
if os.environ.get("KSSD_DISTX_ABLATE"):
    print("ablation", os.environ["KSSD_DISTX_ABLATE"])
for name, fn in (("dist_rows_kernel  ", lambda: ctx.dist_device(off, ids, G, 0, G, sh[0], *pl[0])),
                 ("dist_rows_x_kernel", lambda: ctx.dist_device_x(off, xids, xoff, G, 0, G, sh[1], *pl[1])),
                 ("xorder kernel     ", lambda: ctx.xorder_device(off, ids, G, xids, xoff))):
    for rep in range(2):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ctx.kernel_time(1, reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = ctx.kernel_times(1)
        print("%s back to back %7.1f us per call; dispatch events: mean %6.1f min %6.1f max %6.1f us (%d)" % (
            name, e0.elapsed_time(e1) / 50 * 1e3, (t.mean() * 1e3 if len(t) else 0), (t.min() * 1e3 if len(t) else 0), (t.max() * 1e3 if len(t) else 0), len(t)), flush=True)
