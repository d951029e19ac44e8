"""development tool: distribution of the ulp distance between the device's MashD / AafD planes and the host formula
(oracle, host libm) over the whole 1 000 x 1 000 matrix of the bench workload"""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import bench
import kssd_oracle as ko
import public_kssd_amd as K

dev = torch.device("cuda", 0)
G, L = 1000, 5_000_000
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
ctx = K.GpuCtx(shuf, 0)
cap = int(G * L / 4096 * 1.25) + 4096
packed, mask, chunk_off, _ = bench.make_batch(G, L, 50, 20260101, dev)
off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
ids = torch.zeros(cap, dtype=torch.int32, device=dev)
ctx.sketch_device(packed, mask, chunk_off, off, ids, cap)
rc, total, bad = ctx.sketch_status()
assert rc == 0
shared = torch.zeros(G * G, dtype=torch.int32, device=dev)
planes = [torch.zeros(G * G, dtype=torch.float64, device=dev) for _ in range(4)]
ctx.index_build_device(off, ids, G, total)
ctx.dist_device(off, ids, G, 0, G, shared, *planes)
torch.cuda.synchronize()
sz = (off[1:] - off[:-1]).cpu().numpy().astype(np.uint32)
S = shared.cpu().numpy().view(np.uint32).reshape(G, G)
oJ, oMD, oC, oAD = ko.metrics_batch(sz[None, :], sz[:, None], S, 20)
for name, dev_p, host in (("J", planes[0], oJ), ("MashD", planes[1], oMD), ("C", planes[2], oC), ("AafD", planes[3], oAD)):
    a = dev_p.cpu().numpy().reshape(G, G)
    d = np.abs(a.view(np.int64) - host.view(np.int64))
    sel = S > 0
    print("%-6s pairs with s>0: %d; ulp histogram %s; max %d" % (name, sel.sum(), np.bincount(np.minimum(d[sel], 5), minlength=6).tolist(), d.max()))
