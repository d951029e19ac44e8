#!/usr/bin/env python3
"""development tool (GPU box): where dist_rows_kernel's time goes, per row, on the bench's batch (configs[1]).
Needs the -DKSSD_DEV build (make -C public_kssd_amd tools); run as
    KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DISTTIME=1 python3 profiles/dist_phases.py [genomes]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import public_kssd_amd as K
from public_kssd_amd import capi

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = 5_000_000
dev = torch.device("cuda", 0)
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
packed, mask, chunk_off, _ = bench.make_batch(G, L, max(G // 20, 1), 20260101, dev)
ctx = K.GpuCtx(shuf, 0)
cap = int(G * L / 4096 * 1.25) + 4096
off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
ids = torch.zeros(cap, dtype=torch.int32, device=dev)
shared = torch.zeros(G * G, dtype=torch.int32, device=dev)
planes = [torch.zeros(G * G, dtype=torch.float64, device=dev) for _ in range(4)]
for rep in range(4):
    ctx.sketch_device(packed, mask, chunk_off, off, ids, cap)
    ctx.index_build_device(off, ids, G, cap)
    ctx.dist_device(off, ids, G, 0, G, shared, *planes)
torch.cuda.synchronize()
for which, name in ((1, "dist_rows"),):
    print(name, "kernel_time", ctx.kernel_time(which, reset=True))
ctx.kernel_time(1, reset=True)
for rep in range(10):
    ctx.dist_device(off, ids, G, 0, G, shared, *planes)
torch.cuda.synchronize()
print("dist_rows avg ms over 10 launches", ctx.kernel_time(1))
lib = capi.gpu_lib()
print("matrix checksum", int(shared.to(torch.int64).sum()), float(planes[1].sum()))
if not hasattr(lib, "kssd_gpu_dev_disttimes") or not os.environ.get("KSSD_DEV_DISTTIME"):
    sys.exit(0)
t = np.zeros(G * 4, dtype=np.uint64)
lib.kssd_gpu_dev_disttimes.argtypes = [C.c_void_p, C.c_uint32]
rc = lib.kssd_gpu_dev_disttimes(t.ctypes.data, G)
assert rc == 0, rc
t = t.reshape(G, 4).astype(np.int64)
d = np.diff(t, axis=1)
tot = t[:, 3] - t[:, 0]
for i, nm in enumerate(("start -> ids probed (first pass)", "postings counted (+ later passes)", "epilogue (metrics, 36 KB of stores)")):
    v = np.sort(d[:, i])
    print("  %-40s min %7d  median %7d  mean %9.0f  p90 %7d  max %7d ticks" % (nm, v[0], v[len(v) // 2], v.mean(), v[len(v) * 9 // 10], v[-1]))
v = np.sort(tot)
print("  %-40s min %7d  median %7d  mean %9.0f  p90 %7d  max %7d ticks" % ("whole workgroup", v[0], v[len(v) // 2], v.mean(), v[len(v) * 9 // 10], v[-1]))
print("  (ticks: 10 ns)")
# start skew and span (s_memrealtime: 10 ns ticks on one base for the whole chip)
base = t[:, 0].min()
st = np.sort(t[:, 0] - base) / 100.0
en = np.sort(t[:, 3] - base) / 100.0
print("  workgroup starts after the first one (us): median %.2f  p90 %.2f  max %.2f;  ends: median %.2f  p90 %.2f  last %.2f"
      % (st[len(st) // 2], st[len(st) * 9 // 10], st[-1], en[len(en) // 2], en[len(en) * 9 // 10], en[-1]))
# what a row's walk costs: lists (ids more than one genome holds) and holders on them, per row, against the stamped phases
o = off.cpu().numpy()
x = ids[:int(o[-1])].cpu().numpy().view(np.uint32)
u, inv, cnt = np.unique(x, return_inverse=True, return_counts=True)
c = cnt[inv]
Lr = np.array([(c[o[g]:o[g + 1]] > 1).sum() for g in range(G)])
Hr = np.array([c[o[g]:o[g + 1]][c[o[g]:o[g + 1]] > 1].sum() for g in range(G)])
M16 = np.array([(c[o[g]:o[g + 1]] > 16).sum() for g in range(G)])
print("  per row: ids %d..%d, lists median %d max %d, holders on lists median %d max %d, lists of more than 16 holders median %d max %d"
      % ((o[1:] - o[:-1]).min(), (o[1:] - o[:-1]).max(), np.median(Lr), Lr.max(), np.median(Hr), Hr.max(), np.median(M16), M16.max()))
order = np.argsort(Hr)
for lo, hi in ((0, G // 10), (G // 10, G // 2), (G // 2, G * 9 // 10), (G * 9 // 10, G)):
    r = order[lo:hi]
    print("  rows by holders [%4d, %4d): lists %6.0f  holders %7.0f  >16: %5.0f | probe %6.2f us  walk %6.2f  epilogue %5.2f  whole %6.2f  (means)"
          % (lo, hi, Lr[r].mean(), Hr[r].mean(), M16[r].mean(), d[r, 0].mean() / 100, d[r, 1].mean() / 100, d[r, 2].mean() / 100, tot[r].mean() / 100))
A = np.stack([np.ones(G), Lr, Hr, M16], 1)
coef, *_ = np.linalg.lstsq(A, d[:, 1] / 100.0, rcond=None)
print("  walk us ~ %.2f + %.4f lists + %.5f holders + %.4f long lists (least squares over the rows)" % tuple(coef))
