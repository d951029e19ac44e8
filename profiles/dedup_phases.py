#!/usr/bin/env python3
"""development tool (GPU box): where the per-genome kernel's time goes (sketch_dedup_kernel<K, FUSED>), per workgroup, on the
bench's batch.  Needs the -DKSSD_DEV build; run as
    KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DEDUPTIME=1 [KSSD_DEV_GATHERSPLIT=1] python3 profiles/dedup_phases.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import public_kssd_amd as K
from public_kssd_amd import capi

G, L = 1000, 5_000_000
dev = torch.device("cuda", 0)
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
packed, mask, chunk_off, _ = bench.make_batch(G, L, 50, 20260101, dev)
ctx = K.GpuCtx(shuf, 0)
cap = int(G * L / 4096 * 1.25) + 4096
off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
ids = torch.zeros(cap, dtype=torch.int32, device=dev)
for rep in range(6):
    ctx.sketch_device(packed, mask, chunk_off, off, ids, cap)
torch.cuda.synchronize()
lib = capi.gpu_lib()
t = np.zeros(G * 4, dtype=np.uint64)
lib.kssd_gpu_dev_deduptimes.argtypes = [C.c_void_p, C.c_uint32]
rc = lib.kssd_gpu_dev_deduptimes(t.ctypes.data, G)
assert rc == 0, rc
t = t.reshape(G, 4).astype(np.int64)
d = np.diff(t, axis=1)
split = bool(os.environ.get("KSSD_DEV_GATHERSPLIT"))
names = ("start -> block table in LDS", "-> first round of candidates evaluated", "-> all keys in LDS") if split else \
        ("start -> keys in LDS", "-> sorted", "-> kept ids written")
for i, nm in enumerate(names):
    v = np.sort(d[:, i])
    print("  %-42s min %6.2f  median %6.2f  mean %6.2f  p90 %6.2f  max %6.2f us" % (nm, v[0] / 100, v[len(v) // 2] / 100, v.mean() / 100, v[len(v) * 9 // 10] / 100, v[-1] / 100))
tot = np.sort(t[:, 3] - t[:, 0])
print("  %-42s min %6.2f  median %6.2f  mean %6.2f  p90 %6.2f  max %6.2f us" % ("all stamped", tot[0] / 100, tot[len(tot) // 2] / 100, tot.mean() / 100, tot[len(tot) * 9 // 10] / 100, tot[-1] / 100))
base = t[:, 0].min()
st, en = np.sort(t[:, 0] - base) / 100.0, np.sort(t[:, 3] - base) / 100.0
print("  workgroup starts after the first one (us): median %.2f  p90 %.2f  max %.2f;  last stamp: median %.2f  p90 %.2f  last %.2f" % (st[len(st) // 2], st[len(st) * 9 // 10], st[-1], en[len(en) // 2], en[len(en) * 9 // 10], en[-1]))
