#!/usr/bin/env python3
"""development tool (GPU box): phases of the items of a large genome (sketch_dedup_kernel<K, RANGES>) on a read-set-like
input: a 1 Mb genome at coverage ~190 plus singletons.  Needs the -DKSSD_DEV build:
    KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_DEDUPTIME=1 python3 profiles/ranges_phases.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import public_kssd_amd as K
from public_kssd_amd import capi
from synth import fasta_text

rng = np.random.default_rng(1)
shuf = K.Shuf.generate(8, 5, 2, seed=5)
g = rng.integers(0, 4, 300_000, dtype=np.uint8)
parts = [g[int(o):] for o in rng.integers(0, 1000, 190)] + [rng.integers(0, 4, 10_000_000, dtype=np.uint8)]
text = fasta_text(np.concatenate(parts), b"covered")
ctx = K.GpuCtx(shuf, 0)
ctx.set_lds_sort_limit(1024)
b = K.Batch()
b.add_fasta(text)
for rep in range(3):
    t0 = time.time()
    off, ids, cnt = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY | K.SKETCH_COUNTS)
    print("call %.1f ms, ids %d, max count %d, staged about %d" % ((time.time() - t0) * 1e3, len(ids), cnt.max(), cnt.sum()))
lib = capi.gpu_lib()
N = 4096
t = np.zeros(N * 4, dtype=np.uint64)
lib.kssd_gpu_dev_deduptimes.argtypes = [C.c_void_p, C.c_uint32]
assert lib.kssd_gpu_dev_deduptimes(t.ctypes.data, N) == 0
t = t.reshape(N, 4).astype(np.int64)
t = t[t[:, 3] > 0]
d = np.diff(t, axis=1)
print("items with time stamps:", len(t))
for i, nm in enumerate(("bins found, keys in LDS", "sort", "runs + keep rules + write")):
    v = np.sort(d[:, i])
    print("  %-30s min %8d  median %8d  mean %10.0f  p90 %8d  max %8d ticks" % (nm, v[0], v[len(v) // 2], v.mean(), v[len(v) * 9 // 10], v[-1]))
