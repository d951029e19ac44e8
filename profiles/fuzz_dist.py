"""development tool (GPU box): kssd_gpu_dist on random reference / query sets (clades, duplicates, empty rows) against the oracle"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np
import kssd_oracle as ko
import public_kssd_amd as K
from test_gpu_dist import random_sketches
shuf = K.Shuf.generate(10, 6, 3, seed=1)
ctx = K.GpuCtx(shuf, 0)
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    rng = np.random.default_rng(seed)
    nr = int(rng.integers(1, 80)); nq = int(rng.integers(1, 40))
    roff, rids = random_sketches(rng, nr, 0, 1300, 1 << 28, clades=int(rng.integers(1, 8)))
    qoff, qids = random_sketches(rng, nq, 0, 1200, 1 << 28, clades=int(rng.integers(1, 8)))
    n = len(qids) // 3
    if n and len(rids):
        qids[:n] = rng.choice(rids, size=n)
    for i in range(len(qoff) - 1):
        s, e = int(qoff[i]), int(qoff[i + 1])
        u = np.unique(qids[s:e])
        fill = rng.choice(1 << 28, size=(e - s) - len(u), replace=False).astype(np.uint32)
        qids[s:e] = np.sort(np.concatenate([u, fill]))
    got = ctx.dist(roff, rids, qoff, qids, planes=False)
    want = ko.shared_counts(roff, rids, qoff, qids)
    if not np.array_equal(got, want):
        bad += 1
        d = np.argwhere(got != want)
        print("seed", seed, "nr", nr, "nq", nq, "mismatches", len(d), "first", d[0], got[tuple(d[0])], want[tuple(d[0])], flush=True)
print("bad", bad)
