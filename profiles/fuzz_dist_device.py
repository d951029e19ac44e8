"""development tool (GPU box): the device-level search -- kssd_gpu_index_build_device (the capped build with rooms that are no powers of
two, one- and two-level partitions, the counting build behind an overflow), kssd_gpu_dist_device, kssd_gpu_dist_counts_device +
kssd_gpu_transpose_metrics_device, kssd_gpu_dist_device_transposed, with and without the negative filter -- on random reference / query
sets from a handful of sketches to tens of thousands (indexes of 1 to several thousand buckets), clades, duplicates, empty rows, ids that
do not spread, against the oracle's shared counts and the row-major call's metric bits.  python3 profiles/fuzz_dist_device.py [seeds]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np
import torch
import kssd_oracle as ko
import public_kssd_amd as K
from test_gpu_dist import random_sketches
shuf = K.Shuf.generate(10, 6, 3, seed=1)
dev = torch.device("cuda:0")
def on_dev(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
bad = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for seed in range(n_seeds):
    rng = np.random.default_rng(7000 + seed)
    ctx = K.GpuCtx(shuf, 0)
    shape = int(rng.integers(0, 5))
    if shape == 0:   nr, hi = int(rng.integers(1, 60)), 1300                 # a few sketches of genome size
    elif shape == 1: nr, hi = int(rng.integers(500, 4000)), 300               # thousands of small ones
    elif shape == 2: nr, hi = int(rng.integers(8000, 30000)), 120             # tens of thousands: several thousand buckets, two-level partition
    elif shape == 3: nr, hi = int(rng.integers(200, 1500)), 1300              # the bench's shape, smaller
    else:            nr, hi = int(rng.integers(50, 400)), 40                  # tiny sketches
    clades = int(rng.integers(1, 12))
    universe = 1 << int(rng.choice([12, 16, 22, 28]))                        # (a small universe: ids crowd into few buckets)
    universe = max(universe, hi * 2 * 2)
    roff, rids = random_sketches(rng, nr, 0, hi, universe, clades=clades)
    if shape == 4 and rng.random() < 0.5:                                     # ids that do not spread: one id in (nearly) every sketch
        rows = []
        for g in range(nr):
            r = rids[int(roff[g]):int(roff[g + 1])]
            rows.append(np.unique(np.concatenate([r, np.array([5], np.uint32)])) if len(r) else r)
        rids = np.concatenate(rows).astype(np.uint32) if rows else rids
        roff = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
    nq = int(rng.integers(1, 70))
    own = rng.random() < 0.5
    if own:   # queries = some of the references themselves + foreign ones (the all-pairs shape)
        pick = rng.integers(0, nr, nq)
        qids = np.concatenate([rids[int(roff[g]):int(roff[g + 1])] for g in pick]) if nq else np.zeros(0, np.uint32)
        qoff = np.concatenate([[0], np.cumsum([int(roff[g + 1] - roff[g]) for g in pick])]).astype(np.uint64)
    else:
        qoff, qids = random_sketches(rng, nq, 0, hi, universe, clades=int(rng.integers(1, 6)))
    if len(rids) == 0 or len(qids) == 0:
        ctx.close(); continue
    if os.environ.get("FUZZ_TRACE"): print("seed", seed, "shape", shape, "nr", nr, "nq", nq, "ids", len(rids), len(qids), "universe", universe, "own", own, flush=True)
    want = ko.shared_counts(roff, rids, qoff, qids)
    d_roff, d_rids = on_dev(roff, np.int64), on_dev(rids, np.int32)
    d_qoff, d_qids = on_dev(qoff, np.int64), on_dev(qids, np.int32)
    max_ref = len(rids)  # (the call's bound on the TOTAL number of reference ids)
    use_filter = rng.random() < 0.5
    if use_filter:
        ctx.index_set_filter(True, 0, int(rng.integers(0, nq + 1)))
    if rng.random() < 0.2:
        ctx.index_set_exact(True)
    ctx.index_build_device(d_roff, d_rids, nr, max_ref)
    if os.environ.get("FUZZ_TRACE"): torch.cuda.synchronize(); print("   done", 'ctx.index_build_device', flush=True)
    sh = torch.full((nq, nr), -7, dtype=torch.int32, device=dev)
    pl = [torch.zeros((nq, nr), dtype=torch.float64, device=dev) for _ in range(4)]
    ctx.dist_device(d_qoff, d_qids, nq, 0, nq, sh, *pl)
    if os.environ.get("FUZZ_TRACE"): torch.cuda.synchronize(); print("   done", 'ctx.dist_device', flush=True)
    torch.cuda.synchronize()
    got = sh.cpu().numpy().view(np.uint32)
    ok = np.array_equal(got, want)
    # the counts-only call + the transposing metrics kernel, and the fused transposed call: same counts, same metric bits
    cnt = torch.full((nq, nr), -7, dtype=torch.int32, device=dev)
    ctx.dist_counts_device(d_qoff, d_qids, nq, 0, nq, cnt)
    if os.environ.get("FUZZ_TRACE"): torch.cuda.synchronize(); print("   done", 'ctx.dist_counts_device', flush=True)
    sht = torch.full((nr, nq), -7, dtype=torch.int32, device=dev)
    plt = [torch.zeros((nr, nq), dtype=torch.float64, device=dev) for _ in range(4)]
    ctx.transpose_metrics_device(d_qoff, nq, 0, nq, cnt, nq, sht, *plt)
    if os.environ.get("FUZZ_TRACE"): torch.cuda.synchronize(); print("   done", 'ctx.transpose_metrics_device', flush=True)
    work = torch.zeros((nq, nr), dtype=torch.int32, device=dev)
    sht2 = torch.full((nr, nq), -7, dtype=torch.int32, device=dev)
    plt2 = [torch.zeros((nr, nq), dtype=torch.float64, device=dev) for _ in range(4)]
    ctx.dist_device_transposed(d_qoff, d_qids, nq, 0, nq, work, nq, sht2, *plt2)
    if os.environ.get("FUZZ_TRACE"): torch.cuda.synchronize(); print("   done", 'ctx.dist_device_transposed', flush=True)
    torch.cuda.synchronize()
    ok = ok and np.array_equal(cnt.cpu().numpy().view(np.uint32), want)
    ok = ok and torch.equal(sht.t().contiguous(), sh) and torch.equal(sht2.t().contiguous(), sh)
    for a, b, c in zip(pl, plt, plt2):
        ok = ok and torch.equal(a.view(torch.int64), b.t().contiguous().view(torch.int64)) and torch.equal(a.view(torch.int64), c.t().contiguous().view(torch.int64))
    if not ok:
        bad += 1
        print("seed", seed, "shape", shape, "nr", nr, "nq", nq, "universe", universe, "filter", use_filter, "MISMATCH", flush=True)
    ctx.close()
print("seeds", n_seeds, "bad", bad)
