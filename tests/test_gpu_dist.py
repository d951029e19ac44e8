"""-m gpu: inverted index + intersection + distances on the device against the CPU oracle."""
import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K

pytestmark = pytest.mark.gpu


def random_sketches(rng, n, size_lo, size_hi, universe, clades=4):
    """CSR of n sorted id sets with heavy within-clade sharing"""
    pools = [rng.choice(universe, size=size_hi * 2, replace=False) for _ in range(clades)]
    off = [0]
    ids = []
    for g in range(n):
        sz = int(rng.integers(size_lo, size_hi + 1))
        pool = pools[g % clades]
        own = rng.choice(pool, size=min(sz, len(pool)), replace=False)
        ids.append(np.sort(own).astype(np.uint32))
        off.append(off[-1] + len(own))
    return np.array(off, dtype=np.uint64), np.concatenate(ids) if ids else np.zeros(0, np.uint32)


def ulp_diff(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2 ** 63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2 ** 63) - ib[ib < 0]
    d = np.abs(ia - ib)
    d[both_nan] = 0
    return d


def test_shared_counts_and_metrics(gpu_ctx):
    rng = np.random.default_rng(21)
    roff, rids = random_sketches(rng, 37, 50, 400, 1 << 28)
    qoff, qids = random_sketches(rng, 23, 0, 300, 1 << 28)
    # make some queries share with references
    qids[: len(qids) // 2] = rng.choice(rids, size=len(qids) // 2)
    for i in range(len(qoff) - 1):
        s, e = int(qoff[i]), int(qoff[i + 1])
        u = np.unique(qids[s:e])
        # keep CSR sizes: refill duplicates with fresh ids
        fill = rng.choice(1 << 28, size=(e - s) - len(u), replace=False).astype(np.uint32)
        qids[s:e] = np.sort(np.concatenate([u, fill]))
    shared, J, MD, Cc, AD = gpu_ctx.dist(roff, rids, qoff, qids)
    want = ko.shared_counts(roff, rids, qoff, qids)
    assert np.array_equal(shared, want)
    X = np.diff(roff).astype(np.uint32)[None, :]
    Y = np.diff(qoff).astype(np.uint32)[:, None]
    oJ, oMD, oC, oAD = ko.metrics_arrays(X, Y, want, 20)
    assert ulp_diff(J, oJ).max() == 0          # one IEEE division
    assert ulp_diff(Cc, oC).max() == 0
    assert ulp_diff(MD, oMD).max() <= 1        # north_star tolerance: Mash / Aaf within 1 ulp
    assert ulp_diff(AD, oAD).max() <= 1


def test_all_pairs_self(gpu_ctx):
    rng = np.random.default_rng(4)
    off, ids = random_sketches(rng, 64, 900, 1300, 1 << 28, clades=5)
    shared = gpu_ctx.dist(off, ids, off, ids, planes=False)
    assert np.array_equal(shared, shared.T)
    assert np.array_equal(np.diag(shared), np.diff(off).astype(np.uint32))
    assert np.array_equal(shared, ko.shared_counts(off, ids, off, ids, threads=4))


def test_long_postings_and_empty_rows(gpu_ctx):
    # one id held by every reference (posting as long as the reference set), empty query, empty reference
    R = 300
    roff = np.arange(R + 1, dtype=np.uint64) * 2
    rids = np.empty(2 * R, np.uint32)
    rids[0::2] = 12345
    rids[1::2] = 1000000 + np.arange(R)
    roff = np.concatenate([roff, roff[-1:]])          # last reference is empty
    qoff = np.array([0, 0, 1, 3], dtype=np.uint64)
    qids = np.array([12345, 12345, 1000007], dtype=np.uint32)
    shared, J, MD, Cc, AD = gpu_ctx.dist(roff, rids, qoff, qids)
    want = ko.shared_counts(roff, rids, qoff, qids)
    assert np.array_equal(shared, want)
    assert shared[1, :R].tolist() == [1] * R and shared[2, 7] == 2 and shared[0].sum() == 0
    oJ, oMD, oC, oAD = ko.metrics_arrays(np.diff(roff).astype(np.uint32)[None, :], np.diff(qoff).astype(np.uint32)[:, None], want, 20)
    for a, b in ((J, oJ), (MD, oMD), (Cc, oC), (AD, oAD)):
        assert ulp_diff(a, b).max() <= 1
        assert np.array_equal(np.isnan(a), np.isnan(b))


def test_more_references_than_one_lds_row(gpu_ctx):
    """> 36 864 references: the row kernel tiles the reference axis (two tiles here)"""
    rng = np.random.default_rng(12)
    R = 40_000
    sizes = rng.integers(1, 6, R)
    roff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    common = np.array([7, 99, 4242], dtype=np.uint32)
    rids = np.empty(int(roff[-1]), np.uint32)
    for g in range(R):
        s, e = int(roff[g]), int(roff[g + 1])
        own = (1 << 20) + g * 8 + np.arange(e - s, dtype=np.uint32)
        if g % 3 == 0:
            own[0] = common[g % 3]  # id 7 sits in every third reference: a posting across both tiles
        rids[s:e] = np.sort(own)
    qoff = np.array([0, 3, 5], dtype=np.uint64)
    qids = np.array([7, 99, (1 << 20) + 39_999 * 8, 7, (1 << 20) + 8], dtype=np.uint32)
    qids[:3] = np.sort(qids[:3]); qids[3:] = np.sort(qids[3:])
    shared = gpu_ctx.dist(roff, rids, qoff, qids, planes=False)
    assert shared.shape == (2, R)
    assert np.array_equal(shared, ko.shared_counts(roff, rids, qoff, qids, threads=4))
    assert shared[0, 39_999] >= 1 and shared[:, ::3].min() >= 1
