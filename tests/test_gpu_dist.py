"""-m gpu: inverted index + intersection + distances on the device against the CPU oracle."""
import os
import zlib

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


def test_kernel_timing_is_a_sample_of_the_launches(shuf_l3k10):
    """kssd_gpu_set_kernel_timing: every launch of the rows kernel carries events by default, every n-th one or none after the call;
    the results do not depend on it"""
    rng = np.random.default_rng(5)
    off, ids = random_sketches(rng, 40, 100, 300, 1 << 28, clades=3)
    want = ko.shared_counts(off, ids, off, ids)
    ctx = K.GpuCtx(shuf_l3k10)
    try:
        for every, timed in ((1, 6), (0, 0), (4, 2), (1, 6)):
            ctx.set_kernel_timing(every)
            ctx.kernel_time(1, reset=True)
            for _ in range(6):
                assert np.array_equal(ctx.dist(off, ids, off, ids, planes=False), want)
            ms, n = ctx.kernel_time(1, reset=True)
            assert n == timed and (ms > 0) == (timed > 0), (every, n, ms)
    finally:
        ctx.close()


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


def test_query_rows_sharded_over_a_device_list(gpu_ctx):
    """kssd_gpu_dist_multi: contiguous blocks of query rows, one per entry of the device list, every entry indexes all
    references (command_dist.c:774-785: one owner per row).  On a one-GPU box the entries name device 0 repeatedly;
    the blocks must assemble into exactly what one device computes -- counts and all four planes bit for bit."""
    rng = np.random.default_rng(77)
    roff, rids = random_sketches(rng, 57, 300, 900, 1 << 28, clades=6)
    qoff, qids = random_sketches(rng, 41, 0, 700, 1 << 28, clades=6)
    one = gpu_ctx.dist(roff, rids, qoff, qids)
    assert np.array_equal(one[0], ko.shared_counts(roff, rids, qoff, qids, threads=4))
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 7):
        got = K.dist_multi(devs, 20, roff, rids, qoff, qids)
        for a, b in zip(one, got):
            assert np.array_equal(a.view(np.int64) if a.dtype == np.float64 else a,
                                  b.view(np.int64) if b.dtype == np.float64 else b), devs
    # more entries than query rows, counts only, and an empty query set
    got = K.dist_multi([0] * 5, 20, roff, rids, qoff[:3], qids[:int(qoff[2])], planes=False)
    assert np.array_equal(got, one[0][:2])
    got = K.dist_multi([0, 0], 20, roff, rids, np.zeros(1, np.uint64), np.zeros(0, np.uint32), planes=False)
    assert got.shape == (0, 57)
    with pytest.raises(K.KssdError) as e:                   # a device that does not exist is an error, not a fallback
        K.dist_multi([0, 99], 20, roff, rids, qoff, qids, planes=False)
    assert e.value.code == K.capi.ERR_NO_DEVICE


def test_host_level_search_tiles_its_rows(gpu_ctx):
    """Q x R larger than one device tile (512 MiB of rows): the host-level call works the rows off in tiles with two
    buffers; 6 000 x 30 000 counts + one plane = 2.1 GB of output"""
    rng = np.random.default_rng(5)
    R, Q = 30_000, 6_000
    rsz = rng.integers(1, 4, R)
    roff = np.concatenate([[0], np.cumsum(rsz)]).astype(np.uint64)
    rids = rng.integers(1, 5000, int(roff[-1])).astype(np.uint32)      # small universe: long postings
    for g in range(R):                                                   # distinct inside a sketch
        s, e = int(roff[g]), int(roff[g + 1])
        rids[s:e] = np.sort(rids[s] + np.arange(e - s, dtype=np.uint32))
    qsz = rng.integers(0, 5, Q)
    qoff = np.concatenate([[0], np.cumsum(qsz)]).astype(np.uint64)
    qids = rng.integers(1, 5000, int(qoff[-1])).astype(np.uint32)
    for g in range(Q):
        s, e = int(qoff[g]), int(qoff[g + 1])
        qids[s:e] = np.sort(qids[s] + np.arange(e - s, dtype=np.uint32)) if e > s else qids[s:e]
    shared = np.zeros((Q, R), dtype=np.uint32)
    cont = np.zeros((Q, R), dtype=np.float64)
    K.capi._gck(K.gpu_lib().kssd_gpu_dist(gpu_ctx.h, roff.ctypes.data, rids.ctypes.data, R, qoff.ctypes.data, qids.ctypes.data, Q,
                                          shared.ctypes.data, None, None, cont.ctypes.data, None))
    want = ko.shared_counts(roff, rids, qoff, qids, threads=8)
    assert np.array_equal(shared, want)
    den = np.minimum(np.diff(roff)[None, :], np.diff(qoff)[:, None]).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        wc = want / den
    assert np.array_equal(np.isnan(cont), np.isnan(wc)) and np.array_equal(cont[~np.isnan(wc)], wc[~np.isnan(wc)])


def test_index_buckets_too_large_for_lds(gpu_ctx):
    """the index is built bucket by bucket (top bits of id * 0x9E3779B1); ids crafted to fall into ONE bucket make it
    larger than a workgroup's LDS table and send it down the in-HBM build -- same counts"""
    rng = np.random.default_rng(31)
    pool = rng.choice(1 << 28, size=400_000, replace=False).astype(np.uint64)
    mix = (pool * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    crowd = pool[(mix >> np.uint64(29)) == 0].astype(np.uint32)      # top 3 bits zero: bucket 0 of 8 (and of 2, 4)
    assert len(crowd) > 40_000
    R = 12
    off, ids = [0], []
    for g in range(R):                                               # 12 references x 500 ids, heavy sharing, 6 000 entries
        own = np.sort(rng.choice(crowd[:900], size=500, replace=False))
        ids.append(own)
        off.append(off[-1] + len(own))
    roff, rids = np.array(off, np.uint64), np.concatenate(ids)
    qoff = np.array([0, 700, 700, 1500], np.uint64)
    qids = np.concatenate([np.sort(rng.choice(crowd[:1200], 700, replace=False)), np.sort(rng.choice(crowd, 800, replace=False))]).astype(np.uint32)
    shared = gpu_ctx.dist(roff, rids, qoff, qids, planes=False)
    assert np.array_equal(shared, ko.shared_counts(roff, rids, qoff, qids))
    assert shared[0].min() > 100 and shared[1].sum() == 0
    # and a reference set of ONE huge sketch (a read set as reference): several workgroups walk one genome
    big = np.sort(rng.choice(1 << 28, size=300_000, replace=False)).astype(np.uint32)
    roff1 = np.array([0, len(big)], np.uint64)
    q = np.sort(np.concatenate([big[::7], rng.choice(1 << 28, 1000)])).astype(np.uint32)
    q = np.unique(q)
    got = gpu_ctx.dist(roff1, big, np.array([0, len(q)], np.uint64), q, planes=False)
    assert int(got[0, 0]) == len(np.intersect1d(q, big))
    gpu_ctx.index_set_exact(False)    # (the crafted bucket has sent the session's context to the counting build: back to the ordinary one)


def test_capped_and_exact_index_builds_agree(shuf_l3k10):
    """the ordinary index build gives every bucket the same room (no counting pass; the rows kernel needs no bucket
    descriptor); kssd_gpu_index_set_exact asks for the counting build of rounds 1 - 3.  Same counts and planes from both, on
    one bucket (few ids), 128 buckets and with empty sketches in between."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        for n_ref, lo, hi, clades in ((5, 0, 40, 2), (200, 900, 1300, 8), (40, 0, 3, 3)):
            roff, rids = random_sketches(rng, n_ref, lo, hi, 1 << 28, clades=clades)
            qoff, qids = random_sketches(rng, 31, 0, hi, 1 << 28, clades=clades)
            if len(rids) and len(qids):
                qids[: len(qids) // 2] = rng.choice(rids, size=len(qids) // 2)
                for i in range(len(qoff) - 1):
                    s_, e_ = int(qoff[i]), int(qoff[i + 1])
                    u = np.unique(qids[s_:e_])
                    fill = rng.choice(1 << 28, size=(e_ - s_) - len(u), replace=False).astype(np.uint32)
                    qids[s_:e_] = np.sort(np.concatenate([u, fill]))
            want = ko.shared_counts(roff, rids, qoff, qids, threads=4)
            d = [torch.from_numpy(a).to(dev) for a in (roff.astype(np.int64), rids.view(np.int32), qoff.astype(np.int64), qids.view(np.int32))]
            outs = []
            for exact in (False, True):
                ctx.index_set_exact(exact)
                ctx.index_build_device(d[0], d[1], n_ref, len(rids))
                assert ctx.index_status() == 0
                shared = torch.full((31 * n_ref,), -1, dtype=torch.int32, device=dev)
                planes = [torch.zeros(31 * n_ref, dtype=torch.float64, device=dev) for _ in range(4)]
                ctx.dist_device(d[2], d[3], 31, 0, 31, shared, *planes)
                torch.cuda.synchronize()
                assert np.array_equal(shared.cpu().numpy().view(np.uint32).reshape(31, n_ref), want), (n_ref, exact)
                outs.append([p.cpu().numpy().view(np.int64) for p in planes])
            for x, y in zip(*outs):
                assert np.array_equal(x, y)
    finally:
        ctx.close()


def test_index_overflow_is_reported_and_the_exact_build_takes_over(shuf_l3k10):
    """ids crafted into ONE bucket of the hash: the capped build meets a bucket fuller than its run, kssd_gpu_index_status
    says KSSD_ERR_OVERFLOW (kssd_gpu_dist_device computes nothing on that index), the next build of the context counts first
    and its status is clean -- counts equal the oracle's.  A database of one sketch held by hundreds of genomes (every id with
    hundreds of holders: bucket sizes come in blocks) takes the same way."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    pool = rng.choice(1 << 28, size=400_000, replace=False).astype(np.uint64)
    mix = (pool * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    crowd = pool[(mix >> np.uint64(29)) == 0].astype(np.uint32)
    R = 12
    off, ids = [0], []
    for g in range(R):
        own = np.sort(rng.choice(crowd[:900], size=500, replace=False))
        ids.append(own)
        off.append(off[-1] + len(own))
    roff, rids = np.array(off, np.uint64), np.concatenate(ids)
    qoff = np.array([0, 700, 700, 1500], np.uint64)
    qids = np.concatenate([np.sort(rng.choice(crowd[:1200], 700, replace=False)), np.sort(rng.choice(crowd, 800, replace=False))]).astype(np.uint32)
    want = ko.shared_counts(roff, rids, qoff, qids)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        d = [torch.from_numpy(a).to(dev) for a in (roff.astype(np.int64), rids.view(np.int32), qoff.astype(np.int64), qids.view(np.int32))]
        ctx.index_build_device(d[0], d[1], R, len(rids), check=False)   # (the unchecked call of a timed loop)
        shared = torch.full((3 * R,), 7, dtype=torch.int32, device=dev)
        ctx.dist_device(d[2], d[3], 3, 0, 3, shared)              # nothing is computed on an index that is not whole ...
        torch.cuda.synchronize()
        assert int((shared == -1).sum().item()) == 3 * R          # ... and the stale 7s do not pass for an answer: every count is 0xFFFFFFFF
        assert ctx.index_status() == K.capi.ERR_OVERFLOW          # 6 000 entries in one bucket of four, whose run holds 4 095
        ctx.index_build_device(d[0], d[1], R, len(rids), check=False)   # the context counts first from now on
        assert ctx.index_status() == 0
        ctx.dist_device(d[2], d[3], 3, 0, 3, shared)
        torch.cuda.synchronize()
        assert np.array_equal(shared.cpu().numpy().view(np.uint32).reshape(3, R), want)
        ctx.index_set_exact(False)
        ctx.index_build_device(d[0], d[1], R, len(rids))          # the binding's default: the status is read and the build repeated
        assert ctx.index_status() == 0
        shared.fill_(7)
        ctx.dist_device(d[2], d[3], 3, 0, 3, shared)
        torch.cuda.synchronize()
        assert np.array_equal(shared.cpu().numpy().view(np.uint32).reshape(3, R), want)
        ctx.index_set_exact(False)
        # host level: the retry is inside
        assert np.array_equal(ctx.dist(roff, rids, qoff, qids, planes=False), want)
        ctx.index_set_exact(False)
        # 600 copies of one sketch of 1 200 ids: 720 000 entries in 512 buckets of up to 4 096, ~2.3 ids x 600 holders each
        one = np.sort(rng.choice(1 << 28, size=1200, replace=False)).astype(np.uint32)
        roff2 = (np.arange(601, dtype=np.uint64) * 1200)
        rids2 = np.tile(one, 600)
        q2 = np.sort(np.concatenate([one[::2], rng.choice(1 << 28, 300).astype(np.uint32)]))
        q2 = np.unique(q2)
        got = ctx.dist(roff2, rids2, np.array([0, len(q2)], np.uint64), q2, planes=False)
        assert np.array_equal(got, ko.shared_counts(roff2, rids2, np.array([0, len(q2)], np.uint64), q2, threads=4))
    finally:
        ctx.close()


def _sketchset(names, off, ids, kmerlen=20, dim_rd_len=6):
    return K.SketchSet(123, kmerlen, dim_rd_len, 1, names, off, ids)


@pytest.mark.parametrize("opts", [dict(metric=0, n_max=3), dict(metric=1, n_max=1), dict(metric=0, dthreshold="q30"),
                                  dict(metric=1, dthreshold="q20", correction=1), dict(metric=0, n_max=5, dthreshold="q50"),
                                  dict(metric=1, dthreshold="q40", pfield=0), dict(metric=0, correction=1, n_max=4, pfield=1)])
def test_report_selection_on_the_device_leaves_the_same_text(gpu_ctx, tmp_path, opts):
    """kssd_gpu_dist_select + kssd_distance_print_pairs against the dense report (whose text the golden tests pin to the
    reference's): byte-identical distance.out for -N, -D, --correction, both metrics, and far fewer pairs formatted"""
    rng = np.random.default_rng(zlib.crc32(str(sorted(opts.items())).encode()))   # (hash() of a str differs from process to process)
    roff, rids = random_sketches(rng, 60, 800, 1300, 1 << 28, clades=6)
    qoff, qids = random_sketches(rng, 25, 0, 1200, 1 << 28, clades=6)
    n = len(qids) // 3
    qids[:n] = rng.choice(rids, size=n)                                   # relatives among the references
    for i in range(len(qoff) - 1):
        s, e = int(qoff[i]), int(qoff[i + 1])
        u = np.unique(qids[s:e])
        fill = rng.choice(1 << 28, size=(e - s) - len(u), replace=False).astype(np.uint32)
        qids[s:e] = np.sort(np.concatenate([u, fill]))
    ref = _sketchset(["ref/r%03d.fa" % i for i in range(60)], roff, rids)
    qry = _sketchset(["qry/q%03d.fa" % i for i in range(25)], qoff, qids)
    opts = dict(opts)
    if isinstance(opts.get("dthreshold"), str):       # a threshold that cuts through the related pairs: a quantile of their distances
        sh0 = ko.shared_counts(roff, rids, qoff, qids)
        _, md, _, ad = ko.metrics_batch(np.diff(roff).astype(np.uint32)[None, :], np.diff(qoff).astype(np.uint32)[:, None], sh0, 20)
        dd = (ad if opts.get("metric") else md)[sh0 > 0]
        opts["dthreshold"] = float(np.quantile(dd[np.isfinite(dd)], int(opts["dthreshold"][1:]) / 100.0))
    poff, pref, psh, dense = gpu_ctx.dist_select(roff, rids, qoff, qids, metric=opts.get("metric", 0), correction=opts.get("correction", 0),
                                                 dim_rd_len=6, dthreshold=opts.get("dthreshold", 1.0), n_max=opts.get("n_max", 0), dense=True)
    want_shared = ko.shared_counts(roff, rids, qoff, qids)
    assert np.array_equal(dense, want_shared)
    for q in range(25):                                                       # the candidates carry the right counts, ascending
        r = pref[int(poff[q]):int(poff[q + 1])]
        assert np.all(np.diff(r.astype(np.int64)) > 0)
        assert np.array_equal(psh[int(poff[q]):int(poff[q + 1])], want_shared[q, r])
    if not (opts.get("correction") and not opts.get("n_max")):
        assert int(poff[-1]) < 25 * 60 // 2                                  # a real selection
    # (with --correction and no -N the unrelated pairs stay: shared = 0 gives a negative corrected metric, its distance is
    # NaN, NaN compares false against -D and the reference prints the line, command_dist.c:1262-1267)
    a, b = str(tmp_path / "dense.out"), str(tmp_path / "pairs.out")
    K.distance_print(a, want_shared, ref, qry, threads=2, **opts)
    K.distance_print_pairs(b, poff, pref, psh, ref, qry, threads=2, **opts)
    ta, tb = open(a, "rb").read(), open(b, "rb").read()
    assert ta == tb and ta.count(b"\n") > 1


def test_long_query_rows_are_shared_by_several_workgroups(gpu_ctx):
    """a read set sketched as one genome is ONE query row of hundreds of thousands of ids: beyond 16 384 ids the row is
    split over workgroups (atomic sums in the output row, the last one computes the metrics) -- same counts, same planes"""
    rng = np.random.default_rng(101)
    roff, rids = random_sketches(rng, 300, 900, 1300, 1 << 28, clades=10)
    long_q = np.unique(np.concatenate([rng.choice(rids, 60_000), rng.choice(1 << 28, 140_000)])).astype(np.uint32)
    short_q = np.sort(rng.choice(rids, 700, replace=False)).astype(np.uint32)
    short_q = np.unique(short_q)
    qoff = np.array([0, len(long_q), len(long_q), len(long_q) + len(short_q)], dtype=np.uint64)   # long, empty, short
    qids = np.concatenate([long_q, short_q])
    shared, J, MD, Cc, AD = gpu_ctx.dist(roff, rids, qoff, qids)
    want = ko.shared_counts(roff, rids, qoff, qids, threads=4)
    assert np.array_equal(shared, want) and want[0].min() > 20
    X = np.diff(roff).astype(np.uint32)[None, :]
    Y = np.diff(qoff).astype(np.uint32)[:, None]
    oJ, oMD, oC, oAD = ko.metrics_batch(X, Y, want, 20)
    ok = Y[:, 0] > 0                                                        # (the empty row: 0/0 everywhere, NaN on both sides)
    assert ulp_diff(J[ok], oJ[ok]).max() == 0 and ulp_diff(Cc[ok], oC[ok]).max() == 0
    assert ulp_diff(MD[ok], oMD[ok]).max() <= 1 and ulp_diff(AD[ok], oAD[ok]).max() <= 1
    # device-level entry with the hint, counts only, against the one-workgroup-per-row kernel
    import torch
    dev = torch.device("cuda", 0)
    d_roff = torch.from_numpy(roff.astype(np.int64)).to(dev)
    d_rids = torch.from_numpy(rids.view(np.int32)).to(dev)
    d_qoff = torch.from_numpy(qoff.astype(np.int64)).to(dev)
    d_qids = torch.from_numpy(qids.view(np.int32)).to(dev)
    gpu_ctx.index_build_device(d_roff, d_rids, 300, len(rids))
    a = torch.zeros(3 * 300, dtype=torch.int32, device=dev)
    b = torch.full((3 * 300,), -1, dtype=torch.int32, device=dev)
    gpu_ctx.dist_device(d_qoff, d_qids, 3, 0, 3, a)
    gpu_ctx.dist_device(d_qoff, d_qids, 3, 0, 3, b, max_row_ids=len(long_q))
    torch.cuda.synchronize()
    assert torch.equal(a, b) and np.array_equal(a.cpu().numpy().view(np.uint32).reshape(3, 300), want)


def test_negative_filter_in_front_of_the_index_changes_nothing(shuf_l3k10):
    """kssd_gpu_index_set_filter: a per-bucket Bloom filter consulted before the table is walked (for searches whose rows
    mostly miss: the foreign rows of the multi-GPU partition).  Same counts and planes with it, without it, and with a
    block of rows exempt; also through a bucket built in HBM."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    roff, rids = random_sketches(rng, 200, 900, 1300, 1 << 28, clades=8)
    qoff, qids = random_sketches(rng, 150, 0, 1300, 1 << 28, clades=8)        # other pools: nearly all ids miss
    n = len(qids) // 4
    qids[:n] = rng.choice(rids, size=n)                                        # ... except a quarter
    for i in range(len(qoff) - 1):
        s, e = int(qoff[i]), int(qoff[i + 1])
        u = np.unique(qids[s:e])
        fill = rng.choice(1 << 28, size=(e - s) - len(u), replace=False).astype(np.uint32)
        qids[s:e] = np.sort(np.concatenate([u, fill]))
    want = ko.shared_counts(roff, rids, qoff, qids, threads=4)
    assert want.sum() > 10_000
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        d = [torch.from_numpy(a).to(dev) for a in (roff.astype(np.int64), rids.view(np.int32), qoff.astype(np.int64), qids.view(np.int32))]
        outs = []
        for enable, lo, hi in ((False, 0, 0), (True, 0, 0), (True, 40, 90), (True, 0, 150)):
            ctx.index_set_filter(enable, lo, hi)
            ctx.index_build_device(d[0], d[1], 200, len(rids))
            shared = torch.full((150 * 200,), -1, dtype=torch.int32, device=dev)
            planes = [torch.zeros(150 * 200, dtype=torch.float64, device=dev) for _ in range(4)]
            ctx.dist_device(d[2], d[3], 150, 0, 150, shared, *planes)
            torch.cuda.synchronize()
            assert np.array_equal(shared.cpu().numpy().view(np.uint32).reshape(150, 200), want), (enable, lo, hi)
            outs.append([p.cpu().numpy().view(np.int64) for p in planes])
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert np.array_equal(a, b)
        # ids crafted into one bucket: the bucket (and its filter words) are built in HBM
        pool = rng.choice(1 << 28, size=400_000, replace=False).astype(np.uint64)
        mix = (pool * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
        crowd = pool[(mix >> np.uint64(29)) == 0].astype(np.uint32)
        r2 = np.sort(rng.choice(crowd[:7000], 6000, replace=False))
        roff2 = np.array([0, 3000, 6000], np.uint64)
        r2[:3000].sort(); r2[3000:].sort()
        q2 = np.sort(rng.choice(crowd, 5000, replace=False))
        qoff2 = np.array([0, 5000], np.uint64)
        ctx.index_set_filter(True, 0, 0)
        got = ctx.dist(roff2, r2, qoff2, q2, planes=False)
        assert np.array_equal(got, ko.shared_counts(roff2, r2, qoff2, q2))
    finally:
        ctx.close()


def test_in_process_exchange_over_rccl_with_one_rank(shuf_l3k10):
    """kssd_gpu_allgather_sketches (csrc/kssd_xchg.inc): the C product's exchange.  The box has one GPU, so the communicator has
    one rank: librccl is opened, a communicator made, both all-gathers and the unpacking run -- the gathered CSR is the
    rank's own, padding never surfaces, and a second call re-uses the communicator.  (N > 1 ranks: the N-GPU bench.)"""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    G, unit = 37, 5000
    sizes = rng.integers(0, 120, G)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ids = np.full(unit, -7, dtype=np.int32)
    ids[:off[-1]] = rng.integers(0, 1 << 28, int(off[-1]))
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        d_off, d_ids = torch.from_numpy(off).to(dev), torch.from_numpy(ids).to(dev)
        roff = torch.zeros(G + 1, dtype=torch.int64, device=dev)
        rids = torch.full((unit,), -1, dtype=torch.int32, device=dev)
        for _ in range(2):
            K.GpuCtx.allgather_sketches([ctx], [d_off], [d_ids], G, unit, [roff], [rids])
            torch.cuda.synchronize()
            assert np.array_equal(roff.cpu().numpy(), off)
            assert np.array_equal(rids.cpu().numpy()[:off[-1]], ids[:off[-1]])
        # two ranks on one device: refused before RCCL is asked
        with pytest.raises(K.KssdError):
            K.GpuCtx.allgather_sketches([ctx, ctx], [d_off, d_off], [d_ids, d_ids], G, unit, [roff, roff], [rids, rids])
    finally:
        ctx.close()


def test_sketch_gather_over_the_rccl_backend_with_one_rank(shuf_l3k10):
    """torch.distributed's `nccl` backend IS RCCL on this box: the exchange code of the N-GPU bench (shard.SketchGather: two
    all_gather_into_tensor + the unpacking kernel) run through it with a world of one -- the backend initialises, the
    collectives run on device tensors, the CSR comes out.  (More ranks need more GPUs: the driver's scaling run.)"""
    import socket
    import torch
    import torch.distributed as dist
    from public_kssd_amd.shard import SketchGather
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        assert dist.get_backend() == "nccl"
        rng = np.random.default_rng(9)
        G, unit = 50, 4096
        sizes = rng.integers(0, 80, G)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        ids = np.full(unit, -3, dtype=np.int32)
        ids[:off[-1]] = rng.integers(0, 1 << 28, int(off[-1]))
        g = SketchGather(1, G, unit, dev, ctx)
        roff, rids = g(torch.from_numpy(off).to(dev), torch.from_numpy(ids).to(dev))
        torch.cuda.synchronize()
        assert np.array_equal(roff.cpu().numpy(), off) and np.array_equal(rids.cpu().numpy()[:off[-1]], ids[:off[-1]])
    finally:
        ctx.close()
        dist.destroy_process_group()


def test_rows_written_transposed_carry_the_bits_of_the_rows_kernel(shuf_l3k10):
    """kssd_gpu_dist_device_transposed (the own-index partition of the multi-GPU all-pairs run): counts row-major by query, then
    the transposing metrics kernel -- element (reference r, query q) at r * pitch + (q - q_begin).  Against kssd_gpu_dist_device
    on the same index: counts equal the oracle's, every plane's bits equal the rows kernel's epilogue (it is the same device
    function), for sizes that are no multiples of the 64 x 64 tile, empty sketches on both sides, a sub-range of the rows, a
    pitch wider than the range, planes left out, the negative filter on"""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        for n_ref, n_qry, filt in ((70, 131, False), (1, 5, False), (65, 64, True), (200, 333, True)):
            roff, rids = random_sketches(rng, n_ref, 0, 600, 1 << 28, clades=5)
            qoff, qids = random_sketches(rng, n_qry, 0, 500, 1 << 28, clades=5)
            if len(rids) and len(qids):
                qids[: len(qids) // 2] = rng.choice(rids, size=len(qids) // 2)
                for i in range(n_qry):
                    s_, e_ = int(qoff[i]), int(qoff[i + 1])
                    u = np.unique(qids[s_:e_])
                    fill = rng.choice(1 << 28, size=(e_ - s_) - len(u), replace=False).astype(np.uint32)
                    qids[s_:e_] = np.sort(np.concatenate([u, fill]))
            want = ko.shared_counts(roff, rids, qoff, qids, threads=4)
            d = [torch.from_numpy(a).to(dev) for a in (roff.astype(np.int64), rids.view(np.int32), qoff.astype(np.int64), qids.view(np.int32))]
            ctx.index_set_filter(filt, 3, 9)
            ctx.index_build_device(d[0], d[1], n_ref, len(rids))
            shared = torch.zeros(n_qry * n_ref, dtype=torch.int32, device=dev)
            planes = [torch.zeros(n_qry * n_ref, dtype=torch.float64, device=dev) for _ in range(4)]
            ctx.dist_device(d[2], d[3], n_qry, 0, n_qry, shared, *planes)
            torch.cuda.synchronize()
            assert np.array_equal(shared.cpu().numpy().view(np.uint32).reshape(n_qry, n_ref), want)
            ref_planes = [p.cpu().numpy().view(np.int64).reshape(n_qry, n_ref) for p in planes]
            for q0, q1, pitch, with_planes in ((0, n_qry, n_qry, True), (n_qry // 3, n_qry - 1, n_qry + 13, True), (2, 2, 5, True), (0, n_qry, n_qry, False)):
                rows = q1 - q0
                work = torch.zeros(max(1, rows * n_ref), dtype=torch.int32, device=dev)
                sh_t = torch.full((n_ref * pitch,), -5, dtype=torch.int32, device=dev)
                pl_t = [torch.full((n_ref * pitch,), -5.0, dtype=torch.float64, device=dev) for _ in range(4)] if with_planes else [None] * 4
                ctx.dist_device_transposed(d[2], d[3], n_qry, q0, q1, work, pitch, sh_t, *pl_t)
                torch.cuda.synchronize()
                got = sh_t.cpu().numpy().view(np.uint32).reshape(n_ref, pitch)
                assert np.array_equal(got[:, :rows], want[q0:q1].T), (n_ref, n_qry, q0, q1)
                assert (got[:, rows:] == np.uint32(0xFFFFFFFB)).all()                      # nothing beyond the range is touched
                if with_planes:
                    for p, w in zip(pl_t, ref_planes):
                        g = p.cpu().numpy().view(np.int64).reshape(n_ref, pitch)
                        assert np.array_equal(g[:, :rows], w[q0:q1].T), (n_ref, n_qry, q0, q1)
        ctx.index_set_filter(False)
    finally:
        ctx.close()


def test_index_of_more_than_2048_buckets_is_partitioned_in_two_levels(shuf_l3k10):
    """5.2 M ids: 4 096 buckets of a room that is no power of two (a multiple of 64), the entries partitioned by super-bucket first and
    by bucket inside it then (idx_scatter_tile_kernel + idx_scatter_sub_kernel); KSSD_INDEX_ONE_LEVEL=1 keeps the one-level pass of
    rounds 1 - 4.  Counts against the oracle's posting traversal either way, with and without the negative filter, and the two
    builds' metric planes carry the same bits."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(123)
    roff, rids = random_sketches(rng, 4000, 1250, 1350, 1 << 28, clades=200)
    qoff, qids = random_sketches(rng, 40, 0, 1300, 1 << 28, clades=7)
    n = len(qids) // 2
    qids[:n] = rng.choice(rids, size=n)
    for i in range(len(qoff) - 1):
        s_, e_ = int(qoff[i]), int(qoff[i + 1])
        u = np.unique(qids[s_:e_])
        fill = rng.choice(1 << 28, size=(e_ - s_) - len(u), replace=False).astype(np.uint32)
        qids[s_:e_] = np.sort(np.concatenate([u, fill]))
    want = ko.shared_counts(roff, rids, qoff, qids, threads=8)
    assert want.sum() > 20_000
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        d = [torch.from_numpy(a).to(dev) for a in (roff.astype(np.int64), rids.view(np.int32), qoff.astype(np.int64), qids.view(np.int32))]
        outs = []
        for one_level, filt in ((False, False), (True, False), (False, True)):
            if one_level:
                os.environ["KSSD_INDEX_ONE_LEVEL"] = "1"
            try:
                ctx.index_set_filter(filt, 3, 11)
                for _ in range(2):                              # (twice: the cursors are back at zero after a build)
                    ctx.index_build_device(d[0], d[1], 4000, len(rids))
                assert ctx.index_status() == 0
            finally:
                os.environ.pop("KSSD_INDEX_ONE_LEVEL", None)
            shared = torch.full((40 * 4000,), -1, dtype=torch.int32, device=dev)
            planes = [torch.zeros(40 * 4000, dtype=torch.float64, device=dev) for _ in range(4)]
            ctx.dist_device(d[2], d[3], 40, 0, 40, shared, *planes)
            torch.cuda.synchronize()
            assert np.array_equal(shared.cpu().numpy().view(np.uint32).reshape(40, 4000), want), (one_level, filt)
            outs.append([p.cpu().numpy().view(np.int64) for p in planes])
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert np.array_equal(a, b)
        ctx.index_set_filter(False)
    finally:
        ctx.close()



def test_index_bound_that_is_too_small_is_reported_not_followed(shuf_l3k10):
    """kssd_gpu_index_build_device sizes the index's arrays from max_ref_ids, the caller's bound on d_roff[n_ref] (the TOTAL of the
    references' ids) -- the real total is only read on the device.  A bound that is too small (here: the largest sketch's size where the
    total belongs: what profiles/fuzz_dist_device.py first passed, and the GPU faulted) takes the first max_ref_ids entries and nothing
    behind the arrays; kssd_gpu_index_status says KSSD_ERR_PARAM; the same context then builds and searches with the right bound."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99)
    roff, rids = random_sketches(rng, 812, 100, 1300, 1 << 28, clades=5)
    qoff, qids = roff[:41].copy(), rids[: int(roff[40])].copy()
    want = ko.shared_counts(roff, rids, qoff, qids, threads=4)
    d = [torch.from_numpy(a).to(dev) for a in (roff.astype(np.int64), rids.view(np.int32), qoff.astype(np.int64), qids.view(np.int32))]
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        for exact in (False, True):
            ctx.index_set_exact(exact)
            for bound in (int(np.diff(roff).max()), len(rids) // 2, len(rids) - 1):
                ctx.index_build_device(d[0], d[1], 812, bound, check=False)
                assert ctx.index_status() == K.capi.ERR_PARAM, (exact, bound)
                with pytest.raises(K.KssdError):
                    ctx.index_build_device(d[0], d[1], 812, bound)
            ctx.index_build_device(d[0], d[1], 812, len(rids))
            shared = torch.zeros((40, 812), dtype=torch.int32, device=dev)
            ctx.dist_device(d[2], d[3], 40, 0, 40, shared)
            torch.cuda.synchronize()
            assert np.array_equal(shared.cpu().numpy().view(np.uint32), want)
    finally:
        ctx.close()
