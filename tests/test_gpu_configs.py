"""-m gpu: every BASELINE.json config through the HIP path at (scaled) size, with oracle parity.

  configs[1]  1 000 x 5 Mb, L3K10: ALL 1 000 sketches and the FULL 1 000 x 1 000 shared matrix against the oracle
              (the property checks of this config live in test_gpu_fullsize.py)
  configs[2]  10 000 genomes (the config-2 generator with 500 clades: no GTDB on the box), all-pairs 1e8 on ONE GPU:
              properties on the whole matrix + oracle parity on 128 genomes and their 128 x 10 000 rows
  configs[3]  FASTQ reads -> read-set sketch (-n 1, -n 2) -> containment (-M 1) against the 10 000 sketches.
              Scaled: 1 M x 150 bp reads from 16 of the genomes (the 100 M-read run with a 10 M-read oracle slice is
              `bench.py --workload fastq`, its JSON line is kept under profiles/)
  configs[4]  s7/l5 ("L5K10": .shuf from `shuffle -k 10 -s 7 -l 5`, 1 GiB table, 2^-16 pass rate): 8 x 250 Mb pieces
              (2 Gbase, ~30 000 ids) against the oracle, the reference's capacity abort (hashlimit 4 914,
              iseq2comem.c:262-263) on a 400 Mb piece, and the whole-genome run the reference cannot do
              (KSSD_SKETCH_NO_CAPACITY; scaled to one 1 Gb record, checked against the oracle with the limit lifted
              the same way: not a parity claim about the reference, which aborts)
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K
from synth import fasta_text, fastq_records, host_cores, sample_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORES = host_cores()


def _sketch_retry(ctx, packed, mask, chunk_off, off, ids, cap, flags=K.SKETCH_FASTA, min_occ=1):
    for attempt in range(8):
        ctx.sketch_device(packed, mask, chunk_off, off, ids, cap, flags, min_occ)
        rc, total, bad = ctx.sketch_status()
        if rc == 0:
            return int(total)
        assert rc == K.capi.ERR_OVERFLOW, rc
    raise AssertionError("sketch kept overflowing")


def _oracle_csr_in_slices(shuf, G, L, clades, seed, dev, slice_genomes=100):
    """the bench batch of G genomes on the device + the oracle's CSR of ALL of them (FASTA text of `slice_genomes`
    genomes at a time on the host: 1 000 x 5 Mb of text would be 5 GB)"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    pool = ThreadPoolExecutor(max_workers=min(16, CORES))
    pending, sizes, idchunks = [], [], []

    def flush():
        texts = list(pool.map(lambda a: fasta_text(a[1], b"g%d" % a[0], n_mask=a[2]), pending))
        off, ids = ko.sketch_texts(shuf.table, shuf.k, shuf.subk, shuf.drlevel, texts, threads=CORES)
        for g in range(len(texts)):
            idchunks.append(np.sort(ids[int(off[g]):int(off[g + 1])]))
            sizes.append(len(idchunks[-1]))
        pending.clear()

    def on_genome(gi, codes, nmask):
        pending.append((gi, codes.cpu().numpy(), nmask.cpu().numpy()))
        if len(pending) == slice_genomes:
            flush()
    packed, mask, chunk_off, _ = bench.make_batch(G, L, clades, seed, dev, on_genome=on_genome)
    if pending:
        flush()
    pool.shutdown()
    ooff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    return packed, mask, chunk_off, ooff, np.concatenate(idchunks).astype(np.uint32)


def test_config2_every_sketch_and_the_full_matrix_against_the_oracle(shuf_l3k10):
    """bit-exact vs the CPU oracle on all 1 000 sketches and the full 1 000 x 1 000 shared
    matrix; J and C bit-exact, MashD / AafD within 1 ulp (north_star tolerance) on every pair"""
    import torch
    dev = torch.device("cuda", 0)
    G, L = 1000, 5_000_000
    packed, mask, chunk_off, ooff, oids = _oracle_csr_in_slices(shuf_l3k10, G, L, 50, 20260101, dev)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        cap = int(G * L / 4096 * 1.25) + 4096
        off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
        ids = torch.zeros(cap, dtype=torch.int32, device=dev)
        total = _sketch_retry(ctx, packed, mask, chunk_off, off, ids, cap)
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), ooff), "sketch sizes differ from the oracle's"
        assert np.array_equal(ids[:total].cpu().numpy().view(np.uint32), oids), "sketch ids differ from the oracle's"
        shared = torch.zeros(G * G, dtype=torch.int32, device=dev)
        planes = [torch.zeros(G * G, dtype=torch.float64, device=dev) for _ in range(4)]
        ctx.index_build_device(off, ids, G, total)
        ctx.dist_device(off, ids, G, 0, G, shared, *planes)
        torch.cuda.synchronize()
        want = ko.shared_counts(ooff, oids, ooff, oids, threads=CORES)
        got = shared.cpu().numpy().view(np.uint32).reshape(G, G)
        assert np.array_equal(got, want), "shared-k-mer matrix differs from the oracle's"
        sz = np.diff(ooff).astype(np.uint32)
        oJ, oMD, oC, oAD = ko.metrics_batch(sz[None, :], sz[:, None], want, 20)
        J, MD, C, AD = [p.cpu().numpy().reshape(G, G) for p in planes]

        def ulps(a, b):
            return np.abs(a.view(np.int64) - b.view(np.int64))
        assert ulps(J, oJ).max() == 0 and ulps(C, oC).max() == 0
        assert ulps(MD, oMD).max() <= 1 and ulps(AD, oAD).max() <= 1
    finally:
        ctx.close()


@pytest.fixture(scope="module")
def refs_10k(shuf_l3k10):
    """configs[2] / [3]: 10 000 x 5 Mb (500 clades) packed on the device (18.8 GB) and sketched; keeps the CSR on the
    device, the codes of the first 128 genomes on the host, and frees the packed batch"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    G, L, SAMPLE = 10_000, 5_000_000, 128
    packed, mask, chunk_off, kept = bench.make_batch(G, L, 500, 20260101, dev, keep_codes=SAMPLE)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    cap = int(G * L / 4096 * 1.25) + 4096
    off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    ids = torch.zeros(cap, dtype=torch.int32, device=dev)
    total = _sketch_retry(ctx, packed, mask, chunk_off, off, ids, cap)
    # idempotence at this size: the same batch gives the same CSR, bit for bit
    off2 = torch.zeros_like(off)
    ids2 = torch.zeros_like(ids)
    assert _sketch_retry(ctx, packed, mask, chunk_off, off2, ids2, cap) == total
    assert torch.equal(off, off2) and torch.equal(ids[:total], ids2[:total])
    del packed, mask, off2, ids2
    torch.cuda.empty_cache()
    yield dict(ctx=ctx, G=G, L=L, off=off, ids=ids, total=total, kept=kept, dev=dev)
    ctx.close()


def test_config3_ten_thousand_genomes_all_pairs(refs_10k, shuf_l3k10):
    import torch
    r = refs_10k
    ctx, G, L, off, ids, total, dev = r["ctx"], r["G"], r["L"], r["off"], r["ids"], r["total"], r["dev"]
    sz = off[1:] - off[:-1]
    assert int(off[0]) == 0 and int(off[-1]) == total
    assert 1000 < int(sz.min()) and int(sz.max()) < 1500 and abs(float(sz.double().mean()) - L / 4096) < 10
    v = ids[:total].to(torch.int64)
    inc = v[1:] > v[:-1]
    inc[(off[1:-1] - 1).clamp(min=0)] = True
    assert bool(inc.all()) and int(v.max()) < (1 << 28) and int(v.min()) >= 1
    # oracle parity on the first 128 genomes
    SAMPLE = len(r["kept"])
    texts = [fasta_text(c, b"g%d" % g, n_mask=m) for g, (c, m) in enumerate(r["kept"])]
    ooff, oids = ko.sketch_texts(shuf_l3k10.table, 10, 6, 3, texts, threads=CORES)
    oh = off.cpu().numpy().astype(np.uint64)
    ih = ids[:total].cpu().numpy().view(np.uint32)
    osets = [np.sort(oids[int(ooff[g]):int(ooff[g + 1])]) for g in range(SAMPLE)]
    for g in range(SAMPLE):
        assert np.array_equal(ih[int(oh[g]):int(oh[g + 1])], osets[g]), g
    # all-pairs: 1e8 pairs, shared counts + the four planes (3.6 GB of output)
    shared = torch.zeros(G * G, dtype=torch.int32, device=dev)
    planes = [torch.zeros(G * G, dtype=torch.float64, device=dev) for _ in range(4)]
    ctx.index_build_device(off, ids, G, total)
    ctx.dist_device(off, ids, G, 0, G, shared, *planes)
    torch.cuda.synchronize()
    S32 = shared.view(G, G)
    assert torch.equal(S32, S32.t()), "shared counts are symmetric"
    assert torch.equal(S32.diagonal().to(torch.int64), sz), "a sketch shares all of itself"
    # checksum of checksums (computed without the rows kernel): matrix sum = sum over distinct ids of (holders)^2,
    # row sums = posting lengths of the row's ids
    uniq, inv, cnt = torch.unique(v, return_inverse=True, return_counts=True)
    assert int(S32.sum(dtype=torch.int64)) == int((cnt * cnt).sum())
    gid = torch.repeat_interleave(torch.arange(G, device=dev), sz)
    rows = torch.zeros(G, dtype=torch.int64, device=dev).index_add_(0, gid, cnt[inv])
    assert torch.equal(S32.sum(1, dtype=torch.int64), rows)
    del uniq, inv, cnt, gid
    # oracle rows for the sample: posting traversal of the oracle over the full reference CSR
    want = ko.shared_counts(oh, ih, np.concatenate([[0], np.cumsum([len(s) for s in osets])]).astype(np.uint64),
                            np.concatenate(osets), threads=CORES)
    got = S32[:SAMPLE].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want), "rows of the sample differ from the oracle's"
    szh = sz.cpu().numpy().astype(np.uint32)
    oJ, oMD, oC, oAD = ko.metrics_batch(szh[None, :], szh[:SAMPLE, None], want, 20)
    J, MD, C, AD = [p.view(G, G)[:SAMPLE].cpu().numpy() for p in planes]

    def ulps(a, b):
        return np.abs(a.view(np.int64) - b.view(np.int64))
    assert ulps(J, oJ).max() == 0 and ulps(C, oC).max() == 0
    assert ulps(MD, oMD).max() <= 1 and ulps(AD, oAD).max() <= 1
    # metric identities on the whole planes
    Jf, MDf, Cf, ADf = [p.view(G, G) for p in planes]
    X = sz.view(1, G).double()
    Y = sz.view(G, 1).double()
    Sd = S32.double()
    assert torch.equal(Jf, Sd / (X + Y - Sd))
    assert torch.equal(Cf, Sd / torch.minimum(X, Y))
    assert bool(((MDf >= 0) & (MDf <= 1) & (ADf >= 0) & (ADf <= 1)).all())
    assert bool((MDf[S32 == 0] == 1).all()) and bool((ADf[S32 == 0] == 1).all())


def test_config4_fastq_reads_to_containment(refs_10k, shuf_l3k10):
    """reads -> fastq2co sketch (-n 1 / -n 2, iseq2comem.c:277-356) -> containment row (-M 1, command_dist.c:1262-1265)
    against the 10 000 reference sketches.  1 M x 150 bp from 16 of the genomes, 0.5 % errors, both strands."""
    import torch
    r = refs_10k
    ctx, G, off, ids, total, dev = r["ctx"], r["G"], r["off"], r["ids"], r["total"], r["dev"]
    NSRC, NREADS = 16, 1_000_000
    reads = sample_reads([c for c, _ in r["kept"][:NSRC]], NREADS, 150, seed=404)
    fq = fastq_records(reads)
    del reads
    b = K.Batch()
    assert b.add_fastq(fq, Q=0) == 4 * NREADS
    sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
    oh = off.cpu().numpy().astype(np.uint64)
    ih = ids[:total].cpu().numpy().view(np.uint32)
    szh = np.diff(oh).astype(np.uint32)
    ctx.index_build_device(off, ids, G, total)
    for M in (1, 2):
        want = np.sort(sk.fastq(fq, Q=0, M=M))
        qoff, qids = ctx.sketch_batch(b, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY, min_occ=M)
        assert np.array_equal(qids, want), (M, len(qids), len(want))
        assert len(qids) > (5_000 if M == 1 else 2_500)   # 16 clade mates share most of their k-mers
        # containment of the read set in every reference
        d_qoff = torch.from_numpy(qoff.astype(np.int64)).to(dev)
        d_qids = torch.from_numpy(qids.view(np.int32)).to(dev)
        shared = torch.zeros(G, dtype=torch.int32, device=dev)
        planes = [torch.zeros(G, dtype=torch.float64, device=dev) for _ in range(4)]
        ctx.dist_device(d_qoff, d_qids, 1, 0, 1, shared, *planes)
        torch.cuda.synchronize()
        wshared = ko.shared_counts(oh, ih, qoff, qids, threads=CORES)
        assert np.array_equal(shared.cpu().numpy().view(np.uint32)[None, :], wshared), M
        qsz = np.array([[len(qids)]], dtype=np.uint32)
        oJ, oMD, oC, oAD = ko.metrics_batch(szh[None, :], qsz, wshared, 20)
        C, AD = planes[2].cpu().numpy()[None, :], planes[3].cpu().numpy()[None, :]
        assert np.array_equal(C.view(np.int64), oC.view(np.int64))
        assert np.abs(AD.view(np.int64) - oAD.view(np.int64)).max() <= 1
        # the reads come from 16 of the 20 members of clade 0: that clade holds the hits, the other 9 980 genomes
        # share next to nothing with the read set
        assert wshared[0, :NSRC].min() > (500 if M == 1 else 200) and wshared[0, 20:].max() <= 4


@pytest.fixture(scope="module")
def shuf_s7l5():
    return K.Shuf.generate(10, 7, 5, seed=20260105)


def test_config5_s7l5_chromosome_pieces_capacity_abort_and_whole_genome(shuf_s7l5):
    shuf = shuf_s7l5
    info = K.derive(10, 7, 5)
    assert info.hashsize == 8191 and info.hashlimit == 4914           # primer[5], 0.6 of it
    rng = np.random.default_rng(55)
    ctx = K.GpuCtx(shuf, 0)
    try:
        # (a) 8 chromosome-sized pieces of 250 Mb: ~3 800 ids each, 2 Gbase / ~30 000 ids in total, vs the oracle
        texts = []
        for i in range(8):
            codes = rng.integers(0, 4, 250_000_000, dtype=np.uint8)
            nm = np.zeros(len(codes), dtype=bool)
            nm[rng.integers(0, len(codes), 2000)] = True             # a sprinkle of N
            texts.append(fasta_text(codes, b"chr%d" % i, n_mask=nm))
            del codes, nm
        ooff, oids = ko.sketch_texts(shuf.table, 10, 7, 5, texts, threads=min(8, CORES))
        b = K.Batch()
        for t in texts:
            b.add_fasta(t)
        off, ids = ctx.sketch_batch(b)
        b.close()
        assert int(off[-1]) > 28_000
        for g in range(8):
            got = ids[int(off[g]):int(off[g + 1])]
            want = np.sort(oids[int(ooff[g]):int(ooff[g + 1])])
            assert np.array_equal(got, want), (g, len(got), len(want))
        # (b) a 400 Mb piece holds more than hashlimit = 4 914 distinct ids: the reference aborts ("the context space
        # is too crowd", iseq2comem.c:262-263), so does the oracle, so must the device -- naming the genome
        hdr = b">piece_400Mb\n"
        big = hdr + texts[0][texts[0].index(b"\n") + 1:] + texts[1][texts[1].index(b"\n") + 1:150_000_000] + b"\n"
        del texts
        sk = ko.Sketcher(shuf.table, 10, 7, 5)
        with pytest.raises(ko.OracleError) as oe:
            sk.fasta(big)
        assert oe.value.code == -2
        b = K.Batch()
        b.add_fasta(b">small\n" + big[len(hdr):2_000_000] + b"\n")
        b.add_fasta(big)
        with pytest.raises(K.KssdError) as e:
            ctx.sketch_batch(b)
        assert e.value.code == K.capi.ERR_CAPACITY and e.value.bad_genome == 1
        # (c) the same record with the capacity limit lifted (KSSD_SKETCH_NO_CAPACITY: what a whole 3 Gb genome needs;
        # the reference cannot produce this, so the check is against the oracle's id SET of the two halves it can do)
        off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
        b.close()
        got = ids[int(off[1]):int(off[2])]
        assert len(got) > 4914 and np.all(np.diff(got.astype(np.int64)) > 0)
        half = len(big) // 2
        cut = big.index(b"\n", half) + 1
        a_ids = sk.fasta(big[:cut])
        b_ids = sk.fasta(b">second half\n" + big[cut - 71:])            # overlaps the last full line: no k-mer is lost
        union = np.union1d(a_ids, b_ids)
        assert np.array_equal(got, union)
    finally:
        ctx.close()
