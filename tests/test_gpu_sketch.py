"""-m gpu: the HIP sketch path, through the C ABI, against the CPU oracle (bit-exact id sets)."""
import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K
from synth import clade_genomes, fasta_text, fastq_text

pytestmark = pytest.mark.gpu


def csr_sets(off, ids):
    return [ids[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


def oracle_sets(shuf, texts, uniq=False):
    sk = ko.Sketcher(shuf.table, shuf.k, shuf.subk, shuf.drlevel)
    out = []
    for t in texts:
        ids, comps = sk.fasta(t, uniq=uniq, with_comps=True)
        out.append(np.sort((ids.astype(np.uint64) << np.uint64(sk.p.comp_bits)) | comps.astype(np.uint64)).astype(np.uint32))
    return out


def check(ctx, shuf, texts, flags=K.SKETCH_FASTA, uniq=False):
    b = K.Batch()
    for t in texts:
        b.add_fasta(t)
    off, ids = ctx.sketch_batch(b, flags)
    got = csr_sets(off, ids)
    want = oracle_sets(shuf, texts, uniq)
    assert len(got) == len(want)
    for g, (a, w) in enumerate(zip(got, want)):
        assert np.array_equal(a, w), "genome %d: %d vs %d ids" % (g, len(a), len(w))
    return off, ids


def test_clade_genomes_l3k10(gpu_ctx, shuf_l3k10):
    gs = clade_genomes(3, 4, 300_000, seed=11)
    texts = [fasta_text(c, nm, n_mask=m) for nm, c, m in gs]
    off, ids = check(gpu_ctx, shuf_l3k10, texts)
    sizes = np.diff(off)
    assert sizes.min() > 40 and sizes.max() < 120  # ~ 300000 / 4096


def test_edge_cases_l3k10(gpu_ctx, shuf_l3k10):
    rng = np.random.default_rng(5)

    def rnd(n):
        return rng.integers(0, 4, n, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    lower = bytes(acgt[rnd(50000)]).lower()
    texts = [
        b">only header and a short record\nACGTACGTAC\n",                      # shorter than 2k
        b">a\n" + bytes(acgt[rnd(19)]) + b"\n",                                  # 2k-1 bases: nothing
        b">a\n" + bytes(acgt[rnd(20)]) + b"\n",                                  # exactly one window
        b">lower\n" + lower + b"\n",                                              # soft-masked
        b">crlf\r\n" + b"\r\n".join(bytes(acgt[rnd(60)]) for _ in range(500)) + b"\r\n",
        b">iupac\n" + bytes(acgt[rnd(30000)]) + b"RYKMSWN-*" + bytes(acgt[rnd(30000)]) + b"\n",
        b"".join(b">rec%d some text\n" % i + bytes(acgt[rnd(7000)]) + b"\n" for i in range(9)),  # multi record
        b">mid\n" + bytes(acgt[rnd(20000)]) + b">inline header breaks the run\n" + bytes(acgt[rnd(20000)]) + b"\n",
        b"ACGT" * 5000 + b"\n",                                                   # no header, low complexity
        b">polyA\n" + b"A" * 30000 + b"\n",
        b">exact chunk\n" + bytes(acgt[rnd(4096)]) + b"\n",
        b">two chunks minus one\n" + bytes(acgt[rnd(8191)]) + b"\n",
        b">n-rich\n" + bytes(np.where(rng.random(80000) < 0.02, ord("N"), acgt[rnd(80000)]).astype(np.uint8)) + b"\n",
    ]
    check(gpu_ctx, shuf_l3k10, texts)


def test_uniq_mode(gpu_ctx, shuf_l3k10):
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    unit = bytes(acgt[rng.integers(0, 4, 150000, dtype=np.uint8)])
    texts = [b">dup\n" + unit + b"\nN\n" + unit[:70000] + b"\n", b">single\n" + unit + b"\n"]
    check(gpu_ctx, shuf_l3k10, texts, flags=K.SKETCH_UNIQ, uniq=True)


def test_fastq_min_occ(gpu_ctx, shuf_l3k10):
    rng = np.random.default_rng(3)
    genome = rng.integers(0, 4, 120000, dtype=np.uint8)
    reads = []
    for _ in range(4000):
        s = int(rng.integers(0, len(genome) - 150))
        r = genome[s:s + 150].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        reads.append(r)
    fq = fastq_text(reads)
    sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
    for M in (1, 2, 3):
        want = np.sort(sk.fastq(fq, Q=0, M=M))
        b = K.Batch()
        lines = b.add_fastq(fq, Q=0)
        assert lines == 4 * len(reads)
        off, ids = gpu_ctx.sketch_batch(b, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY, min_occ=M)
        assert np.array_equal(ids, want), (M, len(ids), len(want))
    # quality floor: raw ASCII compare (iseq2comem.c:312); 'I' >= '5' passes, '#' fails
    fq2 = fastq_text(reads[:500], qual=b"#")
    b = K.Batch()
    b.add_fastq(fq2, Q=ord("5"))
    off, ids = gpu_ctx.sketch_batch(b, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY)
    assert len(ids) == 0 and len(sk.fastq(fq2, Q=ord("5"), M=1)) == 0


def test_occurrence_counts_of_the_abundance_sketches(gpu_ctx, shuf_l3k10, monkeypatch):
    """KSSD_SKETCH_COUNTS: every id with its number of occurrences (dist -A, iseq2comem.c:554-615), saturating at
    65535 like the reference's 16-bit counter; through the LDS sort and through the big-genome path"""
    rng = np.random.default_rng(11)
    genome = rng.integers(0, 4, 200000, dtype=np.uint8)
    reads = []
    for _ in range(9000):
        s = int(rng.integers(0, len(genome) - 150))
        r = genome[s:s + 150].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        reads.append(r)
    sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
    # one read that holds a sampled k-mer, 70 000 times: its k-mers saturate the counter
    hot = next(r for r in reads if len(sk.fastq_koc(fastq_text([r]))[0]) > 0)
    fq_a = fastq_text(reads)
    fq_b = fastq_text(reads[:2000] + [hot] * 70000)
    try:
        for big in (False, True):
            if big:
                gpu_ctx.set_lds_sort_limit(64)
            ctx = gpu_ctx
            b = K.Batch()
            assert b.add_reads(fq_a) == len(reads)
            assert b.add_reads(fq_b) == 72000
            off, ids, cnt = ctx.sketch_batch_pos(b, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY | K.SKETCH_COUNTS)
            for g, fq in enumerate((fq_a, fq_b)):
                wi, wc = sk.fastq_koc(fq)
                o = np.argsort(wi)
                lo, hi = int(off[g]), int(off[g + 1])
                assert np.array_equal(ids[lo:hi], wi[o]), (big, g)
                assert np.array_equal(cnt[lo:hi], wc[o].astype(np.uint32)), (big, g)
            assert cnt[int(off[1]):].max() == 65535 and cnt[:int(off[1])].max() < 100
            # positions and counts are one or the other per call
            with pytest.raises(K.KssdError):
                ctx.sketch_batch_pos(b, K.SKETCH_KEEP_ZERO | K.SKETCH_FIRST_POS | K.SKETCH_COUNTS)
    finally:
        gpu_ctx.set_lds_sort_limit(0)


def _byread_text(rng, n_reads, max_len, repeat_every=0):
    out = [fasta_text(rng.integers(0, 4, 900, dtype=np.uint8))[len(b">seq\n"):]]  # bases in front of the first header: read 0
    keep = []
    for i in range(n_reads):
        r = rng.integers(0, 4, int(rng.integers(5, max_len)), dtype=np.uint8)
        if repeat_every and i % repeat_every == repeat_every - 1:
            r = keep[int(rng.integers(0, len(keep)))]  # a repeated read: its k-mers come again
        keep.append(r)
        out.append(fasta_text(r, name=b"read%d" % i, n_mask=(rng.random(len(r)) < 2e-4)))
        if i % 17 == 3:
            out.append(b">no bases\n")
    return b"".join(out)


@pytest.mark.parametrize("k,subk,dr", [(10, 6, 3), (11, 6, 3), (8, 5, 2)])
def test_by_position_stream_of_the_byread_sketches(k, subk, dr, monkeypatch):
    """KSSD_SKETCH_BY_POS: every sampled k-mer in sequence order, repeats and id 0 included, with its position; cut at the
    tokeniser's read starts it is the stream reads2mco writes (dist --byread, iseq2comem.c:78-186), per component file
    and per-read index -- through the LDS sort and through the big-genome path; several genomes per call"""
    shuf = K.Shuf.generate(k, subk, dr, seed=77)
    sk = ko.Sketcher(shuf.table, k, subk, dr)
    rng = np.random.default_rng(5 + k)
    texts = [_byread_text(rng, 60, 40000, repeat_every=7), _byread_text(rng, 400, 300), b">only a header\n", b"ACGTNNNN\n>x\nAC\n"]
    ctx = K.GpuCtx(shuf, 0)
    try:
        for big in (False, True):
            if big:
                ctx.set_lds_sort_limit(16)
            b = K.Batch()
            cuts = [b.add_fasta_reads(t) for t in texts]
            off, ids, pos = ctx.sketch_batch_pos(b, K.SKETCH_BY_POS)
            n_total = 0
            for g, t in enumerate(texts):
                wi, wc, wr, nr = sk.byread(t)
                lo, hi = int(off[g]), int(off[g + 1])
                gi, gp = ids[lo:hi], pos[lo:hi]
                assert nr == len(cuts[g]) == t.count(b">")
                assert np.all(np.diff(gp.astype(np.int64)) > 0), "positions ascend strictly: one k-mer per position"
                full = (wi.astype(np.uint64) << np.uint64(sk.p.comp_bits)) | wc.astype(np.uint64)
                assert np.array_equal(gi.astype(np.uint64), full), (big, g, len(gi), len(full))
                assert np.array_equal(np.searchsorted(cuts[g], gp, side="right"), wr), (big, g)
                n_total += len(wi)
            assert n_total > 150
            assert len(np.unique(ids[:int(off[1])])) < int(off[1]), "the repeated reads must show up as repeated ids"
        with pytest.raises(K.KssdError):  # one kind of second output per call
            ctx.sketch_batch_pos(b, K.SKETCH_BY_POS | K.SKETCH_COUNTS)
        with pytest.raises(K.KssdError):  # no keep rule may interfere
            ctx.sketch_batch_pos(b, K.SKETCH_BY_POS | K.SKETCH_UNIQ)
    finally:
        ctx.close()


@pytest.mark.parametrize("k,subk,dr", [(8, 5, 2), (10, 7, 5), (9, 6, 3), (11, 6, 3), (12, 7, 4)])
def test_other_shuffles(k, subk, dr):
    shuf = K.Shuf.generate(k, subk, dr, seed=77 + k)
    ctx = K.GpuCtx(shuf, 0)
    try:
        gs = clade_genomes(2, 2, 400_000, seed=k)
        texts = [fasta_text(c, nm, n_mask=m) for nm, c, m in gs]
        check(ctx, shuf, texts)
    finally:
        ctx.close()


def test_capacity_error_is_the_references(gpu_ctx, shuf_l3k10):
    """a genome with more distinct k-mers than 0.6*hashsize must fail like the reference (iseq2comem.c:262)"""
    # L3K10 needs > 1.26 M distinct ids (> 5 Gb of sequence): not testable at unit size; use a small-k shuffle
    shuf = K.Shuf.generate(6, 4, 1, seed=5)   # hashsize = primer[5] = 8191, limit 4914, pass rate 1/16
    ctx = K.GpuCtx(shuf, 0)
    try:
        rng = np.random.default_rng(1)
        big = fasta_text(rng.integers(0, 4, 400_000, dtype=np.uint8))
        small = fasta_text(rng.integers(0, 4, 20_000, dtype=np.uint8))
        sk = ko.Sketcher(shuf.table, 6, 4, 1)
        with pytest.raises(ko.OracleError) as oe:
            sk.fasta(big)
        assert oe.value.code == -2
        b = K.Batch()
        b.add_fasta(small)
        b.add_fasta(big)
        with pytest.raises(K.KssdError) as e:
            ctx.sketch_batch(b)
        assert e.value.code == -3 and e.value.bad_genome == 1
        b2 = K.Batch()
        b2.add_fasta(small)
        off, ids = ctx.sketch_batch(b2)
        assert np.array_equal(ids, np.sort(sk.fasta(small)))
    finally:
        ctx.close()


def test_empty_batch_and_empty_genome(gpu_ctx, shuf_l3k10):
    b = K.Batch()
    b.add_fasta(b">nothing but a header\n")
    b.add_fasta(b">x\nACGT\n")
    off, ids = gpu_ctx.sketch_batch(b)
    assert list(off) == [0, 0, 0] and len(ids) == 0


@pytest.fixture
def force_big_path(gpu_ctx, monkeypatch):
    """route every genome with more than ~100 expected ids through the global-memory dedup (rocPRIM sort + run kernels):
    kssd_gpu_set_lds_sort_limit on the session context and on every context created meanwhile"""
    monkeypatch.setattr(K.capi, "DEFAULT_LDS_SORT_LIMIT", 100)
    gpu_ctx.set_lds_sort_limit(100)
    yield
    gpu_ctx.set_lds_sort_limit(0)


def test_big_genome_path_matches_oracle(gpu_ctx, shuf_l3k10, force_big_path):
    """same rules as the LDS dedup: id 0, -u, -n, order, mixed with small genomes in one batch"""
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    gs = clade_genomes(2, 2, 1_500_000, seed=4)            # ~366 ids each: big path (threshold 100)
    texts = [fasta_text(c, nm, n_mask=m) for nm, c, m in gs]
    texts.insert(1, b">tiny\n" + bytes(acgt[rng.integers(0, 4, 50_000, dtype=np.uint8)]) + b"\n")  # stays on the LDS path
    check(gpu_ctx, shuf_l3k10, texts)
    unit = bytes(acgt[rng.integers(0, 4, 900_000, dtype=np.uint8)])
    check(gpu_ctx, shuf_l3k10, [b">dup\n" + unit + b"\nN\n" + unit[:500_000] + b"\n"], flags=K.SKETCH_UNIQ, uniq=True)
    # FASTQ -n through the big path
    genome = rng.integers(0, 4, 1_200_000, dtype=np.uint8)
    reads = []
    for _ in range(20000):
        s = int(rng.integers(0, len(genome) - 150))
        r = genome[s:s + 150].copy()
        reads.append((3 - r)[::-1] if rng.random() < 0.5 else r)
    fq = fastq_text(reads)
    sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
    for M in (1, 2):
        b = K.Batch()
        b.add_fastq(fq, Q=0)
        off, ids = gpu_ctx.sketch_batch(b, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY, min_occ=M)
        assert np.array_equal(ids, np.sort(sk.fastq(fq, Q=0, M=M))), M


def test_big_genome_path_capacity_error(force_big_path):
    shuf = K.Shuf.generate(6, 4, 1, seed=5)   # limit 4914 distinct ids
    ctx = K.GpuCtx(shuf, 0)
    try:
        rng = np.random.default_rng(1)
        b = K.Batch()
        b.add_fasta(fasta_text(rng.integers(0, 4, 20_000, dtype=np.uint8)))
        b.add_fasta(fasta_text(rng.integers(0, 4, 400_000, dtype=np.uint8)))
        with pytest.raises(K.KssdError) as e:
            ctx.sketch_batch(b)
        assert e.value.code == -3 and e.value.bad_genome == 1
    finally:
        ctx.close()


def test_genome_larger_than_the_lds_sort(shuf_l3k10):
    """a 150 Mb record (~36 600 ids > DEDUP_MAX_N = 32768) takes the global-memory path without any override"""
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        rng = np.random.default_rng(8)
        t = fasta_text(rng.integers(0, 4, 150_000_000, dtype=np.uint8))
        b = K.Batch()
        b.add_fasta(t)
        off, ids = ctx.sketch_batch(b)
        want = np.sort(ko.Sketcher(shuf_l3k10.table, 10, 6, 3).fasta(t))
        assert len(want) > 32768 and np.array_equal(ids, want)
        s1, bl = ctx.scan_stats()
        npos = b.n_chunks * 4096
        assert 0.006 < s1 / npos < 0.010 and 0.0004 < bl / npos < 0.0009  # stage 1 / Bloom pass rates
    finally:
        ctx.close()


def test_many_tiny_genomes_share_waves_and_chunks(gpu_ctx, shuf_l3k10):
    """2 000 genomes of 1-3 chunks each: many genomes per scan wave, several genomes per wave of the exact stage,
    k-mers must not leak across genome boundaries"""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base = bytes(acgt[rng.integers(0, 4, 12_000, dtype=np.uint8)])
    texts = []
    for i in range(2000):
        n = int(rng.integers(3000, 12_000))
        s = int(rng.integers(0, 12_000 - n + 1))
        texts.append(b">g%d\n" % i + base[s:s + n] + b"\n")   # overlapping slices: the same k-mers in many genomes
    b = K.Batch()
    for t in texts:
        b.add_fasta(t)
    off, ids = gpu_ctx.sketch_batch(b)
    sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
    want_all = np.sort(sk.fasta(b">all\n" + base + b"\n"))
    for g in rng.choice(2000, 120, replace=False):
        got = ids[int(off[g]):int(off[g + 1])]
        assert np.array_equal(got, np.sort(sk.fasta(texts[g]))), g
        assert np.isin(got, want_all).all()


def test_first_positions_give_the_reference_file_order():
    """combco byte order: the reference dumps its hash table slot by slot; where two ids of a genome probe the same
    slot the earlier one in the sequence keeps it.  With the device's first positions the host replays exactly that.
    A small table (131 071 slots for ~7 800 ids) makes hundreds of such collisions per genome."""
    shuf = K.Shuf.generate(8, 5, 2, seed=3)
    hs = K.derive(8, 5, 2).hashsize
    assert hs == 131071
    ctx = K.GpuCtx(shuf, 0)
    try:
        rng = np.random.default_rng(2)
        texts = [fasta_text(rng.integers(0, 4, n, dtype=np.uint8)) for n in (2_000_000, 700_000, 30_000)]
        b = K.Batch()
        for t in texts:
            b.add_fasta(t)
        off, ids, pos = ctx.sketch_batch_pos(b)
        off2, ids2 = ctx.sketch_batch(b)
        assert np.array_equal(off, off2) and np.array_equal(ids, ids2)  # same sets, ascending, with or without positions
        sk = ko.Sketcher(shuf.table, 8, 5, 2)
        differs = 0
        for g, t in enumerate(texts):
            dump = sk.fasta(t)  # the reference's file order
            mine, p = ids[int(off[g]):int(off[g + 1])], pos[int(off[g]):int(off[g + 1])]
            assert len(np.unique(p)) == len(p)  # one k-mer per position
            assert np.array_equal(K.slot_order_pos(mine, p, hs), dump), g
            differs += int(not np.array_equal(K.slot_order(mine, hs), dump))
        assert differs >= 1  # the ascending-id replay is NOT enough here: the positions matter
    finally:
        ctx.close()


def test_first_positions_through_the_big_genome_path(shuf_l3k10, force_big_path):
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        rng = np.random.default_rng(6)
        t = fasta_text(rng.integers(0, 4, 3_000_000, dtype=np.uint8))
        b = K.Batch()
        b.add_fasta(t)
        off, ids, pos = ctx.sketch_batch_pos(b)
        sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
        assert np.array_equal(K.slot_order_pos(ids, pos, sk.p.hashsize), sk.fasta(t))
    finally:
        ctx.close()


def test_round_trip_sketch_then_reverse(gpu_ctx, shuf_l3k10):
    """encode -> decode at full size: every id of a 5 Mb genome's sketch, turned back into its canonical 20-mer
    (kssd_reverse_id = core_reverse2unituple, command_reverse.c:311-321), occurs in the genome on one strand, sits at the
    first position the device reported, and its sub-context has the rank the id carries."""
    import ctypes as C
    rng = np.random.default_rng(31)
    codes = rng.integers(0, 4, 5_000_000, dtype=np.uint8)
    b = K.Batch()
    b.add_fasta(fasta_text(codes))
    off, ids, pos = gpu_ctx.sketch_batch_pos(b)
    assert 1100 < len(ids) < 1350
    acc = np.zeros(4096, np.uint32)
    tab = shuf_l3k10.table
    sel = np.nonzero(tab < 4096)[0]
    acc[tab[sel]] = sel
    L = K.host_lib()
    L.kssd_reverse_id.restype = C.c_uint64
    L.kssd_reverse_id.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    # all 20-mers of the genome as integers, forward strand, first base in the top bits
    fw = np.zeros(len(codes) - 19, np.uint64)
    for i in range(20):
        fw = (fw << np.uint64(2)) | codes[i:len(codes) - 19 + i].astype(np.uint64)
    rc = np.zeros_like(fw)
    for i in range(20):
        rc |= (np.uint64(3) - ((fw >> np.uint64(2 * i)) & np.uint64(3))) << np.uint64(2 * (19 - i))
    canon = np.minimum(fw, rc)
    for i, p in zip(ids.tolist(), pos.tolist()):
        u = L.kssd_reverse_id(i, 10, 6, 3, acc.ctypes.data)
        start = p - 4                       # the device reports the sub-context start; the k-mer starts `out` = 4 bases before
        assert canon[start] == u, (i, p)
        assert tab[(u >> 8) & 0xFFFFFF] == (i & 0xFFF)


def test_low_complexity_megabases_overflow_and_retry(shuf_l3k10):
    """12 Mb of a period-4 repeat whose 12-mer IS an accepted pattern: EVERY position passes all filters -- candidate
    list and staging regions overflow their estimates by three orders of magnitude, the call reports it, the host-level
    entry point retries with larger buffers (and the global-memory dedup), and the result is still the reference's"""
    tab = shuf_l3k10.table.copy()
    code = 0
    for ch in b"ACGT" * 3:
        code = (code << 2) | b"ACGT".index(ch)
    y = int(np.nonzero(tab == 5)[0][0])          # give the repeat's sub-context rank 5: still a permutation
    tab[y], tab[code] = tab[code], 5
    shuf = K.Shuf((shuf_l3k10.id, 10, 6, 3), tab)
    ctx = K.GpuCtx(shuf, 0)
    try:
        sk = ko.Sketcher(shuf.table, 10, 6, 3)
        texts = [b">str\n" + b"ACGT" * 3_000_000 + b"\n", b">polyA\n" + b"A" * 2_000_000 + b"\n",
                 b">dinuc\n" + b"AC" * 1_500_000 + b"\n"]
        b = K.Batch()
        for t in texts:
            b.add_fasta(t)
        off, ids = ctx.sketch_batch(b)
        s1, bl = ctx.scan_stats()
        assert bl > 2_500_000                     # the flood really happened: every 4th position of the 12 Mb repeat
        for g, t in enumerate(texts):
            assert np.array_equal(ids[int(off[g]):int(off[g + 1])], np.sort(sk.fasta(t))), g
        assert len(ids[int(off[0]):int(off[1])]) >= 1
        # the same context afterwards on ordinary sequence (its oversized buffers shrink back; results unaffected)
        rng = np.random.default_rng(3)
        t2 = [fasta_text(rng.integers(0, 4, 400_000, dtype=np.uint8)) for _ in range(5)]
        for _ in range(2):
            b2 = K.Batch()
            for t in t2:
                b2.add_fasta(t)
            off2, ids2 = ctx.sketch_batch(b2)
            for g, t in enumerate(t2):
                assert np.array_equal(ids2[int(off2[g]):int(off2[g + 1])], np.sort(sk.fasta(t))), g
    finally:
        ctx.close()


def test_a_wave_that_owns_more_than_2048_chunks(shuf_l3k10):
    """A buffered stage-1 candidate names its chunk modulo 2048 and a Bloom round takes the newest 64 entries: without the
    periodic drain the entries at the bottom of a wave's buffer are attributed to a chunk 2048 k too late once the wave owns
    more than 2048 chunks (a 34 GB read set, a batch of mammalian genomes on the full grid).  Here: ONE scan workgroup
    (kssd_gpu_set_scan_grid) over 40 000 chunks = 2 500 per wave, with a long N stretch in the middle so that entries wait
    across a candidate-free run, against the oracle and against the default grid."""
    rng = np.random.default_rng(2048)
    n = 40_000 * 4096 - 1000
    codes = rng.integers(0, 4, n, dtype=np.uint8)
    nm = np.zeros(n, dtype=bool)
    nm[rng.integers(0, n, 300)] = True
    text = fasta_text(codes, b"long", n_mask=nm)
    # a second genome: bases, then 12 Mb of N written as one stretch (one invalid position in the batch), then bases again
    c2 = rng.integers(0, 4, 30_000_000, dtype=np.uint8)
    t2 = fasta_text(c2[:9_000_000], b"gap") + b"N" * 12_000_000 + b"\n" + fasta_text(c2[9_000_000:], b"x")[len(b">x\n"):]
    sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
    want = [np.sort(sk.fasta(text)), np.sort(sk.fasta(t2))]
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        b = K.Batch()
        b.add_fasta(text)
        b.add_fasta(t2)
        assert b.n_chunks > 16 * 2048
        res = {}
        for grid in (1, 0):
            ctx.set_scan_grid(grid)
            off, ids = ctx.sketch_batch(b)
            res[grid] = (off.copy(), ids.copy())
            for g in range(2):
                got = ids[int(off[g]):int(off[g + 1])]
                assert np.array_equal(got, want[g]), (grid, g, len(got), len(want[g]))
        assert np.array_equal(res[0][1], res[1][1])
        b.close()
    finally:
        ctx.set_scan_grid(0)
        ctx.close()


def test_genomes_sorted_in_lds_in_parts(shuf_l3k10):
    """genomes whose staged tuples exceed one LDS sort but split by their ids' top bits into up to 16 ranges that fit
    (DEDUP_PARTS: a 150 Mb record here, the 3 Gb records of BASELINE configs[4] at -s 7): id sets against the oracle, next
    to small genomes and an empty one in the same batch; first positions and occurrence counts through the same path;
    the same genomes with kssd_gpu_set_lds_sort_limit(4096): more genomes take the parts path, the largest the global-memory one"""
    rng = np.random.default_rng(150)
    big = rng.integers(0, 4, 150_000_000, dtype=np.uint8)
    nm = np.zeros(len(big), dtype=bool)
    nm[rng.integers(0, len(big), 500)] = True
    texts = [fasta_text(rng.integers(0, 4, 300_000, dtype=np.uint8), b"small0"), fasta_text(big, b"chr", n_mask=nm), b">empty\n",
             fasta_text(big[:40_000_000], b"arm"), fasta_text(rng.integers(0, 4, 2_000_000, dtype=np.uint8), b"small1")]
    del big, nm
    sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
    want = [np.sort(sk.fasta(t)) for t in texts]
    assert len(want[1]) > 32768
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        b = K.Batch()
        for t in texts:
            b.add_fasta(t)
        for limit in (0, 4096):
            ctx.set_lds_sort_limit(limit)
            off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
            for g in range(len(texts)):
                assert np.array_equal(ids[int(off[g]):int(off[g + 1])], want[g]), (limit, g)
            off, ids, pos = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY | K.SKETCH_FIRST_POS)
            for g in range(len(texts)):
                assert np.array_equal(ids[int(off[g]):int(off[g + 1])], want[g]), (limit, g, "pos")
            off, ids, cnt = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY | K.SKETCH_COUNTS)
            for g in range(len(texts)):
                assert np.array_equal(ids[int(off[g]):int(off[g + 1])], want[g]) and (cnt[int(off[g]):int(off[g + 1])] >= 1).all(), (limit, g, "counts")
        # first positions: the k-mer at that position reduces to that id (checked on the 40 Mb arm through the oracle's position list)
        ctx.set_lds_sort_limit(0)
        off, ids, pos = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY | K.SKETCH_FIRST_POS)
        order = K.slot_order_pos(ids[int(off[3]):int(off[4])], pos[int(off[3]):int(off[4])], 2097143)
        assert np.array_equal(order, sk.fasta(texts[3]))            # the reference's file order: needs every first position right
        # without the flag the 150 Mb record is still below the reference's capacity limit (1 258 285): no abort either way
        off2, ids2 = ctx.sketch_batch(b)
        assert np.array_equal(ids2, ids)
        b.close()
    finally:
        ctx.set_lds_sort_limit(0)
        ctx.close()


def test_large_genomes_sorted_by_ranges_of_their_keys():
    """genomes beyond sixteen LDS sorts (read sets, chromosomes, --byread files): staged keys partitioned by ranges of their
    leading field and sorted in LDS (DEDUP_RANGES, big_rng_* kernels) -- ids, minimum occurrence, first positions -> the
    reference's file order, occurrence counts, the by-position stream; and keys that do NOT spread (one unit 8 000 times:
    a handful of ids, each thousands of times) overflow an item, the status says so, and the repeated call sorts in global
    memory.  -k 8 -s 5 -l 2 samples every 256th position and kssd_gpu_set_lds_sort_limit(1024) makes 12 Mb large."""
    rng = np.random.default_rng(2048)
    shuf = K.Shuf.generate(8, 5, 2, seed=5)
    sk = ko.Sketcher(shuf.table, 8, 5, 2)
    hashsize = K.derive(8, 5, 2).hashsize
    big = rng.integers(0, 4, 12_000_000, dtype=np.uint8)
    nm = np.zeros(len(big), dtype=bool)
    nm[rng.integers(0, len(big), 300)] = True
    part = rng.integers(0, 4, 2_500_000, dtype=np.uint8)
    six = np.concatenate([part[o:] for o in (0, 11, 5, 300, 7, 2)] + [rng.integers(0, 4, 1_000_000, dtype=np.uint8)])
    unit = rng.integers(0, 4, 2_000, dtype=np.uint8)
    texts = [fasta_text(big, b"big", n_mask=nm), fasta_text(rng.integers(0, 4, 100_000, dtype=np.uint8), b"small"), b">empty\n",
             fasta_text(six, b"six copies")]
    heavy = fasta_text(np.tile(unit, 8_000), b"one unit 8000 times")
    del big, nm, six
    ctx, plain = K.GpuCtx(shuf, 0), K.GpuCtx(shuf, 0)
    try:
        ctx.set_lds_sort_limit(1024)
        b = K.Batch()
        for t in texts:
            b.add_fasta(t)
        F = K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY
        want = [np.sort(sk.fasta(t)) for t in texts]
        assert len(want[0]) > 16 * 1024
        off, ids = ctx.sketch_batch(b, F)
        for g in range(len(texts)):
            assert np.array_equal(ids[int(off[g]):int(off[g + 1])], want[g]), g
        off, ids, pos = ctx.sketch_batch_pos(b, F | K.SKETCH_FIRST_POS)
        for g in (0, 3):
            lo, hi = int(off[g]), int(off[g + 1])
            assert np.array_equal(K.slot_order_pos(ids[lo:hi], pos[lo:hi], hashsize), sk.fasta(texts[g])), g    # the reference's file order
        off_c, ids_c, cnt_c = ctx.sketch_batch_pos(b, F | K.SKETCH_COUNTS)
        off_p, ids_p, cnt_p = plain.sketch_batch_pos(b, F | K.SKETCH_COUNTS)
        assert np.array_equal(off_c, off_p) and np.array_equal(ids_c, ids_p) and np.array_equal(cnt_c, cnt_p)
        assert np.array_equal(ids_c, ids) and cnt_c[int(off_c[3]):int(off_c[4])].max() >= 6
        for M in (2, 6, 7):
            off_m, ids_m = ctx.sketch_batch(b, F | K.SKETCH_KEEP_ZERO, min_occ=M)
            keep = cnt_c >= M
            per_genome = [int(keep[int(off_c[g]):int(off_c[g + 1])].sum()) for g in range(len(texts))]
            assert np.array_equal(ids_m, ids_c[keep]) and np.array_equal(np.diff(off_m.astype(np.int64)), per_genome), M
        off_b, ids_b, pos_b = ctx.sketch_batch_pos(b, K.SKETCH_BY_POS)
        off_q, ids_q, pos_q = plain.sketch_batch_pos(b, K.SKETCH_BY_POS)
        assert np.array_equal(off_b, off_q) and np.array_equal(ids_b, ids_q) and np.array_equal(pos_b, pos_q)
        assert (np.diff(pos_b[int(off_b[0]):int(off_b[1])].astype(np.int64)) > 0).all() and off_b[1] > 16 * 1024
        b.close()
        # keys that do not spread
        hb = K.Batch()
        hb.add_fasta(heavy)
        hb.add_fasta(texts[1])
        w = np.sort(sk.fasta(heavy))
        off, ids, cnt = ctx.sketch_batch_pos(hb, F | K.SKETCH_COUNTS)
        assert np.array_equal(ids[:int(off[1])], w) and cnt[:int(off[1])].max() >= 7_999 and len(w) < 64
        assert np.array_equal(ids[int(off[1]):], want[1])
        off, ids = ctx.sketch_batch(hb, F)                      # (the context remembers: global-memory sort at once)
        assert np.array_equal(ids[:int(off[1])], w)
        hb.close()
    finally:
        ctx.close()
        plain.close()


def test_chunks_with_more_stage_one_candidates_than_the_buffer_holds(gpu_ctx, shuf_l3k10):
    """stretches where a large share of the positions passes stage 1: accepted sub-contexts tiled back to back (every twelfth
    position is a true member, 341 per chunk: more than the wave's 256 buffered positions -- the scan's lane-after-lane
    path), homopolymers and short tandem repeats next to ordinary sequence, N runs inside; ids against the oracle"""
    rng = np.random.default_rng(31)
    acc = np.flatnonzero((shuf_l3k10.table >= 0) & (shuf_l3k10.table < 4096))         # accepted sub-contexts (12 bases = 24 bits)
    assert len(acc) == 4096

    def bases_of(x):
        return np.array([(int(x) >> (2 * (11 - i))) & 3 for i in range(12)], dtype=np.uint8)
    pats = [bases_of(x) for x in rng.choice(acc, 6, replace=False)]
    tiled = [np.tile(p, 30_000) for p in pats[:3]]                                      # 360 kb each of one pattern back to back
    mix = np.concatenate([np.tile(pats[3], 2_000), rng.integers(0, 4, 50_000, dtype=np.uint8), np.tile(pats[4], 5_000),
                          np.zeros(40_000, np.uint8), np.tile(np.array([0, 1], np.uint8), 30_000), np.tile(pats[5], 9_000),
                          rng.integers(0, 4, 200_000, dtype=np.uint8)])
    nm = np.zeros(len(mix), dtype=bool)
    nm[rng.integers(0, len(mix), 60)] = True
    texts = [fasta_text(t, b"tiled%d" % i) for i, t in enumerate(tiled)] + [fasta_text(mix, b"mix", n_mask=nm),
                                                                            fasta_text(rng.integers(0, 4, 300_000, dtype=np.uint8), b"plain")]
    check(gpu_ctx, shuf_l3k10, texts, flags=K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
    st1, bl = gpu_ctx.scan_stats()
    assert st1 > 0.05 * sum(len(t) for t in tiled)                                     # the dense path really ran


@pytest.mark.parametrize("params", [(10, 6, 3), (11, 6, 3), (10, 7, 5), (8, 4, 1)])
def test_lengths_around_chunk_and_block_borders(params):
    """batches of genomes whose lengths sit on and around the scan's borders (a packed word, the k-mer, a chunk of 4 096, a
    block of four chunks), with tandem repeats, scattered Ns and long N runs, tiny and empty genomes -- whole tuples against
    the oracle (profiles/fuzz_sketch.py is the long version of this)"""
    k, s, l = params
    shuf = K.Shuf.generate(k, s, l, seed=77 + k)
    sk = ko.Sketcher(shuf.table, k, s, l)
    cb = 4 * max(k - l - 7, 0)
    ctx = K.GpuCtx(shuf, 0)
    try:
        for seed in range(4):
            rng = np.random.default_rng(100 * k + seed)
            texts = []
            for g in range(10):
                n = int(rng.choice([0, 1, 15, 16, 17, 2 * k - 1, 2 * k, 4095, 4096, 4097, 16383, 16384, 16385, 65536, int(rng.integers(100, 300_000))]))
                codes = rng.integers(0, 4, n, dtype=np.uint8)
                kind = int(rng.integers(0, 4))
                if kind == 1 and n > 100:
                    codes = np.tile(codes[: int(rng.integers(1, 50))], n)[:n]
                nm = np.zeros(n, dtype=bool)
                if kind == 2 and n:
                    nm[rng.integers(0, n, max(1, n // 500))] = True
                if kind == 3 and n > 50:
                    a = int(rng.integers(0, n - 10))
                    nm[a:a + int(rng.integers(1, 9000))] = True
                texts.append(fasta_text(codes, b"g%d" % g, n_mask=nm if nm.any() else None))
            b = K.Batch()
            for t in texts:
                b.add_fasta(t)
            off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
            for g, t in enumerate(texts):
                wi, wc = sk.fasta(t, with_comps=True)
                want = np.sort((wi.astype(np.uint64) << np.uint64(cb) | wc.astype(np.uint64)).astype(np.uint32))
                assert np.array_equal(ids[int(off[g]):int(off[g + 1])], want), (params, seed, g, len(t))
            b.close()
    finally:
        ctx.close()


def test_tandem_repeat_whose_region_is_all_its_positions_converges():
    """Found by profiles/fuzz_sketch.py in round 5 (profiles/r05E_fuzz_tandem_repeat.txt): at a dense parameter set (-k 8 -s 4 -l 1: every
    16th sub-context) a tandem repeat of 190 kb stages 38 000 occurrences of a handful of ids.  Its region grows to ALL of its positions
    and can grow no further, the parts of the LDS sort split ids by their top bits -- all in one part -- and every repeated call asked
    for more of a region that was at its limit; an empty genome in the same batch reported the fullest region there can be (`cap - 1`
    wrapped around for a region of no room), which the host multiplied into its growth factor.  Such a genome now leaves the parts path
    after the first overflow, an empty region is 0 % full, and the call converges: ids as the oracle's."""
    k, s, l = 8, 4, 1
    shuf = K.Shuf.generate(k, s, l, seed=105)
    sk = ko.Sketcher(shuf.table, k, s, l)
    rng = np.random.default_rng(5048)
    for unit_len, n, with_empty in ((46, 193_203, True), (46, 193_203, False), (7, 60_000, True)):
        unit = rng.integers(0, 4, unit_len, dtype=np.uint8)
        texts = [fasta_text(np.tile(rng.integers(0, 4, 29, dtype=np.uint8), 16_385)[:16_385], b"g0"),
                 fasta_text(np.tile(unit, n)[:n], b"g1")]
        if with_empty:
            texts += [fasta_text(np.zeros(0, np.uint8), b"g2"), fasta_text(rng.integers(0, 4, 16, dtype=np.uint8), b"g3")]
        ctx = K.GpuCtx(shuf, 0)
        b = K.Batch()
        for t in texts:
            b.add_fasta(t)
        off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
        for g, t in enumerate(texts):
            want = np.zeros(0, np.uint32)
            if len(t):
                wi, wc = sk.fasta(t, with_comps=True)
                want = np.sort((wi.astype(np.uint64) << np.uint64(4 * max(k - l - 7, 0)) | wc.astype(np.uint64)).astype(np.uint32))
            assert np.array_equal(ids[int(off[g]):int(off[g + 1])], want), (unit_len, n, g)
        # the same batch with first positions (8-byte keys: half as many fit the LDS sort): the same ids
        off2, ids2, pos2 = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
        assert np.array_equal(off2, off) and np.array_equal(ids2, ids)
        b.close()
        ctx.close()
    # the fuzzer's second find, as the fuzzer made it (profiles/fuzz_sketch.py, seed base 777 000, parameter set 5, seed 189): the
    # first-position call failed with "hipFuncSetAttribute ... invalid argument"
    rng2 = np.random.default_rng(777_000 + 1000 * 5 + 189)
    texts = []
    for g in range(int(rng2.integers(1, 12))):
        kind = int(rng2.integers(0, 6))
        n = int(rng2.choice([0, 1, 15, 16, 17, 2 * k - 1, 2 * k, 4095, 4096, 4097, 16383, 16384, 16385, 65536, int(rng2.integers(100, 400_000))]))
        codes = rng2.integers(0, 4, n, dtype=np.uint8)
        if kind == 1 and n > 100:
            codes = np.tile(codes[: int(rng2.integers(1, 50))], n)[:n]
        nm = np.zeros(n, dtype=bool)
        if kind == 2 and n:
            nm[rng2.integers(0, n, max(1, n // 500))] = True
        if kind == 3 and n > 50:
            a0 = int(rng2.integers(0, n - 10))
            nm[a0:a0 + int(rng2.integers(1, 9000))] = True
        texts.append(fasta_text(codes, b"g%d" % g, n_mask=nm if nm.any() else None))
    ctx = K.GpuCtx(shuf, 0)
    b = K.Batch()
    for t in texts:
        b.add_fasta(t)
    off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
    off2, ids2, pos2 = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
    assert np.array_equal(off2, off) and np.array_equal(ids2, ids)
    b.close()
    ctx.close()
    # a region of exactly 16 384 positions -- all of the genome's, after an overflow -- asks for an LDS array of 16 384 keys, not of
    # the next power of two (a rounding on the way doubled it: 256 KB of 8-byte keys, more than the LDS; the fuzzer's second find)
    ctx = K.GpuCtx(shuf, 0)
    b = K.Batch()
    texts = [fasta_text(np.tile(rng.integers(0, 4, 23, dtype=np.uint8), 16_384)[:16_384], b"t"), fasta_text(rng.integers(0, 4, 16_384, dtype=np.uint8), b"u")]
    for t in texts:
        b.add_fasta(t)
    off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
    off2, ids2, pos2 = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
    assert np.array_equal(off2, off) and np.array_equal(ids2, ids)
    for g, t in enumerate(texts):
        wi, wc = sk.fasta(t, with_comps=True)
        want = np.sort((wi.astype(np.uint64) << np.uint64(4 * max(k - l - 7, 0)) | wc.astype(np.uint64)).astype(np.uint32))
        assert np.array_equal(ids[int(off[g]):int(off[g + 1])], want), g
    b.close()
    ctx.close()


def test_tandem_repeat_in_one_part_leaves_the_parts_path():
    """Found by profiles/fuzz_sketch.py at its long setting in round 6 (profiles/r06long_fuzz_sketch.txt; parameter set 3 = -k 11 -s 6
    -l 3, seed 450, the LDS sort limited to 256 keys): a 16 kb tandem repeat stages ~360 occurrences of a handful of ids.  Sorted in
    parts, all of them meet in ONE id range of 256 keys inside a region that holds them easily; the host answered by growing the regions
    (x 1.39 x 1.25 per attempt), which moves nothing until the region leaves the parts' range -- at factor 770, more attempts away than
    a call makes: "output or staging buffer too small".  A part that overflows inside a region that did not now says `parts_skew`, and
    the repeated call sorts such genomes in global memory.  The batch as the fuzzer made it, ids against the oracle."""
    pi, (k, s, l) = 3, (11, 6, 3)
    shuf = K.Shuf.generate(k, s, l, seed=100 + pi)
    sk = ko.Sketcher(shuf.table, k, s, l)
    rng = np.random.default_rng(100000 + 1000 * pi + 450)
    texts = []
    for g in range(int(rng.integers(1, 12))):
        kind = int(rng.integers(0, 6))
        n = int(rng.choice([0, 1, 15, 16, 17, 2 * k - 1, 2 * k, 4095, 4096, 4097, 16383, 16384, 16385, 65536, int(rng.integers(100, 400_000))]))
        codes = rng.integers(0, 4, n, dtype=np.uint8)
        if kind == 1 and n > 100:
            codes = np.tile(codes[: int(rng.integers(1, 50))], n)[:n]
        nm = np.zeros(n, dtype=bool)
        if kind == 2 and n:
            nm[rng.integers(0, n, max(1, n // 500))] = True
        if kind == 3 and n > 50:
            a0 = int(rng.integers(0, n - 10))
            nm[a0:a0 + int(rng.integers(1, 9000))] = True
        texts.append(fasta_text(codes, b"g%d" % g, n_mask=nm if nm.any() else None))
    assert any(len(t) > 16_000 for t in texts)
    ctx = K.GpuCtx(shuf, 0)
    try:
        for limit in (256, 64, 1024):
            ctx.set_lds_sort_limit(limit)
            b = K.Batch()
            for t in texts:
                b.add_fasta(t)
            off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
            for g, t in enumerate(texts):
                want = np.zeros(0, np.uint32)
                if len(t):
                    wi, wc = sk.fasta(t, with_comps=True)
                    want = np.sort((wi.astype(np.uint64) << np.uint64(sk.p.comp_bits) | wc.astype(np.uint64)).astype(np.uint32))
                assert np.array_equal(ids[int(off[g]):int(off[g + 1])], want), (limit, g)
            off2, ids2, pos2 = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
            assert np.array_equal(off2, off) and np.array_equal(ids2, ids)
            b.close()
    finally:
        ctx.close()
