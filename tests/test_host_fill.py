"""Host tokeniser, parallel-fill entry points (kssd_batch_reserve / kssd_batch_fill_text): the layout is fixed first and the
genomes are tokenised independently (the CLI does this on all host threads, into page-locked memory); every genome
must come out as the append-style tokeniser leaves it -- same packed words, same mask, same position count -- with
the unused room as padding.  CPU only."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import public_kssd_amd as K
from synth import fasta_text, fastq_text


def _genome_words(b, g):
    co = b.chunk_off()
    p, m = b.packed(), b.mask()
    return (p[int(co[g]) * K.CHUNK_WORDS:int(co[g + 1]) * K.CHUNK_WORDS].copy(),
            m[int(co[g]) * K.CHUNK_MASKW:int(co[g + 1]) * K.CHUNK_MASKW].copy())


def test_fill_equals_append_for_fasta_fastq_and_reads():
    rng = np.random.default_rng(2)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    fa = [fasta_text(rng.integers(0, 4, n, dtype=np.uint8), n_mask=rng.random(n) < 1e-3) for n in (70_000, 4096, 12_289, 5)]
    fa.append(b">a\n" + bytes(acgt[rng.integers(0, 4, 300)]) + b"RYK\n>b x\n" + bytes(acgt[rng.integers(0, 4, 9000)]).lower() + b"\r\n")
    reads = [rng.integers(0, 4, 150, dtype=np.uint8) for _ in range(400)]
    fq = fastq_text(reads)
    cases = [(t, 0) for t in fa] + [(fq, 1), (fq, 2)]
    ref = K.Batch()
    lines_ref = []
    for t, kind in cases:
        if kind == 0:
            ref.add_fasta(t)
            lines_ref.append(0)
        elif kind == 1:
            lines_ref.append(ref.add_fastq(t, Q=0))
        else:
            lines_ref.append(ref.add_reads(t))
    b = K.Batch()
    b.add_fasta(fa[0])                                     # reserve after something is already there
    first = b.reserve([len(t) for t, _ in cases])
    assert first == 1 and b.n_genomes == 1 + len(cases)
    with ThreadPoolExecutor(max_workers=4) as ex:           # concurrently, one genome each
        lines = list(ex.map(lambda a: b.fill_text(first + a[0], a[1][0], kind=a[1][1]), enumerate(cases)))
    assert lines == lines_ref
    for i in range(len(cases)):
        rp, rm = _genome_words(ref, i)
        gp, gm = _genome_words(b, first + i)
        assert b.n_positions(first + i) == ref.n_positions(i)
        assert len(gp) >= len(rp)
        assert np.array_equal(gp[:len(rp)], rp) and np.array_equal(gm[:len(rm)], rm), i
        assert not gp[len(rp):].any() and not gm[len(rm):].any()      # the unused room is padding
    # a cleared batch is reusable and starts from zeroed memory
    b.clear()
    assert b.n_genomes == 0 and b.n_chunks == 0
    g0 = b.reserve([len(fa[1])])
    b.fill_text(g0, fa[1])
    rp, rm = _genome_words(ref, 1)
    gp, gm = _genome_words(b, 0)
    assert np.array_equal(gp[:len(rp)], rp) and np.array_equal(gm[:len(rm)], rm) and not gp[len(rp):].any()


def test_fill_reports_malformed_input_and_leaves_the_slot_empty():
    b = K.Batch()
    g = b.reserve([100, 100])
    try:
        b.fill_text(g, b">header without end")
        assert False, "expected an error"
    except K.KssdError as e:
        assert e.code == -103
    b.fill_text(g + 1, b">ok\nACGTACGTACGTACGTACGTACGT\n")
    p0, m0 = _genome_words(b, g)
    assert not p0.any() and not m0.any() and b.n_positions(g) == 0 and b.n_positions(g + 1) == 24


def test_an_empty_reservation_takes_no_bases():
    """a genome reserved with no room must refuse text instead of writing into its neighbour's first chunk"""
    b = K.Batch()
    g = b.reserve([0, 100])
    b.fill_text(g + 1, b">n\nACGTACGTACGTACGTACGTACGT\n")
    before_p, before_m = [x.copy() for x in _genome_words(b, g + 1)]
    b.fill_text(g, b">only a header\n")                      # nothing to write: fine
    try:
        b.fill_text(g, b">x\nACGTACGT\n")
        assert False, "expected an error"
    except K.KssdError as e:
        assert e.code == -2 or e.code < 0
    after_p, after_m = _genome_words(b, g + 1)
    assert np.array_equal(before_p, after_p) and np.array_equal(before_m, after_m) and b.n_positions(g) == 0
