"""-m gpu: the FASTA tokeniser on the device (kssd_gpu_tokenise_fasta_device, csrc/kssd_tok.inc) against the host
tokeniser (libkssd_host.so, itself pinned against the reference's sketches by the golden tests): the packed words, the
mask words and the position counts must be bit for bit what kssd_batch_fill_text writes for the same bytes, and the
sketches of device-tokenised text must equal the oracle's."""
import os

import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K
from synth import fasta_text, fastq_text

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    rng = np.random.default_rng(12)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rnd(n):
        return bytes(acgt[rng.integers(0, 4, n, dtype=np.uint8)])
    edge = open(os.path.join(G, "qry_fa", "edge.fa"), "rb").read()   # lower case, IUPAC, CRLF, '>' mid-line, a 70 KB header ...
    return [
        fasta_text(rng.integers(0, 4, 300_000, dtype=np.uint8), n_mask=rng.random(300_000) < 1e-3),
        edge,
        b">only a header\n",
        b"ACGT",                                                       # no header, no newline, one word
        b"",                                                           # empty file: an empty genome
        b">h\n" + rnd(4095) + b"\n",                                   # one position short of a chunk
        b">h\n" + rnd(4096) + b"\n" + b"N" * 5000 + rnd(17) + b"\n",   # a break stretch longer than a tile
        rnd(10_000).lower() + b"\n>x" + b"y" * 9000 + b"\n" + rnd(33) + b">tail header\n",   # header longer than two tiles; file ends after a header
        b"\n\n\r\n" + rnd(50) + b"\r\n\r\n" + rnd(50) + b"-*" + rnd(4) + b"\n",
        b"N" * 70 + b"\n" + rnd(100) + b"\n",                          # breaks BEFORE the first base: no leading invalid position
        b">" + b"A" * 5000 + b"\n" + rnd(20_000) + b"\n",              # bases inside a header are not bases
        bytes(rng.integers(0, 256, 60_000, dtype=np.uint8)).replace(b">", b"#") + b"\n",   # arbitrary bytes incl. >= 128
        # files of many 16 KiB groups (the one-pass tokeniser's look-back walks more than its 64-group window): a header of 1.3 MB in
        # the middle (eighty groups that emit nothing), a run of breaks of 1.1 MB, bases without a single line end
        rnd(200_000) + b"\n>" + b"h" * 1_300_000 + b"\n" + rnd(150_000) + b"\n" + b"N" * 1_100_000 + rnd(40_000) + b"\n>last\n" + rnd(70_001),
        rnd(2_500_000),
        b">x\n" + (rnd(59) + b"\n") * 30_000 + b">y" + b" " * 16_383 + b"\n" + rnd(16_384) + b"n" + rnd(16_383) + b"\n",
    ]


def test_device_tokeniser_writes_what_the_host_tokeniser_writes(shuf_l3k10):
    import torch
    dev = torch.device("cuda", 0)
    texts = _cases()
    hb = K.Batch()
    first = hb.reserve([len(t) for t in texts])
    for i, t in enumerate(texts):
        if len(t):
            hb.fill_text(first + i, t)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        buf, offs, lens = ctx._text_layout(texts)
        co = hb.chunk_off()
        d_text = torch.from_numpy(buf).to(dev)
        nchunks = int(co[-1])
        d_packed = torch.full((nchunks * K.CHUNK_WORDS + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)   # zeroed by the call
        d_mask = torch.full((nchunks * K.CHUNK_MASKW + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        rc, bad, npos = ctx.tokenise_fasta_device(d_text, offs, lens, d_packed, d_mask, co)
        assert rc == 0 and bad == -1
        assert np.array_equal(npos, np.array([hb.n_positions(first + i) for i in range(len(texts))], dtype=np.uint64))
        gp = d_packed.cpu().numpy().view(np.uint32)
        gm = d_mask.cpu().numpy().view(np.uint32)
        assert np.array_equal(gp, hb.packed()[:len(gp)])
        assert np.array_equal(gm, hb.mask()[:len(gm)])
        # a header that is not closed: the file index comes back, like the host tokeniser's error
        bad_texts = [texts[0], b">never ends", texts[5], b"ACGT>also open"]
        b2, o2, l2 = ctx._text_layout(bad_texts)
        co2 = np.concatenate([[0], np.cumsum([(len(t) + 4095) // 4096 for t in bad_texts])]).astype(np.uint64)
        rc, bad, _ = ctx.tokenise_fasta_device(torch.from_numpy(b2).to(dev), o2, l2, d_packed, d_mask, co2)
        assert rc == K.capi.ERR_INPUT and bad == 1
        with pytest.raises(K.KssdError) as e:
            ctx.sketch_fasta_texts(bad_texts)
        assert e.value.code == K.capi.ERR_INPUT and e.value.bad_genome == 1
    finally:
        ctx.close()


def test_sketches_of_device_tokenised_text_equal_the_oracle(shuf_l3k10):
    rng = np.random.default_rng(5)
    texts = [t for t in _cases() if t not in (b"",)]
    texts += [fasta_text(rng.integers(0, 4, n, dtype=np.uint8), b"g%d" % n, n_mask=rng.random(n) < 1e-4) for n in (5_000_000, 1_234_567)]
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        off, ids = ctx.sketch_fasta_texts(texts)
        sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
        for g, t in enumerate(texts):
            assert np.array_equal(ids[int(off[g]):int(off[g + 1])], np.sort(sk.fasta(t))), g
        # with first positions: the reference's file order comes out of the same call
        off2, ids2, pos2 = ctx.sketch_fasta_texts(texts, with_pos=True)
        assert np.array_equal(off, off2) and np.array_equal(ids, ids2)
        g = len(texts) - 2
        lo, hi = int(off[g]), int(off[g + 1])
        assert np.array_equal(K.slot_order_pos(ids2[lo:hi], pos2[lo:hi], sk.p.hashsize), sk.fasta(texts[g]))
    finally:
        ctx.close()


# ---- FASTQ (fastq2co with -Q 0) ----------------------------------------------------------------------------------------

def _fastq_cases():
    from synth import fastq_records
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rnd(n):
        return bytes(acgt[rng.integers(0, 4, n, dtype=np.uint8)])

    def rec(seq, name=b"@r", plus=b"+", qual=None, eol=b"\n"):
        return name + eol + seq + eol + plus + eol + (b"I" * len(seq) if qual is None else qual) + eol
    reads = rng.integers(0, 4, (20_000, 150), dtype=np.uint8)
    long_reads = b"".join(rec(rnd(int(n))) for n in rng.integers(1, 12_000, 60))       # lines that cross tiles
    return [
        fastq_records(reads),                                                            # 6.3 MB of equally long reads
        long_reads,
        rec(rnd(100)),                                                                   # one record
        rec(rnd(100)) + rec(rnd(77))[:-1],                                               # the final record lacks its line end: not scanned
        rec(rnd(100)) + b"@r\n" + rnd(50) + b"\n+\n",                                    # three lines of a second record
        rec(rnd(100)) + b"@r\n" + rnd(50),                                               # ... two, the second one open
        rec(rnd(60) + b"N" + rnd(60) + b"nRY-" + rnd(9).lower()) + rec(b"") + rec(b"N" * 40) + rec(rnd(30)),   # breaks, an empty read
        rec(rnd(90), eol=b"\r\n") + rec(rnd(90), eol=b"\r\n"),                           # CRLF: the \r breaks the run at the line end anyway
        rec(rnd(80), name=b"@" + b"ACGT" * 2000, plus=b"+" + b"ACGT" * 1500) + rec(rnd(40), qual=b"ACGT" * 10),   # bases in the other lines
        b"".join(rec(rnd(9)) for _ in range(3000)),                                      # many short records per tile
        rec(rnd(18_900)) + rec(rnd(4096 - 6)) + rec(rnd(4096)),                          # just under the long-line bound; lines ending on tile edges
        rec(rnd(100), qual=b"\x01" * 50) + rec(rnd(10), qual=b"!" * 300),                # quality lines of another length
    ]


def _host_fastq(texts):
    hb = K.Batch()
    first = hb.reserve([len(t) for t in texts])
    lines = [hb.fill_text(first + i, t, kind=1, Q=0) if len(t) else 0 for i, t in enumerate(texts)]
    return hb, first, lines


def test_device_fastq_tokeniser_writes_what_the_host_tokeniser_writes(shuf_l3k10):
    import torch
    dev = torch.device("cuda", 0)
    texts = _fastq_cases()
    hb, first, lines = _host_fastq(texts)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        buf, offs, lens = ctx._text_layout(texts)
        co = hb.chunk_off()
        d_text = torch.from_numpy(buf).to(dev)
        nchunks = int(co[-1])
        d_packed = torch.full((nchunks * K.CHUNK_WORDS + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        d_mask = torch.full((nchunks * K.CHUNK_MASKW + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        rc, bad, npos, nlines = ctx.tokenise_fastq_device(d_text, offs, lens, d_packed, d_mask, co)
        assert rc == 0 and bad == -1
        assert np.array_equal(nlines, np.array(lines, dtype=np.uint64))
        assert np.array_equal(npos, np.array([hb.n_positions(first + i) for i in range(len(texts))], dtype=np.uint64))
        gp = d_packed.cpu().numpy().view(np.uint32)
        gm = d_mask.cpu().numpy().view(np.uint32)
        assert np.array_equal(gm, hb.mask()[:len(gm)])
        assert np.array_equal(gp, hb.packed()[:len(gp)])
    finally:
        ctx.close()


@pytest.mark.parametrize("text", [
    b"@r\nACGTACGTACGTACGTACGTACGT",                            # no complete record (the reference scans what it has)
    b"@ACGTACGTACGTACGTACGTACGTACGT",                           # ... not even a line end
    b"@r\nACGTACGTACGTACGTACGTACGT\n+\nIIII",                   # ... three line ends
    b"@r\n" + b"ACGT" * 5000 + b"\n+\n" + b"I" * 20000 + b"\n",  # lines the reference's buffer splits
    b"@r\nACGTACGTACGTAC\x00GTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIII\n",   # NUL ends the reference's strlen()
    b"@r\nACGTACGTACGTACGTACGT\n+\nIIIIIIIII\xc3IIIIIIIIII\n",    # a negative quality
    b"@r\nACGT\n+\nIIII\n" + b"x" * 19_500,                     # a long tail without line end
])
def test_device_fastq_tokeniser_hands_back_what_only_the_host_does_exactly(shuf_l3k10, text):
    good = b"@r\nACGTTGCAACGTTGCAACGTTGCAAACCGGTT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        with pytest.raises(K.KssdError) as e:
            ctx.sketch_fastq_texts([good, good, text, b""])
        assert e.value.code == K.capi.ERR_UNSUPPORTED and e.value.bad_genome == 2
        off, ids, lines = ctx.sketch_fastq_texts([good, b"", good])      # an empty file is an empty genome
        assert list(lines) == [4, 0, 4] and off[1] == off[2]
    finally:
        ctx.close()


def test_sketches_of_device_tokenised_fastq_equal_the_oracle(shuf_l3k10):
    texts = _fastq_cases()
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        off, ids, lines = ctx.sketch_fastq_texts(texts)
        sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
        for g, t in enumerate(texts):
            assert np.array_equal(ids[int(off[g]):int(off[g + 1])], np.sort(sk.fastq(t, Q=0))), g
    finally:
        ctx.close()


# ---- fuzz: many small random texts, every byte class in every neighbourhood ---------------------------------------------

def _fuzz_texts(kind, n, seed):
    rng = np.random.default_rng(seed)
    alphabets = [b"ACGT", b"ACGTacgt", b"ACGTN", b"ACGT\n", b"ACGTNRY-*\r\n", b"ACGT>\n", b"AC>GT\n\r N", b"\n", b">", b"N>\n"]
    out = []
    for i in range(n):
        ln = int(rng.choice([0, 1, 2, 15, 16, 17, 31, 33, 100, 4095, 4096, 4097, 9000])) if i % 3 == 0 else int(rng.integers(0, 6000))
        al = np.frombuffer(alphabets[int(rng.integers(0, len(alphabets)))], np.uint8)
        w = rng.random(len(al)) ** 3 + 0.02
        t = bytes(al[rng.choice(len(al), size=ln, p=w / w.sum())])
        if kind == "fastq":
            # line structure at random: newline density from "a few long lines" to "mostly empty lines"; at least one complete record
            t = t.replace(b">", b"@")
            extra = int(rng.integers(0, 40))
            pos = np.sort(rng.integers(0, len(t) + 1, extra))
            parts, last = [], 0
            for p in pos:
                parts.append(t[last:int(p)])
                last = int(p)
            parts.append(t[last:])
            t = b"\n".join(parts)
            t = t + b"\n" * max(0, 4 - t.count(b"\n")) + (b"\n" if rng.random() < 0.5 else b"")
        else:
            if t.rfind(b">") > t.rfind(b"\n"):   # an unclosed last header is an error of its own (tested above)
                t += b"\n"
        out.append(t)
    return out


@pytest.mark.parametrize("kind", ["fasta", "fastq"])
def test_device_tokenisers_fuzz(shuf_l3k10, kind):
    import torch
    dev = torch.device("cuda", 0)
    texts = _fuzz_texts(kind, 600, 2026 if kind == "fasta" else 2027)
    hb = K.Batch()
    first = hb.reserve([len(t) for t in texts])
    lines = []
    for i, t in enumerate(texts):
        lines.append(hb.fill_text(first + i, t, kind=1 if kind == "fastq" else 0, Q=0) if len(t) else 0)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        buf, offs, lens = ctx._text_layout(texts)
        co = hb.chunk_off()
        d_text = torch.from_numpy(buf).to(dev)
        nchunks = int(co[-1])
        d_packed = torch.full((nchunks * K.CHUNK_WORDS + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        d_mask = torch.full((nchunks * K.CHUNK_MASKW + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        r = ctx.tokenise_fasta_device(d_text, offs, lens, d_packed, d_mask, co, fastq=(kind == "fastq"))
        assert r[0] == 0 and r[1] == -1, r[:2]
        want_pos = np.array([hb.n_positions(first + i) for i in range(len(texts))], dtype=np.uint64)
        bad = np.nonzero(r[2] != want_pos)[0]
        assert len(bad) == 0, (int(bad[0]), texts[int(bad[0])][:200], int(r[2][bad[0]]), int(want_pos[bad[0]]))
        if kind == "fastq":
            assert np.array_equal(r[3], np.array(lines, dtype=np.uint64))
        gp = d_packed.cpu().numpy().view(np.uint32)
        gm = d_mask.cpu().numpy().view(np.uint32)
        assert np.array_equal(gm, hb.mask()[:len(gm)])
        assert np.array_equal(gp, hb.packed()[:len(gp)])
    finally:
        ctx.close()


# ---- the quality floor (fastq2co -Q, iseq2comem.c:312) on the device --------------------------------------------------------

def _fastq_with_qualities(rng, n_reads, max_len, crlf=False, lower=False):
    """well-formed records (quality line as long as the bases) with qualities scattered around every floor the test uses"""
    out = []
    nl = b"\r\n" if crlf else b"\n"
    for i in range(n_reads):
        ln = int(rng.integers(0, max_len))
        al = np.frombuffer(b"ACGTacgtN" if lower else b"ACGTN", np.uint8)
        w = np.array([1.0] * (len(al) - 1) + [0.03])
        bases = bytes(al[rng.choice(len(al), size=ln, p=w / w.sum())])
        mode = i % 4
        if mode == 0:
            q = rng.integers(33, 75, ln)                      # every quality
        elif mode == 1:
            q = np.full(ln, 73)                               # all high
            q[rng.random(ln) < 0.05] = 35                     # a few low columns
        elif mode == 2:
            q = np.where(np.arange(ln) < ln // 2, 70, 40)     # a low tail
        else:
            q = rng.integers(1, 127, ln)
            q[q == 10] = 11                                   # (no newline inside the quality line)
            q[q == 13] = 14
        out.append(b"@r%d some text" % i + nl + bases + nl + b"+" + (b"r%d" % i if i % 5 == 0 else b"") + nl + bytes(q.astype(np.uint8)) + nl)
    return b"".join(out)


@pytest.mark.parametrize("Q", [1, 40, 53, 74, 127])
def test_device_fastq_quality_floor_equals_the_host_tokeniser(shuf_l3k10, Q):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(900 + Q)
    texts = [_fastq_with_qualities(rng, 300, 400), _fastq_with_qualities(rng, 40, 6000), _fastq_with_qualities(rng, 200, 160, crlf=True),
             _fastq_with_qualities(rng, 500, 40, lower=True), _fastq_with_qualities(rng, 3, 18000), b"",
             b"@a\nACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n",          # a quality line LONGER than the bases is fine
             _fastq_with_qualities(rng, 50, 300)[:-1]]                                                   # the last record lacks its line end: not scanned
    hb = K.Batch()
    first = hb.reserve([len(t) for t in texts])
    lines = [hb.fill_text(first + i, t, kind=1, Q=Q) if len(t) else 0 for i, t in enumerate(texts)]
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        ctx.set_fastq_quality(Q)
        buf, offs, lens = ctx._text_layout(texts)
        co = hb.chunk_off()
        d_text = torch.from_numpy(buf).to(dev)
        nchunks = int(co[-1])
        d_packed = torch.full((nchunks * K.CHUNK_WORDS + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        d_mask = torch.full((nchunks * K.CHUNK_MASKW + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        rc, bad, npos, nlines = ctx.tokenise_fastq_device(d_text, offs, lens, d_packed, d_mask, co)
        assert rc == 0 and bad == -1, (rc, bad)
        assert np.array_equal(nlines, np.array(lines, dtype=np.uint64))
        want_pos = np.array([hb.n_positions(first + i) for i in range(len(texts))], dtype=np.uint64)
        assert np.array_equal(npos, want_pos), (npos, want_pos)
        gp = d_packed.cpu().numpy().view(np.uint32)
        gm = d_mask.cpu().numpy().view(np.uint32)
        assert np.array_equal(gm, hb.mask()[:len(gm)])
        assert np.array_equal(gp, hb.packed()[:len(gp)])
        # and the sketches against the oracle's fastq2co
        off, ids, lines2 = ctx.sketch_fastq_texts(texts)
        sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
        for g, t in enumerate(texts):
            assert np.array_equal(ids[int(off[g]):int(off[g + 1])], np.sort(sk.fastq(t, Q=Q))), (Q, g)
        # the floor is part of the context: without it the same call gives the -Q 0 sketches
        ctx.set_fastq_quality(0)
        off0, ids0, _ = ctx.sketch_fastq_texts(texts[:2])
        for g in range(2):
            assert np.array_equal(ids0[int(off0[g]):int(off0[g + 1])], np.sort(sk.fastq(texts[g], Q=0)))
    finally:
        ctx.close()


def test_a_quality_line_shorter_than_its_bases_goes_back_to_the_host(shuf_l3k10):
    """the reference compares the columns behind the end of a short quality line with whatever an earlier line left in its
    buffer: only its own fgets() sequence reproduces that"""
    good = b"@r\nACGTTGCAACGTTGCAACGTTGCAAACCGGTT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"
    short = b"@r\nACGTTGCAACGTTGCAACGTTGCAAACCGGTT\n+\nIIIIIIIIIIII\n" + good
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        off, ids, lines = ctx.sketch_fastq_texts([good, short])          # without a floor the quality line is never looked at
        assert list(lines) == [4, 8]
        ctx.set_fastq_quality(40)
        with pytest.raises(K.KssdError) as e:
            ctx.sketch_fastq_texts([good, short, good])
        assert e.value.code == K.capi.ERR_UNSUPPORTED and e.value.bad_genome == 1
        off, ids, lines = ctx.sketch_fastq_texts([good, good])
        assert list(lines) == [4, 4]
    finally:
        ctx.close()


# ---- the framing of dist -A (mt_shortreads2koc, iseq2comem.c:552-615) on the device ---------------------------------------

def test_device_reads_framing_equals_the_host_tokeniser_and_the_oracle_counts(shuf_l3k10):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    genome = rng.integers(0, 4, 150_000, dtype=np.uint8)
    reads = []
    for _ in range(6000):
        s = int(rng.integers(0, len(genome) - 150))
        r = genome[s:s + int(rng.integers(20, 151))].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        reads.append(r)
    texts = [fastq_text(reads), fastq_text(reads[:50], qual=b"#"), _fastq_with_qualities(rng, 200, 3000), b""]
    hb = K.Batch()
    first = hb.reserve([len(t) for t in texts])
    for i, t in enumerate(texts):
        if len(t):
            hb.fill_text(first + i, t, kind=2)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        ctx.set_fastq_reads(True)
        ctx.set_fastq_quality(60)                                   # ignored under the reads framing (dist -A has no quality floor)
        buf, offs, lens = ctx._text_layout(texts)
        co = hb.chunk_off()
        d_text = torch.from_numpy(buf).to(dev)
        nchunks = int(co[-1])
        d_packed = torch.full((nchunks * K.CHUNK_WORDS + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        d_mask = torch.full((nchunks * K.CHUNK_MASKW + K.SLACK_WORDS,), -1, dtype=torch.int32, device=dev)
        rc, bad, npos, nlines = ctx.tokenise_fastq_device(d_text, offs, lens, d_packed, d_mask, co)
        assert rc == 0 and bad == -1, (rc, bad)
        assert np.array_equal(npos, np.array([hb.n_positions(first + i) for i in range(len(texts))], dtype=np.uint64))
        gp = d_packed.cpu().numpy().view(np.uint32)
        gm = d_mask.cpu().numpy().view(np.uint32)
        assert np.array_equal(gm, hb.mask()[:len(gm)]) and np.array_equal(gp, hb.packed()[:len(gp)])
        # ids and occurrences against the oracle's mt_shortreads2koc
        off, ids, cnt, lines = ctx.sketch_fastq_texts(texts, K.SKETCH_KEEP_ZERO | K.SKETCH_COUNTS, with_pos=True)
        sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
        for g, t in enumerate(texts):
            if not len(t):
                assert off[g] == off[g + 1]
                continue
            wi, wc = sk.fastq_koc(t)
            o = np.argsort(wi)
            lo, hi = int(off[g]), int(off[g + 1])
            assert np.array_equal(ids[lo:hi], wi[o]) and np.array_equal(cnt[lo:hi], wc[o].astype(np.uint32)), g
        # what fastq2co and mt_shortreads2koc scan differently goes back to the host: an unterminated last line (dist -A
        # scans that record, fastq2co drops it), a partial record, a line longer than the 4 096-byte buffer
        good = fastq_text(reads[:3])
        for t in (good[:-1], good + b"@x\nACGT\n", good + b"@x\n" + b"ACGT" * 1100 + b"\n+\n" + b"I" * 4400 + b"\n"):
            with pytest.raises(K.KssdError) as e:
                ctx.sketch_fastq_texts([good, t])
            assert e.value.code == K.capi.ERR_UNSUPPORTED and e.value.bad_genome == 1
    finally:
        ctx.close()


def test_device_read_starts_equal_the_host_scanner(shuf_l3k10):
    """dist --byread: kssd_gpu_fasta_read_starts (positions of the reads of one file of the batch tokenised last) against
    kssd_batch_add_fasta_reads on the same bytes -- headers across tile and thread borders, '>' inside a header and
    mid-line, reads without bases, text before the first header, CRLF, several files in one batch; and the by-position
    k-mer stream of the same call against the host-tokenised one"""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rnd(n):
        return bytes(acgt[rng.integers(0, 4, n, dtype=np.uint8)])
    many = b"".join(b">r%d len\n" % i + rnd(int(rng.integers(0, 300))) + (b"\n" if i % 7 else b"\r\n") for i in range(20_000))
    texts = [
        _cases()[1],                                                                # edge.fa
        many,
        rnd(500) + b"\n>a\n>b>c\n" + rnd(40) + b">mid line\nNN" + rnd(30) + b"\n>empty\n>last\n" + rnd(9) + b"\n",
        b">" + b"h" * 8190 + b"\n" + rnd(8192 - 3) + b"\n>" + b"x" * 4094 + b"\n" + rnd(1) + b"\n>z\n",   # markers on tile borders
        b"".join(b">\n" + rnd(1) + b"\n" for _ in range(9000)),                       # one base per read
        b">only\n",
        b"ACGT\n",                                                                  # no read at all
    ]
    ctx, ctx2 = K.GpuCtx(shuf_l3k10, 0), K.GpuCtx(shuf_l3k10, 0)
    try:
        off, ids, pos = ctx.sketch_fasta_texts(texts, flags=K.SKETCH_BY_POS, with_pos=True)
        for f, t in enumerate(texts):
            hb = K.Batch()
            want = hb.add_fasta_reads(t)
            got = ctx.fasta_read_starts(f)
            assert np.array_equal(got, want), (f, len(got), len(want))
            h_off, h_ids, h_pos = ctx2.sketch_batch_pos(hb, K.SKETCH_BY_POS)
            assert np.array_equal(ids[off[f]:off[f + 1]], h_ids) and np.array_equal(pos[off[f]:off[f + 1]], h_pos)
        assert len(ctx.fasta_read_starts(1)) == 20_000 and len(ctx.fasta_read_starts(6)) == 0
        with pytest.raises(K.KssdError):
            ctx.fasta_read_starts(len(texts))
    finally:
        ctx.close()
        ctx2.close()
