"""host/kssd_inflate.c against zlib: the bytes `zcat -fc` would write for a gzip'ed input (iseq2comem.c:187,196-208), member by
member, and the CRC-32 by carry-less multiplication against zlib's.  CPU only."""
import ctypes as C
import gzip
import io
import os
import zlib

import numpy as np
import pytest

import public_kssd_amd as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _lib():
    L = K.host_lib()
    L.kssd_gunzip_mem.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.kssd_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
    L.kssd_crc32.restype = C.c_uint32
    L.kssd_host_free.argtypes = [C.c_void_p]
    P2 = C.POINTER(C.c_void_p) * 2
    L.kssd_gunzip_mem2.argtypes = [C.c_char_p * 2, C.c_size_t * 2, P2, C.POINTER(C.c_size_t) * 2, C.POINTER(C.c_size_t) * 2, C.c_int * 2]
    L.kssd_gunzip_mem2.restype = None
    L.kssd_slurp_reuse2.argtypes = [C.c_char_p * 2, P2, C.POINTER(C.c_size_t) * 2, C.POINTER(C.c_size_t) * 2, C.c_int * 2]
    L.kssd_slurp_reuse2.restype = None
    return L


def gunzip(z):
    L = _lib()
    out, cap, n = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
    rc = L.kssd_gunzip_mem(z, len(z), C.byref(out), C.byref(cap), C.byref(n))
    data = C.string_at(out, n.value) if (rc == 0 and n.value) else b""
    if out.value:
        L.kssd_host_free(out)
    return rc, data


def _pair_call(fn, first):
    """fn = kssd_gunzip_mem2 / kssd_slurp_reuse2: the leading arguments, then out / cap / len / rc for two files"""
    L = _lib()
    out = [C.c_void_p(), C.c_void_p()]
    cap = [C.c_size_t(0), C.c_size_t(0)]
    n = [C.c_size_t(0), C.c_size_t(0)]
    rc = (C.c_int * 2)(-99, -99)
    fn(*first, (C.POINTER(C.c_void_p) * 2)(C.pointer(out[0]), C.pointer(out[1])), (C.POINTER(C.c_size_t) * 2)(C.pointer(cap[0]), C.pointer(cap[1])),
       (C.POINTER(C.c_size_t) * 2)(C.pointer(n[0]), C.pointer(n[1])), rc)
    res = []
    for f in range(2):
        res.append((rc[f], C.string_at(out[f], n[f].value) if (rc[f] == 0 and n[f].value) else b""))
        if out[f].value:
            L.kssd_host_free(out[f])
    return res


def gunzip2(z0, z1):
    L = _lib()
    return _pair_call(L.kssd_gunzip_mem2, ((C.c_char_p * 2)(z0, z1), (C.c_size_t * 2)(len(z0), len(z1))))


def dna(rng, n, width=70, alphabet=b"ACGT"):
    a = np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n)]
    rows = (n + width - 1) // width
    buf = np.full((rows, width + 1), ord("\n"), np.uint8)
    flat = np.zeros(rows * width, np.uint8)
    flat[:n] = a
    buf[:, :width] = flat.reshape(rows, width)
    return b">seq some description\n" + buf.reshape(-1)[: n + rows].tobytes()


def test_crc32_by_carry_less_multiplication_equals_zlib():
    L = _lib()
    rng = np.random.default_rng(1)
    for n in list(range(0, 200)) + [255, 256, 1000, 4095, 4096, 65537, 1 << 20, (1 << 20) + 37]:
        b = rng.integers(0, 256, n + 3, dtype=np.uint8).tobytes()
        for off in (0, 1, 3):                                  # unaligned starts
            x = b[off:off + n]
            assert L.kssd_crc32(0, x, len(x)) == zlib.crc32(x), (n, off)
            assert L.kssd_crc32(0x1234ABCD, x, len(x)) == zlib.crc32(x, 0x1234ABCD), (n, off)   # a running value
    # in pieces = in one go
    b = rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes()
    c = 0
    for i in range(0, len(b), 70_001):
        c = L.kssd_crc32(c, b[i:i + 70_001], len(b[i:i + 70_001]))
    assert c == zlib.crc32(b)


@pytest.mark.parametrize("level", [1, 6, 9])
def test_sequence_text_of_every_compression_level(level):
    rng = np.random.default_rng(level)
    for n in (0, 1, 69, 70, 71, 5000, 300_000, 2_000_000):
        txt = dna(rng, n) if n else b""
        z = gzip.compress(txt, level)
        rc, out = gunzip(z)
        assert rc == 0 and out == txt, (level, n)
    # lower case, N runs, IUPAC codes, CRLF: more symbols, longer codes (second-level tables), real matches
    txt = dna(rng, 400_000, alphabet=b"ACGTacgtNNNNRYKM").replace(b"\n", b"\r\n") + b"N" * 100_000 + dna(rng, 50_000)
    rc, out = gunzip(gzip.compress(txt, level))
    assert rc == 0 and out == txt


def test_stored_fixed_and_repetitive_streams():
    rng = np.random.default_rng(7)
    noise = rng.integers(0, 256, 500_000, dtype=np.uint8).tobytes()           # incompressible: stored blocks
    rc, out = gunzip(gzip.compress(noise, 6))
    assert rc == 0 and out == noise
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        for payload in (b"", b"A", b"ACGT" * 10, dna(rng, 20_000), noise[:3000] * 5, b"\0" * 100_000 + b"x",
                        bytes(range(256)) * 300, (b"AC" * 7 + b"GGT") * 9000):
            co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, strategy)
            z = co.compress(payload) + co.flush()
            rc, out = gunzip(z)
            assert rc == 0 and out == payload, (strategy, len(payload))
    # every short distance (the byte-wise copy, the run of one byte) and the longest matches
    for d in range(1, 40):
        unit = rng.integers(0, 256, d, dtype=np.uint8).tobytes()
        payload = unit * (70_000 // d)
        rc, out = gunzip(gzip.compress(payload, 9))
        assert rc == 0 and out == payload, d
    # raw deflate with a preset of flushes: blocks end everywhere (Z_FULL_FLUSH: empty stored blocks between them)
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    parts, payload = [], b""
    for i in range(50):
        piece = dna(rng, int(rng.integers(1, 5000)))
        payload += piece
        parts.append(co.compress(piece) + co.flush(zlib.Z_FULL_FLUSH if i % 3 else zlib.Z_SYNC_FLUSH))
    parts.append(co.flush())
    rc, out = gunzip(b"".join(parts))
    assert rc == 0 and out == payload


def test_stored_blocks_right_behind_huffman_blocks_in_one_stream():
    """text + noise + text through ONE compressobj: a non-empty stored block follows a Huffman block that ended inside the fast
    symbol loop, whose refill leaves bits of the next input byte above the bit count (they must not survive the
    stored block's copy); alone and two in step, and through the file readers"""
    rng = np.random.default_rng(11)
    streams = []
    for i in range(60):
        co = zlib.compressobj((1, 6, 9)[i % 3], zlib.DEFLATED, 31)
        parts = []
        for _ in range(int(rng.integers(2, 5))):
            parts.append(dna(rng, int(rng.integers(100, 120_000))))
            parts.append(rng.integers(0, 256, int(rng.integers(1, 70_000)), dtype=np.uint8).tobytes())
        if i % 2:
            parts.append(dna(rng, 5_000))
        raw = b"".join(parts)
        z = co.compress(raw) + co.flush()
        assert zlib.decompress(z, 31) == raw
        streams.append((z, raw))
        rc, out = gunzip(z)
        assert rc == 0 and out == raw, i
    for (z0, r0), (z1, r1) in zip(streams[::2], streams[1::2]):
        (rc0, o0), (rc1, o1) = gunzip2(z0, z1)
        assert rc0 == 0 and o0 == r0 and rc1 == 0 and o1 == r1


def test_members_headers_and_trailing_zeros():
    rng = np.random.default_rng(3)
    a, b, c = dna(rng, 10_000), b"", dna(rng, 123_456)
    z = gzip.compress(a, 1) + gzip.compress(b, 9) + gzip.compress(c, 6)        # zcat writes all members one behind the other
    rc, out = gunzip(z)
    assert rc == 0 and out == a + b + c
    rc, out = gunzip(z + b"\0" * 513)                                          # tape padding behind the last member
    assert rc == 0 and out == a + b + c
    # header fields: FNAME (gzip.GzipFile writes it), FEXTRA + FCOMMENT + FHCRC by hand
    bio = io.BytesIO()
    with gzip.GzipFile(filename="genome_with_a_name.fasta", mode="wb", fileobj=bio, mtime=12345) as f:
        f.write(a)
    rc, out = gunzip(bio.getvalue())
    assert rc == 0 and out == a
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(c) + raw.flush()
    hdr = bytes([0x1f, 0x8b, 8, 4 | 8 | 16 | 2, 0, 0, 0, 0, 0, 3]) + (5).to_bytes(2, "little") + b"extra" + b"name\0" + b"a comment\0"
    hdr += (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little")
    z2 = hdr + body + zlib.crc32(c).to_bytes(4, "little") + (len(c) & 0xFFFFFFFF).to_bytes(4, "little")
    assert gzip.decompress(z2) == c
    rc, out = gunzip(z2)
    assert rc == 0 and out == c


def test_two_files_in_step_equal_one_at_a_time(tmp_path):
    """kssd_gunzip_mem2 / kssd_slurp_reuse2: every pairing of files of different lengths, levels, block structures and defects gives
    each file what it gets alone -- the bytes or the refusal"""
    rng = np.random.default_rng(21)
    noise = rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes()
    big = dna(rng, 1_500_000)
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    flushed = b"".join(co.compress(dna(rng, int(rng.integers(1, 3000)))) + co.flush(zlib.Z_FULL_FLUSH if i % 3 else zlib.Z_SYNC_FLUSH) for i in range(40)) + co.flush()
    fixed = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    good = gzip.compress(big, 6)
    bad_crc = bytearray(gzip.compress(dna(rng, 50_000), 6)); bad_crc[-6] ^= 1
    bad_mid = bytearray(good); bad_mid[len(good) // 2] ^= 0x55
    corpus = [gzip.compress(b"", 6), gzip.compress(b"A", 1), gzip.compress(dna(rng, 5000), 1), gzip.compress(big, 1), good,
              gzip.compress(dna(rng, 300_000, alphabet=b"ACGTacgtNNNNRYKM"), 9), gzip.compress(noise, 6), flushed,
              fixed.compress(dna(rng, 20_000)) + fixed.flush(), gzip.compress(dna(rng, 10_000), 1) + gzip.compress(b"", 9) + gzip.compress(dna(rng, 70_000), 6) + b"\0" * 100,
              gzip.compress((b"AC" * 7 + b"GGT") * 9000, 9), bytes(bad_crc), bytes(bad_mid), good[: len(good) // 3], b"not gzip at all", b""]
    alone = [gunzip(z) for z in corpus]
    for i, (rc, out) in enumerate(alone):
        if i < 11:
            assert rc == 0 and out == gzip.decompress(corpus[i]), i
        else:
            assert rc != 0, i
    for i in range(len(corpus)):
        for j in range(len(corpus)):
            r = gunzip2(corpus[i], corpus[j])
            assert (r[0][0] == 0) == (alone[i][0] == 0) and r[0][1] == alone[i][1], (i, j)
            assert (r[1][0] == 0) == (alone[j][0] == 0) and r[1][1] == alone[j][1], (i, j)
    # the file-level pair: gzip'ed with gzip'ed in step, anything else one after the other, a missing file refused on its own
    L = _lib()
    paths = {}
    for name, data in (("a.fa.gz", corpus[3]), ("b.fa.gz", corpus[5]), ("plain.fa", big[:100_000]), ("bad.fa.gz", bytes(bad_mid))):
        paths[name] = str(tmp_path / name).encode()
        open(paths[name], "wb").write(data)
    paths["missing"] = str(tmp_path / "missing.fa.gz").encode()
    want = {"a.fa.gz": alone[3], "b.fa.gz": alone[5], "plain.fa": (0, big[:100_000]), "bad.fa.gz": (1, b""), "missing": (1, b"")}
    for x in paths:
        for y in paths:
            r = _pair_call(L.kssd_slurp_reuse2, ((C.c_char_p * 2)(paths[x], paths[y]),))
            for f, name in enumerate((x, y)):
                assert (r[f][0] == 0) == (want[name][0] == 0) and r[f][1] == want[name][1], (x, y, f)


def test_corrupt_and_truncated_streams_are_refused_not_followed():
    rng = np.random.default_rng(9)
    txt = dna(rng, 200_000)
    z = gzip.compress(txt, 6)
    assert gunzip(b"")[0] != 0 and gunzip(b"not gzip at all" * 10)[0] != 0 and gunzip(txt)[0] != 0
    for cut in (1, 5, 9, 10, 11, 20, len(z) // 2, len(z) - 9, len(z) - 8, len(z) - 1):
        assert gunzip(z[:cut])[0] != 0, cut
    bad_crc = bytearray(z); bad_crc[-8] ^= 1
    bad_len = bytearray(z); bad_len[-1] ^= 1
    assert gunzip(bytes(bad_crc))[0] != 0 and gunzip(bytes(bad_len))[0] != 0
    n_refused = 0
    for i in rng.integers(10, len(z) - 8, 400):                                # a flipped bit anywhere in the body: an error (the CRC at the latest)
        y = bytearray(z)
        y[int(i)] ^= 1 << int(rng.integers(0, 8))
        rc, out = gunzip(bytes(y))
        assert rc != 0 or out == txt
        n_refused += rc != 0
    assert n_refused >= 395
    for _ in range(300):                                                        # noise behind a valid header: never a crash
        y = z[:10] + rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8).tobytes()
        assert gunzip(y)[0] != 0


def test_the_reference_test_genomes_and_the_readers():
    """the gzip'ed fixtures of the reference's own test data through kssd_slurp (what `kssd dist` reads its inputs with): the bytes
    zlib gives; KSSD_ZLIB_GUNZIP=1 keeps zlib's decoder"""
    L = K.host_lib()
    L.kssd_slurp.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    names = []
    for d, _, fs in os.walk(G):
        names += [os.path.join(d, f) for f in fs if f.endswith(".gz")]
    assert len(names) >= 4
    for env in (None, "1"):
        if env:
            os.environ["KSSD_ZLIB_GUNZIP"] = env
        try:
            for p in sorted(names):
                out, n = C.c_void_p(), C.c_size_t(0)
                assert L.kssd_slurp(p.encode(), C.byref(out), C.byref(n)) == 0, p
                assert C.string_at(out, n.value) == gzip.open(p).read(), p
                L.kssd_host_free(out)
        finally:
            os.environ.pop("KSSD_ZLIB_GUNZIP", None)


def test_it_is_faster_than_zlib_on_sequence_text():
    import time
    rng = np.random.default_rng(11)
    txt = dna(rng, 8_000_000)
    z = gzip.compress(txt, 1)
    t0 = time.perf_counter(); rc, out = gunzip(z); t_ours = time.perf_counter() - t0
    t0 = time.perf_counter(); ref = zlib.decompress(z, 31); t_zlib = time.perf_counter() - t0
    assert rc == 0 and out == ref
    print("inflate of %.1f MB of sequence text: ours %.0f MB/s, zlib %.0f MB/s" % (len(txt) / 1e6, len(txt) / t_ours / 1e6, len(txt) / t_zlib / 1e6))
    assert t_ours < t_zlib
