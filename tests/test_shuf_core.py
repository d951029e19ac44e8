"""The .shuf core cache of the command line (host/kssd_host.c kssd_shuf_read_core): the accepted sub-contexts out of the
mapped table, kept beside the .shuf, re-used only while they describe the file as it is."""
import os
import time

import numpy as np

import public_kssd_amd as K


def _accepted(shuf):
    d = max(16 ** (shuf.subk - shuf.drlevel), 4096)
    acc = np.full(d, 0xFFFFFFFF, dtype=np.uint32)
    x = np.nonzero((shuf.table >= 0) & (shuf.table < d))[0]
    acc[shuf.table[x]] = x
    return acc


def test_core_is_scanned_cached_and_invalidated(tmp_path, shuf_l3k10):
    p = str(tmp_path / "L3K10.shuf")
    shuf_l3k10.write(p)
    want = _accepted(shuf_l3k10)
    hdr, acc, cached = K.Shuf.read_core(p)
    assert hdr == (shuf_l3k10.id, 10, 6, 3) and not cached and np.array_equal(acc, want)
    assert os.path.getsize(p + ".core") == 48 + 4 * 4096
    hdr, acc, cached = K.Shuf.read_core(p)
    assert cached and np.array_equal(acc, want)
    # another shuffle under the same name: the core no longer describes the file and is replaced
    other = K.Shuf.generate(10, 6, 3, seed=99)
    time.sleep(0.01)
    other.write(p)
    hdr, acc, cached = K.Shuf.read_core(p)
    assert not cached and hdr[0] == other.id and np.array_equal(acc, _accepted(other))
    hdr, acc, cached = K.Shuf.read_core(p)
    assert cached and np.array_equal(acc, _accepted(other))
    # a damaged core is ignored
    with open(p + ".core", "r+b") as f:
        f.seek(4)
        f.write(b"\x07\x00\x00\x00")
    hdr, acc, cached = K.Shuf.read_core(p)
    assert not cached and np.array_equal(acc, _accepted(other))


def test_core_of_a_level_that_keeps_more_than_4096_ranks(tmp_path):
    s = K.Shuf.generate(8, 5, 1, seed=5)          # 16^(5-1) = 65 536 accepted ranks
    p = str(tmp_path / "x.shuf")
    s.write(p)
    hdr, acc, cached = K.Shuf.read_core(p)
    assert len(acc) == 65536 and np.array_equal(acc, _accepted(s))


def test_a_table_that_is_not_a_permutation_is_refused(tmp_path, shuf_l3k10):
    t = shuf_l3k10.table.copy()
    i = int(np.nonzero(t == 5)[0][0])
    t[(i + 1) % len(t)] = 5                        # two sub-contexts with rank 5
    bad = K.Shuf((1, 10, 6, 3), t)
    p = str(tmp_path / "bad.shuf")
    bad.write(p)
    try:
        K.Shuf.read_core(p)
        assert False, "accepted a non-permutation"
    except K.KssdError:
        pass
    assert not os.path.exists(p + ".core")


def test_a_core_whose_values_are_off_is_not_believed(tmp_path, shuf_l3k10):
    """the header of a cached core may match while its payload does not: a flipped value (checksum), a value twice or beyond
    16^subk with the checksum recomputed (plausibility), a table rewritten inside one mtime tick at the same size (the 64 sampled
    ranks are read back from the .shuf itself) -- every time the table is scanned again and the core rewritten"""
    import struct
    p = str(tmp_path / "L3K10.shuf")
    shuf_l3k10.write(p)
    want = _accepted(shuf_l3k10)
    K.Shuf.read_core(p)

    def fnv(acc):
        h = 2166136261
        for b in acc.astype("<u4").tobytes():
            h = ((h ^ b) * 16777619) & 0xFFFFFFFF
        return h

    def rewrite(acc, fix_sum):
        raw = bytearray(open(p + ".core", "rb").read())
        raw[48:] = acc.astype("<u4").tobytes()
        if fix_sum:
            raw[44:48] = struct.pack("<I", fnv(acc))
        open(p + ".core", "wb").write(bytes(raw))

    good = open(p + ".core", "rb").read()
    assert struct.unpack("<I", good[44:48])[0] == fnv(want)
    for name, mutate, fix in (("flipped value", lambda a: a.__setitem__(7, a[7] ^ 1), False),
                              ("a value twice", lambda a: a.__setitem__(9, a[10]), True),
                              ("a value beyond 16^subk", lambda a: a.__setitem__(11, 1 << 24), True),
                              ("two ranks swapped", lambda a: a.__setitem__(slice(0, 2), a[1::-1].copy()), True)):
        acc = want.copy()
        mutate(acc)
        rewrite(acc, fix)
        hdr, got, cached = K.Shuf.read_core(p)
        assert not cached and np.array_equal(got, want), name
        assert open(p + ".core", "rb").read() == good, name          # rewritten from the table
