"""CPU check of the host/device-shared bit manipulation in public_kssd_amd/csrc/kssd_core.h.

tests/emu/emu_sketch.cpp (a test helper, not product code) runs the stage-1 group filter, the stage-1.5
Bloom test and the stage-2 exact evaluation lane by lane on the packed layout the host tokeniser writes; the emitted (genome, id) pairs
must equal the oracle's sketches, and stage 1 must never lose a k-mer that stage 2 on every position finds.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K
from synth import fasta_text

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(HERE, "emu", "libemu_sketch.so")
    src = os.path.join(HERE, "emu", "emu_sketch.cpp")
    core = os.path.join(HERE, "..", "public_kssd_amd", "csrc", "kssd_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", src, "-o", so])
    L = C.CDLL(so)
    L.emu_sketch.restype = C.c_long
    L.emu_sketch.argtypes = [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_uint64, C.c_void_p, C.c_int, C.c_void_p,
                                                                  C.c_uint64, C.c_void_p, C.c_int]
    L.emu_log_over_k.restype = None
    L.emu_log_over_k.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_uint64]
    L.emu_check_exact_table.restype = C.c_long
    L.emu_check_exact_table.argtypes = [C.c_int] * 3 + [C.c_uint32, C.c_int, C.c_int]
    L.emu_sketch_where.restype = C.c_long
    L.emu_sketch_where.argtypes = [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    return L


def run_emu(L, shuf, batch, brute=0, gw=0):
    p, m, co = batch.packed(), batch.mask(), batch.chunk_off()
    gid = np.repeat(np.arange(batch.n_genomes, dtype=np.uint32), np.diff(co).astype(np.int64))
    out = np.zeros(batch.n_chunks * 4096 + 16, dtype=np.uint64)
    nc = (C.c_uint64 * 2)(0, 0)
    n = L.emu_sketch(shuf.k, shuf.subk, shuf.drlevel, shuf.table.ctypes.data, p.ctypes.data, m.ctypes.data,
                     batch.n_chunks, gid.ctypes.data, brute, out.ctypes.data, len(out), nc, gw)
    assert n >= 0, n
    return out[:n], (nc[0], nc[1])


def texts(rng):
    def rnd(n, pn=0.0):
        return fasta_text(rng.integers(0, 4, n, dtype=np.uint8), n_mask=(rng.random(n) < pn) if pn else None)
    multi = b"".join(rnd(60000, 0.0005) for _ in range(3))
    return [multi, rnd(8000), b">e\nACGT\n", rnd(4096 * 3 - 7), rnd(4096 * 2)]


# (one subk = 7 case: its 1 GiB .shuf takes most of this module's time; 10/7/5 is covered by the -m gpu tests)
CASES = [(k, s, d, 0) for k, s, d in [(10, 6, 3), (8, 5, 2), (9, 6, 3), (11, 6, 3), (8, 4, 1), (12, 7, 4), (9, 3, 1)]]
CASES += [(10, 6, 3, 4)]  # gw = 0: the kernel's KSSD_GW; 4: another instantiation of the same templates


@pytest.mark.parametrize("k,subk,dr,gw", CASES)
def test_stage1_stage2_match_oracle(emu, k, subk, dr, gw):
    rng = np.random.default_rng(100 * k + subk)
    shuf = K.Shuf.generate(k, subk, dr, seed=42 + k)
    b = K.Batch()
    tx = texts(rng)
    for t in tx:
        b.add_fasta(t)
    got, ncand = run_emu(emu, shuf, b, gw=gw)
    assert ncand[0] != 2 ** 64 - 1, "the k-mer carried from the scanning lane differs from the packed stream's"
    sk = ko.Sketcher(shuf.table, k, subk, dr)
    for g, t in enumerate(tx):
        ids, comps = sk.fasta(t, with_comps=True)
        want = np.sort((ids.astype(np.uint64) << np.uint64(sk.p.comp_bits)) | comps.astype(np.uint64))
        mine = np.unique(got[(got >> np.uint64(32)) == g] & np.uint64(0xFFFFFFFF))
        mine = mine[mine != 0]
        assert np.array_equal(mine, want), (g, len(mine), len(want))
    brute, _ = run_emu(emu, shuf, b, brute=1)
    assert np.array_equal(np.sort(got), np.sort(brute))
    npos = b.n_chunks * 4096
    print("k=%d subk=%d gw=%d candidates after stage 1: %.3f %%, after the Bloom test: %.4f %%"
          % (k, subk, gw, 100.0 * ncand[0] / npos, 100.0 * ncand[1] / npos))
    if subk == 6 and gw == 0:
        # two alignments of 5-window groups over a 17-bit index let ~0.8 % of the positions through; the Bloom test
        # of the exact pattern leaves the ~0.05 % that are in S plus ~2 % of the rest
        assert 0.006 < ncand[0] / npos < 0.010
        assert 0.0004 < ncand[1] / npos < 0.0009


def test_byread_host_side_against_the_reference_goldens(emu, tmp_path):
    """dist --byread without a GPU: the tokeniser's cut points (kssd_batch_add_fasta_reads) and the file writer
    (kssd_byread_write), fed with the k-mer stream the CPU emulation of the device arithmetic samples in position order,
    leave the files the reference binary wrote (tests/golden/byread.npz), 1 and 16 components"""
    import gzip
    import json
    G = os.path.join(HERE, "golden")
    B = np.load(os.path.join(G, "byread.npz"))
    seed = json.load(open(os.path.join(G, "golden.json")))["seed"]
    inputs = {"byread.fa": gzip.open(os.path.join(G, "byread.fa.gz"), "rb").read(),
              "edge.fa": open(os.path.join(G, "qry_fa", "edge.fa"), "rb").read()}
    for tag, (k, s, l) in {"L3K10": (10, 6, 3), "L3K11": (11, 6, 3)}.items():
        shuf = K.Shuf.generate(k, s, l, seed=seed)
        for name, text in inputs.items():
            b = K.Batch()
            cuts = b.add_fasta_reads(text)
            assert len(cuts) == text.count(b">") and np.all(np.diff(cuts.astype(np.int64)) >= 0)
            p, m = b.packed(), b.mask()
            gid = np.zeros(b.n_chunks, np.uint32)
            out = np.zeros(b.n_chunks * 4096 + 16, np.uint64)
            where = np.zeros_like(out)
            n = emu.emu_sketch_where(k, s, l, shuf.table.ctypes.data, p.ctypes.data, m.ctypes.data, b.n_chunks,
                                     gid.ctypes.data, out.ctypes.data, where.ctypes.data, len(out))
            assert n >= 0
            ids, pos = (out[:n] & np.uint64(0xFFFFFFFF)).astype(np.uint32), where[:n].astype(np.uint32)
            assert np.all(np.diff(pos.astype(np.int64)) > 0)
            d = str(tmp_path / (tag + name))
            K.byread_write(d, shuf, name, ids, pos, cuts)
            ncomp = 16 if k == 11 else 1
            for c in range(ncomp):
                assert np.array_equal(np.fromfile(os.path.join(d, "combco.%d" % c), np.uint32), B["%s/%s/co.%d" % (tag, name, c)])
                assert np.array_equal(np.fromfile(os.path.join(d, "combco.index.%d" % c), np.int64), B["%s/%s/idx.%d" % (tag, name, c)])
            assert not os.path.exists(os.path.join(d, "combco.%d" % ncomp))
            stat = np.fromfile(os.path.join(d, "cofiles.stat"), np.uint8)
            keep = np.r_[4:5, 8:32]  # shuf id aside (the goldens' shuffle carries the reference's own random id)
            assert np.array_equal(stat[:32][keep], B["%s/%s/stat" % (tag, name)][keep])
            assert len(stat) == 32 + 4 + 256 and bytes(stat[36:36 + len(name)]) == name.encode()


def test_distance_epilogue_is_within_one_ulp_of_the_host_formula(emu):
    """kssd_log_over_k (kssd_core.h; the device runs the same IEEE + fma code): log(x) / 2k against the host's
    fl(fl(log x) / 2k) with glibc -- never more than ONE ulp apart (north_star tolerance), equal bit for bit in > 97 % of
    the cases; and, like the host value, within 1.33 ulp of the exact quotient (60-digit decimals) on a sample"""
    import math
    from decimal import Decimal, getcontext
    rng = np.random.default_rng(8)

    def ours(x, k):
        x = np.ascontiguousarray(x, np.float64)
        y = np.empty_like(x)
        emu.emu_log_over_k(x.ctypes.data, k, y.ctypes.data, len(x))
        return y

    n = 400_000
    X = rng.integers(1000, 1500, n).astype(np.float64)
    Y = rng.integers(1000, 1500, n).astype(np.float64)
    S = np.minimum(rng.integers(1, 1400, n), np.minimum(X, Y)).astype(np.float64)
    J, Cc = S / (X + Y - S), S / np.minimum(X, Y)
    cases = {"mash": 1 / (2 * J) + 0.5, "aaf": 1 / Cc, "uniform": rng.uniform(1, 1e4, n), "near 1": 1 + rng.uniform(0, 1e-6, n),
             "wide": np.exp(rng.uniform(0, 700, n))}
    for k in (20.0, 16.0, 30.0, 14.0, 24.0):
        for name, arg in cases.items():
            got = ours(arg, k)
            want = np.array([math.log(v) for v in arg]) / k
            d = np.abs(got.view(np.int64) - want.view(np.int64))
            assert d.max() <= 1, (k, name, int(d.max()))
            assert (d == 0).mean() > 0.95, (k, name, float((d == 0).mean()))
    d = np.abs(ours(cases["mash"], 20.0).view(np.int64) - (np.array([math.log(v) for v in cases["mash"]]) / 20.0).view(np.int64))
    assert (d == 0).mean() > 0.97
    # special values: x = 1 -> 0, inf -> inf (clamped to 1 by the caller), nan stays nan
    sp = ours(np.array([1.0, np.inf, np.nan, 1 + 2.0 ** -52]), 20.0)
    assert sp[0] == 0.0 and np.isinf(sp[1]) and np.isnan(sp[2]) and sp[3] == math.log(1 + 2.0 ** -52) / 20.0
    # against the exact quotient
    getcontext().prec = 60
    arg = np.concatenate([cases["mash"][:4000], cases["aaf"][:4000], cases["uniform"][:2000]])
    got = ours(arg, 20.0)
    worst = 0.0
    for v, o in zip(arg, got):
        exact = Decimal(float(v)).ln() / Decimal(20)
        ulp = Decimal(math.ulp(o))
        worst = max(worst, float(abs(Decimal(float(o)) - exact) / ulp))
    assert worst < 1.35, worst  # like the host, whose two roundings stay within 0.52 * 1.6 + 0.5 ulp of the exact quotient


def test_exact_table_placement_finds_every_key_and_nothing_else(emu):
    """kssd_build_tables places every accepted sub-context in one of its two buckets (moving at most one key to make room) and
    kssd_g_find reads them back: all ranks right, no sub-context outside the set found, for several parameter sets and many sets"""
    for k, s, l in ((10, 6, 3), (8, 5, 2), (10, 7, 5), (8, 4, 1), (9, 5, 2)):
        assert emu.emu_check_exact_table(k, s, l, 7 * k + s, 12, 20000) == 0, (k, s, l)
