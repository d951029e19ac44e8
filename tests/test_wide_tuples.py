"""k - drlevel = 9 (-k 12 -s 6 -l 3): 36-bit reduced tuples, 256 component files (iseq2comem.c:63-64,527,542-543).  The
goldens are the REAL reference's files (tests/golden/make_golden_k12.py; its own index builder crashes with 256 components,
so sketches are all there is to compare)."""
import os
import subprocess

import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K
from synth import k12_genomes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K12, SEED = (12, 6, 3), 20260312
BIN = os.path.join(os.path.dirname(G), "..", "public_kssd_amd", "kssd")


def golden_tuples():
    """per genome (by file name): the reference's component files as {component: stored ids in file order}, and all tuples"""
    B = np.load(os.path.join(G, "k12.npz"))
    names = [str(x) for x in B["names"]]
    out = {}
    for g, nm in enumerate(names):
        per = {c: B["co.%d" % c][int(B["idx.%d" % c][g]):int(B["idx.%d" % c][g + 1])] for c in range(256)}
        tup = np.concatenate([(per[c].astype(np.uint64) << np.uint64(8)) | np.uint64(c) for c in range(256)])
        out[nm] = (per, np.sort(tup))
    return out


def test_derived_constants_and_the_oracle_at_nine_reduced_bases():
    d = K.derive(*K12)
    assert (d.comp_num, d.comp_bits, d.hashsize, d.hashlimit) == (256, 8, 536870909, 322122545)
    B = np.load(os.path.join(G, "k12.npz"))
    assert "double free" in str(B["reference_stage2"])          # the reference's stage II at 256 components (recorded, not ours)
    assert sum(len(t) for _, t in golden_tuples().values()) == 371 + 367 + 138


def test_writer_leaves_the_references_component_files(tmp_path):
    """kssd_slot_order_pos64 + kssd_sketchset_write with the tuples' low four bits beside 32-bit ids: combco.<c>,
    combco.index.<c> and cofiles.stat byte for byte the reference's (genomes in the reference's order)"""
    B = np.load(os.path.join(G, "k12.npz"))
    gold = golden_tuples()
    names = [str(x) for x in B["names"]]
    off, ids, sub = [0], [], []
    for nm in names:
        t = gold[nm][1]
        t = K.slot_order_pos64(t, np.arange(len(t), dtype=np.uint32), 536870909)   # (no two of a few hundred tuples meet in 2^29 slots)
        ids.append((t >> np.uint64(4)).astype(np.uint32))
        sub.append((t & np.uint64(15)).astype(np.uint8))
        off.append(off[-1] + len(t))
    shuf_id = int(np.frombuffer(B["stat"][:4].tobytes(), np.uint32)[0])
    s = K.SketchSet(shuf_id, 24, 6, 256, ["fa/" + n for n in names], off, np.concatenate(ids), sub=np.concatenate(sub))
    d = str(tmp_path / "db")
    s.write(d, 536870909, slot_order=False)
    for c in range(256):
        assert np.array_equal(np.fromfile(os.path.join(d, "combco.%d" % c), np.uint32), B["co.%d" % c]), c
        assert np.array_equal(np.fromfile(os.path.join(d, "combco.index.%d" % c), np.uint64), B["idx.%d" % c]), c
    stat = np.fromfile(os.path.join(d, "cofiles.stat"), np.uint8)
    keep = np.r_[0:5, 8:len(B["stat"])]                                            # (bytes 5..7: padding of the bool)
    assert np.array_equal(stat[:len(B["stat"])][keep], B["stat"][keep])
    with pytest.raises(K.KssdError):                                               # read back / indexed / searched: refused
        K.SketchSet.read(d)


@pytest.mark.gpu
def test_sixteen_passes_give_the_references_tuples():
    """pass s keeps the tuples with low bits s, ids = tuple >> 4: full sketch calls per pass, and the passes 1 .. 15 over the
    candidates of ONE scan (KSSD_PHASE_REPASS) -- both equal the reference's files, genome by genome"""
    import torch
    shuf = K.Shuf.generate(*K12, seed=SEED)
    texts = k12_genomes()
    gold = golden_tuples()
    names = sorted(texts)
    ctx = K.GpuCtx(shuf, 0)
    try:
        assert ctx.tuple_passes() == 16
        ctx._last_n = 0
        with pytest.raises(K.KssdError):          # nothing sketched yet: there is no batch to go over again
            ctx.sketch_again()
        with pytest.raises(K.KssdError):
            ctx.set_tuple_pass(16)
        b = K.Batch()
        for nm in names:
            b.add_fasta(texts[nm])
        got = [[] for _ in names]
        for s in range(16):
            ctx.set_tuple_pass(s)
            off, ids = ctx.sketch_batch(b)
            for g in range(len(names)):
                got[g].append((ids[int(off[g]):int(off[g + 1])].astype(np.uint64) << np.uint64(4)) | np.uint64(s))
        for g, nm in enumerate(names):
            assert np.array_equal(np.sort(np.concatenate(got[g])), gold[nm][1]), nm
        # host level, one scan: kssd_gpu_sketch_again for the passes 1 .. 15 (what the command line does), with first positions
        texts_l = [texts[nm] for nm in names]
        ctx.set_tuple_pass(0)
        off0, ids0, pos0 = ctx.sketch_fasta_texts(texts_l, with_pos=True)
        again = [(off0, ids0, pos0)] + [None] * 15
        for s in range(1, 16):
            ctx.set_tuple_pass(s)
            again[s] = ctx.sketch_again(with_pos=True)
        for g, nm in enumerate(names):
            t = np.concatenate([(i[int(o[g]):int(o[g + 1])].astype(np.uint64) << np.uint64(4)) | np.uint64(s) for s, (o, i, p) in enumerate(again)])
            pp = np.concatenate([p[int(o[g]):int(o[g + 1])] for (o, i, p) in again])
            assert np.array_equal(np.sort(t), gold[nm][1]), nm
            order = K.slot_order_pos64(t, pp, 536870909)                     # the reference's file order, component by component
            for c in range(256):
                assert np.array_equal((order[(order & np.uint64(255)) == np.uint64(c)] >> np.uint64(8)).astype(np.uint32), gold[nm][0][c]), (nm, c)
        # one scan, sixteen passes over its candidates
        dev = torch.device("cuda", 0)
        packed = torch.from_numpy(b.packed().view(np.int32)).to(dev)
        mask = torch.from_numpy(b.mask().view(np.int32)).to(dev)
        co = b.chunk_off()
        cap = 4096
        outs = []
        for s in range(16):
            ctx.set_tuple_pass(s)
            d_off = torch.zeros(len(names) + 1, dtype=torch.int64, device=dev)
            d_ids = torch.zeros(cap, dtype=torch.int32, device=dev)
            ctx.sketch_plan(packed, mask, co, d_off, d_ids, cap)
            for ph in ((K.PHASE_PREP, K.PHASE_SCAN) if s == 0 else (K.PHASE_REPASS,)) + (K.PHASE_EXACT, K.PHASE_FINISH):
                ctx.sketch_phase(ph)
            rc, total, bad = ctx.sketch_status()
            assert rc == 0 and total == int(d_off[-1].item())
            outs.append((d_off.cpu().numpy().astype(np.uint64), d_ids.cpu().numpy().view(np.uint32)))
        for g, nm in enumerate(names):
            t = np.concatenate([(i[int(o[g]):int(o[g + 1])].astype(np.uint64) << np.uint64(4)) | np.uint64(s) for s, (o, i) in enumerate(outs)])
            assert np.array_equal(np.sort(t), gold[nm][1]), nm
        ctx.set_tuple_pass(0)
        b.close()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_passes_after_a_region_overflow_keep_the_scanned_layout():
    """A batch whose staging regions overflow at pass 0 (2 Mb of a period-4 repeat whose 12-mer is an accepted sub-context:
    every fourth position is sampled, 500 x the rate the regions are sized for) is repeated with larger regions; the passes
    1 .. 15 then run on the candidate list and the layout tables of THAT scan -- the layout may not move between them (the
    regions used to shrink back after the first pass that filled them to under a quarter, which every pass of a sixteenth
    does).  All sixteen passes against the oracle, twice over the same context, ordinary genomes beside the repeat."""
    from synth import fasta_text
    shuf0 = K.Shuf.generate(*K12, seed=SEED)
    tab = shuf0.table.copy()
    code = 0
    for ch in b"ACGT" * 3:
        code = (code << 2) | b"ACGT".index(ch)
    y = int(np.nonzero(tab == 7)[0][0])          # the repeat's sub-context gets rank 7: still a permutation
    tab[y], tab[code] = tab[code], 7
    shuf = K.Shuf((shuf0.id,) + K12, tab)
    rng = np.random.default_rng(99)
    texts = [fasta_text(rng.integers(0, 4, 700_000, dtype=np.uint8)), b">str\n" + b"ACGT" * 500_000 + b"\n",
             fasta_text(rng.integers(0, 4, 300_000, dtype=np.uint8)), b">polyA\n" + b"A" * 300_000 + b"\n"]
    sk = ko.Sketcher(shuf.table, *K12)
    want = []
    for t in texts:
        ids, comps = sk.fasta(t, with_comps=True)
        want.append(np.sort((ids.astype(np.uint64) << np.uint64(8)) | comps.astype(np.uint64)))
    assert len(want[1]) >= 1
    ctx = K.GpuCtx(shuf, 0)
    try:
        for rep in range(2):
            ctx.set_tuple_pass(0)
            passes = [ctx.sketch_fasta_texts(texts, with_pos=True)]
            if rep == 0:
                assert ctx.scan_stats()[1] > 400_000          # the flood happened
            for s in range(1, 16):
                ctx.set_tuple_pass(s)
                passes.append(ctx.sketch_again(with_pos=True))
            for g in range(len(texts)):
                t = np.concatenate([(i[int(o[g]):int(o[g + 1])].astype(np.uint64) << np.uint64(4)) | np.uint64(s)
                                    for s, (o, i, p) in enumerate(passes)])
                assert np.array_equal(np.sort(t), want[g]), (rep, g)
        ctx.set_tuple_pass(0)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_command_line_writes_the_references_256_component_files(tmp_path):
    d = str(tmp_path)
    B = np.load(os.path.join(G, "k12.npz"))
    gold = golden_tuples()
    K.Shuf.generate(*K12, seed=SEED).write(os.path.join(d, "k12.shuf"))
    os.mkdir(os.path.join(d, "fa"))
    for nm, t in k12_genomes().items():
        open(os.path.join(d, "fa", nm), "wb").write(t)
    r = subprocess.run([BIN, "dist", "-L", "k12.shuf", "-o", "db", "fa"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()
    assert b"hashsize=536870909\thashlimit=322122545" in r.stdout
    stat = open(os.path.join(d, "db", "cofiles.stat"), "rb").read()
    n = int(np.frombuffer(stat[20:24], np.int32)[0])
    assert n == 3 and int(np.frombuffer(stat[16:20], np.int32)[0]) == 256
    names = [stat[32 + 4 * n + 256 * i: 32 + 4 * n + 256 * (i + 1)].split(b"\0")[0].decode() for i in range(n)]
    for g, nm in enumerate(names):                       # (our input order is sorted, the reference's is shuffled by the clock)
        per = gold[os.path.basename(nm)][0]
        for c in range(256):
            idx = np.fromfile(os.path.join(d, "db", "combco.index.%d" % c), np.uint64)
            co = np.fromfile(os.path.join(d, "db", "combco.%d" % c), np.uint32)
            assert np.array_equal(co[int(idx[g]):int(idx[g + 1])], per[c]), (nm, c)
    sizes = np.frombuffer(stat[32:32 + 4 * n], np.uint32)
    assert sorted(sizes.tolist()) == [138, 367, 371]
    # indexing / searching such a directory is refused with the reason (the reference crashes there)
    r = subprocess.run([BIN, "dist", "-r", "db", "-o", "res", "db"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode != 0 and b"256 components" in r.stdout


def _modes_golden():
    return np.load(os.path.join(G, "k12_modes.npz"))


def test_oracle_equals_the_reference_in_the_modes_beyond_the_plain_one():
    """-u, fastq -n 2 and -A at k - drlevel = 9: the oracle's tuples (and abundances) against what the reference binary wrote
    (tests/golden/make_golden_k12_modes.py), component by component, in file order"""
    from synth import k12_mode_inputs
    W = _modes_golden()
    inp = k12_mode_inputs()
    sk = ko.Sketcher(K.Shuf.generate(*K12, seed=SEED).table, *K12)
    ids, comps = sk.fasta(inp["dup.fa"], uniq=True, with_comps=True)
    for c in range(256):
        assert np.array_equal(ids[comps == c], W["u.co.%d" % c]), c
    ids, comps = sk.fastq(inp["reads.fq"], M=2, with_comps=True)
    for c in range(256):
        assert np.array_equal(ids[comps == c], W["n2.co.%d" % c]), c
    assert sum(len(W["u.co.%d" % c]) for c in range(256)) == 153 and sum(len(W["n2.co.%d" % c]) for c in range(256)) == 140


@pytest.mark.gpu
def test_command_line_in_the_modes_beyond_the_plain_one(tmp_path):
    """`kssd dist -L k12.shuf` with -u, -n 2 and -A: one scan, sixteen passes for the tuples and sixteen for their occurrences;
    combco.<0..255>, combco.index.*, combco.*.a and cofiles.stat byte for byte the reference binary's"""
    from synth import k12_mode_inputs
    W = _modes_golden()
    d = str(tmp_path)
    K.Shuf.generate(*K12, seed=SEED).write(os.path.join(d, "k12.shuf"))
    for nm, t in k12_mode_inputs().items():
        open(os.path.join(d, nm), "wb").write(t)
    for tag, extra, inp in (("u", ["-u"], "dup.fa"), ("n2", ["-n", "2"], "reads.fq"), ("A", ["-A"], "reads.fq")):
        r = subprocess.run([BIN, "dist", "-L", "k12.shuf"] + extra + ["-o", "db_" + tag, inp], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=900)
        assert r.returncode == 0, r.stdout.decode()
        for c in range(256):
            assert np.array_equal(np.fromfile(os.path.join(d, "db_" + tag, "combco.%d" % c), np.uint32), W["%s.co.%d" % (tag, c)]), (tag, c)
            assert np.array_equal(np.fromfile(os.path.join(d, "db_" + tag, "combco.index.%d" % c), np.uint64), W["%s.idx.%d" % (tag, c)]), (tag, c)
            if tag == "A":
                assert np.array_equal(np.fromfile(os.path.join(d, "db_A", "combco.%d.a" % c), np.uint16), W["A.a.%d" % c]), c
        stat = np.fromfile(os.path.join(d, "db_" + tag, "cofiles.stat"), np.uint8)
        want = W[tag + ".stat"]
        keep = np.r_[0:5, 8:len(want)]                                              # (bytes 5..7: padding of the bool)
        assert np.array_equal(stat[:len(want)][keep], want[keep]), tag


def test_slot_order_of_wide_tuples_with_a_keep_rule():
    """kssd_slot_order_pos64_keep: every tuple takes its slot of the reference's double-hashing table (insertions in sequence order,
    global_basic.h:228-230), only the kept ones come back, in slot order -- against a plain replay on a table small enough for
    hundreds of collisions; with every tuple kept it is kssd_slot_order_pos64"""
    import ctypes as C
    L = K.host_lib()
    L.kssd_slot_order_pos64_keep.restype = C.c_uint64
    L.kssd_slot_order_pos64_keep.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    rng = np.random.default_rng(12)
    S = 509
    for n in (0, 1, 2, 300):
        t = np.unique(rng.integers(1, 1 << 36, n, dtype=np.uint64))
        n = len(t)
        pos = rng.permutation(10 * n + 5)[:n].astype(np.uint32)
        keep = (rng.random(n) < 0.6).astype(np.uint8)
        table = {}
        for i in np.argsort(pos):
            key = int(t[i])
            h1, h2 = key % S, 1 + key % (S - 1)
            k = 0
            while (h1 + k * h2) % S in table:
                k += 1
            table[(h1 + k * h2) % S] = i
        want = np.array([t[table[s]] for s in sorted(table) if keep[table[s]]], dtype=np.uint64)
        got = t.copy()
        m = L.kssd_slot_order_pos64_keep(got.ctypes.data, pos.ctypes.data, keep.ctypes.data, n, S)
        assert m == len(want) and np.array_equal(got[:m], want), n
        if n:
            allk = np.ones(n, np.uint8)
            got2 = t.copy()
            assert L.kssd_slot_order_pos64_keep(got2.ctypes.data, pos.ctypes.data, allk.ctypes.data, n, S) == n
            assert np.array_equal(got2, K.slot_order_pos64(t, pos, S))
