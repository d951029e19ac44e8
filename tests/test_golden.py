"""Golden vectors written by the real reference binary (tests/golden/make_golden.py):
   not-gpu: the oracle and the host C layer reproduce them;  gpu: the HIP path reproduces them."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(G, "golden.json")))
SK = np.load(os.path.join(G, "sketches.npz"))
SH = np.load(os.path.join(G, "shared.npz"))


def fasta_files(sub):
    d = os.path.join(G, sub)
    return sorted(os.path.join(d, f) for f in os.listdir(d))


def read_any(p):
    return gzip.open(p, "rb").read() if p.endswith(".gz") else open(p, "rb").read()


@pytest.fixture(scope="module")
def shuf(shuf_l3k10):
    assert shuf_l3k10.id == META["shuf"]["id"]
    return shuf_l3k10


def test_shuf_is_the_one_the_reference_consumed(shuf, tmp_path):
    p = str(tmp_path / "L3K10.shuf")
    shuf.write(p)
    assert hashlib.sha256(open(p, "rb").read()).hexdigest() == META["shuf"]["sha256"]
    back = K.Shuf.read(p)
    assert np.array_equal(back.table, shuf.table) and (back.k, back.subk, back.drlevel) == (10, 6, 3)
    h, t = ko.read_shuf(p)
    assert h == dict(id=shuf.id, k=10, subk=6, drlevel=3) and np.array_equal(t, shuf.table)


def test_oracle_sketches_equal_reference(shuf):
    sk = ko.Sketcher(shuf.table, 10, 6, 3)
    for sub in ("ref", "qry"):
        for p in fasta_files(sub + "_fa"):
            want = SK["%s/%s" % (sub, os.path.basename(p))]
            assert np.array_equal(np.sort(sk.fasta(read_any(p))), want), p
            assert np.array_equal(np.sort(sk.file(p)), want), p
    fq = read_any(os.path.join(G, "reads.fq.gz"))
    for M in (1, 2):
        assert np.array_equal(np.sort(sk.fastq(fq, 0, M)), SK["fq%d/reads.fq.gz" % M])
    assert len(SK["fq1/reads.fq.gz"]) > len(SK["fq2/reads.fq.gz"]) > 0
    assert len(SK["qry/edge.fa"]) > 20


def test_oracle_abundance_sketch_equals_reference(shuf, tmp_path):
    """dist -A (mt_shortreads2koc + write_fqkoc2files): ids in file order and their u16 occurrences; the host reader
    and writer carry combco.<c>.a and the koc flag of cofiles.stat"""
    A = np.load(os.path.join(G, "abund.npz"))
    sk = ko.Sketcher(shuf.table, 10, 6, 3)
    fq = read_any(os.path.join(G, "reads.fq.gz"))
    ids, counts = sk.fastq_koc(fq)
    assert np.array_equal(ids, A["ids"]) and np.array_equal(counts, A["counts"]) and int(A["koc"]) == 1
    assert counts.max() > 1
    # the same set as the plain FASTQ sketch with -n 1
    assert np.array_equal(np.sort(ids), SK["fq1/reads.fq.gz"])
    b = K.Batch()
    assert b.add_reads(fq) == 3000 and b.n_genomes == 1
    # writer / reader: slot order restored from sorted ids, abundances follow their ids
    o = np.argsort(ids)
    s = K.SketchSet(shuf.id, 20, 6, 1, ["reads.fq.gz"], [0, len(ids)], ids[o], counts[o])
    s.write(str(tmp_path / "koc"), sk.p.hashsize, slot_order=True)
    assert np.array_equal(np.fromfile(str(tmp_path / "koc" / "combco.0"), np.uint32), A["ids"])
    assert np.array_equal(np.fromfile(str(tmp_path / "koc" / "combco.0.a"), np.uint16), A["counts"])
    assert np.array_equal(np.fromfile(str(tmp_path / "koc" / "combco.index.0"), np.uint64), A["index"])
    head = np.fromfile(str(tmp_path / "koc" / "cofiles.stat"), np.uint8)[:32]
    keep = np.r_[0:5, 8:32]  # bytes 5..7 pad the reference's bool
    assert np.array_equal(head[keep], A["stat_head"][keep])
    back = K.SketchSet.read(str(tmp_path / "koc"))
    assert np.array_equal(back.ids, A["ids"]) and np.array_equal(back.counts, A["counts"])


def golden_sets():
    rn, qn = [str(x) for x in SH["ref_names"]], [str(x) for x in SH["qry_names"]]
    roff = np.cumsum([0] + [len(SK["ref/" + n]) for n in rn]).astype(np.uint64)
    qoff = np.cumsum([0] + [len(SK["qry/" + n]) for n in qn]).astype(np.uint64)
    rids = np.concatenate([SK["ref/" + n] for n in rn])
    qids = np.concatenate([SK["qry/" + n] for n in qn])
    ref = K.SketchSet(META["shuf"]["id"], 20, 6, 1, ["REF/" + n for n in rn], roff, rids)
    qry = K.SketchSet(META["shuf"]["id"], 20, 6, 1, ["QRY/" + n for n in qn], qoff, qids)
    return ref, qry


BYREAD_SHUFS = {"L3K10": (10, 6, 3), "L3K11": (11, 6, 3)}


def byread_inputs():
    return {"byread.fa": read_any(os.path.join(G, "byread.fa.gz")), "edge.fa": read_any(os.path.join(G, "qry_fa", "edge.fa"))}


def test_oracle_byread_stream_equals_reference():
    """dist --byread (reads2mco, iseq2comem.c:78-186): the k-mer stream per component and the cumulative per-read index,
    as the reference binary wrote them (tests/golden/make_golden_byread.py), 1 and 16 components"""
    B = np.load(os.path.join(G, "byread.npz"))
    for tag, (k, s, l) in BYREAD_SHUFS.items():
        sh = K.Shuf.generate(k, s, l, seed=META["seed"])
        sk = ko.Sketcher(sh.table, k, s, l)
        for name, text in byread_inputs().items():
            files = sk.byread_files(text)
            assert len(files) == (16 if k == 11 else 1)
            for c, (ids, idx) in files.items():
                assert np.array_equal(ids, B["%s/%s/co.%d" % (tag, name, c)]), (tag, name, c)
                assert np.array_equal(idx, B["%s/%s/idx.%d" % (tag, name, c)]), (tag, name, c)
    assert len(B["L3K10/byread.fa/co.0"]) > 80 and len(B["L3K10/byread.fa/idx.0"]) == 98


def test_oracle_shared_counts_equal_reference():
    ref, qry = golden_sets()
    assert np.array_equal(ref.off[1:] - ref.off[:-1], SH["ref_sz"]) and np.array_equal(qry.off[1:] - qry.off[:-1], SH["qry_sz"])
    assert np.array_equal(ko.shared_counts(ref.off, ref.ids, qry.off, qry.ids), SH["shared"])


CASES = {"M0_O2": dict(metric=0, pfield=2), "M1_O1": dict(metric=1, pfield=1),
         "M0_N2_D": dict(metric=0, pfield=2, n_max=2, dthreshold=0.2, correction=1)}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_distance_report_text_equals_reference(tag, tmp_path):
    ref, qry = golden_sets()
    want = open(os.path.join(G, "distance_%s.out" % tag), "rb").read()
    c = CASES[tag]
    # the oracle's printer
    p1 = str(tmp_path / "o.out")
    ko.dist_print(p1, SH["shared"], SH["ref_sz"], SH["qry_sz"], ref.names, qry.names, 20, 6, c["metric"], c["pfield"],
                  c.get("correction", 0), c.get("dthreshold", 1.0), c.get("n_max", 0))
    assert open(p1, "rb").read() == want
    # the product's host C printer
    p2 = str(tmp_path / "h.out")
    K.distance_print(p2, SH["shared"], ref, qry, threads=3, **c)
    assert open(p2, "rb").read() == want


@pytest.mark.gpu
def test_hip_sketches_equal_reference(gpu_ctx, shuf):
    for sub in ("ref", "qry"):
        b = K.Batch()
        files = fasta_files(sub + "_fa")
        for p in files:
            b.add_file(p)
        off, ids = gpu_ctx.sketch_batch(b)
        for g, p in enumerate(files):
            assert np.array_equal(ids[int(off[g]):int(off[g + 1])], SK["%s/%s" % (sub, os.path.basename(p))]), p
    for M in (1, 2):
        b = K.Batch()
        assert b.add_file(os.path.join(G, "reads.fq.gz"), is_fastq=True) == 12000
        off, ids = gpu_ctx.sketch_batch(b, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY, min_occ=M)
        assert np.array_equal(ids, SK["fq%d/reads.fq.gz" % M])


@pytest.mark.gpu
def test_hip_shared_counts_equal_reference(gpu_ctx):
    ref, qry = golden_sets()
    shared, J, MD, Cc, AD = gpu_ctx.dist(ref.off, ref.ids, qry.off, qry.ids)
    assert np.array_equal(shared, SH["shared"])
    # the printed Jaccard / MashD columns of the reference (6 decimals) agree with the device planes
    lines = open(os.path.join(G, "distance_M0_O2.out")).read().splitlines()[1:]
    assert len(lines) == shared.size
    for i, ln in enumerate(lines):
        f = ln.split("\t")
        q, r = divmod(i, shared.shape[1])
        assert abs(float(f[3]) - J[q, r]) <= 5.1e-7 and abs(float(f[4]) - MD[q, r]) <= 5.1e-7


def test_set_operation_goldens_pin_the_numpy_restatement():
    """tests/test_gpu_set.py checks the device against numpy's unique / isin; here that restatement is pinned against what
    the reference binary's `kssd set` wrote for the golden sketches (tests/golden/make_golden_set.py)."""
    import numpy as np
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sk = np.load(os.path.join(g, "sketches.npz"))
    gold = np.load(os.path.join(g, "set_ops.npz"))
    ref_ids = np.concatenate([sk[k] for k in sk.files if k.startswith("ref/")])
    v, n = np.unique(ref_ids, return_counts=True)
    assert np.array_equal(v, gold["union"])
    assert np.array_equal(v[n == 1], gold["uniq"])
    for tag, pan, keep in (("sub", gold["union"], False), ("int", gold["union"], True), ("intq", gold["uniq"], True)):
        for j, nm in enumerate(gold[tag + "_names"]):
            got = gold[tag + "_ids"][int(gold[tag + "_index"][j]):int(gold[tag + "_index"][j + 1])]
            mine = sk["qry/" + str(nm)]
            want = mine[np.isin(mine, pan) == keep]
            assert np.array_equal(np.sort(got), want), (tag, nm)  # the reference keeps its file (hash-slot) order
            assert gold[tag + "_sizes"][j] == len(want)
