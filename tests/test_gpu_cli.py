"""-m gpu: the `kssd` command line end to end on the golden inputs: same directory protocol and files as the
reference, distance.out byte-identical to what the reference binary printed (tests/golden)."""
import json
import os
import subprocess

import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "public_kssd_amd", "kssd")
META = json.load(open(os.path.join(G, "golden.json")))
SK = np.load(os.path.join(G, "sketches.npz"))


def run(args, cwd, env=None):
    r = subprocess.run([BIN] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stdout.decode()
    return r.stdout.decode()


def test_tutorial_flow_on_golden_inputs(tmp_path):
    d = str(tmp_path)
    out = run(["shuffle", "-k", 10, "-s", 6, "-l", 3, "-o", "L3K10", "--seed", META["seed"]], d)
    assert "shuf_id=%d" % META["shuf"]["id"] in out
    out = run(["dist", "-L", "L3K10.shuf", "-o", "ref", os.path.join(G, "ref_fa")], d)
    assert "hashsize=2097143\thashlimit=1258285" in out
    run(["dist", "-L", "L3K10.shuf", "-o", "qry", os.path.join(G, "qry_fa")], d)
    for sub in ("ref", "qry"):
        sets = ko.sketch_sets_by_name(os.path.join(d, sub))
        assert len(sets) == len(os.listdir(os.path.join(G, sub + "_fa")))
        for nm, ids in sets.items():
            assert np.array_equal(ids, SK["%s/%s" % (sub, nm)]), nm
        hdr, sizes, names = ko.read_stat(os.path.join(d, sub, "cofiles.stat"))
        assert (hdr["shuf_id"], hdr["kmerlen"], hdr["dim_rd_len"], hdr["comp_num"]) == (META["shuf"]["id"], 20, 6, 1)
        assert hdr["all_ctx_ct"] == int(sizes.sum())
    # file order inside a genome = the reference's hash-slot order
    sk = ko.Sketcher(K.Shuf.read(os.path.join(d, "L3K10.shuf")).table, 10, 6, 3)
    hdr, names, off, ids = ko.read_sketch_dir(os.path.join(d, "qry"))
    for i, nm in enumerate(names):
        assert np.array_equal(ids[int(off[i]):int(off[i + 1])], sk.file(nm)), nm
    # search in the three renderings the goldens hold; names differ only by the directory prefix
    for tag, extra in (("M0_O2", []), ("M1_O1", ["-M", 1, "-O", 1]), ("M0_N2_D", ["-N", 2, "-D", "0.2", "--correction", 1])):
        run(["dist", "-r", "ref"] + extra + ["-o", "dist_" + tag, "qry"], d)
        got = open(os.path.join(d, "dist_" + tag, "distance.out")).read()
        got = got.replace(os.path.join(G, "qry_fa"), "QRY").replace(os.path.join(G, "ref_fa"), "REF")
        want = open(os.path.join(G, "distance_%s.out" % tag)).read()
        # the reference shuffles its file order with time(NULL); ours is sorted: compare as sets of lines
        assert got.splitlines()[0] == want.splitlines()[0]
        if "N2" in tag:   # -N keeps rank order inside a query
            assert sorted(got.splitlines()[1:]) == sorted(want.splitlines()[1:])
        else:
            assert sorted(got.splitlines()[1:]) == sorted(want.splitlines()[1:])
        assert not os.path.exists(os.path.join(d, "dist_" + tag, "sharedk_ct.dat"))
    # --keepskf + -f: the kept shared-k-mer matrix reproduces the report
    run(["dist", "-r", "ref", "--keepskf", "-o", "keep", "qry"], d)
    sh = np.fromfile(os.path.join(d, "keep", "sharedk_ct.dat"), np.uint32)
    assert sh.size == 3 * 6 and sh.sum() > 0
    run(["dist", "-r", "ref", "-f", os.path.join(d, "keep", "sharedk_ct.dat"), "-o", "again", "qry"], d)
    assert open(os.path.join(d, "again", "distance.out")).read() == open(os.path.join(d, "keep", "distance.out")).read()


def test_fastq_and_auto_shuf(tmp_path):
    d = str(tmp_path)
    run(["shuffle", "-k", 10, "-s", 6, "-l", 3, "-o", "L3K10", "--seed", META["seed"]], d)
    for M in (1, 2):
        out = run(["dist", "-n", M, "-L", "L3K10.shuf", "-o", "fq%d" % M, os.path.join(G, "reads.fq.gz")], d)
        assert "12000 reads detected" in out
        (nm, ids), = ko.sketch_sets_by_name(os.path.join(d, "fq%d" % M)).items()
        assert np.array_equal(ids, SK["fq%d/reads.fq.gz" % M])
    # -L <level>: a default.shuf is generated in the output directory (get_dim_shuffle, command_dist.c:193-216)
    run(["dist", "-k", 9, "-L", 3, "--seed", 7, "-o", "auto", os.path.join(G, "ref_fa")], d)
    sh = K.Shuf.read(os.path.join(d, "auto", "default.shuf"))
    assert (sh.k, sh.subk, sh.drlevel) == (9, 6, 3)
    sk = ko.Sketcher(sh.table, 9, 6, 3)
    for nm, ids in ko.sketch_sets_by_name(os.path.join(d, "auto")).items():
        assert np.array_equal(ids, np.sort(sk.file(os.path.join(G, "ref_fa", nm))))


def test_abundance_sketch_files_are_the_references(tmp_path):
    """dist -A on reads: combco.0, combco.0.a, combco.index.0 byte for byte what the reference wrote with one thread,
    koc set in cofiles.stat; a search with such a query ignores the abundances like the reference does
    (command_dist.c:880)"""
    d = str(tmp_path)
    A = np.load(os.path.join(G, "abund.npz"))
    run(["shuffle", "-k", 10, "-s", 6, "-l", 3, "-o", "L3K10", "--seed", META["seed"]], d)
    out = run(["dist", "-A", "-L", "L3K10.shuf", "-o", "koc", os.path.join(G, "reads.fq.gz")], d)
    assert "running mt_shortreads2koc()" in out
    assert np.array_equal(np.fromfile(os.path.join(d, "koc", "combco.0"), np.uint32), A["ids"])
    assert np.array_equal(np.fromfile(os.path.join(d, "koc", "combco.0.a"), np.uint16), A["counts"])
    assert np.array_equal(np.fromfile(os.path.join(d, "koc", "combco.index.0"), np.uint64), A["index"])
    head = np.fromfile(os.path.join(d, "koc", "cofiles.stat"), np.uint8)[:32]
    keep = np.r_[0:5, 8:32]
    assert np.array_equal(head[keep], A["stat_head"][keep])
    # -A with a FASTA among the inputs closes the mode with the reference's warning
    out = run(["dist", "-A", "-L", "L3K10.shuf", "-o", "mixed", os.path.join(G, "reads.fq.gz"),
               os.path.join(G, "qry_fa", "edge.fa")], d)
    assert "Warning: close abundance mode (-A) since non-fastq file input." in out
    assert not os.path.exists(os.path.join(d, "mixed", "combco.0.a"))
    assert np.fromfile(os.path.join(d, "mixed", "cofiles.stat"), np.uint8)[4] == 0
    # search: reads against the references, same numbers with and without abundances
    run(["dist", "-L", "L3K10.shuf", "-o", "ref", os.path.join(G, "ref_fa")], d)
    run(["dist", "-L", "L3K10.shuf", "-o", "plain", os.path.join(G, "reads.fq.gz")], d)
    run(["dist", "-r", "ref", "-o", "d_koc", "koc"], d)
    run(["dist", "-r", "ref", "-o", "d_plain", "plain"], d)
    assert open(os.path.join(d, "d_koc", "distance.out")).read() == open(os.path.join(d, "d_plain", "distance.out")).read()


def test_byread_sketch_files_are_the_references(tmp_path):
    """dist --byread: combco.<c> and combco.index.<c> byte for byte what the reference binary wrote (tests/golden/
    byread.npz), 1 and 16 components; with two inputs the second overwrites the first, as in the reference"""
    import gzip
    d = str(tmp_path)
    B = np.load(os.path.join(G, "byread.npz"))
    open(os.path.join(d, "byread.fa"), "wb").write(gzip.open(os.path.join(G, "byread.fa.gz"), "rb").read())
    edge = os.path.join(G, "qry_fa", "edge.fa")
    for tag, k in (("L3K10", 10), ("L3K11", 11)):
        run(["shuffle", "-k", k, "-s", 6, "-l", 3, "-o", tag, "--seed", META["seed"]], d)
        for name, path in (("byread.fa", "byread.fa"), ("edge.fa", edge)):
            out = run(["dist", "--byread", "-L", tag + ".shuf", "-o", "o_" + tag + name, path], d)
            assert "decomposing %s by reads is complete!" % path in out
            o = os.path.join(d, "o_" + tag + name)
            ncomp = 16 if k == 11 else 1
            for c in range(ncomp):
                assert np.array_equal(np.fromfile(os.path.join(o, "combco.%d" % c), np.uint32), B["%s/%s/co.%d" % (tag, name, c)])
                assert np.array_equal(np.fromfile(os.path.join(o, "combco.index.%d" % c), np.int64), B["%s/%s/idx.%d" % (tag, name, c)])
            stat = np.fromfile(os.path.join(o, "cofiles.stat"), np.uint8)
            keep = np.r_[0:5, 8:32]
            assert np.array_equal(stat[:32][keep], B["%s/%s/stat" % (tag, name)][keep])
    # gzip'ed input is unpacked (documented deviation); two inputs: the last one's stream stays, both are named
    run(["dist", "--byread", "-L", "L3K10.shuf", "-o", "two", edge, os.path.join(G, "byread.fa.gz")], d)
    assert np.array_equal(np.fromfile(os.path.join(d, "two", "combco.0"), np.uint32), B["L3K10/byread.fa/co.0"])
    assert os.path.getsize(os.path.join(d, "two", "cofiles.stat")) == 32 + 2 * 4 + 2 * 256


def test_errors_match_the_reference(tmp_path):
    d = str(tmp_path)
    r = subprocess.run([BIN, "shuffle", "-k", "10", "-s", "8", "-l", "5", "-o", "x"], cwd=d, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"subk shoud smaller than 8" in r.stderr
    r = subprocess.run([BIN, "dist", "-k", "10", "-L", "5", "-o", "o", os.path.join(G, "ref_fa")], cwd=d, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"subk shoud smaller than 8" in r.stderr     # auto -L 5 is rejected


def test_config1_tutorial_on_real_test_fna_genomes(tmp_path):
    """BASELINE configs[0] through the HIP command line on real sequence: four of the reference's own test_fna genomes
    (B. cereus AE016877, its 10 %, 25 % and 29 % mutated copies; tests/golden/test_fna) and the multi-record edge-case file
    as a third query, the README quick-tutorial flow; goldens written by the reference binary (make_golden_testfna.py).
    Every genome's slice of combco.0 byte for byte (the reference's file order inside a genome; across genomes its input
    order is shuffled by the clock, ours is sorted: by name), sharedk_ct.dat by names, distance.out as a set of lines."""
    d = str(tmp_path)
    F = os.path.join(G, "test_fna")
    want = np.load(os.path.join(G, "test_fna.npz"))
    os.mkdir(os.path.join(d, "qin"))
    os.mkdir(os.path.join(d, "rin"))
    for fn in ("25_AE016877.fasta.gz", "29_AE016877.fasta.gz"):   # (the directory holds all of test_fna since round 6: test_tutorial_in_full)
        os.symlink(os.path.join(F, "seqs2", fn), os.path.join(d, "qin", fn))
    for fn in ("10_AE016877.fasta.gz", "AE016877.fasta.gz"):
        os.symlink(os.path.join(F, "seqs1", fn), os.path.join(d, "rin", fn))
    os.symlink(os.path.join(G, "qry_fa", "edge.fa"), os.path.join(d, "qin", "edge.fa"))
    run(["shuffle", "-k", 10, "-s", 6, "-l", 3, "-o", "L3K10", "--seed", META["seed"]], d)
    run(["dist", "-L", "L3K10.shuf", "-r", "rin", "-o", "refdb"], d)      # sketch + index files
    run(["dist", "-L", "L3K10.shuf", "-o", "qry", "qin"], d)
    run(["dist", "-r", "refdb", "-o", "out", "--keepskf", "qry"], d)
    order = {}
    for sub, dd in (("ref", "refdb"), ("qry", "qry")):
        hdr, names, off, ids = ko.read_sketch_dir(os.path.join(d, dd))
        order[sub] = [os.path.basename(n) for n in names]
        assert len(names) == (2 if sub == "ref" else 3)
        for i, nm in enumerate(order[sub]):
            assert np.array_equal(ids[int(off[i]):int(off[i + 1])], want["%s/%s" % (sub, nm)]), (sub, nm)
    sh = np.fromfile(os.path.join(d, "out", "sharedk_ct.dat"), np.uint32).reshape(3, 2)
    qi = [order["qry"].index(str(n)) for n in want["qry_names"]]
    ri = [order["ref"].index(str(n)) for n in want["ref_names"]]
    assert np.array_equal(sh[np.ix_(qi, ri)], want["shared"])
    got = open(os.path.join(d, "out", "distance.out")).read().replace("rin/", "REF/").replace("qin", "QRY")
    wt = bytes(want["distance_out"]).decode()
    assert got.splitlines()[0] == wt.splitlines()[0] and sorted(got.splitlines()[1:]) == sorted(wt.splitlines()[1:])
    assert os.path.getsize(os.path.join(d, "refdb", "mco.index.0")) == 8 << 28      # the reference's dense index file


def test_tutorial_in_full(tmp_path):
    """BASELINE configs[0] exactly as the reference's README runs it (README.md:33-45): ALL of test_fna -- seqs1's 20 genomes as
    references, seqs2's 11 as queries -- through the tutorial's five commands with this build's command line, against what the
    reference binary left for the same commands (make_golden_tutorial.py): every genome's ids in the reference's file order, both
    shared matrices by name, both distance.out texts line for line (as sets: the reference orders its inputs by the clock)."""
    d = str(tmp_path)
    F = os.path.join(G, "test_fna")
    want = np.load(os.path.join(G, "tutorial.npz"))
    for sub in ("seqs1", "seqs2"):
        os.symlink(os.path.join(F, sub), os.path.join(d, sub))
    run(["shuffle", "-k", 10, "-s", 6, "-l", 3, "-o", "L3K10", "--seed", META["seed"]], d)
    run(["dist", "-L", "L3K10.shuf", "-o", "reference", "seqs1"], d)
    run(["dist", "-o", "reference", "reference"], d)
    run(["dist", "-L", "L3K10.shuf", "-o", "query", "seqs2"], d)
    run(["dist", "-r", "reference", "-o", "distout", "--keepskf", "query"], d)
    run(["dist", "-r", "reference", "-o", "distout2", "--keepskf", "reference"], d)
    order = {}
    for sub, dd, n in (("ref", "reference", 20), ("qry", "query", 11)):
        hdr, names, off, ids = ko.read_sketch_dir(os.path.join(d, dd))
        order[sub] = [os.path.basename(x) for x in names]
        assert len(names) == n
        for i, nm in enumerate(order[sub]):
            assert np.array_equal(ids[int(off[i]):int(off[i + 1])], want["%s/%s" % (sub, nm)]), (sub, nm)
    ri = [order["ref"].index(str(x)) for x in want["ref_names"]]
    sh = np.fromfile(os.path.join(d, "distout", "sharedk_ct.dat"), np.uint32).reshape(11, 20)
    assert np.array_equal(sh[np.ix_([order["qry"].index(str(x)) for x in want["qry_names"]], ri)], want["shared"])
    sh2 = np.fromfile(os.path.join(d, "distout2", "sharedk_ct.dat"), np.uint32).reshape(20, 20)
    assert np.array_equal(sh2[np.ix_([order["ref"].index(str(x)) for x in want["refq_names"]], ri)], want["shared_refs"])
    for out, key, lines in (("distout", "distance_out", 221), ("distout2", "distance_out_refs", 401)):
        got = open(os.path.join(d, out, "distance.out")).read().splitlines()
        wt = bytes(want[key]).decode().splitlines()
        assert len(got) == lines and got[0] == wt[0] and sorted(got[1:]) == sorted(wt[1:]), out


def test_file_order_of_uniq_and_min_occ_modes(tmp_path):
    """`kssd dist -u` and `kssd dist -n 2` write combco.0 in the reference's order even where kept ids collide with ids
    the dump drops (those keep their slots in the reference's table): small table, hundreds of collisions; the oracle's
    dump order is pinned against the real binary on the same inputs in tests/test_interop_ref.py"""
    from test_interop_ref import _collision_inputs
    d = str(tmp_path)
    shuf = K.Shuf.generate(8, 5, 2, seed=3)
    shuf.write(os.path.join(d, "s.shuf"))
    fa, fq = _collision_inputs()
    open(os.path.join(d, "rep.fasta"), "wb").write(fa)
    open(os.path.join(d, "reads.fastq"), "wb").write(fq)
    sk = ko.Sketcher(shuf.table, 8, 5, 2)
    hs = K.derive(8, 5, 2).hashsize
    run(["dist", "-u", "-L", "s.shuf", "-o", "u", "rep.fasta"], d)
    want = sk.fasta(fa, uniq=True)
    got = np.fromfile(os.path.join(d, "u", "combco.0"), np.uint32)
    assert np.array_equal(got, want)
    assert not np.array_equal(K.slot_order(np.sort(want), hs), want)       # the naive replay of the kept ids alone is NOT this order
    run(["dist", "-n", 2, "-L", "s.shuf", "-o", "n2", "reads.fastq"], d)
    want = sk.fastq(fq, Q=0, M=2)
    assert np.array_equal(np.fromfile(os.path.join(d, "n2", "combco.0"), np.uint32), want)


def test_fastq_inputs_the_device_tokeniser_hands_back(tmp_path):
    """plain .fastq files go to the device as raw bytes (csrc/kssd_tok.inc); a file it cannot do exactly as fastq2co (here:
    a truncated first record, a line the reference's 20 000-byte buffer splits) sends its batch through the host
    tokeniser -- either way combco.0 is the oracle's, and KSSD_HOST_FASTQ=1 (host tokeniser for everything) agrees"""
    from test_gpu_tokenise import _fastq_cases
    d = str(tmp_path)
    rng = np.random.default_rng(4)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    long_line = bytes(acgt[rng.integers(0, 4, 30_000, dtype=np.uint8)])
    files = {"a.fastq": _fastq_cases()[1], "b.fastq": _fastq_cases()[0],
             "long.fastq": b"@r\n" + long_line + b"\n+\n" + b"I" * 30_000 + b"\n" + _fastq_cases()[9],
             "open.fq": b"@r\n" + long_line[:5000]}
    for name, text in files.items():
        open(os.path.join(d, name), "wb").write(text)
    shuf = K.Shuf.generate(10, 6, 3, seed=11)
    shuf.write(os.path.join(d, "s.shuf"))
    sk = ko.Sketcher(shuf.table, 10, 6, 3)
    names = sorted(files)
    out = run(["dist", "-L", "s.shuf", "-o", "dev"] + names, d)
    assert "reads detected" in out
    run(["dist", "-L", "s.shuf", "-o", "host"] + names, d, env={"KSSD_HOST_FASTQ": "1"})
    _, nm_d, off_d, ids_d = ko.read_sketch_dir(os.path.join(d, "dev"))
    _, nm_h, off_h, ids_h = ko.read_sketch_dir(os.path.join(d, "host"))
    assert list(nm_d) == list(nm_h) and len(nm_d) == len(files)
    assert np.array_equal(off_d, off_h) and np.array_equal(ids_d, ids_h)
    for g, name in enumerate(nm_d):     # the reference's file order, genome by genome
        assert np.array_equal(ids_d[int(off_d[g]):int(off_d[g + 1])], sk.fastq(files[os.path.basename(name)], Q=0, M=1)), name


def test_long_plain_files_are_streamed_to_the_device(tmp_path):
    """a long plain file is not read into a host buffer of its size: slices of it pass through a ring of eight page-locked
    buffers (kssd_gpu_text_put) while the worker reads on -- forced here for files of a few megabytes with 64 KiB slices
    (the ring wraps a dozen times), FASTA and FASTQ, next to a batch of ordinary small files; the result is the same
    directory, byte for byte, as without streaming"""
    from test_gpu_tokenise import _fastq_cases
    from synth import fasta_text
    d = str(tmp_path)
    rng = np.random.default_rng(9)
    fq = _fastq_cases()
    files = {"a_reads.fastq": fq[0], "b_long.fastq": fq[1], "c_small.fastq": fq[2],
             "open.fq": b"@r\n" + b"ACGT" * 300_000,      # streamed AND handed back to the host tokeniser
             "g1.fasta": fasta_text(rng.integers(0, 4, 3_000_000, dtype=np.uint8), n_mask=rng.random(3_000_000) < 1e-4),
             "g2.fasta": fasta_text(rng.integers(0, 4, 40_000, dtype=np.uint8)),
             "g3.fasta": fasta_text(rng.integers(0, 4, 1_200_001, dtype=np.uint8))}
    import gzip
    # gzip'ed inputs stream too: inflated slice by slice into the ring.  Their size is an estimate (the trailer's length field,
    # or four times the compressed size): a two-member file whose trailer only speaks for its short last member and
    # constant-quality reads that pack 15 : 1 make the device buffer grow on the way
    files["z_reads.fastq.gz"] = gzip.compress(fq[0][:len(fq[0]) // 2], 1) + gzip.compress(fq[0][len(fq[0]) // 2:][:314 * 10], 1)
    files["z_g.fasta.gz"] = gzip.compress(files["g3.fasta"], 1)
    for name, text in files.items():
        open(os.path.join(d, name), "wb").write(text)
    K.Shuf.generate(10, 6, 3, seed=11).write(os.path.join(d, "s.shuf"))
    names = sorted(files)
    run(["dist", "-L", "s.shuf", "-o", "plain"] + names, d, env={"KSSD_STREAM_MIN": str(1 << 40), "KSSD_STREAM_MIN_GZ": str(1 << 40)})
    run(["dist", "-L", "s.shuf", "-o", "streamed"] + names, d,
        env={"KSSD_STREAM_MIN": str(1 << 20), "KSSD_STREAM_MIN_GZ": "1024", "KSSD_STREAM_SLICE": str(1 << 16)})
    for fn in ("combco.0", "combco.index.0", "cofiles.stat"):
        a = open(os.path.join(d, "plain", fn), "rb").read()
        b = open(os.path.join(d, "streamed", fn), "rb").read()
        assert a == b and len(a) > 0, fn


def test_quality_floor_and_abundance_inputs_through_the_device_tokeniser(tmp_path):
    """`kssd dist -Q 45` and `kssd dist -A` on plain .fastq files: tokenised on the device (the quality column gathered two
    lines down; the framing of mt_shortreads2koc), a file with a short quality line or an unterminated last record handed back to
    the host tokeniser -- combco.0 is the oracle's either way, and KSSD_HOST_FASTQ=1 (host tokeniser for everything) agrees byte for
    byte"""
    from test_gpu_tokenise import _fastq_with_qualities
    d = str(tmp_path)
    rng = np.random.default_rng(45)
    genome = rng.integers(0, 4, 80_000, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def reads_file(n, qual_of):
        out = []
        for i in range(n):
            s = int(rng.integers(0, len(genome) - 150))
            r = bytes(acgt[genome[s:s + 150]])
            out.append(b"@r%d\n" % i + r + b"\n+\n" + bytes(qual_of(150)) + b"\n")
        return b"".join(out)
    files = {"hi.fastq": reads_file(3000, lambda n: np.full(n, 70, np.uint8)),
             "mixed.fastq": reads_file(3000, lambda n: rng.integers(35, 75, n).astype(np.uint8)),
             "tail.fastq": reads_file(3000, lambda n: np.where(np.arange(n) < 100, 72, 36).astype(np.uint8)),
             "wild.fastq": _fastq_with_qualities(rng, 400, 500)}
    files["short_q.fastq"] = files["hi.fastq"][:400] + b"@s\n" + bytes(acgt[genome[:120]]) + b"\n+\nIIII\n" + files["mixed.fastq"][:50_000]
    for name, text in files.items():
        open(os.path.join(d, name), "wb").write(text)
    shuf = K.Shuf.generate(10, 6, 3, seed=12)
    shuf.write(os.path.join(d, "s.shuf"))
    sk = ko.Sketcher(shuf.table, 10, 6, 3)
    names = sorted(files)
    for Q in (45, 0):
        run(["dist", "-Q", Q, "-L", "s.shuf", "-o", "dev%d" % Q] + names, d)
        run(["dist", "-Q", Q, "-L", "s.shuf", "-o", "host%d" % Q] + names, d, env={"KSSD_HOST_FASTQ": "1"})
        _, nm_d, off_d, ids_d = ko.read_sketch_dir(os.path.join(d, "dev%d" % Q))
        _, nm_h, off_h, ids_h = ko.read_sketch_dir(os.path.join(d, "host%d" % Q))
        assert list(nm_d) == list(nm_h) and np.array_equal(off_d, off_h) and np.array_equal(ids_d, ids_h)
        for g, name in enumerate(nm_d):
            assert np.array_equal(ids_d[int(off_d[g]):int(off_d[g + 1])], sk.fastq(files[os.path.basename(name)], Q=Q, M=1)), (Q, name)
    _, nm45, off45, _ = ko.read_sketch_dir(os.path.join(d, "dev45"))
    _, nm0, off0, _ = ko.read_sketch_dir(os.path.join(d, "dev0"))
    assert int(off45[-1]) < int(off0[-1])                           # the floor removes k-mers
    # -A: occurrences; the unterminated last record of open.fastq is scanned by mt_shortreads2koc (and handed back by the device)
    afiles = {"a.fastq": files["hi.fastq"][:200_000], "open.fastq": files["mixed.fastq"][:100_000].rstrip(b"\n")}
    os.mkdir(os.path.join(d, "ab"))
    for name, text in afiles.items():
        open(os.path.join(d, "ab", name), "wb").write(text)
    run(["dist", "-A", "-L", "s.shuf", "-o", "koc_dev", "ab"], d)
    run(["dist", "-A", "-L", "s.shuf", "-o", "koc_host", "ab"], d, env={"KSSD_HOST_FASTQ": "1"})
    for fn in ("combco.0", "combco.0.a", "combco.index.0", "cofiles.stat"):
        assert open(os.path.join(d, "koc_dev", fn), "rb").read() == open(os.path.join(d, "koc_host", fn), "rb").read(), fn
    got = np.fromfile(os.path.join(d, "koc_dev", "combco.0"), np.uint32)
    cnt = np.fromfile(os.path.join(d, "koc_dev", "combco.0.a"), np.uint16)
    idx = np.fromfile(os.path.join(d, "koc_dev", "combco.index.0"), np.uint64)
    for g, name in enumerate(sorted(afiles)):
        wi, wc = sk.fastq_koc(afiles[name])
        lo, hi = int(idx[g]), int(idx[g + 1])
        assert np.array_equal(got[lo:hi], wi) and np.array_equal(cnt[lo:hi], wc), name


def test_waves_of_gzipped_files_give_the_files_of_the_plain_inputs(tmp_path):
    """host/kssd_cli_stage1.c: gzip'ed inputs are unpacked two waves at a time, file i of the one in step with file i of the other
    (kssd_slurp_reuse2), and ahead of the runtime's start.  Several waves at -p 4 -- whole, short and mixed ones, a file of two members,
    an empty member, a plain file between them -- must leave the combco.* of the same inputs plain: by default, one file at a time, with
    zlib, with one set of wave buffers ahead.  A damaged file is named."""
    import gzip
    d = str(tmp_path)
    rng = np.random.default_rng(17)
    run(["shuffle", "-k", 10, "-s", 6, "-l", 3, "-o", "L3K10", "--seed", META["seed"]], d)
    os.mkdir(os.path.join(d, "gz")); os.mkdir(os.path.join(d, "fa"))
    def fasta(n, name):
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)]
        rows = (n + 69) // 70
        buf = np.full((rows, 71), 10, np.uint8); flat = np.full(rows * 70, ord("A"), np.uint8); flat[:n] = a; buf[:, :70] = flat.reshape(rows, 70)
        return b">" + name.encode() + b" x\n" + buf.tobytes()
    for i in range(21):      # five waves of four files + one: whole waves, the last one short
        t = fasta(int(rng.integers(20_000, 120_000)), "g%02d" % i)
        if i in (6, 17):     # two members
            z = gzip.compress(t[: len(t) // 3], 1) + gzip.compress(t[len(t) // 3:], 6)
        elif i == 9:         # an empty member behind the text
            z = gzip.compress(t, 6) + gzip.compress(b"", 6)
        else:
            z = gzip.compress(t, 1 if i % 2 else 6)
        if i == 14:          # a plain file in the middle of a wave
            open(os.path.join(d, "gz", "s%02d.fasta" % i), "wb").write(t)
        else:
            open(os.path.join(d, "gz", "s%02d.fasta.gz" % i), "wb").write(z)
        open(os.path.join(d, "fa", "s%02d.fasta" % i), "wb").write(t)
    def sets(sub):
        return {os.path.basename(nm).replace(".gz", ""): ids for nm, ids in ko.sketch_sets_by_name(os.path.join(d, sub)).items()}
    run(["dist", "-p", 4, "-L", "L3K10.shuf", "-o", "o_fa", "fa"], d)
    want = sets("o_fa")
    assert len(want) == 21 and all(len(v) > 0 for v in want.values())
    for tag, env in (("pairs", {}), ("single", {"KSSD_GZ_ONE_AT_A_TIME": "1"}), ("zlib", {"KSSD_ZLIB_GUNZIP": "1"}), ("ahead1", {"KSSD_GZ_AHEAD": "1"}),
                     ("spinning", {"OMP_WAIT_POLICY": "active"})):
        run(["dist", "-p", 4, "-L", "L3K10.shuf", "-o", "o_" + tag, "gz"], d, env=env)
        got = sets("o_" + tag)
        assert sorted(got) == sorted(want), tag
        for nm in want:
            assert np.array_equal(got[nm], want[nm]), (tag, nm)
    bad = bytearray(open(os.path.join(d, "gz", "s03.fasta.gz"), "rb").read()); bad[len(bad) // 2] ^= 0x40
    open(os.path.join(d, "gz", "s03.fasta.gz"), "wb").write(bytes(bad))
    r = subprocess.run([BIN, "dist", "-p", "4", "-L", "L3K10.shuf", "-o", "o_bad", "gz"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode != 0 and b"s03.fasta.gz" in r.stdout, r.stdout.decode()[-500:]


def test_many_inputs_take_four_workers_per_device_and_leave_the_same_files(tmp_path):
    """from 2 048 inputs on `kssd dist` runs four sketch workers per device (host/kssd_cli_stage1.c: the copy engine idles a third of
    the time with two once start-up is amortised): 2 100 names hard-linked onto a dozen small files, the default against
    KSSD_WORKERS_PER_DEVICE=1 -- combco.0, combco.index.0 and cofiles.stat byte for byte, and every name's sketch the oracle's"""
    from synth import fasta_text
    d = str(tmp_path)
    rng = np.random.default_rng(2100)
    shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
    shuf.write(os.path.join(d, "L3K10.shuf"))
    os.mkdir(os.path.join(d, "src"))
    os.mkdir(os.path.join(d, "fa"))
    sk = ko.Sketcher(shuf.table, 10, 6, 3)
    want = []
    for i in range(12):
        t = fasta_text(rng.integers(0, 4, int(rng.integers(20_000, 120_000)), dtype=np.uint8), b"g%d" % i)
        open(os.path.join(d, "src", "g%02d.fasta" % i), "wb").write(t)
        want.append(np.sort(sk.fasta(t)))
    for r in range(175):
        for i in range(12):
            os.link(os.path.join(d, "src", "g%02d.fasta" % i), os.path.join(d, "fa", "h%03d_g%02d.fasta" % (r, i)))
    run(["dist", "-p", 8, "-L", "L3K10.shuf", "-o", "o4", "fa"], d)
    run(["dist", "-p", 8, "-L", "L3K10.shuf", "-o", "o1", "fa"], d, env={"KSSD_WORKERS_PER_DEVICE": "1"})
    for fn in ("combco.0", "combco.index.0", "cofiles.stat"):
        assert open(os.path.join(d, "o4", fn), "rb").read() == open(os.path.join(d, "o1", fn), "rb").read(), fn
    sets = ko.sketch_sets_by_name(os.path.join(d, "o4"))
    assert len(sets) == 2100
    for nm, ids in sets.items():
        assert np.array_equal(ids, want[int(nm[-8:-6])]), nm
