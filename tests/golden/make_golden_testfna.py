#!/usr/bin/env python3
"""BASELINE configs[0] on real sequence: two of the reference's own test genomes (test_fna: B. cereus AE016877 and a
mutated copy, single-record gzip'ed FASTA -- data files the reference ships for its quick tutorial, README.md:33-45)
copied as fixtures, and what the REAL reference binary (oracle/_ref/kssd) writes for the tutorial flow on them:
combco.0 of both sketch directories, sharedk_ct.dat, distance.out.  Run in the dev container only:

    python tests/golden/make_golden_testfna.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kssd_oracle as ko  # noqa: E402
import public_kssd_amd as K  # noqa: E402

SEED = 20260101
SRC = "/root/reference/test_fna"
# round 4: the original genome beside the 10 % copy as references; the 25 % and the 29 % copy (the most mutated one seqs2 holds)
# and the multi-record edge-case file of tests/golden/qry_fa (lower case, N, IUPAC, CRLF, a header across a 64 KiB boundary) as
# queries -- a 3 x 2 matrix.  The reference shuffles its input order by the clock, so everything is stored by file name.
PICK = {"seqs1": ["10_AE016877.fasta.gz", "AE016877.fasta.gz"], "seqs2": ["25_AE016877.fasta.gz", "29_AE016877.fasta.gz"]}
EXTRA_QRY = os.path.join(HERE, "qry_fa", "edge.fa")


def main():
    assert ko.have_ref(), "oracle/_ref/kssd missing: run `make -C oracle` in the dev container"
    dst = os.path.join(HERE, "test_fna")                     # (holds ALL of test_fna since round 6: make_golden_tutorial.py)
    for sub, fns in PICK.items():
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
        for fn in fns:
            if not os.path.exists(os.path.join(dst, sub, fn)):
                shutil.copyfile(os.path.join(SRC, sub, fn), os.path.join(dst, sub, fn))
    tmp = tempfile.mkdtemp(prefix="kssd_golden_fna_")
    try:
        shuf = K.Shuf.generate(10, 6, 3, seed=SEED)
        sp = os.path.join(tmp, "L3K10.shuf")
        shuf.write(sp)
        qdir = os.path.join(tmp, "qin")                      # the queries: seqs2 + the edge-case file (not copied into test_fna: it is a fixture already)
        os.mkdir(qdir)
        for fn in PICK["seqs2"]:
            os.symlink(os.path.join(dst, "seqs2", fn), os.path.join(qdir, fn))
        os.symlink(EXTRA_QRY, os.path.join(qdir, "edge.fa"))
        # the tutorial (README.md:37-44): reference database from seqs1, query sketches from seqs2, search
        rdir = os.path.join(tmp, "rin")                      # the two picked references only
        os.mkdir(rdir)
        for fn in PICK["seqs1"]:
            os.symlink(os.path.join(dst, "seqs1", fn), os.path.join(rdir, fn))
        ko.run_ref(["dist", "-p", 1, "-L", sp, "-r", rdir, "-o", "refdb"], cwd=tmp)
        ko.run_ref(["dist", "-p", 1, "-L", sp, "-o", "qry", qdir], cwd=tmp)
        ko.run_ref(["dist", "-p", 1, "-r", "refdb", "-o", "out", "--keepskf", "qry"], cwd=tmp)
        text = open(os.path.join(tmp, "out", "distance.out"), "rb").read().decode()
        text = text.replace(rdir, "REF").replace(qdir, "QRY")
        out = {}
        for sub, d in (("ref", "refdb"), ("qry", "qry")):
            hdr, names, off, ids = ko.read_sketch_dir(os.path.join(tmp, d))
            for i, nm in enumerate(names):                  # a genome's ids in the reference's FILE order
                out["%s/%s" % (sub, os.path.basename(nm))] = ids[int(off[i]):int(off[i + 1])]
        _, rsz, rnames = ko.read_stat(os.path.join(tmp, "refdb", "mcofiles.stat"), mco=True)
        _, qsz, qnames = ko.read_stat(os.path.join(tmp, "qry", "cofiles.stat"))
        sh = np.fromfile(os.path.join(tmp, "out", "sharedk_ct.dat"), dtype=np.uint32).reshape(len(qnames), len(rnames))
        np.savez_compressed(os.path.join(HERE, "test_fna.npz"), shared=sh,
                            ref_names=np.array([os.path.basename(n) for n in rnames]), qry_names=np.array([os.path.basename(n) for n in qnames]),
                            distance_out=np.frombuffer(text.encode(), dtype=np.uint8), **out)
        print("test_fna goldens written; sketch sizes", rsz, qsz, "\n" + text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
