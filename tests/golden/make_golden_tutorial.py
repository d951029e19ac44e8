#!/usr/bin/env python3
"""BASELINE configs[0] exactly as the reference's README runs it (README.md:33-45, "Quick-Tutorial"): ALL of test_fna -- seqs1
(20 genomes: B. cereus AE016877 and its 1 % .. 30 % mutated copies) as references, seqs2 (11 genomes) as queries -- through the
five commands of the tutorial with the REAL reference binary (oracle/_ref/kssd):

    kssd dist -L L3K10.shuf -o reference seqs1 ; kssd dist -o reference reference
    kssd dist -L L3K10.shuf -o query seqs2
    kssd dist -r reference -o distout query ; kssd dist -r reference -o distout2 reference

The 31 input files are data the reference ships for its own tutorial; they are copied as fixtures (tests/golden/test_fna) because
/root/reference does not exist on the GPU box.  Stored: every genome's ids in the reference's FILE order, both shared matrices by
name, both distance.out texts.  Run in the dev container only:

    python tests/golden/make_golden_tutorial.py
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

SEED = 20260101   # the same L3K10 permutation as make_golden_testfna.py
SRC = "/root/reference/test_fna"


def main():
    assert ko.have_ref(), "oracle/_ref/kssd missing: run `make -C oracle` in the dev container"
    dst = os.path.join(HERE, "test_fna")
    for sub in ("seqs1", "seqs2"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
        for fn in sorted(os.listdir(os.path.join(SRC, sub))):
            if not os.path.exists(os.path.join(dst, sub, fn)):
                shutil.copyfile(os.path.join(SRC, sub, fn), os.path.join(dst, sub, fn))
    tmp = tempfile.mkdtemp(prefix="kssd_golden_tut_")
    try:
        K.Shuf.generate(10, 6, 3, seed=SEED).write(os.path.join(tmp, "L3K10.shuf"))
        for sub in ("seqs1", "seqs2"):
            os.symlink(os.path.join(dst, sub), os.path.join(tmp, sub))
        ko.run_ref(["dist", "-p", 8, "-L", "L3K10.shuf", "-o", "reference", "seqs1"], cwd=tmp)
        ko.run_ref(["dist", "-p", 8, "-o", "reference", "reference"], cwd=tmp)
        ko.run_ref(["dist", "-p", 8, "-L", "L3K10.shuf", "-o", "query", "seqs2"], cwd=tmp)
        ko.run_ref(["dist", "-p", 8, "-r", "reference", "-o", "distout", "--keepskf", "query"], cwd=tmp)
        ko.run_ref(["dist", "-p", 8, "-r", "reference", "-o", "distout2", "--keepskf", "reference"], cwd=tmp)
        out = {}
        for sub, d in (("ref", "reference"), ("qry", "query")):
            hdr, names, off, ids = ko.read_sketch_dir(os.path.join(tmp, d))
            for i, nm in enumerate(names):
                out["%s/%s" % (sub, os.path.basename(nm))] = ids[int(off[i]):int(off[i + 1])]
        _, rsz, rnames = ko.read_stat(os.path.join(tmp, "reference", "mcofiles.stat"), mco=True)
        _, qsz, qnames = ko.read_stat(os.path.join(tmp, "query", "cofiles.stat"))
        _, r2sz, r2names = ko.read_stat(os.path.join(tmp, "reference", "cofiles.stat"))
        sh = np.fromfile(os.path.join(tmp, "distout", "sharedk_ct.dat"), dtype=np.uint32).reshape(len(qnames), len(rnames))
        sh2 = np.fromfile(os.path.join(tmp, "distout2", "sharedk_ct.dat"), dtype=np.uint32).reshape(len(r2names), len(rnames))
        base = lambda v: np.array([os.path.basename(n) for n in v])
        np.savez_compressed(os.path.join(HERE, "tutorial.npz"), shared=sh, shared_refs=sh2, ref_names=base(rnames), qry_names=base(qnames),
                            refq_names=base(r2names),
                            distance_out=np.frombuffer(open(os.path.join(tmp, "distout", "distance.out"), "rb").read(), dtype=np.uint8),
                            distance_out_refs=np.frombuffer(open(os.path.join(tmp, "distout2", "distance.out"), "rb").read(), dtype=np.uint8), **out)
        print("tutorial goldens written:", len(rnames), "references,", len(qnames), "queries; sketch sizes", sorted(rsz)[:3], "..", sorted(qsz)[:3])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
