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
PICK = {"seqs1": "10_AE016877.fasta.gz", "seqs2": "25_AE016877.fasta.gz"}


def main():
    assert ko.have_ref(), "oracle/_ref/kssd missing: run `make -C oracle` in the dev container"
    dst = os.path.join(HERE, "test_fna")
    shutil.rmtree(dst, ignore_errors=True)
    for sub, fn in PICK.items():
        os.makedirs(os.path.join(dst, sub))
        shutil.copyfile(os.path.join(SRC, sub, fn), os.path.join(dst, sub, fn))
    tmp = tempfile.mkdtemp(prefix="kssd_golden_fna_")
    try:
        shuf = K.Shuf.generate(10, 6, 3, seed=SEED)
        sp = os.path.join(tmp, "L3K10.shuf")
        shuf.write(sp)
        # the tutorial (README.md:37-44): reference database from seqs1, query sketches from seqs2, search
        ko.run_ref(["dist", "-p", 1, "-L", sp, "-r", os.path.join(dst, "seqs1"), "-o", "refdb"], cwd=tmp)
        ko.run_ref(["dist", "-p", 1, "-L", sp, "-o", "qry", os.path.join(dst, "seqs2")], cwd=tmp)
        ko.run_ref(["dist", "-p", 1, "-r", "refdb", "-o", "out", "--keepskf", "qry"], cwd=tmp)
        text = open(os.path.join(tmp, "out", "distance.out"), "rb").read().decode()
        text = text.replace(os.path.join(dst, "seqs1"), "REF").replace(os.path.join(dst, "seqs2"), "QRY")
        np.savez_compressed(os.path.join(HERE, "test_fna.npz"),
                            ref_combco=np.fromfile(os.path.join(tmp, "refdb", "combco.0"), dtype=np.uint32),
                            qry_combco=np.fromfile(os.path.join(tmp, "qry", "combco.0"), dtype=np.uint32),
                            shared=np.fromfile(os.path.join(tmp, "out", "sharedk_ct.dat"), dtype=np.uint32),
                            distance_out=np.frombuffer(text.encode(), dtype=np.uint8))
        print("test_fna goldens written; sketch sizes", os.path.getsize(os.path.join(tmp, "refdb", "combco.0")) // 4,
              os.path.getsize(os.path.join(tmp, "qry", "combco.0")) // 4, "\n" + text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
