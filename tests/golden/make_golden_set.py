#!/usr/bin/env python3
"""Golden vectors of `kssd set` (union, uniq union, subtract, intersect) from the REAL reference (oracle/_ref/kssd) on
the committed golden FASTA inputs.  Run in the dev container only:  python tests/golden/make_golden_set.py
Writes tests/golden/set_ops.npz."""
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


def main():
    assert ko.have_ref(), "oracle/_ref/kssd missing: run `make -C oracle` in the dev container"
    tmp = tempfile.mkdtemp(prefix="kssd_golden_set_")
    try:
        K.Shuf.generate(10, 6, 3, seed=SEED).write(os.path.join(tmp, "L3K10.shuf"))
        ko.run_ref(["dist", "-p", 2, "-L", "L3K10.shuf", "-o", "ref", os.path.join(HERE, "ref_fa")], cwd=tmp)
        ko.run_ref(["dist", "-p", 2, "-L", "L3K10.shuf", "-o", "qry", os.path.join(HERE, "qry_fa")], cwd=tmp)
        ko.run_ref(["set", "-u", "-o", "U", "ref"], cwd=tmp)
        ko.run_ref(["set", "-q", "-o", "Q", "ref"], cwd=tmp)
        ko.run_ref(["set", "-s", "U", "-o", "S", "qry"], cwd=tmp)
        ko.run_ref(["set", "-i", "U", "-o", "I", "qry"], cwd=tmp)
        ko.run_ref(["set", "-i", "Q", "-o", "IQ", "qry"], cwd=tmp)  # a uniq_pan as the pan-sketch
        out = {"union": np.fromfile(os.path.join(tmp, "U", "pan.0"), np.uint32),
               "uniq": np.fromfile(os.path.join(tmp, "Q", "uniq_pan.0"), np.uint32)}
        for tag, d in (("sub", "S"), ("int", "I"), ("intq", "IQ")):
            _, sizes, names = ko.read_stat(os.path.join(tmp, d, "cofiles.stat"))
            out[tag + "_ids"] = np.fromfile(os.path.join(tmp, d, "combco.0"), np.uint32)
            out[tag + "_index"] = np.fromfile(os.path.join(tmp, d, "combco.index.0"), np.uint64)
            out[tag + "_sizes"] = sizes
            out[tag + "_names"] = np.array([os.path.basename(n) for n in names])
        np.savez_compressed(os.path.join(HERE, "set_ops.npz"), **out)
        print({k: (v.shape, v.dtype) for k, v in out.items()})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
